// adil_stem.hip — the frozen classifier's ResNet stem, either side of the ADiL hot path (gfx950).
//
// The attack hands x_adv = x + D v (NCHW, bf16/fp32) to the classifier and takes dLoss/dx_adv back.  For the
// torchvision ResNets the reference attacks (demo_dL_attack.py:41-59) the first stage
//     Normalize -> conv 7x7/2 (3 -> 64) -> BatchNorm(eval) -> ReLU -> maxpool 3x3/2
// is where a library convolution is weakest: 3 input channels (MIOpen pads / transposes in and out, zero-fills,
// 1.4 ms forward and 4.4 ms input gradient at B = 512) and a 9-read pooling kernel.  These kernels do the stage
// directly on the attack's own tensor layouts:
//   stem_conv_fwd   x_adv (NCHW)  -> y1 (NHWC bf16) = relu(bn(conv(normalize(x))))     implicit GEMM on MFMA
//   maxpool_fwd     y1 -> p (NHWC), argmax index (1 byte / element)
//   stem_pool_bwd   g_p, idx, p -> g_y1 = route(g_p) * [p > 0] * bn_scale             (maxpool + ReLU + BN backward)
//   stem_conv_bwd   g_y1 -> g_x (NCHW, the layout adil_grad consumes)                  implicit GEMM on MFMA
// Not part of the ADiL maths: the parity target is the plain PyTorch stem (tests/test_gpu_stem.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adil_common.h"
#include "adil_hip.h"
#include "adil_mfma.h"

namespace {

struct StemNorm { float mean[3]; float inv_std[3]; };

template <typename T> __device__ __forceinline__ float ld_px(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_px<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_px<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }

// =========================================================================================================== //
// stem_conv_fwd.  Workgroup (4 waves) = a 16 x 16 tile of conv outputs of one image, all 64 channels.
//   K order per tap row kh: (kw 0..7, ci 0..3) = 32 values, kw = 7 and ci = 3 carrying zero weights, so that with the
//   input tile staged in LDS as [row][col][4 ch] a fragment (8 consecutive k) of output pixel (oh, ow) is ONE aligned
//   16-byte read at element 8*ow + 16*g + 8*h of row 2*oh + kh.  K = 7 * 32 = 224 (14 k-groups of 16).
//   MFMA roles: A = weights (rows = output channels), B = patches (columns = pixels), so a lane ends up with 4
//   consecutive channels of ONE pixel per register quad; a per-wave LDS transpose then gives 16-byte NHWC stores.
// =========================================================================================================== //
#define ST_T 16                        // conv-output tile edge
#define ST_ROWS (2 * ST_T + 5)         // 37 input rows
#define ST_COLS 40                     // 38 input columns used (kw = 7 pad reaches column 2*15 + 7), padded
#define ST_RS (ST_COLS * 4 + 8)        // input row stride (elements): 336 B = 21 x 16 B
#define ST_K 224
#define ST_WS (ST_K + 8)               // weight row stride (elements): 464 B = 29 x 16 B
#define ST_OS 72                       // staged-output pixel stride (elements)

template <typename TX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void stem_conv_fwd_kernel(const TX* __restrict__ x, const bf16_t* __restrict__ wf,
                                                            StemNorm nm, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                            int H, int W, int OH, int OW) {
    // one LDS block: [input tile | weights]; the per-wave output-transpose scratch aliases it once the MFMAs are done
    // (42 KB instead of 60 KB: 3 workgroups per CU)
    __shared__ __attribute__((aligned(16))) bf16_t smem[ST_ROWS * ST_RS + 64 * ST_WS];
    static_assert(4 * 32 * ST_OS <= ST_ROWS * ST_RS + 64 * ST_WS, "output scratch must fit in the tile buffers");
    bf16_t* sin = smem;
    bf16_t* sw = smem + ST_ROWS * ST_RS;
    bf16_t* sout = smem;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int n = blockIdx.z, oh0 = blockIdx.y * ST_T, ow0 = blockIdx.x * ST_T;
    // weights: 64 rows x 224 bf16 = 28 chunks of 16 B per row
    {
        u32x4 wv[7];                                       // 64 * 28 = 7 * 256 chunks: all in flight together
#pragma unroll
        for (int j = 0; j < 7; ++j) wv[j] = *reinterpret_cast<const u32x4*>(wf + (size_t)(tid + 256 * j) * 8);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int q = tid + 256 * j, co = q / 28, off = q - co * 28;
            *reinterpret_cast<u32x4*>(sw + co * ST_WS + off * 8) = wv[j];
        }
    }
    // input tile, normalised, zero outside the image (the padding of the normalised tensor), 4th channel zero
    const TX* xn = x + (size_t)n * 3 * H * W;
    // (all 18 loads of a thread are issued before the first conversion: a rolled loop pays one memory round trip per
    // position)
    constexpr int NPOS = (ST_ROWS * ST_COLS + 255) / 256;
    float raw[NPOS][3], okm[NPOS];
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int pos = tid + 256 * j < ST_ROWS * ST_COLS ? tid + 256 * j : ST_ROWS * ST_COLS - 1;
        const int row = pos / ST_COLS, col = pos - row * ST_COLS;
        const int ih = 2 * oh0 - 3 + row, iw = 2 * ow0 - 3 + col;
        const bool ok = (ih >= 0) && (ih < H) && (iw >= 0) && (iw < W);
        const size_t at = (size_t)(ok ? ih : 0) * W + (ok ? iw : 0);
        okm[j] = ok ? 1.0f : 0.0f;
        raw[j][0] = ld_px<TX>(xn, at);
        raw[j][1] = ld_px<TX>(xn + (size_t)H * W, at);
        raw[j][2] = ld_px<TX>(xn + (size_t)2 * H * W, at);
    }
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
        const int pos = tid + 256 * j;
        const int row = pos / ST_COLS, col = pos - row * ST_COLS;
        const float v0 = (raw[j][0] - nm.mean[0]) * nm.inv_std[0] * okm[j];
        const float v1 = (raw[j][1] - nm.mean[1]) * nm.inv_std[1] * okm[j];
        const float v2 = (raw[j][2] - nm.mean[2]) * nm.inv_std[2] * okm[j];
        u32x2 t;
        t[0] = pack2_bf16(v0, v1);
        t[1] = pack2_bf16(v2, 0.0f);
        if (pos < ST_ROWS * ST_COLS) *reinterpret_cast<u32x2*>(sin + row * ST_RS + col * 4) = t;
    }
    lds_sync();
    // wave w: pixel tiles 2w, 2w+1 (tile t = conv rows 2t, 2t+1 x 16 columns; lane c <-> (c / 16, c % 16))
    f32x16 acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][ct][r] = 0.0f;
    const int prow = c >> 4, pcol = c & 15;
#pragma unroll 1
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) a[ct] = lds8(sw + (ct * 32 + c) * ST_WS + kh * 32 + 16 * g + 8 * h);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int orow = 2 * (2 * w + p) + prow;
                b[p] = lds8(sin + (2 * orow + kh) * ST_RS + 8 * pcol + 16 * g + 8 * h);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) mma16(acc[p][ct], a[ct], b[p]);
        }
    }
    // epilogue: BN affine + ReLU, per-wave transpose through LDS, 16-byte NHWC stores
    lds_sync();                                            // every wave is done reading the tiles the scratch aliases
    bf16_t* so = sout + w * 32 * ST_OS;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = ct * 32 + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[p][ct][4 * q + j] * scale[co + j] + shift[co + j], 0.0f);
                u32x2 t;
                t[0] = pack2_bf16(v[0], v[1]);
                t[1] = pack2_bf16(v[2], v[3]);
                *reinterpret_cast<u32x2*>(so + c * ST_OS + co) = t;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = lane + 64 * i, px = id >> 3, ch = (id & 7) * 8;
            const int oh = oh0 + 2 * (2 * w + p) + (px >> 4), ow = ow0 + (px & 15);
            const u32x4 t = *reinterpret_cast<const u32x4*>(so + px * ST_OS + ch);
            if (oh < OH && ow < OW) *reinterpret_cast<u32x4*>(y + (((size_t)n * OH + oh) * OW + ow) * 64 + ch) = t;
        }
    }
}

// =========================================================================================================== //
// maxpool 3x3 / stride 2 / pad 1 on NHWC bf16, 8 channels per thread; also records the argmax position
// (kh * 3 + kw, first maximum in scan order, NaN wins — the rule of torch's max_pool2d) for the backward.
// =========================================================================================================== //

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ y, bf16_t* __restrict__ p,
                                                          uint8_t* __restrict__ idx, int B, int OH, int OW, int PH, int PW,
                                                          int C8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * PH * PW * C8;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int pw = (int)(r % PW); r /= PW;
    const int ph = (int)(r % PH);
    const int n = (int)(r / PH);
    float best[8];
    unsigned bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0xffu; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int oh = 2 * ph - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ow = 2 * pw - 1 + kw;
            if (oh >= 0 && oh < OH && ow >= 0 && ow < OW) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(y + ((((size_t)n * OH + oh) * OW + ow) * C8 + c8) * 8);
                float f[8];
                unpack8(t, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool take = (f[j] > best[j]) || (f[j] != f[j]) || (bi[j] == 0xffu);
                    best[j] = take ? f[j] : best[j];
                    bi[j] = take ? (unsigned)(kh * 3 + kw) : bi[j];
                }
            }
        }
    }
    *reinterpret_cast<u32x4*>(p + i * 8) = pack8(best);
    u32x2 o;
    o[0] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    o[1] = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<u32x2*>(idx + i * 8) = o;
}

// maxpool backward (gather form: every conv-output pixel looks at the <= 4 windows that contain it) fused with the
// ReLU mask (the routed element equals the pooled value p, so [y1 > 0] == [p > 0]: y1 itself is not needed) and
// the eval-BatchNorm scale.  g_y1 is the gradient wrt the convolution output.
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(const bf16_t* __restrict__ g, const uint8_t* __restrict__ idx,
                                                            const bf16_t* __restrict__ p, const float* __restrict__ scale,
                                                            bf16_t* __restrict__ gy, int B, int OH, int OW, int PH, int PW,
                                                            int C8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * OH * OW * C8;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const int ph0 = oh >> 1, nph = 1 + (oh & 1), pw0 = ow >> 1, npw = 1 + (ow & 1);
    for (int a = 0; a < nph; ++a) {
        const int ph = ph0 + a;
        if (ph >= PH) continue;
        const int kh = oh - (2 * ph - 1);
        for (int b = 0; b < npw; ++b) {
            const int pw = pw0 + b;
            if (pw >= PW) continue;
            const unsigned want = (unsigned)(kh * 3 + (ow - (2 * pw - 1)));
            const size_t at = ((((size_t)n * PH + ph) * PW + pw) * C8 + c8) * 8;
            const u32x2 id = *reinterpret_cast<const u32x2*>(idx + at);
            const u32x4 gt = *reinterpret_cast<const u32x4*>(g + at);
            const u32x4 pt = *reinterpret_cast<const u32x4*>(p + at);
            float gf[8], pf[8];
            unpack8(gt, gf);
            unpack8(pt, pf);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned k = (id[j >> 2] >> (8 * (j & 3))) & 0xffu;
                acc[j] += (k == want && pf[j] > 0.0f) ? gf[j] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= scale[c8 * 8 + j];
    *reinterpret_cast<u32x4*>(gy + i * 8) = pack8(acc);
}

// =========================================================================================================== //
// stem_conv_bwd: input gradient of the 7x7/2 convolution (64 -> 3 channels), times 1/std of the normalisation.
//   g_x[ci][ih][iw] = inv_std[ci] * sum_{kh,kw,co} g_y1[(ih+3-kh)/2][(iw+3-kw)/2][co] * w[co][ci][kh][kw]
//   over the taps with even ih+3-kh, iw+3-kw: the input pixels split into 4 parity classes (ih&1, iw&1) with
//   3 or 4 taps per axis.  Workgroup = 16 x 32 input pixels of one image = 4 classes x (8 x 16) pixels; wave w takes
//   sub-rows 2w, 2w+1 of every class.  MFMA roles: A = weights (rows = ci, 3 of 16 used, mfma_16x16x32), B = g_y1 patches
//   (columns = pixels; 8 consecutive co = one aligned 16-byte LDS read).
// =========================================================================================================== //
#define SB_TH 16
#define SB_TW 32
#define SB_GR (SB_TH / 2 + 3)          // 11 g_y1 rows
#define SB_GC (SB_TW / 2 + 3)          // 19 g_y1 columns
#define SB_PS 72                       // g_y1 pixel stride in LDS (elements): 144 B = 9 x 16 B

// v_mfma_f32_16x16x32_bf16: A lane l -> row l&15, k = 8*(l>>4)+j; B lane l -> column l&15, same k; C register r of
// lane l -> row 4*(l>>4)+r, column l&15 (layout checked on hardware).  With only 3 useful rows (ci) the 16-row shape
// wastes half as many MFMA cycles as 32x32x16: 16 pixels x 32 channels of K per instruction, 4 passes.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int A, int BB>
__device__ __forceinline__ void stem_bwd_class(f32x4 (&acc)[2], const bf16_t* sg, const bf16_t* swb, int w, int l16, int ks) {
    const int wrow = (l16 < 3 ? l16 : 0) * 49 * 64;
    const bool wlane = l16 < 3;                            // only 3 of the 16 A rows exist: the other lanes skip the LDS read
#pragma unroll 1                                           // rolled: a fully unrolled 196-MFMA body hoists LDS reads into spills
    for (int kh = 1 - A; kh < 7; kh += 2) {
#pragma unroll
        for (int kw = 1 - BB; kw < 7; kw += 2) {
            const int col = l16 + (BB + 3 - kw) / 2 + 1;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                bf16x8 a = {};
                if (wlane) a = lds8(swb + wrow + (kh * 7 + kw) * 64 + 32 * cg + 8 * ks);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = 2 * w + q + (A + 3 - kh) / 2 + 1;
                    const bf16x8 b = lds8(sg + (row * SB_GC + col) * SB_PS + 32 * cg + 8 * ks);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
                }
            }
        }
    }
}

template <typename TX> __device__ __forceinline__ void st_pair(TX* p, float a, float b);
template <> __device__ __forceinline__ void st_pair<float>(float* p, float a, float b) {
    *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
template <> __device__ __forceinline__ void st_pair<bf16_t>(bf16_t* p, float a, float b) {
    *reinterpret_cast<unsigned*>(p) = pack2_bf16(a, b);
}

template <typename TX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void stem_conv_bwd_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ wb,
                                                            StemNorm nm, TX* __restrict__ gx, int H, int W, int OH,
                                                            int OW) {
    __shared__ __attribute__((aligned(16))) bf16_t sg[SB_GR * SB_GC * SB_PS];
    __shared__ __attribute__((aligned(16))) bf16_t swb[3 * 49 * 64];   // the zero row (ci = 3) of w_bwd is never read
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int n = blockIdx.z, ih0 = blockIdx.y * SB_TH, iw0 = blockIdx.x * SB_TW;
    {   // weights (3 * 49 * 8 = 1176 chunks) and the g_y1 tile (11 * 19 * 8 = 1672 chunks): every load of a thread is in
        // flight before the first LDS store (rolled loops paid one memory round trip per chunk: 11 per workgroup)
        constexpr int NWQ = (3 * 49 * 8 + 255) / 256, NGQ = (SB_GR * SB_GC * 8 + 255) / 256;
        u32x4 wv[NWQ], gv[NGQ];
        bool gok[NGQ];
#pragma unroll
        for (int j = 0; j < NWQ; ++j) {
            const int q = tid + 256 * j < 3 * 49 * 8 ? tid + 256 * j : 3 * 49 * 8 - 1;
            wv[j] = *reinterpret_cast<const u32x4*>(wb + q * 8);
        }
        const int ohb = ih0 / 2 - 1, owb = iw0 / 2 - 1;
#pragma unroll
        for (int j = 0; j < NGQ; ++j) {
            const int q = tid + 256 * j < SB_GR * SB_GC * 8 ? tid + 256 * j : SB_GR * SB_GC * 8 - 1;
            const int ch = q & 7, px = q >> 3;
            const int row = px / SB_GC, col = px - row * SB_GC;
            const int oh = ohb + row, ow = owb + col;
            gok[j] = (oh >= 0) && (oh < OH) && (ow >= 0) && (ow < OW);
            gv[j] = *reinterpret_cast<const u32x4*>(gy + (((size_t)n * OH + (gok[j] ? oh : 0)) * OW + (gok[j] ? ow : 0)) * 64 + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < NWQ; ++j) {
            const int q = tid + 256 * j;
            if (q < 3 * 49 * 8) *reinterpret_cast<u32x4*>(swb + q * 8) = wv[j];
        }
#pragma unroll
        for (int j = 0; j < NGQ; ++j) {
            const int q = tid + 256 * j;
            const int ch = q & 7, px = q >> 3;
            u32x4 t = gv[j];
            if (!gok[j]) t = u32x4{0u, 0u, 0u, 0u};
            if (q < SB_GR * SB_GC * 8) *reinterpret_cast<u32x4*>(sg + px * SB_PS + ch * 8) = t;
        }
    }
    lds_sync();
    const int l16 = lane & 15, ks = lane >> 4;             // pixel column of the class sub-grid / K slice of this lane
    f32x4 acc[4][2];                                       // [parity class][sub-row 2w + q]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[k][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    stem_bwd_class<0, 0>(acc[0], sg, swb, w, l16, ks);
    stem_bwd_class<0, 1>(acc[1], sg, swb, w, l16, ks);
    stem_bwd_class<1, 0>(acc[2], sg, swb, w, l16, ks);
    stem_bwd_class<1, 1>(acc[3], sg, swb, w, l16, ks);
    if (ks == 0) {                                         // accumulator rows 0..2 (= ci) are registers 0..2 of lanes 0..15
        const int iw = iw0 + 2 * l16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ih = ih0 + 2 * (2 * w + q) + a;
                if (ih < H && iw < W) {
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        TX* o = gx + (((size_t)n * 3 + ci) * H + ih) * W + iw;
                        st_pair<TX>(o, acc[2 * a][q][ci] * nm.inv_std[ci], acc[2 * a + 1][q][ci] * nm.inv_std[ci]);
                    }
                }
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------ //
extern "C" int adil_stem_conv_fwd(const void* x, int x_dtype, const void* w_fwd, float mean0, float mean1, float mean2,
                                  float inv_std0, float inv_std1, float inv_std2, const float* scale, const float* shift,
                                  void* y, int B, int H, int W, void* stream) {
    ADIL_ENTER();
    if (!x || !w_fwd || !scale || !shift || !y || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ADIL_EINVAL;
    const StemNorm nm = {{mean0, mean1, mean2}, {inv_std0, inv_std1, inv_std2}};
    const int OH = H / 2, OW = W / 2;
    const dim3 grid((OW + ST_T - 1) / ST_T, (OH + ST_T - 1) / ST_T, B);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == ADIL_F32)
        hipLaunchKernelGGL(stem_conv_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const bf16_t*)w_fwd, nm,
                           scale, shift, (bf16_t*)y, H, W, OH, OW);
    else if (x_dtype == ADIL_BF16)
        hipLaunchKernelGGL(stem_conv_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w_fwd, nm,
                           scale, shift, (bf16_t*)y, H, W, OH, OW);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_maxpool_fwd(const void* y, void* p, uint8_t* idx, int B, int OH, int OW, int C, void* stream) {
    ADIL_ENTER();
    if (!y || !p || !idx || B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 7)) return ADIL_EINVAL;
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const size_t total = (size_t)B * PH * PW * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, (bf16_t*)p, idx, B, OH, OW, PH, PW, C / 8);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_stem_pool_bwd(const void* g, const uint8_t* idx, const void* p, const float* scale, void* gy, int B,
                                  int OH, int OW, int C, void* stream) {
    ADIL_ENTER();
    if (!g || !idx || !p || !scale || !gy || B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 7)) return ADIL_EINVAL;
    const int PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1;
    const size_t total = (size_t)B * OH * OW * (C / 8);
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)g, idx, (const bf16_t*)p, scale, (bf16_t*)gy, B, OH, OW, PH, PW, C / 8);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_stem_conv_bwd(const void* gy, const void* w_bwd, float inv_std0, float inv_std1, float inv_std2, void* gx,
                                  int gx_dtype, int B, int H, int W, void* stream) {
    ADIL_ENTER();
    if (!gy || !w_bwd || !gx || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ADIL_EINVAL;
    const StemNorm nm = {{0.0f, 0.0f, 0.0f}, {inv_std0, inv_std1, inv_std2}};
    const int OH = H / 2, OW = W / 2;
    const dim3 grid((W + SB_TW - 1) / SB_TW, (H + SB_TH - 1) / SB_TH, B);
    hipStream_t st = (hipStream_t)stream;
    if (gx_dtype == ADIL_F32)
        hipLaunchKernelGGL(stem_conv_bwd_kernel<float>, grid, dim3(256), 0, st, (const bf16_t*)gy, (const bf16_t*)w_bwd, nm,
                           (float*)gx, H, W, OH, OW);
    else if (gx_dtype == ADIL_BF16)
        hipLaunchKernelGGL(stem_conv_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)gy, (const bf16_t*)w_bwd, nm,
                           (bf16_t*)gx, H, W, OH, OW);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}
