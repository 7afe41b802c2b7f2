// adil_convs.hip — the frozen ResNet's convolutions behind the stem, on channels_last bf16 storage (gfx950):
//   pw_conv_fwd / pw_conv_bwd   1x1 convolutions (stride 1, and the stride-2 downsample ones through an in-kernel
//                               gather) as GEMMs with the BatchNorm / residual / ReLU epilogue (and
//                               the previous layer's BatchNorm + ReLU as a prologue) applied on chip, forward and input
//                               gradient
//   conv3x3                     3x3 / stride-1 convolutions as an implicit GEMM with linear pixel tiling
// Not part of the ADiL maths: the parity target is plain PyTorch (tests/test_gpu_stem.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adil_common.h"
#include "adil_hip.h"
#include "adil_mfma.h"

// =========================================================================================================== //
// Pointwise (1x1, stride 1) convolution of the frozen ResNet with its eval-BatchNorm / residual / ReLU epilogue
// in ONE kernel.  On channels_last storage the convolution is the row-major GEMM
//     Y[M][N] = act( (X[M][K] . W[N][K]^T) * scale[n] + shift[n] (+ R[M][N]) ),   M = B*H*W, K = Cin, N = Cout
// and at ResNet-50 / B = 512 it is HBM-bound (K <= 2048, activations of 0.1 - 0.8 GB): what matters is that every
// activation crosses HBM once.  A library GEMM + a separate epilogue kernel writes and re-reads the pre-activation
// tensor (the epilogue passes were 10.5 ms of a 51 ms step); here the epilogue runs on the accumulators.
//   Workgroup = 128 pixels x BN channels (BN = 128 or 64), K in chunks of 64 through LDS (the next chunk waits in
//   registers: one 37 KB buffer, 3-4 workgroups per CU).
//   MFMA roles as in the stem: A = W (rows = channels -> accumulator registers), B = X (columns = pixels -> lanes),
//   so a lane owns 4 consecutive channels of one pixel per register quad (8-byte residual loads); the finished tile
//   goes through a per-wave LDS transpose to 16-byte NHWC stores.
// =========================================================================================================== //
namespace {

#define PW_BM 128
#define PW_BK 64
#define PW_LS (PW_BK + 8)              // LDS row stride (elements): 144 B = 9 x 16 B
#define PW_PK 512                      // most input channels a fused prologue / backward epilogue supports

template <int BN, bool PRO, bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PRO && BN == 128 ? 2 : (RES ? 3 : 4)))) void pw_conv_fwd_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ wgt, const float* __restrict__ scale,
    const float* __restrict__ shift, const bf16_t* __restrict__ res, bf16_t* __restrict__ y, int M, int K, int N,
    int relu, int MT, int NT, const float* __restrict__ pscale, const float* __restrict__ pshift, int sub_w, int sub_hw) {
    constexpr int CT = BN / 32;                          // channel tiles per wave
    constexpr int XCH = PW_BM * PW_BK / 8 / 256;         // 16-byte chunks of the X tile per thread (4)
    constexpr int WCH = BN * PW_BK / 8 / 256;            // ... of the W tile (4 or 2)
    constexpr int OS = BN + 8;                           // transposed-output pixel stride (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sx = reinterpret_cast<bf16_t*>(smem_raw);    // [128][PW_LS]   (single buffer: the next chunk waits in
    bf16_t* sw = sx + PW_BM * PW_LS;                     // [BN][PW_LS]     registers; 37 KB -> 3-4 workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    // workgroups are dealt round-robin to the 8 XCDs: consecutive workgroups of ONE XCD share the pixel tile, so the
    // NT reads of an X tile meet in that XCD's L2
    int mt, nt;
    if ((MT & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        nt = j % NT;
        mt = (j / NT) * 8 + xcd;
    } else {
        nt = blockIdx.x % NT;
        mt = blockIdx.x / NT;
    }
    const int m0 = mt * PW_BM, n0 = nt * BN;
    const int nk = K / PW_BK;
    const int m = m0 + w * 32 + c;                       // this lane's pixel in the epilogue
    const bool mok = m < M;
    __shared__ __attribute__((aligned(16))) float ssc[2 * BN];   // this tile's scale | shift (visible after the first barrier)
    if (tid < BN) { ssc[tid] = scale[n0 + tid]; ssc[BN + tid] = shift[n0 + tid]; }
    // optional prologue: the X operand is the RAW output of the previous (library) convolution and its eval-BatchNorm
    // + ReLU is applied on the way from registers to LDS:  x' = relu(x * pscale[k] + pshift[k])   (K <= PW_PK)
    __shared__ __attribute__((aligned(16))) float spro[PRO ? 2 * PW_PK : 4];
    constexpr bool pro = PRO;
    if (pro) {
        const int i2 = 2 * tid < K ? 2 * tid : 0;                            // K <= 512: one float2 of each per thread
        const float2 a = *reinterpret_cast<const float2*>(pscale + i2), b = *reinterpret_cast<const float2*>(pshift + i2);
        if (2 * tid < K) {
            *reinterpret_cast<float2*>(spro + 2 * tid) = a;
            *reinterpret_cast<float2*>(spro + PW_PK + 2 * tid) = b;
        }
        __syncthreads();
    }

    // rows of X this thread stages (fixed over the K loop).  sub_w > 0: the convolution has stride 2 — output pixel
    // m = (n, oh, ow) of an (OH = sub_hw / sub_w) x (OW = sub_w) grid reads input pixel (n, 2 oh, 2 ow) of the
    // 2 OH x 2 OW tensor x points to (the stride-2 downsample convolutions gather, nothing is copied)
    size_t xrow[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int id = tid + 256 * i, row = id >> 3;
        const int mm = (m0 + row < M) ? m0 + row : M - 1;
        if (sub_w > 0) {
            const int n = mm / sub_hw, r = mm - n * sub_hw, oh = r / sub_w, ow = r - oh * sub_w;
            xrow[i] = (size_t)4 * n * sub_hw + (size_t)4 * oh * sub_w + 2 * ow;
        } else {
            xrow[i] = (size_t)mm;
        }
    }
    u32x4 xr[XCH], wr[WCH];
    auto load_tiles = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i, ch = id & 7;
            xr[i] = *reinterpret_cast<const u32x4*>(x + xrow[i] * K + kc * PW_BK + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            wr[i] = *reinterpret_cast<const u32x4*>(wgt + (size_t)(n0 + row) * K + kc * PW_BK + ch * 8);
        }
    };
    auto store_tiles = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            u32x4 t = xr[i];
            if (pro) {
                const f32x2* ps = reinterpret_cast<const f32x2*>(spro + kc * PW_BK + ch * 8);
                const f32x2* pb = reinterpret_cast<const f32x2*>(spro + PW_PK + kc * PW_BK + ch * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = relu_bf2(f32x2_to_bf2(bf2_to_f32x2(t[j]) * ps[j] + pb[j]));
            }
            *reinterpret_cast<u32x4*>(sx + row * PW_LS + ch * 8) = t;
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            *reinterpret_cast<u32x4*>(sw + row * PW_LS + ch * 8) = wr[i];
        }
    };

    load_tiles(0);
    // the residual does not depend on the GEMM: its loads fly under the whole K loop
    u32x2 rr[RES ? CT : 1][4];
    if (RES) {
        const bf16_t* rp = res + (size_t)(mok ? m : M - 1) * N + n0 + 4 * h;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[ct][q] = *reinterpret_cast<const u32x2*>(rp + 32 * ct + 8 * q);
    }
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    for (int it = 0; it < nk; ++it) {
        store_tiles(it);
        if (it + 1 < nk) load_tiles(it + 1);
        lds_barrier();
        const bf16_t* bx = sx + (w * 32 + c) * PW_LS + 8 * h;
        const bf16_t* bw = sw + c * PW_LS + 8 * h;
#pragma unroll
        for (int ks = 0; ks < PW_BK / 16; ++ks) {
            const bf16x8 b = lds8(bx + 16 * ks);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) mma16(acc[ct], lds8(bw + ct * 32 * PW_LS + 16 * ks), b);
        }
        lds_barrier();
    }
    // epilogue on the accumulators: lane = pixel m, register quad q of tile ct = channels n0 + 32ct + 8q + 4h .. +3
    bf16_t* so = reinterpret_cast<bf16_t*>(smem_raw) + w * 32 * OS;     // the tile buffers are idle now
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = 32 * ct + 8 * q + 4 * h;
            const f32x2* sc = reinterpret_cast<const f32x2*>(ssc + co);
            const f32x2* sh = reinterpret_cast<const f32x2*>(ssc + BN + co);
            f32x2 v0 = f32x2{acc[ct][4 * q], acc[ct][4 * q + 1]} * sc[0] + sh[0];
            f32x2 v1 = f32x2{acc[ct][4 * q + 2], acc[ct][4 * q + 3]} * sc[1] + sh[1];
            if (RES) {
                v0 += bf2_to_f32x2(rr[ct][q][0]);
                v1 += bf2_to_f32x2(rr[ct][q][1]);
            }
            u32x2 t;
            t[0] = f32x2_to_bf2(v0);
            t[1] = f32x2_to_bf2(v1);
            if (relu) { t[0] = relu_bf2(t[0]); t[1] = relu_bf2(t[1]); }
            *reinterpret_cast<u32x2*>(so + c * OS + co) = t;
        }
    }
    constexpr int CPP = BN / 8;                           // 16-byte chunks per pixel
#pragma unroll
    for (int i = 0; i < 32 * CPP / 64; ++i) {
        const int id = lane + 64 * i, px = id / CPP, ch = id - px * CPP;
        const u32x4 t = *reinterpret_cast<const u32x4*>(so + px * OS + ch * 8);
        const int mm = m0 + w * 32 + px;
        if (mm < M) *reinterpret_cast<u32x4*>(y + (size_t)mm * N + n0 + ch * 8) = t;
    }
}

// Input gradient of the same layer, again ONE kernel:  with v = g (+ g2), mask = [y > 0] (all ones without ReLU),
//     gres[M][N] = v * mask                      (gradient of the residual input; optional)
//     gx[M][K]   = (v * mask * scale[n]) . W     (W given transposed: wt[K][N], so the reduction index n is contiguous)
// The epilogue backward is applied to the X-operand chunk on its way from registers to LDS (it would otherwise be a
// separate kernel writing and re-reading an M x N tensor), and g2 lets the caller hand over the two gradients that
// meet at a residual join without adding them first (autograd's add kernels were 3 ms of a 46 ms step).
template <int BO, bool G3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BO == 128 || G3) ? 2 : 3))) void pw_conv_bwd_kernel(
    const bf16_t* __restrict__ g, const bf16_t* __restrict__ g2, const bf16_t* __restrict__ y,
    const float* __restrict__ scale, const bf16_t* __restrict__ wt, bf16_t* __restrict__ gx, bf16_t* __restrict__ gres,
    int M, int K, int N, int relu, int MT, int OT, const bf16_t* __restrict__ xin, const float* __restrict__ pscale,
    const float* __restrict__ pshift, const bf16_t* __restrict__ g3, int sub_w, int sub_hw) {
    constexpr int CT = BO / 32;
    constexpr int XCH = PW_BM * PW_BK / 8 / 256;         // 4
    constexpr int WCH = BO * PW_BK / 8 / 256;
    constexpr int OS = BO + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sx = reinterpret_cast<bf16_t*>(smem_raw);    // [128][PW_LS]  gz chunk
    bf16_t* sw = sx + PW_BM * PW_LS;                     // [BO][PW_LS]   wt chunk
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    int mt, ot;
    if ((MT & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        ot = j % OT;
        mt = (j / OT) * 8 + xcd;
    } else {
        ot = blockIdx.x % OT;
        mt = blockIdx.x / OT;
    }
    const int m0 = mt * PW_BM, k0 = ot * BO;
    const int nn = N / PW_BK;
    const bool write_res = (gres != nullptr) && (ot == 0);
    __shared__ __attribute__((aligned(16))) float ssc[2048];     // BatchNorm scale of every reduction channel
    {   // N <= 2048 floats = at most two float4 per thread, both in flight together (a rolled scalar loop pays up to
        // eight memory round trips before the first tile load is even issued)
        float4 sv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i4 = tid + 256 * j;
            sv[j] = *reinterpret_cast<const float4*>(scale + (4 * i4 < N ? 4 * i4 : 0));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i4 = tid + 256 * j;
            if (4 * i4 < N) *reinterpret_cast<float4*>(ssc + 4 * i4) = sv[j];
        }
    }
    __syncthreads();

    // G3: a third incoming gradient that lives on the stride-2 grid (it comes back from a stride-2 downsample
    // convolution reading this layer's output): pixel (n, h, w) of the 2 OH x 2 OW grid receives g3[(n, h/2, w/2)] when
    // h and w are even, nothing otherwise — the zero-upsampled tensor is never materialised.
    size_t g3row[G3 ? XCH : 1];
    float g3on[G3 ? XCH : 1];
    if (G3) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3;
            const int mm = (m0 + row < M) ? m0 + row : M - 1;
            const int hw4 = 4 * sub_hw, w2 = 2 * sub_w;
            const int n = mm / hw4, r = mm - n * hw4, hh = r / w2, ww = r - hh * w2;
            g3on[i] = ((hh | ww) & 1) ? 0.0f : 1.0f;
            g3row[i] = (size_t)n * sub_hw + (size_t)(hh >> 1) * sub_w + (ww >> 1);
        }
    }
    u32x4 gr[XCH], hr[XCH], yr[XCH], tr[G3 ? XCH : 1], wr[WCH];
    auto load_tiles = [&](int nc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const int mm = m0 + row;
            const size_t at = (size_t)(mm < M ? mm : M - 1) * N + nc * PW_BK + ch * 8;
            gr[i] = *reinterpret_cast<const u32x4*>(g + at);
            if (g2 != nullptr) hr[i] = *reinterpret_cast<const u32x4*>(g2 + at);
            if (relu) yr[i] = *reinterpret_cast<const u32x4*>(y + at);
            if (G3) tr[i] = *reinterpret_cast<const u32x4*>(g3 + g3row[i] * N + nc * PW_BK + ch * 8);   // always a valid address
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            wr[i] = *reinterpret_cast<const u32x4*>(wt + (size_t)(k0 + row) * N + nc * PW_BK + ch * 8);
        }
    };
    auto store_tiles = [&](int nc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            const f32x2* s2 = reinterpret_cast<const f32x2*>(ssc + nc * PW_BK + ch * 8);
            u32x4 rs, gz;                                             // gres chunk, gz chunk (packed bf16)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 v = bf2_to_f32x2(gr[i][j]);
                if (g2 != nullptr) v += bf2_to_f32x2(hr[i][j]);
                if (G3) v += bf2_to_f32x2(tr[i][j]) * g3on[i];
                const unsigned m = relu ? pos_mask_bf2(yr[i][j]) : 0xffffffffu;
                rs[j] = f32x2_to_bf2(v) & m;
                gz[j] = f32x2_to_bf2(v * s2[j]) & m;
            }
            const int mm = m0 + row;
            if (write_res && mm < M) *reinterpret_cast<u32x4*>(gres + (size_t)mm * N + nc * PW_BK + ch * 8) = rs;
            *reinterpret_cast<u32x4*>(sx + row * PW_LS + ch * 8) = gz;
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int id = tid + 256 * i, row = id >> 3, ch = id & 7;
            *reinterpret_cast<u32x4*>(sw + row * PW_LS + ch * 8) = wr[i];
        }
    };

    load_tiles(0);
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    for (int it = 0; it < nn; ++it) {
        store_tiles(it);
        if (it + 1 < nn) load_tiles(it + 1);
        lds_barrier();
        const bf16_t* bx = sx + (w * 32 + c) * PW_LS + 8 * h;
        const bf16_t* bw = sw + c * PW_LS + 8 * h;
#pragma unroll
        for (int ks = 0; ks < PW_BK / 16; ++ks) {
            const bf16x8 b = lds8(bx + 16 * ks);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) mma16(acc[ct], lds8(bw + ct * 32 * PW_LS + 16 * ks), b);
        }
        lds_barrier();
    }
    bf16_t* so = reinterpret_cast<bf16_t*>(smem_raw) + w * 32 * OS;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x2 t;
            t[0] = pack2_bf16(acc[ct][4 * q], acc[ct][4 * q + 1]);
            t[1] = pack2_bf16(acc[ct][4 * q + 2], acc[ct][4 * q + 3]);
            *reinterpret_cast<u32x2*>(so + c * OS + 32 * ct + 8 * q + 4 * h) = t;
        }
    }
    // optional epilogue: the forward fed this layer relu(xin * pscale + pshift) computed on the fly (xin = raw output
    // of the previous convolution), so the gradient handed back is wrt xin:  gx * [xin*pscale+pshift > 0] * pscale
    constexpr int CPP = BO / 8;
#pragma unroll
    for (int i = 0; i < 32 * CPP / 64; ++i) {
        const int id = lane + 64 * i, px = id / CPP, ch = id - px * CPP;
        u32x4 t = *reinterpret_cast<const u32x4*>(so + px * OS + ch * 8);
        const int mm = m0 + w * 32 + px;
        if (mm < M) {
            if (xin != nullptr) {
                float v[8], xv[8];
                unpack8(t, v);
                unpack8(*reinterpret_cast<const u32x4*>(xin + (size_t)mm * K + k0 + ch * 8), xv);
                const float4 a0 = *reinterpret_cast<const float4*>(pscale + k0 + ch * 8);
                const float4 a1 = *reinterpret_cast<const float4*>(pscale + k0 + ch * 8 + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(pshift + k0 + ch * 8);
                const float4 b1 = *reinterpret_cast<const float4*>(pshift + k0 + ch * 8 + 4);
                const float ps[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float pb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (xv[j] * ps[j] + pb[j] > 0.0f) ? v[j] * ps[j] : 0.0f;
                t = pack8(v);
            }
            *reinterpret_cast<u32x4*>(gx + (size_t)mm * K + k0 + ch * 8) = t;
        }
    }
}

template <int BO, bool G3>
int launch_pw_bwd(const void* g, const void* g2, const void* y, const float* scale, const void* wt, void* gx, void* gres,
                  int M, int K, int N, int relu, const void* xin, const float* pscale, const float* pshift, const void* g3,
                  int sub_w, int sub_hw, hipStream_t st) {
    const int MT = (M + PW_BM - 1) / PW_BM, OT = K / BO;
    const size_t tiles = (size_t)(PW_BM + BO) * PW_LS * sizeof(bf16_t);
    const size_t outb = (size_t)4 * 32 * (BO + 8) * sizeof(bf16_t);
    const size_t lds = tiles > outb ? tiles : outb;
    hipLaunchKernelGGL((pw_conv_bwd_kernel<BO, G3>), dim3((unsigned)(MT * OT)), dim3(256), lds, st, (const bf16_t*)g,
                       (const bf16_t*)g2, (const bf16_t*)y, scale, (const bf16_t*)wt, (bf16_t*)gx, (bf16_t*)gres, M, K, N,
                       relu, MT, OT, (const bf16_t*)xin, pscale, pshift, (const bf16_t*)g3, sub_w, sub_hw);
    ADIL_CHECK_LAUNCH();
    return 0;
}

template <int BN, bool PRO, bool RES>
int launch_pw_fwd_r(const void* x, const void* w, const float* scale, const float* shift, const void* res, void* y, int M,
                  int K, int N, int relu, const float* pscale, const float* pshift, int sub_w, int sub_hw, hipStream_t st) {
    const int MT = (M + PW_BM - 1) / PW_BM, NT = N / BN;
    const size_t tiles = (size_t)(PW_BM + BN) * PW_LS * sizeof(bf16_t);
    const size_t outb = (size_t)4 * 32 * (BN + 8) * sizeof(bf16_t);
    const size_t lds = tiles > outb ? tiles : outb;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)pw_conv_fwd_kernel<BN, PRO, RES>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((pw_conv_fwd_kernel<BN, PRO, RES>), dim3((unsigned)(MT * NT)), dim3(256), lds, st, (const bf16_t*)x,
                       (const bf16_t*)w, scale, shift, (const bf16_t*)res, (bf16_t*)y, M, K, N, relu, MT, NT, pscale, pshift, sub_w, sub_hw);
    ADIL_CHECK_LAUNCH();
    return 0;
}

template <int BN, bool PRO>
int launch_pw_fwd(const void* x, const void* w, const float* scale, const float* shift, const void* res, void* y, int M,
                  int K, int N, int relu, const float* pscale, const float* pshift, int sub_w, int sub_hw, hipStream_t st) {
    // without a residual the 32 registers of its prefetch are free: 4 workgroups per CU instead of 3
    if (res != nullptr)
        return launch_pw_fwd_r<BN, PRO, true>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
    return launch_pw_fwd_r<BN, PRO, false>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
}

}  // namespace

extern "C" int adil_pw_conv_bwd(const void* g, const void* g2, const void* y, const float* scale, const void* wt, void* gx,
                                void* gres, int M, int K, int N, int relu, const void* xin, const float* pscale,
                                const float* pshift, const void* g3, int sub_w, int sub_hw, void* stream) {
    ADIL_ENTER();
    if (!g || !scale || !wt || !gx || (relu && !y) || M <= 0 || K <= 0 || N <= 0 || (N % PW_BK) || (K % 64) || N > 2048)
        return ADIL_EINVAL;
    if (xin && (!pscale || !pshift)) return ADIL_EINVAL;
    if (g3 && (sub_w <= 0 || sub_hw <= 0 || sub_hw % sub_w || M % (4 * sub_hw))) return ADIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (g3) {
        if (K % 128 == 0)
            return launch_pw_bwd<128, true>(g, g2, y, scale, wt, gx, gres, M, K, N, relu, xin, pscale, pshift, g3, sub_w, sub_hw, st);
        return launch_pw_bwd<64, true>(g, g2, y, scale, wt, gx, gres, M, K, N, relu, xin, pscale, pshift, g3, sub_w, sub_hw, st);
    }
    if (K % 128 == 0)
        return launch_pw_bwd<128, false>(g, g2, y, scale, wt, gx, gres, M, K, N, relu, xin, pscale, pshift, nullptr, 0, 0, st);
    return launch_pw_bwd<64, false>(g, g2, y, scale, wt, gx, gres, M, K, N, relu, xin, pscale, pshift, nullptr, 0, 0, st);
}

extern "C" int adil_pw_conv_fwd(const void* x, const void* w, const float* scale, const float* shift, const void* res,
                                void* y, int M, int K, int N, int relu, const float* pscale, const float* pshift,
                                int sub_w, int sub_hw, void* stream) {
    ADIL_ENTER();
    if (!x || !w || !scale || !shift || !y || M <= 0 || K <= 0 || N <= 0 || (K % PW_BK) || (N % 64)) return ADIL_EINVAL;
    if ((pscale != nullptr) != (pshift != nullptr) || (pscale && K > PW_PK)) return ADIL_EINVAL;
    if (sub_w < 0 || (sub_w > 0 && (sub_hw <= 0 || sub_hw % sub_w || M % sub_hw))) return ADIL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (pscale) {
        if (N % 128 == 0)
            return launch_pw_fwd<128, true>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
        return launch_pw_fwd<64, true>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
    }
    if (N % 128 == 0) return launch_pw_fwd<128, false>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
    return launch_pw_fwd<64, false>(x, w, scale, shift, res, y, M, K, N, relu, pscale, pshift, sub_w, sub_hw, st);
}

// =========================================================================================================== //
// 3x3 / stride 1 / pad 1 convolution of the frozen ResNet (conv2 of every bottleneck), NHWC bf16, raw output:
// its BatchNorm + ReLU live in the next pointwise kernel's prologue, so this kernel is a pure implicit GEMM
//     Y[m][n] = sum_{tap, c} X[m + (kh-1)*W + (kw-1)][c] * Wp[n][tap][c]        (taps crossing an image edge masked)
// and the input gradient is the same kernel on flipped / transposed weights.  Pixels are tiled LINEARLY (128
// consecutive (n,h,w) indices, any H, W): the halo of a tile is the contiguous range [m0-W-1, m0+128+W+1), staged
// once per 64-channel chunk; the fragment of a lane's pixel for tap (kh,kw) is the LDS row pl + kh*W + kw.
// Per tap one weight tile (double buffered, next tile's global loads in flight), ONE barrier, 16 MFMAs per wave on
// 64 px x 64 (32) channel wave tiles.  No zero-fill launch, no epilogue pass.
// =========================================================================================================== //
namespace {

#define C3_LS 72

template <int BN, int WPX>                              // WPX waves along pixels (64 each) x 4/WPX along channels
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv3x3_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, bf16_t* __restrict__ y, int M, int H, int W, int C, int N,
    int MT, int NT) {
    constexpr int C3_BM = WPX * 64;                      // pixels per workgroup: 128 (2 x 2 waves) or 256 (4 x 1 waves)
    constexpr int CTW = BN / 32 / (4 / WPX);             // channel tiles per wave
    constexpr int XCHK = (C3_BM + 128) * 8 / 256;        // halo chunks per thread (NP <= C3_BM + 128: W <= 63)
    constexpr int WCH = BN * 8 / 256;                    // 16-byte chunks of a weight tile per thread
    constexpr int OS = BN + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NP = C3_BM + 2 * W + 2;                    // halo pixels
    bf16_t* sx = reinterpret_cast<bf16_t*>(smem_raw);    // [NP][C3_LS]  (later: [128][OS] output transpose)
    const int sx_elems = (NP * C3_LS > C3_BM * OS ? NP * C3_LS : C3_BM * OS);
    bf16_t* sw = sx + ((sx_elems + 7) & ~7);             // [2][BN][C3_LS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int wpx = w % WPX, wch = w / WPX;
    int mt, nt;
    if ((MT & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        nt = j % NT;
        mt = (j / NT) * 8 + xcd;
    } else {
        nt = blockIdx.x % NT;
        mt = blockIdx.x / NT;
    }
    const int m0 = mt * C3_BM, n0 = nt * BN;
    const int nci = C >> 6;

    // this lane's two pixels and the validity of their 9 taps
    int pl[2];
    unsigned vmask[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        pl[p] = (wpx * 2 + p) * 32 + c;
        const int m = m0 + pl[p];
        const int ww = m % W, hh = (m / W) % H;
        unsigned vm = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - 3 * kh;
            const bool ok = (m < M) && (hh + kh - 1 >= 0) && (hh + kh - 1 < H) && (ww + kw - 1 >= 0) && (ww + kw - 1 < W);
            vm |= ok ? (1u << t) : 0u;
        }
        vmask[p] = vm;
    }

    u32x4 xr[XCHK], wr[3][WCH];                           // weight tiles are requested THREE taps ahead (an L2 round trip
    auto load_x = [&](int cc) {                          // is ~5x the 16 MFMAs of one tap)
#pragma unroll
        for (int i = 0; i < XCHK; ++i) {
            const int q = tid + 256 * i, px = q >> 3, ch = q & 7;
            int gm = m0 - W - 1 + (px < NP ? px : NP - 1);
            gm = gm < 0 ? 0 : (gm >= M ? M - 1 : gm);
            xr[i] = *reinterpret_cast<const u32x4*>(x + (size_t)gm * C + cc * 64 + ch * 8);
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < XCHK; ++i) {
            const int q = tid + 256 * i, px = q >> 3, ch = q & 7;
            if (px < NP) *reinterpret_cast<u32x4*>(sx + px * C3_LS + ch * 8) = xr[i];
        }
    };
    const int nit = nci * 9;
    auto load_w = [&](int it, u32x4 (&r)[WCH]) {         // it = cc * 9 + tap (clamped: the surplus loads are never stored)
        const int itc = it < nit ? it : nit - 1;
        const int cc = itc / 9, tap = itc - 9 * cc;
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, ch = q & 7;
            r[i] = *reinterpret_cast<const u32x4*>(wp + ((size_t)(n0 + row) * 9 + tap) * C + cc * 64 + ch * 8);
        }
    };
    auto store_w = [&](int buf, const u32x4 (&r)[WCH]) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int q = tid + 256 * i, row = q >> 3, ch = q & 7;
            *reinterpret_cast<u32x4*>(sw + (buf * BN + row) * C3_LS + ch * 8) = r[i];
        }
    };

    f32x16 acc[2][CTW];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][ct][r] = 0.0f;

    auto tap_body = [&](int it, u32x4 (&rfree)[WCH], const u32x4 (&rnext)[WCH]) {
        // on entry: sw[it & 1] holds tile `it`, rnext holds tile it+1, the third register set holds tile it+2 (in
        // flight), rfree's tile is already in LDS
        const int buf = it & 1;
        const int cc = it / 9, tap = it - 9 * cc;
        const int kh = tap / 3, kw = tap - 3 * kh;
        load_w(it + 3, rfree);
        if (tap == 4 && cc + 1 < nci) load_x(cc + 1);                 // next channel chunk's halo flies under taps 4..8
        const bf16_t* bw = sw + (buf * BN + (wch * CTW) * 32 + c) * C3_LS + 8 * h;
        const int shift = kh * W + kw;
        const bool ok0 = (vmask[0] >> tap) & 1u, ok1 = (vmask[1] >> tap) & 1u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 b0 = lds8(sx + (pl[0] + shift) * C3_LS + 16 * ks + 8 * h);
            bf16x8 b1 = lds8(sx + (pl[1] + shift) * C3_LS + 16 * ks + 8 * h);
            u32x4 z0 = __builtin_bit_cast(u32x4, b0), z1 = __builtin_bit_cast(u32x4, b1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { z0[j] = ok0 ? z0[j] : 0u; z1[j] = ok1 ? z1[j] : 0u; }
            b0 = __builtin_bit_cast(bf16x8, z0);
            b1 = __builtin_bit_cast(bf16x8, z1);
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                const bf16x8 a = lds8(bw + ct * 32 * C3_LS + 16 * ks);
                mma16(acc[0][ct], a, b0);
                mma16(acc[1][ct], a, b1);
            }
        }
        if (it + 1 < nit) store_w(buf ^ 1, rnext);
        if (tap == 8 && cc + 1 < nci) {                               // all waves are done with this halo after the barrier
            lds_barrier();
            store_x();
        }
        lds_barrier();
    };

    load_x(0);
    load_w(0, wr[0]);
    load_w(1, wr[1]);
    load_w(2, wr[2]);
    store_x();
    store_w(0, wr[0]);
    __syncthreads();
    for (int it = 0; it < nit; it += 3) {                             // nit = 9 * nci: a multiple of 3
        tap_body(it, wr[0], wr[1]);
        tap_body(it + 1, wr[1], wr[2]);
        tap_body(it + 2, wr[2], wr[0]);
    }
    // epilogue: transpose through LDS (all waves share one [128][OS] tile), 16-byte NHWC stores
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 t;
                t[0] = pack2_bf16(acc[p][ct][4 * q], acc[p][ct][4 * q + 1]);
                t[1] = pack2_bf16(acc[p][ct][4 * q + 2], acc[p][ct][4 * q + 3]);
                *reinterpret_cast<u32x2*>(sx + pl[p] * OS + (wch * CTW + ct) * 32 + 8 * q + 4 * h) = t;
            }
        }
    }
    __syncthreads();
    constexpr int CPP = BN / 8;
#pragma unroll
    for (int i = 0; i < C3_BM * CPP / 256; ++i) {
        const int id = tid + 256 * i, px = id / CPP, ch = id - px * CPP;
        const int mm = m0 + px;
        if (mm < M) *reinterpret_cast<u32x4*>(y + (size_t)mm * N + n0 + ch * 8) = *reinterpret_cast<const u32x4*>(sx + px * OS + ch * 8);
    }
}

template <int BN, int WPX>
int launch_conv3x3(const void* x, const void* wp, void* y, int M, int H, int W, int C, int N, hipStream_t st) {
    constexpr int C3_BM = WPX * 64;
    const int MT = (M + C3_BM - 1) / C3_BM, NT = N / BN;
    const int NP = C3_BM + 2 * W + 2;
    size_t sx_elems = (size_t)NP * C3_LS;
    if (sx_elems < (size_t)C3_BM * (BN + 8)) sx_elems = (size_t)C3_BM * (BN + 8);
    sx_elems = (sx_elems + 7) & ~(size_t)7;
    const size_t lds = (sx_elems + (size_t)2 * BN * C3_LS) * sizeof(bf16_t);
    if (lds > 160 * 1024) return ADIL_EINVAL;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv3x3_kernel<BN, WPX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((conv3x3_kernel<BN, WPX>), dim3((unsigned)(MT * NT)), dim3(256), lds, st, (const bf16_t*)x,
                       (const bf16_t*)wp, (bf16_t*)y, M, H, W, C, N, MT, NT);
    ADIL_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int adil_conv3x3(const void* x, const void* wp, void* y, int B, int H, int W, int C, int N, void* stream) {
    ADIL_ENTER();
    if (!x || !wp || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || (C % 64) || (N % 64) || W > 63) return ADIL_EINVAL;   // halo = 128 + 2W + 2 pixels <= 256
    const long long M = (long long)B * H * W;
    if (M > 0x7fffffffLL) return ADIL_EINVAL;
    if (N % 128 == 0) return launch_conv3x3<128, 2>(x, wp, y, (int)M, H, W, C, N, (hipStream_t)stream);
    return launch_conv3x3<64, 4>(x, wp, y, (int)M, H, W, C, N, (hipStream_t)stream);   // 64 px x 64 ch wave tiles as well
}
