// ADiL dictionary contractions for gfx950 on the matrix cores.
//   synth :  out = x + vp D^T                 (B x Kp)(Kp x P), fused add / clamps        [K1, K9]
//   grad  :  grad_d = g^T vp (P x K)  and  grad_vb = g D (B x K) in ONE pass over g       [K2, K3, K6]
// D is P x K row-major (atom innermost), image streams are B x P row-major (fp32 or bf16).
//
// Both kernels are HBM-bound streaming kernels; MFMA is used only so that the small dense contractions do not
// become ALU-bound.  One code path per stream type, selected by Mma<T>:
//   T = float  : v_mfma_f32_32x32x2_f32  (exact fp32 fmaf chain; the parity path)
//   T = bf16   : v_mfma_f32_32x32x16_bf16 (D and V rounded to bf16 on the fly, fp32 accumulate; the throughput path)
// Every operand fragment covers a "k-group" of 16 reduction indices: lane (r = lane&31, h = lane>>5) supplies the
// 8 consecutive indices 16g + 8h + j, j = 0..7, of row/column r — identical for both instruction shapes, so the
// data movement is shared and only Mma<T>::mma differs (8 f32 MFMAs vs 1 bf16 MFMA per k-group).
// C/D layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "adil_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma;

template <> struct Mma<float> {
    struct Frag { float v[8]; };
    typedef float Elem;                         // element type of operands staged in LDS / workspace
    static constexpr int PAD = 4;               // LDS row padding (elements): (stride/4) odd -> conflict-free b128
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag from8(const float (&f)[8]) {
        Frag r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[j] = f[j];
        return r;
    }
    static __device__ __forceinline__ Elem to_elem(float x) { return x; }
    static __device__ __forceinline__ Frag load8(const Elem* p) {      // 8 consecutive elements, 16-B aligned
        const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
        Frag r;
        r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
        r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
        return r;
    }
};

template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    typedef bf16_t Elem;
    static constexpr int PAD = 8;               // (stride*2/16) odd
    static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag from8(const float (&f)[8]) {
        Frag r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (__bf16)f[j];
        return r;
    }
    static __device__ __forceinline__ Elem to_elem(float x) { return f32_to_bf16(x); }
    static __device__ __forceinline__ Frag load8(const Elem* p) { return *reinterpret_cast<const Frag*>(p); }
};

// ---- image-stream access: N consecutive pixels of one row as floats ---------------------------------------- //
template <typename T, int N> struct PixVec;
template <> struct PixVec<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct PixVec<bf16_t, 4> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[4]) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
        o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[4]) {
        uint2 t;
        t.x = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
        t.y = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
};
template <> struct PixVec<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p); o[0] = t.x; o[1] = t.y;
    }
};
template <> struct PixVec<bf16_t, 2> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&o)[2]) {
        const unsigned t = *reinterpret_cast<const unsigned*>(p);
        o[0] = __uint_as_float(t << 16); o[1] = __uint_as_float(t & 0xffff0000u);
    }
};
template <typename T> struct PixVec<T, 1> {
    static __device__ __forceinline__ void load(const T* p, float (&o)[1]) { o[0] = Elem<T>::load(p, 0); }
};

// N pixels starting at p[0]; `nvalid` of them exist (tile tail); vec = rows are aligned for the vector form
template <typename T, int N>
__device__ __forceinline__ void load_px(const T* p, int nvalid, bool vec, float (&o)[N]) {
    if (vec && nvalid >= N) {
        PixVec<T, N>::load(p, o);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = (i < nvalid) ? Elem<T>::load(p, i) : 0.0f;
    }
}
template <typename T>
__device__ __forceinline__ void store_px4(T* p, int nvalid, bool vec, const float (&o)[4]) {
    if (vec && nvalid >= 4) {
        PixVec<T, 4>::store(p, o);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nvalid) Elem<T>::store(p, i, o[i]);
    }
}
// 8 consecutive pixels of one row as an MFMA fragment (the A operand of grad_v)
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag load_frag8(const T* p, int nvalid, bool vec) {
    if (vec && nvalid >= 8) {
        return Mma<T>::load8(reinterpret_cast<const typename Mma<T>::Elem*>(p));   // stream type == Elem type
    }
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (i < nvalid) ? Elem<T>::load(p, i) : 0.0f;
    return Mma<T>::from8(f);
}

// 8 consecutive fp32 values (16-B aligned, e.g. a row of the packed codes) as an MFMA fragment
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag frag_from_f32x8(const float* p) {
    const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
    const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return Mma<T>::from8(f);
}

__device__ __forceinline__ int c_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// =========================================================================================================== //
// K1 (+K9) synthesis.  Workgroup = 4 waves = one 128-pixel slice of D (converted + staged once in LDS) swept
// over the whole batch; wave w takes batch blocks w, w+4, ...  Per 32x128 output block a wave issues 16 vector
// loads of x straight into the accumulators (C operand), Kp/16 k-groups of MFMAs against the LDS-resident D
// slice, and 16 vector stores.  Lane c of pixel tile t owns pixel p0 + 4c + t, so that one lane's four tiles
// are 4 CONSECUTIVE pixels: 16-byte (fp32) / 8-byte (bf16) accesses, 512 / 256 contiguous bytes per row.
// LDS row of pixel r: (r&3)*32 + (r>>2)  (tile-major), row stride Kp + PAD elements.
// =========================================================================================================== //
#define SYNTH_TILE 128

template <typename T, bool XACC>
__global__ __launch_bounds__(256) void synth_mfma_kernel(const T* __restrict__ x, const float* __restrict__ d,
                                                         const float* __restrict__ vp, T* __restrict__ out, int B,
                                                         int P, int K, int Kp, float delta_clamp, int pixel_clamp,
                                                         int vec) {
    using M = Mma<T>;
    using E = typename M::Elem;
    using Frag = typename M::Frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* sd = reinterpret_cast<E*>(smem_raw);
    const int Ks = Kp + M::PAD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int p0 = blockIdx.x * SYNTH_TILE;
    for (int i = tid; i < SYNTH_TILE * Kp; i += 256) {
        const int r = i / Kp, k = i - r * Kp;
        const int p = p0 + r;
        const float val = (p < P && k < K) ? d[(size_t)p * K + k] : 0.0f;
        sd[((r & 3) * 32 + (r >> 2)) * Ks + k] = M::to_elem(val);
    }
    __syncthreads();
    const int NG = Kp >> 4;
    const int nbb = (B + 31) >> 5;
    const int px = p0 + 4 * c;
    const int nvalid = max(0, min(4, P - px));
    const bool v4 = vec != 0;
    for (int bb = w; bb < nbb; bb += 4) {
        const int b0 = bb << 5;
        f32x16 acc[4];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = b0 + c_row(reg, h);
            float xv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (XACC && row < B && nvalid > 0) load_px<T, 4>(x + (size_t)row * P + px, nvalid, v4, xv);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t][reg] = xv[t];
        }
        const float* arow = vp + (size_t)(b0 + c) * Kp + 8 * h;
        Frag a = frag_from_f32x8<T>(arow);
        for (int g = 0; g < NG; ++g) {
            Frag an = a;
            if (g + 1 < NG) an = frag_from_f32x8<T>(arow + 16 * (g + 1));      // prefetch the next k-group's codes
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const Frag bf = M::load8(sd + (t * 32 + c) * Ks + 16 * g + 8 * h);
                M::mma(acc[t], a, bf);
            }
            a = an;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = b0 + c_row(reg, h);
            if (row < B && nvalid > 0) {
                float r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = acc[t][reg];
                if (!XACC) {
                    if (delta_clamp >= 0.0f) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) r[t] = fminf(fmaxf(r[t], -delta_clamp), delta_clamp);
                    }
                    if (x != nullptr) {
                        float xv[4];
                        load_px<T, 4>(x + (size_t)row * P + px, nvalid, v4, xv);
#pragma unroll
                        for (int t = 0; t < 4; ++t) r[t] += xv[t];
                    }
                }
                if (pixel_clamp) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) r[t] = fminf(fmaxf(r[t], 0.0f), 1.0f);
                }
                store_px4<T>(out + (size_t)row * P + px, nvalid, v4, r);
            }
        }
    }
}

// =========================================================================================================== //
// K2 + K3 fused: one pass over g.  Workgroup = 8 waves (2 per SIMD); each WAVE owns pixel tiles of PXT*32
// pixels and sweeps ALL batch blocks for them:
//   grad_d tile (PXT*32 px x AT*32 atoms) lives in the wave's accumulators for the whole sweep (A = g^T read in
//     the coalesced "lane = pixel, 8 batch rows per lane" layout, B = codes from the transposed packed matrix);
//   grad_v partials (32 rows x AT*32 atoms per batch block) come from the same g block re-read in the
//     "lane = batch row, 8 pixels per lane" layout (L1/L2 hits: the wave touched those lines a moment ago),
//     B = this tile's D fragments held in registers; they are accumulated across tiles and waves in an LDS
//     array [Bp][AT*32] with ds_add_f32 and flushed once per workgroup to a slab that a small kernel reduces.
// =========================================================================================================== //
template <typename T, int PXT, int AT, bool WD, bool WV>
__global__ __launch_bounds__(512) void grad_mfma_kernel(const T* __restrict__ g, const float* __restrict__ d,
                                                        const typename Mma<T>::Elem* __restrict__ vpt,
                                                        int vstride, float* __restrict__ grad_d,
                                                        float* __restrict__ slab, int B, int Bp, int P, int K,
                                                        int accumulate_d, int vec, int ntiles) {
    using M = Mma<T>;
    using Frag = typename M::Frag;
    constexpr int KA = AT * 32;
    constexpr int TW = PXT * 32;
    constexpr int NG3 = TW / 16;
    extern __shared__ __attribute__((aligned(16))) float sacc[];          // [Bp][KA], WV only
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    if (WV) {
        for (int i = tid; i < Bp * KA; i += 512) sacc[i] = 0.0f;
        __syncthreads();
    }
    const bool vok = vec != 0;
    const int nbb = Bp >> 5;
    const int nwaves = gridDim.x * 8;
    for (int tile = blockIdx.x * 8 + w; tile < ntiles; tile += nwaves) {
        const int p0 = tile * TW;
        Frag dfr[NG3][AT];
        if (WV) {
#pragma unroll
            for (int g3 = 0; g3 < NG3; ++g3) {
#pragma unroll
                for (int at = 0; at < AT; ++at) {
                    float f[8];
                    const int atom = at * 32 + c;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pix = p0 + 16 * g3 + 8 * h + j;
                        f[j] = (pix < P && atom < K) ? d[(size_t)pix * K + atom] : 0.0f;
                    }
                    dfr[g3][at] = M::from8(f);
                }
            }
        }
        f32x16 accd[PXT][AT];
        if (WD) {
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int at = 0; at < AT; ++at)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accd[t][at][r] = 0.0f;
        }
        const int px2 = p0 + PXT * c;                            // layout 2: this lane's PXT pixels
        const int nv2 = max(0, min(PXT, P - px2));
        for (int bb = 0; bb < nbb; ++bb) {
            const int b0 = bb << 5;
            if (WD) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    float raw[8][PXT];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = b0 + 16 * g2 + 8 * h + j;
#pragma unroll
                        for (int t = 0; t < PXT; ++t) raw[j][t] = 0.0f;
                        if (row < B && nv2 > 0) load_px<T, PXT>(g + (size_t)row * P + px2, nv2, vok, raw[j]);
                    }
                    Frag afr[PXT];
#pragma unroll
                    for (int t = 0; t < PXT; ++t) {
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = raw[j][t];
                        afr[t] = M::from8(f);
                    }
#pragma unroll
                    for (int at = 0; at < AT; ++at) {
                        const Frag bfr = M::load8(vpt + (size_t)(at * 32 + c) * vstride + b0 + 16 * g2 + 8 * h);
#pragma unroll
                        for (int t = 0; t < PXT; ++t) M::mma(accd[t][at], afr[t], bfr);
                    }
                }
            }
            if (WV) {
                f32x16 accv[AT];
#pragma unroll
                for (int at = 0; at < AT; ++at)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accv[at][r] = 0.0f;
                const int row = b0 + c;
#pragma unroll
                for (int g3 = 0; g3 < NG3; ++g3) {
                    const int px3 = p0 + 16 * g3 + 8 * h;
                    const int nv3 = (row < B) ? max(0, min(8, P - px3)) : 0;
                    Frag a3;
                    if (nv3 > 0) {
                        a3 = load_frag8<T>(g + (size_t)row * P + px3, nv3, vok);
                    } else {
                        const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        a3 = M::from8(z);
                    }
#pragma unroll
                    for (int at = 0; at < AT; ++at) M::mma(accv[at], a3, dfr[g3][at]);
                }
#pragma unroll
                for (int at = 0; at < AT; ++at)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        atomicAdd(&sacc[(b0 + c_row(r, h)) * KA + at * 32 + c], accv[at][r]);
            }
        }
        if (WD) {
#pragma unroll
            for (int t = 0; t < PXT; ++t)
#pragma unroll
                for (int at = 0; at < AT; ++at) {
                    const int atom = at * 32 + c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pix = p0 + PXT * c_row(r, h) + t;
                        if (pix < P && atom < K) {
                            const size_t o = (size_t)pix * K + atom;
                            grad_d[o] = accumulate_d ? grad_d[o] + accd[t][at][r] : accd[t][at][r];
                        }
                    }
                }
        }
    }
    if (WV) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(slab + (size_t)blockIdx.x * Bp * KA);
        const float4* src = reinterpret_cast<const float4*>(sacc);
        for (int i = tid; i < Bp * KA / 4; i += 512) dst[i] = src[i];
    }
}

// codes transposed + converted to the MFMA element type: vpt[a][b] = vp[b][a]  (a < KA, b < Bp)
template <typename E>
__global__ __launch_bounds__(256) void transpose_codes_kernel(const float* __restrict__ vp, int Bp, int Kp, int KA,
                                                              E* __restrict__ vpt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KA * Bp) return;
    const int a = i / Bp, b = i - a * Bp;
    const float v = (a < Kp) ? vp[(size_t)b * Kp + a] : 0.0f;
    if constexpr (sizeof(E) == 4) vpt[i] = v; else vpt[i] = f32_to_bf16(v);
}

__global__ __launch_bounds__(256) void grad_v_reduce_kernel(const float* __restrict__ slab, int nslabs, int Bp, int KA,
                                                            int B, int K, float* __restrict__ grad_vb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    float acc = 0.0f;
    for (int s = 0; s < nslabs; ++s) acc += slab[((size_t)s * Bp + b) * KA + k];
    grad_vb[i] = acc;
}

// =========================================================================================================== //
// C ABI
// =========================================================================================================== //
static const int kNumCU = 256;                     // MI355X
static const size_t kLdsAccumBytes = 128 * 1024;   // LDS budget of the grad_v accumulator (of 160 KiB per CU)

static int set_lds(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

static inline int atom_tiles(int K) { return (K + 31) / 32; }
static inline int grad_at(int K) { const int a = atom_tiles(K); return a <= 2 ? a : 4; }       // instantiated: 1, 2, 4
static inline int grad_pxt(int K) { return grad_at(K) <= 2 ? 2 : 1; }
static inline int grad_chunk_rows(int B, int K) {                                              // batch rows per launch
    const int KA = grad_at(K) * 32;
    int rows = (int)(kLdsAccumBytes / (KA * sizeof(float))) / 32 * 32;
    const int Bp = round_up(B, 32);
    return rows < Bp ? rows : Bp;
}
static inline int grad_num_wgs(int P, int K) {
    const int ntiles = (P + grad_pxt(K) * 32 - 1) / (grad_pxt(K) * 32);
    int wgs = (ntiles + 7) / 8;
    return wgs < kNumCU ? wgs : kNumCU;
}

extern "C" size_t adil_grad_workspace_bytes(int B, int P, int K) {
    const size_t KA = grad_at(K) * 32, Bp = round_up(B, 32);
    const size_t vpt = KA * Bp * sizeof(float);
    const size_t slab = (size_t)grad_num_wgs(P, K) * grad_chunk_rows(B, K) * KA * sizeof(float);
    return ((vpt + 255) / 256) * 256 + slab;
}

template <typename T>
static int launch_synth(const void* x, const float* d, const float* vp, void* out, int B, int P, int K,
                        float delta_clamp, int pixel_clamp, hipStream_t st) {
    using E = typename Mma<T>::Elem;
    const int Kp = round_up(K, 16);
    const size_t lds = (size_t)SYNTH_TILE * (Kp + Mma<T>::PAD) * sizeof(E);
    const int vec = (P % 4 == 0) && (((uintptr_t)out | (uintptr_t)x) % 16 == 0);
    const bool xacc = (x != nullptr) && (delta_clamp < 0.0f);
    const dim3 grid((P + SYNTH_TILE - 1) / SYNTH_TILE), block(256);
    int rc;
    if (xacc) {
        if ((rc = set_lds((const void*)synth_mfma_kernel<T, true>, lds))) return rc;
        hipLaunchKernelGGL((synth_mfma_kernel<T, true>), grid, block, lds, st, (const T*)x, d, vp, (T*)out, B, P, K, Kp,
                           delta_clamp, pixel_clamp, vec);
    } else {
        if ((rc = set_lds((const void*)synth_mfma_kernel<T, false>, lds))) return rc;
        hipLaunchKernelGGL((synth_mfma_kernel<T, false>), grid, block, lds, st, (const T*)x, d, vp, (T*)out, B, P, K, Kp,
                           delta_clamp, pixel_clamp, vec);
    }
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_synth(const void* x, const float* d, const float* vp, void* out, int B, int P, int K, int dtype,
                          float delta_clamp, int pixel_clamp, void* stream) {
    ADIL_ENTER();
    if (!d || !vp || !out || B <= 0 || P <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (dtype == ADIL_F32) return launch_synth<float>(x, d, vp, out, B, P, K, delta_clamp, pixel_clamp, (hipStream_t)stream);
    if (dtype == ADIL_BF16) return launch_synth<bf16_t>(x, d, vp, out, B, P, K, delta_clamp, pixel_clamp, (hipStream_t)stream);
    return ADIL_EINVAL;
}

template <typename T, int PXT, int AT>
static int launch_grad_cfg(const T* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                           int K, int accumulate_d, void* ws, hipStream_t st) {
    using E = typename Mma<T>::Elem;
    constexpr int KA = AT * 32;
    const int Kp = round_up(K, 16), Bp = round_up(B, 32);
    const int ntiles = (P + PXT * 32 - 1) / (PXT * 32);
    const int nwg = grad_num_wgs(P, K);
    E* vpt = reinterpret_cast<E*>(ws);
    float* slab = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ws) + (((size_t)KA * Bp * sizeof(float) + 255) / 256) * 256);
    const int vec = (P % 8 == 0) && ((uintptr_t)g % 16 == 0);
    if (grad_d != nullptr) {
        hipLaunchKernelGGL((transpose_codes_kernel<E>), dim3((KA * Bp + 255) / 256), dim3(256), 0, st, vp, Bp, Kp, KA, vpt);
        ADIL_CHECK_LAUNCH();
    }
    const int chunk = grad_chunk_rows(B, K);
    for (int r0 = 0; r0 < Bp; r0 += chunk) {
        const int rows_p = (Bp - r0 < chunk) ? (Bp - r0) : chunk;           // padded rows in this chunk
        const int rows = (B - r0 < rows_p) ? (B - r0) : rows_p;             // real rows
        const T* gc = g + (size_t)r0 * P;
        const int acc_d = accumulate_d || (r0 > 0);
        int rc;
        // NOTE: vpt is indexed [atom][Bp] with the chunk's column offset r0 folded into the pointer
        if (grad_d != nullptr && grad_vb != nullptr) {
            const size_t lds = (size_t)rows_p * KA * sizeof(float);
            if ((rc = set_lds((const void*)grad_mfma_kernel<T, PXT, AT, true, true>, lds))) return rc;
            hipLaunchKernelGGL((grad_mfma_kernel<T, PXT, AT, true, true>), dim3(nwg), dim3(512), lds, st, gc, d, vpt + r0,
                               Bp, grad_d, slab, rows, rows_p, P, K, acc_d, vec, ntiles);
        } else if (grad_d != nullptr) {
            hipLaunchKernelGGL((grad_mfma_kernel<T, PXT, AT, true, false>), dim3(nwg), dim3(512), 0, st, gc, d, vpt + r0,
                               Bp, grad_d, slab, rows, rows_p, P, K, acc_d, vec, ntiles);
        } else {
            const size_t lds = (size_t)rows_p * KA * sizeof(float);
            if ((rc = set_lds((const void*)grad_mfma_kernel<T, PXT, AT, false, true>, lds))) return rc;
            hipLaunchKernelGGL((grad_mfma_kernel<T, PXT, AT, false, true>), dim3(nwg), dim3(512), lds, st, gc, d, vpt + r0,
                               Bp, grad_d, slab, rows, rows_p, P, K, acc_d, vec, ntiles);
        }
        ADIL_CHECK_LAUNCH();
        if (grad_vb != nullptr) {
            hipLaunchKernelGGL(grad_v_reduce_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, st, (const float*)slab, nwg,
                               rows_p, KA, rows, K, grad_vb + (size_t)r0 * K);
            ADIL_CHECK_LAUNCH();
        }
    }
    return 0;
}

template <typename T>
static int launch_grad(const void* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                       int K, int accumulate_d, void* ws, hipStream_t st) {
    const int at = grad_at(K);
    if (at == 1) return launch_grad_cfg<T, 2, 1>((const T*)g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
    if (at == 2) return launch_grad_cfg<T, 2, 2>((const T*)g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
    return launch_grad_cfg<T, 1, 4>((const T*)g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
}

extern "C" int adil_grad(const void* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                         int K, int dtype, int accumulate_d, void* ws, size_t ws_bytes, void* stream) {
    ADIL_ENTER();
    if (!g || B <= 0 || P <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (grad_d == nullptr && grad_vb == nullptr) return ADIL_EINVAL;
    if (grad_d != nullptr && !vp) return ADIL_EINVAL;
    if (grad_vb != nullptr && !d) return ADIL_EINVAL;
    if (!ws || ws_bytes < adil_grad_workspace_bytes(B, P, K)) return ADIL_EWORKSPACE;
    if (dtype == ADIL_F32) return launch_grad<float>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, (hipStream_t)stream);
    if (dtype == ADIL_BF16) return launch_grad<bf16_t>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, (hipStream_t)stream);
    return ADIL_EINVAL;
}
