// Shared device helpers for the ADiL gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adil_hip.h"

#define ADIL_MAX_ATOMS 128
#define ADIL_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

// two floats -> one dword of packed bf16 (lo = a, hi = b), round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950
// (converting element-wise and OR-ing the halves costs six VALU instructions per pair)
typedef float adil_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 adil_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
    const adil_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, adil_bf16x2));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(const float* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void store(float* p, size_t i, float v) { p[i] = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
    static __device__ __forceinline__ void store(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// max of non-negative floats through their (order-preserving) integer bit pattern
__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
}

// --------------------------------------------------------------------------- //
// AdamW (torch.optim.AdamW single-tensor semantics) on one element
// --------------------------------------------------------------------------- //
struct AdamWHyper {
    float decay;      // 1 - lr*wd
    float b1, b2;     // betas
    float eps;
    float step_size;  // lr / (1 - b1^t)
    float bc2_sqrt;   // sqrt(1 - b2^t)
};

__device__ __forceinline__ float adamw_elem(float p, float g, float& m, float& s, const AdamWHyper& h) {
    // No FMA contraction in here: the function is inlined into many kernels (and template instantiations of one kernel), and
    // hipcc's default -ffp-contract=fast fused `s*b2 + (1-b2)*g*g` one way in one of them and another way in the next — the
    // "same" update then differed in the last bit between two launch routes (round 4: a four-wave variant of
    // adamw_l1ball_kernel against the one-wave form — the variant summed its slabs in one memory round trip instead of
    // eight and measured no faster, 10.2 vs 10.5 us: the slab reads are bound by their 256-byte access pattern, not by
    // the round trips; it was removed, this guard stays).  Every product and sum below is rounded on its own, as torch's
    // AdamW does on the host.
#pragma clang fp contract(off)
    p *= h.decay;
    m = m + (1.0f - h.b1) * (g - m);                 // exp_avg.lerp_(grad, 1-b1)
    s = s * h.b2 + (1.0f - h.b2) * g * g;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    float denom = sqrtf(s) / h.bc2_sqrt + h.eps;
    return p - h.step_size * (m / denom);
}


// --------------------------------------------------------------------------- //
// Sum of one entry over the per-workgroup partial-sum slabs of a grad_v pass, in a FIXED order: slab s goes to
// accumulator s % 32, the 32 accumulators meet in a pairwise tree.  Bitwise reproducible, and the same function serves
// the stand-alone reduce kernel and the consumers that fold the reduction into their own launch (adamw_l1ball,
// pack_codes), so both routes give identical bits.  p = address of the entry in slab 0, stride = floats per slab.
// 32 independent loads are in flight per lane; the tail uses clamped addresses and 0/1 weights, never a branch around
// a load (hipcc would serialise them, FINDINGS.md 2).
// --------------------------------------------------------------------------- //
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, int nslabs, size_t stride) {
    float acc[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) acc[u] = 0.0f;
    int s0 = 0;
    for (; s0 + 32 <= nslabs; s0 += 32) {
        float t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = p[(size_t)(s0 + u) * stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u] += t[u];
    }
    if (s0 < nslabs) {
        float t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int s = s0 + u;
            t[u] = p[(size_t)(s < nslabs ? s : nslabs - 1) * stride];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u] += ((s0 + u < nslabs) ? 1.0f : 0.0f) * t[u];
    }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    }
    return acc[0];
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Compute units of the current device (grid sizing, per-workgroup scratch), asked once per device.  256 (MI355X) when
// no device can be queried — the workspace-size entry points are also called on boxes without a GPU (build check).
static inline int adil_num_cu() {
    static int cached[32] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) { (void)hipGetLastError(); return 256; }
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            n = 256;
        }
        cached[dev] = n;
    }
    return cached[dev];
}

// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated HIP call (e.g. PyTorch's lazy device
// probing) left behind so that ADIL_CHECK_LAUNCH reports only our own launch failures.
#define ADIL_ENTER() (void)hipGetLastError()

#define ADIL_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)
