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
    p *= h.decay;
    m = m + (1.0f - h.b1) * (g - m);                 // exp_avg.lerp_(grad, 1-b1)
    s = s * h.b2 + (1.0f - h.b2) * g * g;            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    float denom = sqrtf(s) / h.bc2_sqrt + h.eps;
    return p - h.step_size * (m / denom);
}


static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated HIP call (e.g. PyTorch's lazy device
// probing) left behind so that ADIL_CHECK_LAUNCH reports only our own launch failures.
#define ADIL_ENTER() (void)hipGetLastError()

#define ADIL_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)
