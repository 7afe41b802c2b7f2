// adil_mfma.h — small device helpers shared by the frozen-classifier kernels (adil_stem.hip, adil_convs.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adil_common.h"

namespace {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 lds8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void mma16(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void lds_sync() { __syncthreads(); }
// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() carries a fence that hipcc
// lowers to s_waitcnt vmcnt(0), which drains every prefetched tile at every barrier (FINDINGS.md 3a).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void unpack8(const u32x4& t, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(t[i] << 16);
        f[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
    return t;
}

// ---- packed elementwise helpers: two values per VALU instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 on
// fp32 pairs, v_pk_*_i16 on packed bf16).  The GEMM kernels' pro/epilogues were VALU-bound with scalar code: 670 VALU
// instructions per wave against 64 MFMAs on a stage-3 pointwise layer (SQ_INSTS_VALU / SQ_INSTS_MFMA).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bf2_to_f32x2(unsigned u) {
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ unsigned f32x2_to_bf2(f32x2 v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// ReLU on a packed bf16 pair: as int16 a negative float is a negative integer -> one v_pk_max_i16
__device__ __forceinline__ unsigned relu_bf2(unsigned p) {
    const i16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, p), z));
}
// 0xffff per half where the bf16 value is > 0 (y never holds -0.0: it is a ReLU output), else 0
__device__ __forceinline__ unsigned pos_mask_bf2(unsigned y) {
    const i16x2 z = {0, 0};
    const i16x2 s = z - __builtin_bit_cast(i16x2, y);
    return __builtin_bit_cast(unsigned, s >> 15);
}

}  // namespace
