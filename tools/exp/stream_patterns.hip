// What does the ACCESS PATTERN of the bf16 synth cost?  (round 4: AdamW(D) streams 6.3 TB/s and the fp32 z-step 5.8 TB/s at
// the same read / write mix at which the bf16 synth stays at 5.0 TB/s; the z-step moves 16 bytes per lane and 512 contiguous
// bytes per image row, the bf16 synth 8 bytes per lane and 256 bytes per row.)  Pure streams out = x + 1 on (B x P) bf16 images,
// B = 512, P = 150528, no dictionary and no MFMA, same workgroup shape / sweep order / residency as synth_mfma_kernel (4 waves,
// LDS sized for 4 workgroups per CU, wave w takes 32-row blocks w, w+4, ...; all loads of a block in flight, then all stores):
//   v0  lane = (32 columns x 2 rows), 8 B per lane, 16 loads: 128-pixel tile, 256 B per row   (the kernel as it is)
//   v1  lane = (16 columns x 4 rows), 16 B per lane, 8 loads: 128-pixel tile, 256 B per row   (16x16 MFMA tile shape)
//   v2  lane = (32 columns x 2 rows), 16 B per lane, 16 loads: 256-pixel tile, 512 B per row, batch halves on blockIdx.y
//   v3  lane = (64 columns x 1 row),  16 B per lane, 16 loads: 512-pixel tile, 1 KB per row, batch quarters on blockIdx.y
//   lin grid-stride 16 B per lane over the flat buffer (what AdamW's pattern looks like)
// Buffers rotate (NBUF pairs, 1.8 GB) so that no launch finds its data in the 256 MB Infinity Cache.
// hipcc --offload-arch=gfx950 -O3 tools/exp/stream_patterns.hip -o tools/exp/bin/stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned short bf16_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc block_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned bump(unsigned v) {        // two bf16 -> +1.0 each -> two bf16 (truncating; the point is the stream)
    const float a = __uint_as_float(v << 16) + 1.0f, b = __uint_as_float(v & 0xffff0000u) + 1.0f;
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}

// VB = bytes per lane (8 / 16), CL = lanes along the row (16 / 32 / 64), NL = loads per 32-row block
template <int VB, int CL>
__global__ __launch_bounds__(256) void sweep(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int P, int rows_per_y) {
    constexpr int RL = 64 / CL;                  // rows per instruction
    constexpr int NL = 32 / RL;                  // instructions per 32-row block
    constexpr int TW = CL * VB / 2;              // pixels per tile
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane % CL, r = lane / CL;
    if (threadIdx.x == 0) lds[0] = 1;
    const unsigned rowb = (unsigned)P * 2u;
    const int p0 = blockIdx.x * TW;
    const int voff = (int)((unsigned)r * rowb) + (p0 + c * (VB / 2)) * 2;
    const int row_lo = blockIdx.y * rows_per_y, row_hi = min(B, row_lo + rows_per_y);
    for (int b0 = row_lo + 32 * w; b0 < row_hi; b0 += 128) {
        const int rows = row_hi - b0 < 32 ? row_hi - b0 : 32;
        const buf_rsrc rx = block_rsrc(x + (size_t)b0 * P, (unsigned)rows * rowb);
        const buf_rsrc ro = block_rsrc(out + (size_t)b0 * P, (unsigned)rows * rowb);
        if constexpr (VB == 8) {
            u32x2 v[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(rx, voff, (int)((unsigned)(i * RL) * rowb), 0);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                u32x2 t = v[i];
                t[0] = bump(t[0]); t[1] = bump(t[1]);
                __builtin_amdgcn_raw_buffer_store_b64(t, ro, voff, (int)((unsigned)(i * RL) * rowb), 0);
            }
        } else {
            u32x4 v[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (int)((unsigned)(i * RL) * rowb), 0);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                u32x4 t = v[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = bump(t[e]);
                __builtin_amdgcn_raw_buffer_store_b128(t, ro, voff, (int)((unsigned)(i * RL) * rowb), 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

__global__ __launch_bounds__(256) void linear(const u32x4* __restrict__ x, u32x4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; v[u] = x[j < n ? j : n - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + u * stride;
            u32x4 t = v[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = bump(t[e]);
            if (j < n) out[j] = t;
        }
    }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int REP = argc > 1 ? atoi(argv[1]) : 36, B = argc > 2 ? atoi(argv[2]) : 512, P = 150528;
    const int NBUF = B <= 512 ? 6 : 3;            // >= 1.8 GB in rotation either way
    const size_t n = (size_t)B * P, bytes = n * 2;
    std::vector<bf16_t*> xs(NBUF), os(NBUF);
    for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&xs[i], bytes)); CK(hipMalloc(&os[i], bytes)); CK(hipMemset(xs[i], 0x3c, bytes)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 36 * 1024;                                   // 4 workgroups per CU, as the synth kernel's 112 registers allow
    CK(hipFuncSetAttribute((const void*)sweep<8, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)sweep<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)sweep<16, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)sweep<16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < NBUF; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < REP; ++i) launch(i % NBUF);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / REP;
        printf("| %-58s | %7.1f us | %5.2f TB/s |\n", name, us, 2.0 * bytes / (us * 1e-6) / 1e12);
        fflush(stdout);
    };
    if (argc > 3) {
        // does the RELATIVE placement of the output batch matter (DRAM bank / channel interleaving of a read stream and a write
        // stream that advance together)?  One arena per pair; out = x + bytes + delta.
        for (int i = 0; i < NBUF; ++i) { CK(hipFree(xs[i])); CK(hipFree(os[i])); }
        const size_t slack = 64u << 20;
        std::vector<char*> arena(NBUF);
        for (int i = 0; i < NBUF; ++i) { CK(hipMalloc(&arena[i], 2 * bytes + slack)); CK(hipMemset(arena[i], 0x3c, 2 * bytes + slack)); }
        printf("| out = x + %zu B + delta | synth pattern (v0) | flat copy |\n|---|---|---|\n", bytes);
        const size_t deltas[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576,
                                 2097152, 4194304, 8388608, 16777216, 33554432};
        for (size_t dl : deltas) {
            double us[2];
            for (int k = 0; k < 2; ++k) {
                auto launch = [&](int i) {
                    const bf16_t* x = (const bf16_t*)arena[i];
                    bf16_t* o = (bf16_t*)(arena[i] + bytes + dl);
                    if (k == 0) hipLaunchKernelGGL((sweep<8, 32>), dim3(P / 128, 1), dim3(256), lds, 0, x, o, B, P, B);
                    else hipLaunchKernelGGL(linear, dim3(2048), dim3(256), 0, 0, (const u32x4*)x, (u32x4*)o, n / 8);
                };
                for (int i = 0; i < NBUF; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < REP; ++i) launch(i % NBUF);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[k] = ms * 1e3 / REP;
            }
            printf("| %9zu | %6.1f us = %4.2f TB/s | %6.1f us = %4.2f TB/s |\n", dl, us[0], 2.0 * bytes / us[0] / 1e6, us[1], 2.0 * bytes / us[1] / 1e6);
            fflush(stdout);
        }
        return 0;
    }
    printf("| pattern (B = %d x P = 150528 bf16, out = x + 1, %.0f MB in + %.0f MB out) | time | bytes / time |\n|---|---|---|\n", B, bytes / 1e6, bytes / 1e6);
    for (int pass = 0; pass < 2; ++pass) {
        run("v0  8 B/lane, 256 B rows, 2 rows/instr, tile 128 (now)", [&](int i) {
            hipLaunchKernelGGL((sweep<8, 32>), dim3(P / 128, 1), dim3(256), lds, 0, xs[i], os[i], B, P, B); });
        run("v1 16 B/lane, 256 B rows, 4 rows/instr, tile 128", [&](int i) {
            hipLaunchKernelGGL((sweep<16, 16>), dim3(P / 128, 1), dim3(256), lds, 0, xs[i], os[i], B, P, B); });
        run("v2 16 B/lane, 512 B rows, 2 rows/instr, tile 256, 2 row halves", [&](int i) {
            hipLaunchKernelGGL((sweep<16, 32>), dim3(P / 256, 2), dim3(256), lds, 0, xs[i], os[i], B, P, B / 2); });
        run("v2' the same, whole batch per workgroup (588 workgroups)", [&](int i) {
            hipLaunchKernelGGL((sweep<16, 32>), dim3(P / 256, 1), dim3(256), lds, 0, xs[i], os[i], B, P, B); });
        run("v3 16 B/lane, 1 KB rows, 1 row/instr, tile 512, 4 row quarters", [&](int i) {
            hipLaunchKernelGGL((sweep<16, 64>), dim3(P / 512, 4), dim3(256), lds, 0, xs[i], os[i], B, P, B / 4); });
        run("v0' as v0, 2 row halves (2352 workgroups)", [&](int i) {
            hipLaunchKernelGGL((sweep<8, 32>), dim3(P / 128, 2), dim3(256), lds, 0, xs[i], os[i], B, P, B / 2); });
        run("lin grid-stride 16 B/lane, 2048 workgroups", [&](int i) {
            hipLaunchKernelGGL(linear, dim3(2048), dim3(256), 0, 0, (const u32x4*)xs[i], (u32x4*)os[i], n / 8); });
    }
    return 0;
}
