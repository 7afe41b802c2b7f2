// How much would an in-launch cross-workgroup reduction of grad_v cost?  (VERDICT r1 #5a asked for it; DESIGN §4 keeps the
// per-workgroup slabs + a reduce kernel.)  Every workgroup of the fused grad kernel ends with a 512 x 50 fp32 partial of
// grad_v; 236 workgroups.  This times ONLY that epilogue, three ways, on the same grid:
//   slab   : each workgroup stores its partial contiguously (what the product does; the 6 us reduce kernel comes on top)
//   i64    : deterministic fixed-point: atomicAdd of int64 into ONE 512 x 50 accumulator (device scope)
//   f32    : float atomicAdd into one accumulator (NOT reproducible; shown for scale)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/exp/atomics_vs_slabs.hip -o /tmp/avs && /tmp/avs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROWS = 512, K = 50, N = ROWS * K, NWG = 236, NT = 512;

__global__ __launch_bounds__(NT) void slab_kernel(float* slab, float seed) {
    float* dst = slab + (size_t)blockIdx.x * N;
    for (int i = threadIdx.x; i < N; i += NT) dst[i] = seed + (float)i;
}
__global__ __launch_bounds__(NT) void i64_kernel(unsigned long long* acc, float seed) {
    for (int i = threadIdx.x; i < N; i += NT) {
        const long long q = (long long)((seed + (float)i) * 1048576.0f);
        atomicAdd(acc + i, (unsigned long long)q);
    }
}
__global__ __launch_bounds__(NT) void f32_kernel(float* acc, float seed) {
    for (int i = threadIdx.x; i < N; i += NT) atomicAdd(acc + i, seed + (float)i);
}
__global__ void reduce_kernel(const float* slab, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float a = 0.0f;
    for (int s = 0; s < NWG; ++s) a += slab[(size_t)s * N + i];
    out[i] = a;
}
template <typename F> static float time_us(F&& launch, int reps = 50) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0f / reps;
}
int main() {
    float *slab, *accf, *out; unsigned long long* acci;
    CHECK(hipMalloc(&slab, (size_t)NWG * N * 4)); CHECK(hipMalloc(&accf, N * 4)); CHECK(hipMalloc(&out, N * 4));
    CHECK(hipMalloc(&acci, N * 8));
    CHECK(hipMemset(accf, 0, N * 4)); CHECK(hipMemset(acci, 0, N * 8));
    printf("epilogue of %d workgroups x %d partial sums (fp32)\n", NWG, N);
    printf("slab stores            %8.1f us\n", time_us([&] { hipLaunchKernelGGL(slab_kernel, dim3(NWG), dim3(NT), 0, 0, slab, 1.0f); }));
    printf("reduce kernel          %8.1f us\n", time_us([&] { hipLaunchKernelGGL(reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, slab, out); }));
    printf("int64 atomics (fixed)  %8.1f us\n", time_us([&] { hipLaunchKernelGGL(i64_kernel, dim3(NWG), dim3(NT), 0, 0, acci, 1.0f); }));
    printf("float atomics          %8.1f us\n", time_us([&] { hipLaunchKernelGGL(f32_kernel, dim3(NWG), dim3(NT), 0, 0, accf, 1.0f); }));
    return 0;
}
