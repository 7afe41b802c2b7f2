// Which XCD does block b of a 1-D grid land on?  (speed-only question: the atom-split workgroup pairs of the K > 64 gradient
// pass sit 8 block ids apart and count on sharing an L2.)  hipcc --offload-arch=gfx950 -O2 tools/exp/xcc_map.hip -o /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(int* xcc, int* cu) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        xcc[blockIdx.x] = (int)(v & 0xf);
        unsigned h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        cu[blockIdx.x] = (int)h;
        lds[0] = 1;
    }
    // keep the workgroup resident for a while so that all 256 are placed side by side
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
int main() {
    for (int lds_kb : {8, 122}) {
        const int n = 256;
        int *dx, *dc;
        hipMalloc(&dx, n * 4); hipMalloc(&dc, n * 4);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
        hipLaunchKernelGGL(probe, dim3(n), dim3(512), lds_kb * 1024, 0, dx, dc);
        std::vector<int> x(n), c(n);
        hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        printf("LDS %d KB per workgroup: XCC id of blocks 0..31:", lds_kb);
        for (int i = 0; i < 32; ++i) printf(" %d", x[i]);
        int same = 0;
        for (int i = 0; i + 8 < n; ++i) same += (x[i] == x[i + 8]);
        printf("\n  blocks b and b+8 on the same XCC: %d of %d; b and b+1: ", same, n - 8);
        int adj = 0;
        for (int i = 0; i + 1 < n; ++i) adj += (x[i] == x[i + 1]);
        printf("%d of %d\n", adj, n - 1);
    }
    return 0;
}
