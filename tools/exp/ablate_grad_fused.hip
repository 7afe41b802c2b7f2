// Experiment (not product): where do the ~50 us of grad_fused_mfma_kernel<bf16, AT=2, NW=8, RB=2> (grad_d + grad_v of one learning
// step: 512 x 150528 bf16, 50 atoms) go?  The kernel body (aligned interior, first row chunk, no atom split) is repeated here
// with phases that can be switched off at compile time; timing only — the ablated variants compute garbage.  The g batch
// rotates over four buffers so that no launch finds it in the Infinity Cache.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I dl_attack_on_imagenet_amd/csrc tools/exp/ablate_grad_fused.hip -o tools/exp/bin/ablate_grad_fused
#include "../../dl_attack_on_imagenet_amd/csrc/adil_contract.hip"
#include <cstdio>
#include <cstring>
#include <vector>

enum { NO_FLIP = 1024, NO_GV_MFMA = 1, NO_GV_READS = 2, NO_GD_MFMA = 4, NO_GD_READS = 8, NO_IMG_WRITE = 16, NO_G_LOAD = 32, NO_D_STORE = 64,
       NO_RED = 128, NO_D_TILE = 256, NO_BARRIERS = 512 };

template <int ABL>
__global__ __launch_bounds__(512) void ablate_kernel(const bf16_t* __restrict__ g, const float* __restrict__ d,
                                                     const bf16_t* __restrict__ vpt, int vstride, float* __restrict__ grad_d,
                                                     float* __restrict__ slab, int B, int Bp, int P, int K, int tile_end,
                                                     int tiles_per_wg, int k_split, int nranges) {
    int range = blockIdx.x, k0 = 0, kn = K, flip = 0;
    if (k_split > 0) {                                            // atom-split workgroup pairs, as in the product kernel
        const int bid = blockIdx.x, half = (bid >> 3) & 1;
        range = (bid >> 4) * 8 + (bid & 7);
        if (range >= nranges) return;
        k0 = half ? k_split : 0;
        kn = half ? K - k_split : k_split;
        flip = (ABL & NO_FLIP) ? 0 : (half ^ (range & 1));
    }
    using T = bf16_t;
    using M = Mma<T>;
    using E = typename M::Elem;
    using Frag = typename M::Frag;
    constexpr int AT = 2, NW = 8, RB = 2;
    constexpr int KA = AT * 32;
    constexpr int GS = GV_TW + M::PAD;
    constexpr int EPL = 16 / sizeof(E), LPR = GV_TW / EPL, RPI = 64 / LPR, NLD = 32 / RPI;
    constexpr int NT = NW * 64;
    constexpr int DPT = (GV_TW * KA + NT - 1) / NT;
    constexpr int NBLK = NW * RB;
    constexpr int NTILE = 2 * AT;
    constexpr int KS = NW / NTILE;
    constexpr int RS = NBLK * 32 / KS;
    constexpr int NKG = RS / 16;
    constexpr int RPW = 16 / KS;
    constexpr int GD = GV_TW + DPAD, DPL = KA * GD, DBUF = DImg<T>::PLANES * DPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sdt = reinterpret_cast<bf16_t*>(smem_raw);
    E* simg = reinterpret_cast<E*>(sdt + 2 * DBUF);
    float* red = reinterpret_cast<float*>(simg + NBLK * 32 * GS);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int t0 = range * tiles_per_wg;
    const int t1 = min(tile_end, t0 + tiles_per_wg);
    const int nt = t1 - t0;
    auto tile_at = [&](int i) __attribute__((always_inline)) { const int j = i ^ flip; return t0 + (j < nt ? j : i); };
    const int ti = w % NTILE, ks = w / NTILE, tp = ti & 1, ta = ti >> 1;
    auto barrier = [&]() __attribute__((always_inline)) { if constexpr (!(ABL & NO_BARRIERS)) lds_barrier(); };

    f32x16 accv[RB][AT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int at = 0; at < AT; ++at)
#pragma unroll
            for (int r = 0; r < 16; ++r) accv[rb][at][r] = 0.0f;
    const int lrow = lane / LPR, lcol = (lane - lrow * LPR) * EPL;
    typename M::Raw vfr[NKG];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) vfr[kg] = M::load8_raw(vpt + (size_t)(k0 + ta * 32 + c) * vstride + ks * RS + 16 * kg + 8 * h);
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) M::touch_raw(vfr[kg]);

    float dreg[DPT];
    u32x4 blk[RB][NLD];
    if (nt > 0) {
        const int tf = tile_at(0);
        gv_load_d<T, AT, NW, true>(d, tf, P, K, tid, dreg, k0, kn);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) gv_load_g<T, true>(g, tf, (w * RB + rb) * 32, B, P, lrow, lcol, blk[rb]);
        gv_write_d<T, AT, NW, true>(sdt, tf, P, K, tid, dreg, kn);
    }
    for (int ti_ = 0; ti_ < nt; ++ti_) {
        const int tile = tile_at(ti_), tnext = tile_at(ti_ + 1);
        const int buf = ti_ & 1;
        const bool more = ti_ + 1 < nt;
        const int p0 = tile * GV_TW;
        if constexpr (!(ABL & NO_IMG_WRITE)) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                E* sg = simg + (size_t)(w * RB + rb) * 32 * GS;
#pragma unroll
                for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(sg + (i * RPI + lrow) * GS + lcol) = blk[rb][i];
            }
        }
        if (more) {
            if constexpr (!(ABL & NO_D_TILE)) gv_load_d<T, AT, NW, true>(d, tnext, P, K, tid, dreg, k0, kn);
            if constexpr (!(ABL & NO_G_LOAD)) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) gv_load_g<T, true>(g, tnext, (w * RB + rb) * 32, B, P, lrow, lcol, blk[rb]);
            }
        }
        barrier();
        {
            const bf16_t* sdb = sdt + buf * DBUF;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const E* sg = simg + (size_t)(w * RB + rb) * 32 * GS;
#pragma unroll
                for (int g3 = 0; g3 < GV_TW / 16; ++g3) {
                    Frag a;
                    if constexpr (ABL & NO_GV_READS) { a = M::expand(vfr[g3]); } else { a = M::load8(sg + c * GS + 16 * g3 + 8 * h); }
#pragma unroll
                    for (int at = 0; at < AT; ++at) {
                        Frag bfr;
                        if constexpr (ABL & NO_GV_READS) { bfr = M::expand(vfr[4 + g3 + at]); }
                        else { bfr = DImg<T>::load8(sdb + (at * 32 + c) * GD + 16 * g3 + 8 * h, DPL); }
                        if constexpr (ABL & NO_GV_MFMA) {
                            accv[rb][at][g3] += __uint_as_float(((const unsigned*)&a)[0] ^ ((const unsigned*)&bfr)[0]);
                        } else {
                            M::mma(accv[rb][at], a, bfr);
                        }
                    }
                }
            }
        }
        f32x16 accd;
#pragma unroll
        for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            const int r0 = ks * RS + 16 * kg;
            Frag a;
            if constexpr (ABL & NO_GD_READS) { a = M::expand(vfr[(kg + 1) % NKG]); }
            else { a = ColFrag<T>::load(simg + (size_t)(r0 >> 5) * 32 * GS, GS, r0 & 16, tp * 32, lane); }
            if constexpr (ABL & NO_GD_MFMA) {
                accd[kg] += __uint_as_float(((const unsigned*)&a)[0] ^ ((const unsigned*)&vfr[kg])[0]);
            } else {
                M::mma(accd, a, M::expand(vfr[kg]));
            }
        }
        if constexpr (!(ABL & NO_RED)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = accd[r];
        }
        barrier();
        {
            const int atom = ta * 32 + c;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int reg = ks * RPW + rr;
                float sum;
                if constexpr (!(ABL & NO_RED)) {
                    sum = 0.0f;
#pragma unroll
                    for (int q = 0; q < KS; ++q) sum += red[((ti + NTILE * q) * 16 + reg) * 64 + lane];
                } else {
                    sum = accd[reg];
                }
                const int pix = p0 + tp * 32 + c_row(reg, h);
                if constexpr (ABL & NO_D_STORE) {
                    if (sum == 123.456f) grad_d[(size_t)pix * K + k0 + atom] = sum;     // keeps the value alive, never stores
                } else {
                    if (atom < kn) grad_d[(size_t)pix * K + k0 + atom] = sum;
                }
            }
        }
        if constexpr (!(ABL & NO_D_TILE)) {
            if (more) gv_write_d<T, AT, NW, true>(sdt + (buf ^ 1) * DBUF, tnext, P, K, tid, dreg, kn);
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int b0 = (w * RB + rb) * 32;
        float* dst = slab + (size_t)range * Bp * K + k0;
#pragma unroll
        for (int at = 0; at < AT; ++at)
            if (at * 32 + c < kn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)(b0 + c_row(r, h)) * K + at * 32 + c] = accv[rb][at][r];
            }
    }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

struct Ctx {
    int B = 512, P = 150528, K = 50, nwg, tpw, ntiles, k_split = 0, nranges = 0;
    std::vector<bf16_t*> g;
    float *d, *grad_d, *slab;
    bf16_t* vpt;
    hipEvent_t e0, e1;
    size_t lds;
};

template <int ABL>
static void run(Ctx& c, const char* name, int rep) {
    CK(hipFuncSetAttribute((const void*)ablate_kernel<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds));
    auto launch = [&](int i) {
        hipLaunchKernelGGL((ablate_kernel<ABL>), dim3(c.nwg), dim3(512), c.lds, 0, c.g[i % c.g.size()], c.d, c.vpt, c.B, c.grad_d,
                           c.slab, c.B, c.B, c.P, c.K, c.ntiles, c.tpw, c.k_split, c.nranges);
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(c.e0));
    for (int i = 0; i < rep; ++i) launch(i);
    CK(hipEventRecord(c.e1));
    CK(hipEventSynchronize(c.e1));
    float ms;
    CK(hipEventElapsedTime(&ms, c.e0, c.e1));
    printf("| %-86s | %6.1f us |\n", name, ms * 1e3 / rep);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rep = argc > 1 ? atoi(argv[1]) : 40;
    for (int K : {50, 100}) {
        Ctx c;
        c.K = K;
        c.ntiles = c.P / GV_TW;
        if (K <= 64) {
            c.tpw = (c.ntiles + 255) / 256;
            c.nwg = (c.ntiles + c.tpw - 1) / c.tpw;
        } else {                                                  // launch_grad_fused_split: two workgroups per range, one per CU
            c.tpw = (c.ntiles + 127) / 128;
            c.nranges = (c.ntiles + c.tpw - 1) / c.tpw;
            c.nwg = 16 * ((c.nranges + 7) / 8);
            c.k_split = (K + 1) / 2;
        }
        c.lds = grad_fused_lds_bytes<bf16_t, 2, 8, 2>();
        const size_t n = (size_t)c.B * c.P;
        c.g.resize(4);
        for (auto& p : c.g) { CK(hipMalloc(&p, n * 2)); CK(hipMemset(p, 0x3c, n * 2)); }
        CK(hipMalloc(&c.d, (size_t)c.P * c.K * 4)); CK(hipMemset(c.d, 0, (size_t)c.P * c.K * 4));
        CK(hipMalloc(&c.grad_d, (size_t)c.P * c.K * 4));
        CK(hipMalloc(&c.vpt, (size_t)128 * c.B * 2)); CK(hipMemset(c.vpt, 0x3c, (size_t)128 * c.B * 2));
        CK(hipMalloc(&c.slab, (size_t)(c.nwg + 2) * c.B * c.K * 4));
        CK(hipEventCreate(&c.e0)); CK(hipEventCreate(&c.e1));
        printf("\ngrad_fused_mfma_kernel<bf16, 2, 8, 2> ablation, K = %d%s: %d workgroups x %d tiles of 64 pixels, LDS %zu B, 512 rows\n\n",
               c.K, c.k_split ? " (atom-split workgroup pairs)" : "", c.nwg, c.tpw, c.lds);
        printf("| variant | time |\n|---|---|\n");
        for (int pass = 0; pass < 2; ++pass) {
            run<0>(c, "the kernel as it is", rep);
            if (c.k_split) run<NO_FLIP>(c, "both halves of a pair walk their tiles in the SAME order", rep);
            if (c.k_split) run<NO_FLIP | NO_GV_MFMA | NO_GV_READS | NO_GD_MFMA | NO_GD_READS | NO_RED>(c, "the same, stream only", rep);
            run<NO_GV_MFMA | NO_GV_READS>(c, "without grad_v (MFMAs and LDS fragment reads)", rep);
            run<NO_GD_MFMA | NO_GD_READS>(c, "without grad_d's MFMAs and transposing reads", rep);
            run<NO_RED>(c, "without the exchange of the row-split partials through LDS", rep);
            run<NO_D_STORE>(c, "without the grad_d stores", rep);
            run<NO_D_TILE>(c, "without the D tile (loads + LDS writes) after the first", rep);
            run<NO_G_LOAD>(c, "without the g loads after the first tile (HBM stream removed)", rep);
            run<NO_BARRIERS>(c, "without the two barriers per tile", rep);
            run<NO_GV_MFMA | NO_GV_READS | NO_GD_MFMA | NO_GD_READS | NO_RED>(c, "stream only: loads, image writes, barriers, stores (no fragment reads, no MFMAs)", rep);
            run<NO_G_LOAD | NO_D_TILE | NO_D_STORE>(c, "compute only: no global traffic after the first tile", rep);
        }
        for (auto p : c.g) CK(hipFree(p));
        CK(hipFree(c.d)); CK(hipFree(c.grad_d)); CK(hipFree(c.vpt)); CK(hipFree(c.slab));
    }
    return 0;
}
