// Experiment (not product): where do the 87 us of grad_v_f32_kernel<2,16> (v = z D_dagger^T, 512 x 50, fp32) go?
// The kernel body is repeated here with phases that can be switched off at compile time; timing only, the
// ablated variants compute garbage.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I dl_attack_on_imagenet_amd/csrc tools/exp/ablate_grad_v_f32.hip -o tools/exp/bin/ablate_grad_v_f32
#include "../../dl_attack_on_imagenet_amd/csrc/adil_contract.hip"
#include <cstdio>
#include <cstring>
#include <vector>

enum { NO_MFMA = 1, NO_SPLIT = 2, NO_IMGLOAD = 4, NO_BARRIER = 8, NO_IMGWRITE = 16, NO_FRAGREAD = 32, NO_DWORK = 64 };

template <int AT, int NW, int ABL>
__global__ __launch_bounds__(NW * 64) void ablate_kernel(const float* __restrict__ g, const float* __restrict__ d,
                                                         float* __restrict__ slab, int B, int Bp, int P, int K, int ntiles,
                                                         int tiles_per_wg) {
    using M = Mma<float>;
    using Frag = M::Frag;
    constexpr int TW = 32, KA = AT * 32, NT = NW * 64;
    constexpr int GI = TW + DPAD, IPL = NW * 32 * GI;
    constexpr int GD = TW + DPAD, DPL = KA * GD, DBUF = 3 * DPL;
    constexpr int LPR = TW / 4, RPI = 64 / LPR, NLD = 32 / RPI;
    constexpr int DPT = (TW * KA + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sdt = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* simg = sdt + 2 * DBUF;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = min(ntiles, t0 + tiles_per_wg);
    const int tlast = max(t1 - 1, 0);
    const int b0 = w * 32;
    const int lrow = lane / LPR, lcol = (lane - lrow * LPR) * 4;
    f32x16 accv[AT];
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[at][r] = 0.0f;

    struct Stage { float dreg[DPT]; u32x4 blk[NLD]; };
    auto load_stage = [&](Stage& st, int tile, bool first) __attribute__((always_inline)) {
        if (!(ABL & NO_DWORK) || first) {
#pragma unroll
            for (int e = 0; e < DPT; ++e) {
                const int i = tid + e * NT;
                const int px = i / KA, a = i - px * KA;
                st.dreg[e] = d[(size_t)(tile * TW + (px < TW ? px : TW - 1)) * K + (a < K ? a : K - 1)];
            }
        }
        if (!(ABL & NO_IMGLOAD) || first) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int row = b0 + i * RPI + lrow;
                st.blk[i] = *reinterpret_cast<const u32x4*>(g + (size_t)(row < B ? row : B - 1) * P + tile * TW + lcol);
            }
        }
    };
    auto write_d = [&](bf16_t* dst, const Stage& st, float valid) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            if (px < TW) DImg<float>::put(dst, a * GD + px, DPL, st.dreg[e] * ((a < K) ? valid : 0.0f));
        }
    };
    Stage sa, sb;
    load_stage(sa, min(t0, tlast), true);
    load_stage(sb, min(t0 + 1, tlast), true);
    write_d(sdt, sa, t0 < t1 ? 1.0f : 0.0f);

    auto tile_step = [&](int tile, int dbuf, Stage& cur, Stage& oth) __attribute__((always_inline)) {
        if (!(ABL & NO_IMGWRITE)) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if (ABL & NO_SPLIT) {
                    const int off = (b0 + i * RPI + lrow) * GI + lcol;
                    *reinterpret_cast<u32x2*>(simg + off) = u32x2{cur.blk[i][0], cur.blk[i][1]};
                    *reinterpret_cast<u32x2*>(simg + off + IPL) = u32x2{cur.blk[i][2], cur.blk[i][3]};
                    *reinterpret_cast<u32x2*>(simg + off + 2 * IPL) = u32x2{cur.blk[i][1], cur.blk[i][2]};
                } else {
                    const float f4[4] = {__uint_as_float(cur.blk[i][0]), __uint_as_float(cur.blk[i][1]),
                                         __uint_as_float(cur.blk[i][2]), __uint_as_float(cur.blk[i][3])};
                    DImg<float>::put4(simg, (b0 + i * RPI + lrow) * GI + lcol, IPL, f4);
                }
            }
        }
        load_stage(cur, min(tile + 2, tlast), false);
        if (!(ABL & NO_BARRIER)) lds_barrier();
        const bf16_t* sdb = sdt + dbuf * DBUF;
#pragma unroll
        for (int g3 = 0; g3 < TW / 16; ++g3) {
            Frag a;
            if (ABL & NO_FRAGREAD) {
                u32x4 t = {(unsigned)tile, (unsigned)g3, 0x3f803f80u, 0x3f803f80u};
                asm volatile("" : "+v"(t));
                a.h = a.m = a.l = __builtin_bit_cast(bf16x8, t);
            } else {
                a = DImg<float>::load8(simg + (b0 + c) * GI + 16 * g3 + 8 * h, IPL);
            }
#pragma unroll
            for (int at = 0; at < AT; ++at) {
                Frag b;
                if (ABL & NO_FRAGREAD) b = a;
                else b = DImg<float>::load8(sdb + (at * 32 + c) * GD + 16 * g3 + 8 * h, DPL);
                if (ABL & NO_MFMA) {
                    M::touch(b);
                    Frag a2 = a;
                    M::touch(a2);
                } else {
                    M::mma(accv[at], a, b);
                }
            }
        }
        if (!(ABL & NO_DWORK)) write_d(sdt + (dbuf ^ 1) * DBUF, oth, tile + 1 < t1 ? 1.0f : 0.0f);
    };
    for (int tile = t0; tile < t1; tile += 2) {
        tile_step(tile, 0, sa, sb);
        tile_step(tile + 1, 1, sb, sa);
    }
    if (b0 < Bp) {
        float* dst = slab + (size_t)blockIdx.x * Bp * K;
#pragma unroll
        for (int at = 0; at < AT; ++at)
            if (at * 32 + c < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)(b0 + c_row(r, h)) * K + at * 32 + c] = accv[at][r];
            }
    }
}


// ---- software-pipelined variant: ONE register stage; the wave reads all its A fragments of tile t first, then the
// split + ds_write of tile t+1 (into the image rows it has just read) is interleaved with the MFMAs of tile t.
template <int AT, int NW, int SCHED, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void pipe_kernel(const float* __restrict__ g, const float* __restrict__ d,
                                                       float* __restrict__ slab, int B, int Bp, int P, int K, int ntiles,
                                                       int tiles_per_wg) {
    using M = Mma<float>;
    using Frag = M::Frag;
    constexpr int TW = 32, KA = AT * 32, NT = NW * 64;
    constexpr int GI = TW + DPAD, IPL = NW * 32 * GI;
    constexpr int GD = TW + DPAD, DPL = KA * GD, DBUF = 3 * DPL;
    constexpr int LPR = TW / 4, RPI = 64 / LPR, NLD = 32 / RPI;
    constexpr int DPT = (TW * KA + NT - 1) / NT;
    constexpr int NG = (TW / 16) * AT;                           // MFMA groups per tile (6 MFMAs each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sdt = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* simg = sdt + 2 * DBUF;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = min(ntiles, t0 + tiles_per_wg);
    const int tlast = max(t1 - 1, 0);
    const int b0 = w * 32;
    const int lrow = lane / LPR, lcol = (lane - lrow * LPR) * 4;
    f32x16 accv[AT];
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[at][r] = 0.0f;
    float dreg[DPT];
    u32x4 blk[NLD];
    const float* grow[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = b0 + i * RPI + lrow;
        grow[i] = g + (size_t)(row < B ? row : B - 1) * P + lcol;
    }
    auto load_d = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            dreg[e] = d[(size_t)(tile * TW + (px < TW ? px : TW - 1)) * K + (a < K ? a : K - 1)];
        }
    };
    auto load_img = [&](int i, int tile) __attribute__((always_inline)) {
        blk[i] = *reinterpret_cast<const u32x4*>(grow[i] + tile * TW);
    };
    auto split_img = [&](int i) __attribute__((always_inline)) {
        const float f4[4] = {__uint_as_float(blk[i][0]), __uint_as_float(blk[i][1]), __uint_as_float(blk[i][2]),
                             __uint_as_float(blk[i][3])};
        DImg<float>::put4(simg, (b0 + i * RPI + lrow) * GI + lcol, IPL, f4);
    };
    auto write_d = [&](bf16_t* dst, float valid) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            if (px < TW) DImg<float>::put(dst, a * GD + px, DPL, dreg[e] * ((a < K) ? valid : 0.0f));
        }
    };
    load_d(min(t0, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(i, min(t0, tlast));
    write_d(sdt, t0 < t1 ? 1.0f : 0.0f);
#pragma unroll
    for (int i = 0; i < NLD; ++i) split_img(i);
    load_d(min(t0 + 1, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(i, min(t0 + 1, tlast));

    auto tile_step = [&](int tile, int dbuf) __attribute__((always_inline)) {
        if (!(ABL & NO_BARRIER)) lds_barrier();   // D[dbuf] staged by everyone; everyone is done reading D[dbuf^1]
        const bf16_t* sdb = sdt + dbuf * DBUF;
        Frag a[TW / 16];
#pragma unroll
        for (int g3 = 0; g3 < TW / 16; ++g3) a[g3] = DImg<float>::load8(simg + (b0 + c) * GI + 16 * g3 + 8 * h, IPL);
        const int tnext = min(tile + 2, tlast);
        const float valid = tile + 1 < t1 ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int g3 = q / AT, at = q - g3 * AT;
            {
                Frag b = DImg<float>::load8(sdb + (at * 32 + c) * GD + 16 * g3 + 8 * h, DPL);
                if (ABL & NO_MFMA) { M::touch(b); M::touch(a[g3]); }
                else M::mma(accv[at], a[g3], b);
            }
            // the LDS executes a wave's instructions in order: these writes land after the reads of a[] above
#pragma unroll
            for (int i = q * NLD / NG; i < (q + 1) * NLD / NG; ++i) {
                split_img(i);
                if (!(ABL & NO_IMGLOAD)) load_img(i, tnext);
            }
            if (q == NG - 1 && !(ABL & NO_DWORK)) {
                write_d(sdt + (dbuf ^ 1) * DBUF, valid);
                load_d(tnext);
            }
            if (SCHED) __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int tile = t0; tile < t1; tile += 2) {
        tile_step(tile, 0);
        tile_step(tile + 1, 1);
    }
    if (b0 < Bp) {
        float* dst = slab + (size_t)blockIdx.x * Bp * K;
#pragma unroll
        for (int at = 0; at < AT; ++at)
            if (at * 32 + c < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)(b0 + c_row(r, h)) * K + at * 32 + c] = accv[at][r];
            }
    }
}


// ---- pipelined (one stage) + UNPADDED, XOR-swizzled LDS images: rows of 32 bf16 = 64 B, the 16-B chunk q of row r lives at
// chunk q ^ ((r >> 2) & 3).  ds_write_b64 banks are (a/4) mod 32 over groups of 16 contiguous lanes = two rows of 64 B:
// with the 80-B padded rows the second row overlaps the first by 4 banks (2-way conflict on every image write).
template <int AT, int NW, int PAIR, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void swz_kernel(const float* __restrict__ g, const float* __restrict__ d,
                                                       float* __restrict__ slab, int B, int Bp, int P, int K, int ntiles,
                                                       int tiles_per_wg) {
    using M = Mma<float>;
    using Frag = M::Frag;
    constexpr int TW = 32, KA = AT * 32, NT = NW * 64;
    constexpr int GI = TW, IPL = NW * 32 * GI;
    constexpr int GD = TW, DPL = KA * GD, DBUF = 3 * DPL;
    constexpr int LPR = TW / 4, RPI = 64 / LPR, NLD = 32 / RPI;
    constexpr int DPT = (TW * KA + NT - 1) / NT;
    constexpr int NG = (TW / 16) * AT;                           // MFMA groups per tile (6 MFMAs each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sdt = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* simg = sdt + 2 * DBUF;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = min(ntiles, t0 + tiles_per_wg);
    const int tlast = max(t1 - 1, 0);
    const int b0 = w * 32;
    const int lrow = lane / LPR, lcol = (lane - lrow * LPR) * 4;
    f32x16 accv[AT];
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[at][r] = 0.0f;
    float dreg[DPT];
    u32x4 blk[NLD];
    const float* grow[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = b0 + i * RPI + lrow;
        grow[i] = g + (size_t)(row < B ? row : B - 1) * P + lcol;
    }
    auto load_d = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            dreg[e] = d[(size_t)(tile * TW + (px < TW ? px : TW - 1)) * K + (a < K ? a : K - 1)];
        }
    };
    auto load_img = [&](int i, int tile) __attribute__((always_inline)) {
        blk[i] = *reinterpret_cast<const u32x4*>(grow[i] + tile * TW);
    };
    auto split_img = [&](int i) __attribute__((always_inline)) {
        const float f4[4] = {__uint_as_float(blk[i][0]), __uint_as_float(blk[i][1]), __uint_as_float(blk[i][2]),
                             __uint_as_float(blk[i][3])};
        const int r = i * RPI + lrow;
        DImg<float>::put4(simg, (b0 + r) * GI + ((((lcol >> 3) ^ (r >> 2)) & 3) << 3) + (lcol & 7), IPL, f4);
    };
    auto write_d = [&](bf16_t* dst, float valid) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            if (px < TW) DImg<float>::put(dst, a * GD + ((((px >> 3) ^ (a >> 2)) & 3) << 3) + (px & 7), DPL, dreg[e] * ((a < K) ? valid : 0.0f));
        }
    };
    load_d(min(t0, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(i, min(t0, tlast));
    write_d(sdt, t0 < t1 ? 1.0f : 0.0f);
#pragma unroll
    for (int i = 0; i < NLD; ++i) split_img(i);
    load_d(min(t0 + 1, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(i, min(t0 + 1, tlast));

    auto tile_step = [&](int tile, int dbuf) __attribute__((always_inline)) {
        if (!(ABL & NO_BARRIER)) lds_barrier();   // D[dbuf] staged by everyone; everyone is done reading D[dbuf^1]
        const bf16_t* sdb = sdt + dbuf * DBUF;
        Frag a[TW / 16];
#pragma unroll
        for (int g3 = 0; g3 < TW / 16; ++g3) a[g3] = DImg<float>::load8(simg + (b0 + c) * GI + ((((2 * g3 + h) ^ (c >> 2)) & 3) << 3), IPL);
        const int tnext = min(tile + 2, tlast);
        const float valid = tile + 1 < t1 ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int g3 = q / AT, at = q - g3 * AT;
            {
                Frag b = DImg<float>::load8(sdb + (at * 32 + c) * GD + ((((2 * g3 + h) ^ (c >> 2)) & 3) << 3), DPL);
                if (ABL & NO_MFMA) { M::touch(b); M::touch(a[g3]); }
                else M::mma(accv[at], a[g3], b);
            }
            // the LDS executes a wave's instructions in order: these writes land after the reads of a[] above
#pragma unroll
            for (int i = q * NLD / NG; i < (q + 1) * NLD / NG; ++i) {
                split_img(i);
                if (!(ABL & NO_IMGLOAD)) load_img(i, tnext);
            }
            if (q == NG - 1 && !(ABL & NO_DWORK)) {
                write_d(sdt + (dbuf ^ 1) * DBUF, valid);
                load_d(tnext);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int tile = t0; tile < t1; tile += 2) {
        tile_step(tile, 0);
        tile_step(tile + 1, 1);
    }
    if (b0 < Bp) {
        float* dst = slab + (size_t)blockIdx.x * Bp * K;
#pragma unroll
        for (int at = 0; at < AT; ++at)
            if (at * 32 + c < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)(b0 + c_row(r, h)) * K + at * 32 + c] = accv[at][r];
            }
    }
}


// ---- pipelined, TWO register stages: raw buffer loads (one VGPR offset for all blocks, SGPR offsets per block/tile; rows
// past the batch read as zeros by the range check) and per-wave plane images (one LDS address register) pay for them.
template <int AT, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void pipe2_kernel(const float* __restrict__ g, const float* __restrict__ d,
                                                        float* __restrict__ slab, int B, int Bp, int P, int K, int ntiles,
                                                        int tiles_per_wg) {
    using M = Mma<float>;
    using Frag = M::Frag;
    constexpr int TW = 32, KA = AT * 32, NT = NW * 64;
    constexpr int GI = TW + DPAD, WPL = 32 * GI;                 // per-wave image: [3 planes][32 rows][GI]
    constexpr int GD = TW + DPAD, DPL = KA * GD, DBUF = 3 * DPL;
    constexpr int LPR = TW / 4, RPI = 64 / LPR, NLD = 32 / RPI;
    constexpr int DPT = (TW * KA + NT - 1) / NT;
    constexpr int NG = (TW / 16) * AT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sdt = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    bf16_t* simg = sdt + 2 * DBUF + w * 3 * WPL;
    const int t0 = blockIdx.x * tiles_per_wg;
    const int t1 = min(ntiles, t0 + tiles_per_wg);
    const int tlast = max(t1 - 1, 0);
    const int b0 = w * 32;
    const int lrow = lane / LPR, lcol = (lane - lrow * LPR) * 4;
    const buf_rsrc rg = block_rsrc(g, (unsigned)((size_t)B * P * sizeof(float)));
    const int voff = ((b0 + lrow) * P + lcol) * (int)sizeof(float);
    f32x16 accv[AT];
#pragma unroll
    for (int at = 0; at < AT; ++at)
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[at][r] = 0.0f;
    struct Stage { float dreg[DPT]; u32x4 blk[NLD]; };
    auto load_d = [&](Stage& st, int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            st.dreg[e] = d[(size_t)(tile * TW + (px < TW ? px : TW - 1)) * K + (a < K ? a : K - 1)];
        }
    };
    auto load_img = [&](Stage& st, int i, int tile) __attribute__((always_inline)) {
        st.blk[i] = BufPx<float>::load(rg, voff, (i * RPI * P + tile * TW) * (int)sizeof(float));
    };
    auto split_img = [&](const Stage& st, int i) __attribute__((always_inline)) {
        const float f4[4] = {__uint_as_float(st.blk[i][0]), __uint_as_float(st.blk[i][1]), __uint_as_float(st.blk[i][2]),
                             __uint_as_float(st.blk[i][3])};
        DImg<float>::put4(simg, (i * RPI + lrow) * GI + lcol, WPL, f4);
    };
    auto write_d = [&](bf16_t* dst, const Stage& st, float valid) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const int i = tid + e * NT;
            const int px = i / KA, a = i - px * KA;
            if (px < TW) DImg<float>::put(dst, a * GD + px, DPL, st.dreg[e] * ((a < K) ? valid : 0.0f));
        }
    };
    Stage sa, sb;
    load_d(sa, min(t0, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(sa, i, min(t0, tlast));
    load_d(sb, min(t0 + 1, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(sb, i, min(t0 + 1, tlast));
    write_d(sdt, sa, t0 < t1 ? 1.0f : 0.0f);
#pragma unroll
    for (int i = 0; i < NLD; ++i) split_img(sa, i);
    load_d(sa, min(t0 + 2, tlast));
#pragma unroll
    for (int i = 0; i < NLD; ++i) load_img(sa, i, min(t0 + 2, tlast));
    // now: LDS holds tile t0, sb holds tile t0+1, sa holds tile t0+2

    auto tile_step = [&](int tile, int dbuf, Stage& nxt) __attribute__((always_inline)) {   // nxt holds tile + 1
        if (!(ABL & NO_BARRIER)) lds_barrier();
        const bf16_t* sdb = sdt + dbuf * DBUF;
        Frag a[TW / 16];
#pragma unroll
        for (int g3 = 0; g3 < TW / 16; ++g3) a[g3] = DImg<float>::load8(simg + c * GI + 16 * g3 + 8 * h, WPL);
        const int tnext = min(tile + 3, tlast);
        const float valid = tile + 1 < t1 ? 1.0f : 0.0f;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int g3 = q / AT, at = q - g3 * AT;
            {
                Frag b = DImg<float>::load8(sdb + (at * 32 + c) * GD + 16 * g3 + 8 * h, DPL);
                if (ABL & NO_MFMA) { M::touch(b); M::touch(a[g3]); }
                else M::mma(accv[at], a[g3], b);
            }
#pragma unroll
            for (int i = q * NLD / NG; i < (q + 1) * NLD / NG; ++i) {
                split_img(nxt, i);
                if (!(ABL & NO_IMGLOAD)) load_img(nxt, i, tnext);
            }
            if (q == NG - 1 && !(ABL & NO_DWORK)) {
                write_d(sdt + (dbuf ^ 1) * DBUF, nxt, valid);
                load_d(nxt, tnext);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int tile = t0; tile < t1; tile += 2) {
        tile_step(tile, 0, sb);
        tile_step(tile + 1, 1, sa);
    }
    if (b0 < Bp) {
        float* dst = slab + (size_t)blockIdx.x * Bp * K;
#pragma unroll
        for (int at = 0; at < AT; ++at)
            if (at * 32 + c < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(size_t)(b0 + c_row(r, h)) * K + at * 32 + c] = accv[at][r];
            }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ABL>
static int run(const char* name, const float* g, const float* d, float* slab, int B, int P, int K) {
    constexpr int AT = 2, NW = 16, TW = 32;
    const size_t lds = (2 * 3 * (size_t)AT * 32 * (TW + DPAD) + 3 * (size_t)NW * 32 * (TW + DPAD)) * sizeof(bf16_t);
    CK(hipFuncSetAttribute((const void*)ablate_kernel<AT, NW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nt = P / 32, tpw = (nt + 255) / 256, nwg = (nt + tpw - 1) / tpw;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((ablate_kernel<AT, NW, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((ablate_kernel<AT, NW, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %8.1f us\n", name, ms * 1e3 / n);
    fflush(stdout);
    return 0;
}

template <int SCHED, int ABL = 0>
static int run_pipe(const char* name, const float* g, const float* d, float* slab, float* ref, int B, int P, int K) {
    constexpr int AT = 2, NW = 16, TW = 32;
    const size_t lds = (2 * 3 * (size_t)AT * 32 * (TW + DPAD) + 3 * (size_t)NW * 32 * (TW + DPAD)) * sizeof(bf16_t);
    CK(hipFuncSetAttribute((const void*)pipe_kernel<AT, NW, SCHED, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)ablate_kernel<AT, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nt = P / 32, tpw = (nt + 255) / 256, nwg = (nt + tpw - 1) / tpw;
    hipLaunchKernelGGL((ablate_kernel<AT, NW, 0>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, ref, B, B, P, K, nt, tpw);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((pipe_kernel<AT, NW, SCHED, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((pipe_kernel<AT, NW, SCHED, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t n_out = (size_t)nwg * B * K;
    std::vector<float> a(n_out), b(n_out);
    CK(hipMemcpy(a.data(), slab, n_out * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, n_out * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < n_out; ++i) diff += (memcmp(&a[i], &b[i], 4) != 0);
    printf("%-52s %8.1f us   slab entries differing from the full kernel: %zu of %zu\n", name, ms * 1e3 / n, diff, n_out);
    fflush(stdout);
    return 0;
}

template <int PAIR, int ABL = 0>
static int run_swz(const char* name, const float* g, const float* d, float* slab, float* ref, int B, int P, int K) {
    constexpr int AT = 2, NW = 16, TW = 32;
    const size_t lds = (2 * 3 * (size_t)AT * 32 * TW + 3 * (size_t)NW * 32 * TW) * sizeof(bf16_t);
    const size_t lds_ref = (2 * 3 * (size_t)AT * 32 * (TW + DPAD) + 3 * (size_t)NW * 32 * (TW + DPAD)) * sizeof(bf16_t);
    CK(hipFuncSetAttribute((const void*)swz_kernel<AT, NW, PAIR, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)ablate_kernel<AT, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ref));
    const int nt = P / 32, tpw = (nt + 255) / 256, nwg = (nt + tpw - 1) / tpw;
    hipLaunchKernelGGL((ablate_kernel<AT, NW, 0>), dim3(nwg), dim3(NW * 64), lds_ref, 0, g, d, ref, B, B, P, K, nt, tpw);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((swz_kernel<AT, NW, PAIR, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((swz_kernel<AT, NW, PAIR, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t n_out = (size_t)nwg * B * K;
    std::vector<float> a(n_out), b(n_out);
    CK(hipMemcpy(a.data(), slab, n_out * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, n_out * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < n_out; ++i) diff += (memcmp(&a[i], &b[i], 4) != 0);
    printf("%-52s %8.1f us   slab entries differing from the full kernel: %zu of %zu\n", name, ms * 1e3 / n, diff, n_out);
    fflush(stdout);
    return 0;
}

template <int ABL>
static int run_pipe2(const char* name, const float* g, const float* d, float* slab, float* ref, int B, int P, int K) {
    constexpr int AT = 2, NW = 16, TW = 32;
    const size_t lds = (2 * 3 * (size_t)AT * 32 * (TW + DPAD) + 3 * (size_t)NW * 32 * (TW + DPAD)) * sizeof(bf16_t);
    CK(hipFuncSetAttribute((const void*)pipe2_kernel<AT, NW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nt = P / 32, tpw = (nt + 255) / 256, nwg = (nt + tpw - 1) / tpw;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((pipe2_kernel<AT, NW, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((pipe2_kernel<AT, NW, ABL>), dim3(nwg), dim3(NW * 64), lds, 0, g, d, slab, B, B, P, K, nt, tpw);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t n_out = (size_t)nwg * B * K;
    std::vector<float> a(n_out), b(n_out);
    CK(hipMemcpy(a.data(), slab, n_out * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, n_out * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < n_out; ++i) diff += (memcmp(&a[i], &b[i], 4) != 0);
    printf("%-52s %8.1f us   slab entries differing from the full kernel: %zu of %zu\n", name, ms * 1e3 / n, diff, n_out);
    fflush(stdout);
    return 0;
}

int main() {
    const int B = 512, P = 150528, K = 50;
    float *g, *d, *slab;
    CK(hipMalloc(&g, (size_t)B * P * 4));
    CK(hipMalloc(&d, (size_t)P * K * 4));
    CK(hipMalloc(&slab, (size_t)256 * B * K * 4));
    float* ref;
    CK(hipMalloc(&ref, (size_t)256 * B * K * 4));
    std::vector<float> hg((size_t)B * P), hd((size_t)P * K);
    unsigned s = 12345;
    for (auto& x : hg) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.0f * 0.02f - 0.01f; }
    for (auto& x : hd) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; }
    CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    if (run<0>("full kernel", g, d, slab, B, P, K)) return 1;
    if (run_pipe<0>("pipelined (compiler's schedule)", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe<1>("pipelined (sched_barrier per MFMA group)", g, d, slab, ref, B, P, K)) return 1;
    if (run_swz<0>("pipelined + swizzled unpadded LDS images", g, d, slab, ref, B, P, K)) return 1;
    if (run_swz<0, NO_MFMA>("pipelined + swizzled, no MFMA", g, d, slab, ref, B, P, K)) return 1;
    if (run_swz<0, NO_IMGLOAD>("pipelined + swizzled, no image loads in the loop", g, d, slab, ref, B, P, K)) return 1;
    if (run_swz<0, NO_DWORK>("pipelined + swizzled, no D loads / staging", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe2<0>("pipelined, two stages", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe2<NO_IMGLOAD>("pipelined, two stages, no image loads in the loop", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe2<NO_MFMA>("pipelined, two stages, no MFMA", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe2<NO_DWORK>("pipelined, two stages, no D loads / staging", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe2<NO_BARRIER>("pipelined, two stages, no barrier", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe<1, NO_MFMA>("pipelined, no MFMA", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe<1, NO_IMGLOAD>("pipelined, no image loads in the loop", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe<1, NO_BARRIER>("pipelined, no barrier", g, d, slab, ref, B, P, K)) return 1;
    if (run_pipe<1, NO_DWORK>("pipelined, no D loads / staging", g, d, slab, ref, B, P, K)) return 1;
    if (run<NO_MFMA>("no MFMA", g, d, slab, B, P, K)) return 1;
    if (run<NO_SPLIT>("no split (raw ds_write)", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGLOAD>("no image loads in the loop", g, d, slab, B, P, K)) return 1;
    if (run<NO_BARRIER>("no barrier", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGWRITE>("no image split + ds_write", g, d, slab, B, P, K)) return 1;
    if (run<NO_FRAGREAD>("no fragment ds_reads", g, d, slab, B, P, K)) return 1;
    if (run<NO_DWORK>("no D loads / D staging", g, d, slab, B, P, K)) return 1;
    if (run<NO_MFMA | NO_FRAGREAD>("no MFMA, no fragment reads", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGWRITE | NO_DWORK>("no image write, no D work (loads + MFMA only)", g, d, slab, B, P, K)) return 1;
    if (run<NO_MFMA | NO_FRAGREAD | NO_IMGWRITE | NO_DWORK>("loads + barrier only", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGLOAD | NO_IMGWRITE | NO_DWORK>("fragment reads + MFMA + barrier only", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGLOAD | NO_IMGWRITE | NO_DWORK | NO_BARRIER>("fragment reads + MFMA only", g, d, slab, B, P, K)) return 1;
    if (run<NO_IMGLOAD | NO_IMGWRITE | NO_DWORK | NO_BARRIER | NO_FRAGREAD>("MFMA only", g, d, slab, B, P, K)) return 1;
    return 0;
}
