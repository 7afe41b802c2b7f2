// ADiL parameter-update kernels for gfx950: fused AdamW + projections, prox, atom
// constraints, Gram / pseudo-inverse helpers and evaluation sums.  All HBM-bound
// elementwise / row-wise work: 16-byte vector accesses, one pass per stream.
#include "adil_common.h"

// ---- K4 / K8: flat AdamW + clamp[lo,hi] (+ max|delta|) --------------------- //
// four floats -> four fp8 (OCP e4m3) bytes of 256 x, saturating at the e4m3 range: the operand encoding of the fp8
// synthesis (adil_contract.hip, Mma<fp8_t>::pack4 with dscale = 256) — the two must stay bit-identical
__device__ __forceinline__ unsigned fp8x4_of_dict(float a, float b, float c, float d) {
    const float s = 256.0f;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a * s, -448.0f), 448.0f), fminf(fmaxf(b * s, -448.0f), 448.0f), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c * s, -448.0f), 448.0f), fminf(fmaxf(d * s, -448.0f), 448.0f), w, true);
    return (unsigned)w;
}

__global__ __launch_bounds__(256) void dict_to_fp8_kernel(const float* __restrict__ p, size_t n4, unsigned* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        out[i] = fp8x4_of_dict(v.x, v.y, v.z, v.w);
    }
}

template <typename GT, bool FP8COPY = false>
__global__ __launch_bounds__(256) void adamw_clamp_kernel(float* __restrict__ p, const GT* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ s, size_t n,
                                                          AdamWHyper h, float lo, float hi, float* max_abs_delta,
                                                          const float* __restrict__ dyn, unsigned* __restrict__ p_fp8 = nullptr) {
    if (dyn != nullptr) { h.step_size = dyn[0]; h.bc2_sqrt = dyn[1]; }   // step-dependent scalars from device memory (graphs)
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float local_max = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<const float4*>(p)[i];
        float4 mv = reinterpret_cast<const float4*>(m)[i];
        float4 sv = reinterpret_cast<const float4*>(s)[i];
        float gv[4];
        if constexpr (sizeof(GT) == 4) {
            float4 t = reinterpret_cast<const float4*>(g)[i];
            gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
        } else {
            ushort4 t = reinterpret_cast<const ushort4*>(g)[i];
            gv[0] = bf16_to_f32(t.x); gv[1] = bf16_to_f32(t.y); gv[2] = bf16_to_f32(t.z); gv[3] = bf16_to_f32(t.w);
        }
        float po[4] = {pv.x, pv.y, pv.z, pv.w};
        float mo[4] = {mv.x, mv.y, mv.z, mv.w};
        float so[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q = adamw_elem(po[j], gv[j], mo[j], so[j], h);
            q = fminf(fmaxf(q, lo), hi);
            local_max = fmaxf(local_max, fabsf(q - po[j]));
            po[j] = q;
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(po[0], po[1], po[2], po[3]);
        if constexpr (FP8COPY) p_fp8[i] = fp8x4_of_dict(po[0], po[1], po[2], po[3]);   // the persistent fp8 copy (n % 4 == 0)
        reinterpret_cast<float4*>(m)[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
        reinterpret_cast<float4*>(s)[i] = make_float4(so[0], so[1], so[2], so[3]);
    }
    // tail (n % 4 elements), handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        size_t i = n4 * 4 + threadIdx.x;
        float mm = m[i], ss = s[i], p0 = p[i];
        float q = adamw_elem(p0, Elem<GT>::load(g, i), mm, ss, h);
        q = fminf(fmaxf(q, lo), hi);
        local_max = fmaxf(local_max, fabsf(q - p0));
        p[i] = q; m[i] = mm; s[i] = ss;
    }
    if (max_abs_delta != nullptr) {
        local_max = wave_max(local_max);
        if ((threadIdx.x & 63) == 0 && local_max > 0.0f) atomic_max_nonneg(max_abs_delta, local_max);
    }
}

// --------------------------------------------------------------------------- //
// Row-wise l1-ball projection inside one wavefront (Duchi et al.), K <= 64*EPL.
// Sort-free: every element finds its descending rank and the prefix sum at that
// rank by sweeping the row once through v_readlane broadcasts.
// --------------------------------------------------------------------------- //
template <int EPL>
__device__ __forceinline__ void l1ball_row(float (&x)[EPL], int lane, float radius) {
    float a[EPL];
    float l1 = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { a[e] = fabsf(x[e]); l1 += a[e]; }
    l1 = wave_sum(l1);
    if (l1 < radius) return;                      // strict '<' (utils.py:33): rows inside the ball are untouched
    int rank[EPL];
    float pre[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { rank[e] = 1; pre[e] = a[e]; }
#pragma unroll
    for (int je = 0; je < EPL; ++je) {
#pragma unroll
        for (int jl = 0; jl < ADIL_WAVE; ++jl) {
            const float aj = __shfl(a[je], jl, 64);
            const int jidx = je * ADIL_WAVE + jl;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int iidx = e * ADIL_WAVE + lane;
                const bool before = (aj > a[e]) || (aj == a[e] && jidx < iidx);
                rank[e] += before ? 1 : 0;
                pre[e] += before ? aj : 0.0f;
            }
        }
    }
    // rho = max{ rank : mu_rank * rank > cumsum_rank - radius }   (utils.py:37)
    int rho = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e)
        if (a[e] * (float)rank[e] > pre[e] - radius) rho = max(rho, rank[e]);
    rho = wave_max_i(rho);
    float c = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) c += (rank[e] == rho) ? pre[e] : 0.0f;
    c = wave_sum(c);                               // ranks are unique: exactly one contributor
    const float theta = (c - radius) / (float)rho; // utils.py:38
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float pr = fmaxf(a[e] - theta, 0.0f);
        x[e] = (x[e] > 0.0f) ? pr : ((x[e] < 0.0f) ? -pr : 0.0f * pr);
    }
}

// ---- K5: AdamW on all N rows of V + l1-ball projection --------------------- //
template <int EPL>
__global__ __launch_bounds__(256) void adamw_l1ball_kernel(float* __restrict__ v, const float* __restrict__ grad_vb,
                                                           int32_t* __restrict__ pos, float* __restrict__ m,
                                                           float* __restrict__ s, int N, int K, AdamWHyper h,
                                                           float radius, float* max_abs_delta, int do_adam,
                                                           int reset_pos, const float* skip_if_below,
                                                           float skip_threshold, float* clear,
                                                           const float* __restrict__ dyn,
                                                           const float* __restrict__ slabs, int nslabs, int slab_rows) {
    if (dyn != nullptr) { h.step_size = dyn[0]; h.bc2_sqrt = dyn[1]; }   // step-dependent scalars from device memory (graphs)
    // device-side stop test of the solver loop (adil.py:614), see zstep_mfma_kernel
    if (skip_if_below != nullptr && *skip_if_below < skip_threshold) {
        if (max_abs_delta != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *max_abs_delta = 0.0f;   // stay stopped
        return;
    }
    if (clear != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *clear = 0.0f;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;                          // whole wave exits together
    int slot = row;
    if (pos != nullptr) {
        slot = pos[row];
        // one wave owns the row: hand the slot table back all -1 for the next batch's adil_pack_codes
        if (reset_pos && slot >= 0 && lane == 0) pos[row] = -1;
    }
    float x[EPL], x_old[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = e * ADIL_WAVE + lane;
        float val = 0.0f;
        if (k < K) {
            const size_t i = (size_t)row * K + k;
            val = v[i];
            x_old[e] = val;
            if (do_adam) {
                float mm = m[i], ss = s[i];
                // the batch gradient row: dense, or still in the producer's per-workgroup slabs (summed here in the fixed
                // order of slab_sum: the reduction launch between adil_grad and this kernel is gone)
                float g = 0.0f;
                if (slot >= 0)
                    g = (nslabs > 0) ? slab_sum(slabs + (size_t)slot * K + k, nslabs, (size_t)slab_rows * K)
                                     : grad_vb[(size_t)slot * K + k];
                val = adamw_elem(val, g, mm, ss, h);
                m[i] = mm; s[i] = ss;
            }
        } else {
            x_old[e] = 0.0f;
        }
        x[e] = val;
    }
    if (radius >= 0.0f) l1ball_row<EPL>(x, lane, radius);
    float dmax = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = e * ADIL_WAVE + lane;
        if (k < K) {
            v[(size_t)row * K + k] = x[e];
            dmax = fmaxf(dmax, fabsf(x[e] - x_old[e]));
        }
    }
    if (max_abs_delta != nullptr) {
        dmax = wave_max(dmax);
        if (lane == 0 && dmax > 0.0f) atomic_max_nonneg(max_abs_delta, dmax);
    }
}

template <int EPL>
__global__ __launch_bounds__(256) void l2ball_kernel(float* __restrict__ x, int N, int K, float radius) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    float val[EPL];
    float ss = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = e * ADIL_WAVE + lane;
        val[e] = (k < K) ? x[(size_t)row * K + k] : 0.0f;
        ss += val[e] * val[e];
    }
    const float nrm = sqrtf(wave_sum(ss));
    const float den = fmaxf(nrm, radius);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int k = e * ADIL_WAVE + lane;
        if (k < K) x[(size_t)row * K + k] = radius * val[e] / den;
    }
}

// ---- K10: ISTA step  v = softshrink(v - step*g, lam) ------------------------ //
__global__ __launch_bounds__(256) void ista_kernel(float* __restrict__ v, const float* __restrict__ g, size_t n,
                                                   float step, float lam) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float t = v[i];
        if (g != nullptr) t = t - step * g[i];
        v[i] = (t > lam) ? (t - lam) : ((t < -lam) ? (t + lam) : 0.0f);
    }
}

// ---- gather + pad the batch's code rows ------------------------------------ //
// vp[b][k] = source row b (k < K, b < B; else 0) with the source either v[index[b]] or the SUM over the per-workgroup
// slabs of a grad_v pass (slab_sum: the reduce launch behind adil_grad folded into this one).  Optionally the same
// values transposed, vpt[a][b] for a < A (rows >= K zero), in fp32 or bf16: the B operand of grad_d, which adil_grad
// otherwise produces with a launch of its own.
template <typename E>
__global__ __launch_bounds__(256) void pack_codes_kernel(const float* __restrict__ v, const int64_t* __restrict__ index,
                                                         int B, int K, int Kp, int Bp, float* __restrict__ vp,
                                                         int32_t* __restrict__ pos, E* __restrict__ vpt, int A,
                                                         const float* __restrict__ slabs, int nslabs, int slab_rows) {
    const int W = (vpt != nullptr) ? A : Kp;                       // A >= Kp always
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * W) return;
    const int b = i / W, k = i - b * W;
    float val = 0.0f;
    if (b < B && k < K) {
        if (nslabs > 0) {
            val = slab_sum(slabs + (size_t)b * K + k, nslabs, (size_t)slab_rows * K);
        } else {
            const int64_t row = (index != nullptr) ? index[b] : (int64_t)b;
            val = v[row * K + k];
            if (k == 0 && pos != nullptr) pos[row] = b;     // batch slot of code row `row` (consumed + reset by K5)
        }
    }
    if (k < Kp) vp[(size_t)b * Kp + k] = val;
    if (vpt != nullptr) {
        if constexpr (sizeof(E) == 4) vpt[(size_t)k * Bp + b] = val; else vpt[(size_t)k * Bp + b] = f32_to_bf16(val);
    }
}

// ---- batched image gather (the data step in front of the path) -------------- //
// dst[b][:] = convert(src[index[b]][:]); 8 elements per thread: 16-byte accesses on the 2-byte side, 2 x 16 B on the
// 4-byte side.  P % 8 == 0.  One pass: B*P*(sizeof(S)+sizeof(D)) bytes.
template <typename S, typename D>
__global__ __launch_bounds__(256) void gather_images_kernel(const S* __restrict__ src, const int64_t* __restrict__ index,
                                                            D* __restrict__ dst, int B, int P8) {
    const int b = blockIdx.y;
    const int64_t row = (index != nullptr) ? index[b] : (int64_t)b;
    const S* s = src + (size_t)row * P8 * 8;
    D* d = dst + (size_t)b * P8 * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P8; i += gridDim.x * blockDim.x) {
        float f[8];
        if constexpr (sizeof(S) == 4) {
            const float4 a = reinterpret_cast<const float4*>(s)[2 * i], c = reinterpret_cast<const float4*>(s)[2 * i + 1];
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
        } else {
            const uint4 a = reinterpret_cast<const uint4*>(s)[i];
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
        }
        if constexpr (sizeof(D) == 4) {
            reinterpret_cast<float4*>(d)[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
            reinterpret_cast<float4*>(d)[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            uint4 o;
            o.x = pack2_bf16(f[0], f[1]); o.y = pack2_bf16(f[2], f[3]); o.z = pack2_bf16(f[4], f[5]); o.w = pack2_bf16(f[6], f[7]);
            reinterpret_cast<uint4*>(d)[i] = o;
        }
    }
}

// ---- K7: inverse of the K x K Gram matrix (symmetric positive definite) ------ //
// One workgroup, in-place Gauss-Jordan without pivoting (stable for SPD matrices) in fp64: the result is the correctly
// rounded fp32 inverse for any conditioning an fp32 LAPACK inverse can handle at all.
// Round 3: the matrix lives in REGISTERS — thread (ti, tj) of a Kt x Kt grid, Kt = ceil(K / 4), owns the 4 x 4 tile of
// rows 4ti.., columns 4tj.. (identity beyond K) — and a step only moves the pivot row and the pivot column through LDS
// (2 x 128 doubles, double-buffered: ONE barrier per step).  Only the Kt^2 threads that own a tile are launched (K = 50:
// 169 threads = 3 waves; K = 100: 625 = 10 waves): a step is ~100 fp64 instructions per wave, and fp64 issues at a
// quarter of the fp32 rate, so idle tiles are not free.  The round-2 kernel kept the whole matrix in LDS and pushed all
// K^2 entries through it in every step (240 KB of LDS traffic and four barriers per step: 46 us at K = 50, ~250 us at
// K = 100).
__global__ __launch_bounds__(1024) void spd_inverse_kernel(const float* __restrict__ a, int K, int Kt, float* __restrict__ out) {
    __shared__ double rowbuf[2][128];
    __shared__ double colbuf[2][128];
    const bool active = (int)threadIdx.x < Kt * Kt;              // the tail of the last wave only keeps the barriers company
    const int ti = active ? threadIdx.x / Kt : 0, tj = active ? threadIdx.x - ti * Kt : 0;
    double t[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * ti + r, j = 4 * tj + c;
            t[r][c] = (i < K && j < K) ? (double)a[(size_t)i * K + j] : (i == j ? 1.0 : 0.0);
        }
    }
    for (int k = 0; k < K; ++k) {
        const int par = k & 1, kt = k >> 2, kr = k & 3;
        if (active && ti == kt) {                          // owners of pivot row k publish their four entries of it
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double val = t[0][c];
#pragma unroll
                for (int r = 1; r < 4; ++r) val = (kr == r) ? t[r][c] : val;
                rowbuf[par][4 * tj + c] = val;
            }
        }
        if (active && tj == kt) {                          // owners of pivot column k
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double val = t[r][0];
#pragma unroll
                for (int c = 1; c < 4; ++c) val = (kr == c) ? t[r][c] : val;
                colbuf[par][4 * ti + r] = val;
            }
        }
        __syncthreads();                                   // the only barrier of the step (buffers alternate)
        if (!active) continue;
        // 1 / a_kk sits on the serial chain of the elimination (every step waits for it): hardware reciprocal + two
        // Newton steps (4 dependent FMAs, ~1e-16 relative) instead of the IEEE division sequence (scale, rcp, 6 FMAs, fix-up)
        const double akk = rowbuf[par][k];
        double piv = __builtin_amdgcn_rcp(akk);
        piv = __builtin_fma(__builtin_fma(-akk, piv, 1.0), piv, piv);
        piv = __builtin_fma(__builtin_fma(-akk, piv, 1.0), piv, piv);
        double rw[4], cl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rw[c] = (4 * tj + c == k) ? piv : rowbuf[par][4 * tj + c] * piv;   // scaled pivot row; pivot -> 1/a_kk
#pragma unroll
        for (int r = 0; r < 4; ++r) cl[r] = colbuf[par][4 * ti + r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * ti + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 4 * tj + c;
                double val;
                if (i == k) val = rw[c];                                           // pivot row: scaled; pivot: its reciprocal
                else val = ((j == k) ? 0.0 : t[r][c]) - cl[r] * rw[c];             // elimination (pivot column: -a_ik / a_kk)
                t[r][c] = val;
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * ti + r, j = 4 * tj + c;
            if (i < K && j < K) out[(size_t)i * K + j] = (float)t[r][c];
        }
    }
}

// ---- K11: per-atom norms / scaling ------------------------------------------ //
// partial[block][k] = sum over the block's rows of d[p][k]^2 ; then a tiny reduce.
__global__ __launch_bounds__(256) void atom_sumsq_partial_kernel(const float* __restrict__ d, int P, int K, int KT,
                                                                 int rows_per_block, float* __restrict__ partial) {
    __shared__ float red[256];
    const int k = threadIdx.x % KT, r = threadIdx.x / KT, R = 256 / KT;
    const int p_begin = blockIdx.x * rows_per_block;
    const int p_end = min(P, p_begin + rows_per_block);
    float acc = 0.0f;
    if (k < K)
        for (int p = p_begin + r; p < p_end; p += R) { const float t = d[(size_t)p * K + k]; acc += t * t; }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (r == 0) {
        for (int rr = 1; rr < R; ++rr) acc += red[rr * KT + k];
        if (k < K) partial[(size_t)blockIdx.x * K + k] = acc;
    }
}
__global__ void atom_norm_finish_kernel(const float* __restrict__ partial, int nblocks, int K, float* __restrict__ norms) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float acc = 0.0f;
    for (int b = 0; b < nblocks; ++b) acc += partial[(size_t)b * K + k];
    norms[k] = sqrtf(acc);
}
__global__ __launch_bounds__(256) void atom_scale_kernel(float* __restrict__ d, size_t n, int K,
                                                         const float* __restrict__ norms, int sphere) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float nk = norms[i % K];
        d[i] = d[i] / (sphere ? nk : fmaxf(nk, 1.0f));
    }
}

// ---- K11: constraint_dict's l1 branch (utils.py:55-56): every (channel, atom) row of H*W pixels onto the l1 ball ---- //
// One workgroup per row; the row is strided in D's layout (element (c, p, k) at (c*HW + p)*K + k).  Rows are far longer
// than the wave-sized code rows of l1ball_row, so the threshold is found without sorting (Michelot 1986): start from all
// entries active, theta = (sum of active |x| - r) / #active, drop the entries with |x| <= theta, repeat until nothing
// drops — at the fixed point theta is Duchi's (cumsum[rho] - r) / rho of the sort-based reference.  Sums in fp64 with a
// fixed reduction tree (bitwise reproducible).  No caller upstream: correctness first, HBM passes second.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    __syncthreads();                                   // red may still be read from the previous call
    if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    a = 0.0; b = 0.0;
    for (int i = 0; i < nw; ++i) { a += red[2 * i]; b += red[2 * i + 1]; }
}

__global__ __launch_bounds__(1024) void atom_l1ball_kernel(float* __restrict__ d, int HW, int K, float radius) {
    __shared__ double red[32];
    const int c = blockIdx.x / K, k = blockIdx.x - c * K;
    float* row = d + (size_t)c * HW * K + k;
    double l1 = 0.0, cnt = 0.0;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) { l1 += (double)fabsf(row[(size_t)p * K]); cnt += 1.0; }
    block_sum2(l1, cnt, red);
    if ((float)l1 < radius) return;                    // strict '<' (utils.py:33): rows inside the ball are untouched
    double theta = (l1 - (double)radius) / cnt;
    for (int it = 0; it < 4096; ++it) {                // terminates after at most HW rounds; a handful in practice
        double s = 0.0, n = 0.0;
        for (int p = threadIdx.x; p < HW; p += blockDim.x) {
            const double a = (double)fabsf(row[(size_t)p * K]);
            if (a > theta) { s += a; n += 1.0; }
        }
        block_sum2(s, n, red);
        if (n == cnt) break;                           // nothing dropped: theta is the threshold
        cnt = n;
        theta = (s - (double)radius) / n;
    }
    const float th = (float)theta;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const float x = row[(size_t)p * K];
        const float pr = fmaxf(fabsf(x) - th, 0.0f);
        row[(size_t)p * K] = (x > 0.0f) ? pr : ((x < 0.0f) ? -pr : 0.0f * pr);
    }
}

// ---- K7: the Gram matrix and D * M^T are MFMA kernels in adil_contract.hip (adil_gram, adil_dict_rightmul) ---- //

// ---- K12: per-image evaluation sums ----------------------------------------- //
// One workgroup of 1024 threads per image; 16-byte loads (8 bf16 / 4 fp32 elements) when the image size and both base
// pointers allow (VEC), element-wise otherwise; per-thread sums -> wave sums -> the 16 wave partials added in a fixed order
// (bitwise reproducible).  Round 3: the round-1 kernel (256 threads per image, 2-byte loads) read its 2 x 154 MB at
// 1.1 TB/s (272 us per 512-image batch in profiles/r03_transfer_bench_stats.md).
template <typename T, bool VEC>
__global__ __launch_bounds__(1024) void image_metrics_kernel(const T* __restrict__ adv, const T* __restrict__ x, int P,
                                                             float* __restrict__ sq_err, float* __restrict__ sq_norm) {
    __shared__ float red[2][16];
    const size_t base = (size_t)blockIdx.x * P;
    float e = 0.0f, q = 0.0f;
    if constexpr (VEC) {
        constexpr int N = 16 / sizeof(T);
        const uint4* av = reinterpret_cast<const uint4*>(adv + base);
        const uint4* xv = reinterpret_cast<const uint4*>(x + base);
        for (int i = threadIdx.x; i < P / N; i += 1024) {
            const uint4 a4 = av[i], x4 = xv[i];
            const unsigned aw[4] = {a4.x, a4.y, a4.z, a4.w}, xw[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (sizeof(T) == 4) {
                    const float xf = __uint_as_float(xw[j]), dv = __uint_as_float(aw[j]) - xf;
                    e += dv * dv; q += xf * xf;
                } else {
                    const float x0 = __uint_as_float(xw[j] << 16), x1 = __uint_as_float(xw[j] & 0xffff0000u);
                    const float d0 = __uint_as_float(aw[j] << 16) - x0, d1 = __uint_as_float(aw[j] & 0xffff0000u) - x1;
                    e += d0 * d0; e += d1 * d1; q += x0 * x0; q += x1 * x1;
                }
            }
        }
    } else {
        for (int p = threadIdx.x; p < P; p += 1024) {
            const float xv = Elem<T>::load(x, base + p);
            const float dv = Elem<T>::load(adv, base + p) - xv;
            e += dv * dv;
            q += xv * xv;
        }
    }
    e = wave_sum(e); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = e; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float se = 0.0f, sn = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { se += red[0][w]; sn += red[1][w]; }
        sq_err[blockIdx.x] = se;
        sq_norm[blockIdx.x] = sn;
    }
}

// =========================================================================== //
// C ABI
// =========================================================================== //
static inline int pow2_at_least(int k) { int t = 16; while (t < k) t <<= 1; return t; }
static inline int stream_grid(size_t work_items, int per_block) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;                          // 256 CUs x 8 blocks, grid-stride the rest
    return (int)b;
}

extern "C" int adil_abi_version(void) { return 7; }
extern "C" int adil_max_atoms(void) { return ADIL_MAX_ATOMS; }

extern "C" int adil_pack_codes(const float* v, const int64_t* index, int B, int K, float* vp, int32_t* pos, void* vpt,
                               int vpt_dtype, const float* slabs, int nslabs, int slab_rows, void* stream) {
    ADIL_ENTER();
    if (!vp || B <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (nslabs > 0 ? (!slabs || slab_rows < B || pos != nullptr) : !v) return ADIL_EINVAL;
    if (nslabs < 0 || (vpt != nullptr && ((uintptr_t)vpt & 15))) return ADIL_EINVAL;
    const int Kp = round_up(K, 16), Bp = round_up(B, 32), A = adil_grad_code_rows(K);
    const int total = Bp * (vpt != nullptr ? A : Kp);
    const int threads = nslabs > 0 ? 64 : 256;                   // slab sums: spread the latency-bound lanes over all CUs
    const dim3 grid((total + threads - 1) / threads), block(threads);
    hipStream_t st = (hipStream_t)stream;
    if (vpt == nullptr || vpt_dtype == ADIL_F32)
        hipLaunchKernelGGL(pack_codes_kernel<float>, grid, block, 0, st, v, index, B, K, Kp, Bp, vp, pos, (float*)vpt, A, slabs,
                           nslabs, slab_rows);
    else if (vpt_dtype == ADIL_BF16)
        hipLaunchKernelGGL(pack_codes_kernel<bf16_t>, grid, block, 0, st, v, index, B, K, Kp, Bp, vp, pos, (bf16_t*)vpt, A, slabs,
                           nslabs, slab_rows);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_gather_images(const void* src, int src_dtype, const int64_t* index, void* dst, int dst_dtype, int B,
                                  int P, void* stream) {
    ADIL_ENTER();
    if (!src || !dst || B <= 0 || P <= 0 || (P & 7)) return ADIL_EINVAL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return ADIL_EINVAL;
    const int P8 = P / 8;
    int gx = (P8 + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid(gx, B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (src_dtype == ADIL_F32 && dst_dtype == ADIL_F32)
        hipLaunchKernelGGL((gather_images_kernel<float, float>), grid, block, 0, st, (const float*)src, index, (float*)dst, B, P8);
    else if (src_dtype == ADIL_F32 && dst_dtype == ADIL_BF16)
        hipLaunchKernelGGL((gather_images_kernel<float, bf16_t>), grid, block, 0, st, (const float*)src, index, (bf16_t*)dst, B, P8);
    else if (src_dtype == ADIL_BF16 && dst_dtype == ADIL_F32)
        hipLaunchKernelGGL((gather_images_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)src, index, (float*)dst, B, P8);
    else if (src_dtype == ADIL_BF16 && dst_dtype == ADIL_BF16)
        hipLaunchKernelGGL((gather_images_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)src, index, (bf16_t*)dst, B, P8);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_spd_inverse(const float* a, int K, float* out, void* stream) {
    ADIL_ENTER();
    if (!a || !out || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    const int Kt = (K + 3) / 4;
    hipLaunchKernelGGL(spd_inverse_kernel, dim3(1), dim3(round_up(Kt * Kt, 64)), 0, (hipStream_t)stream, a, K, Kt, out);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_adamw_clamp(float* p, const void* g, int g_dtype, float* m, float* s, size_t n, float decay,
                                float b1, float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi,
                                float* max_abs_delta, const float* dyn_scalars, void* stream) {
    ADIL_ENTER();
    if (!p || !g || !m || !s || n == 0) return ADIL_EINVAL;
    if (((uintptr_t)p | (uintptr_t)m | (uintptr_t)s | (uintptr_t)g) & 15) return ADIL_EINVAL;  // 16-B vector access
    AdamWHyper h{decay, b1, b2, eps, step_size, bc2_sqrt};
    const int grid = stream_grid(n / 4 + 1, 256);
    if (g_dtype == ADIL_F32)
        hipLaunchKernelGGL((adamw_clamp_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p,
                           (const float*)g, m, s, n, h, lo, hi, max_abs_delta, dyn_scalars, (unsigned*)nullptr);
    else if (g_dtype == ADIL_BF16)
        hipLaunchKernelGGL((adamw_clamp_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p,
                           (const bf16_t*)g, m, s, n, h, lo, hi, max_abs_delta, dyn_scalars, (unsigned*)nullptr);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_dict_to_fp8(const float* d, size_t n, void* d_fp8, void* stream) {
    ADIL_ENTER();
    if (!d || !d_fp8 || n == 0 || (n & 3) || ((uintptr_t)d & 15) || ((uintptr_t)d_fp8 & 3)) return ADIL_EINVAL;
    hipLaunchKernelGGL(dict_to_fp8_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, d, n / 4,
                       (unsigned*)d_fp8);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_adamw_clamp_fp8(float* p, const void* g, int g_dtype, float* m, float* s, size_t n, float decay,
                                    float b1, float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi,
                                    float* max_abs_delta, const float* dyn_scalars, void* p_fp8, void* stream) {
    ADIL_ENTER();
    if (!p || !g || !m || !s || !p_fp8 || n == 0 || (n & 3)) return ADIL_EINVAL;
    if ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)s | (uintptr_t)g) & 15) || ((uintptr_t)p_fp8 & 3)) return ADIL_EINVAL;
    AdamWHyper h{decay, b1, b2, eps, step_size, bc2_sqrt};
    const int grid = stream_grid(n / 4 + 1, 256);
    if (g_dtype == ADIL_F32)
        hipLaunchKernelGGL((adamw_clamp_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (const float*)g, m, s,
                           n, h, lo, hi, max_abs_delta, dyn_scalars, (unsigned*)p_fp8);
    else if (g_dtype == ADIL_BF16)
        hipLaunchKernelGGL((adamw_clamp_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (const bf16_t*)g, m,
                           s, n, h, lo, hi, max_abs_delta, dyn_scalars, (unsigned*)p_fp8);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

static int launch_adamw_l1ball(float* v, const float* grad_vb, int32_t* pos, float* m, float* s, int N, int K,
                               AdamWHyper h, float radius, float* max_abs_delta, int do_adam, int reset_pos,
                               hipStream_t st, const float* skip_if_below = nullptr, float skip_threshold = 0.0f,
                               float* clear = nullptr, const float* dyn = nullptr, const float* slabs = nullptr,
                               int nslabs = 0, int slab_rows = 0) {
    // one wave per row; with the slab reduction inside, one wave per WORKGROUP: the batch rows each walk ~240 slabs in
    // dependent rounds of 32 loads, and 512 single-wave workgroups spread over all CUs where 128 four-wave ones fill half
    const int rows_per_block = nslabs > 0 ? 1 : 4;
    const dim3 grid((N + rows_per_block - 1) / rows_per_block), block(64 * rows_per_block);
    if (K <= 64)
        hipLaunchKernelGGL(adamw_l1ball_kernel<1>, grid, block, 0, st, v, grad_vb, pos, m, s, N, K, h, radius,
                           max_abs_delta, do_adam, reset_pos, skip_if_below, skip_threshold, clear, dyn, slabs, nslabs, slab_rows);
    else
        hipLaunchKernelGGL(adamw_l1ball_kernel<2>, grid, block, 0, st, v, grad_vb, pos, m, s, N, K, h, radius,
                           max_abs_delta, do_adam, reset_pos, skip_if_below, skip_threshold, clear, dyn, slabs, nslabs, slab_rows);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_adamw_l1ball(float* v, const float* grad_vb, int32_t* pos, int reset_pos, float* m, float* s, int N,
                                 int K, float decay, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                                 float radius, float* max_abs_delta, const float* skip_if_below, float skip_threshold,
                                 float* clear, const float* dyn_scalars, const float* slabs, int nslabs, int slab_rows,
                                 void* stream) {
    ADIL_ENTER();
    if (!v || !m || !s || N <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (nslabs < 0 || (nslabs > 0 && (!slabs || slab_rows <= 0))) return ADIL_EINVAL;
    if (!grad_vb && !pos && nslabs == 0) return ADIL_EINVAL;      // without a slot table every row reads its own gradient row
    if (!pos && nslabs > 0 && slab_rows < N) return ADIL_EINVAL;
    AdamWHyper h{decay, b1, b2, eps, step_size, bc2_sqrt};
    return launch_adamw_l1ball(v, grad_vb, pos, m, s, N, K, h, radius, max_abs_delta, 1, reset_pos, (hipStream_t)stream,
                               skip_if_below, skip_threshold, clear, dyn_scalars, slabs, nslabs, slab_rows);
}

extern "C" int adil_l1ball_project(float* x, int N, int K, float radius, void* stream) {
    ADIL_ENTER();
    if (!x || N <= 0 || K <= 0 || K > ADIL_MAX_ATOMS || radius < 0.0f) return ADIL_EINVAL;
    AdamWHyper h{};
    return launch_adamw_l1ball(x, nullptr, nullptr, nullptr, nullptr, N, K, h, radius, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int adil_l2ball_project(float* x, int N, int K, float radius, void* stream) {
    ADIL_ENTER();
    if (!x || N <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    const dim3 grid((N + 3) / 4), block(256);
    if (K <= 64) hipLaunchKernelGGL(l2ball_kernel<1>, grid, block, 0, (hipStream_t)stream, x, N, K, radius);
    else hipLaunchKernelGGL(l2ball_kernel<2>, grid, block, 0, (hipStream_t)stream, x, N, K, radius);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_ista_step(float* v, const float* g, size_t n, float step, float lam, void* stream) {
    ADIL_ENTER();
    if (!v || n == 0) return ADIL_EINVAL;
    hipLaunchKernelGGL(ista_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, v, g, n, step, lam);
    ADIL_CHECK_LAUNCH();
    return 0;
}

static const int kAtomBlocks = 512;
extern "C" size_t adil_atom_workspace_bytes(int P, int K) { (void)P; return (size_t)kAtomBlocks * K * sizeof(float); }

extern "C" int adil_atom_norms(const float* d, int P, int K, float* norms, void* ws, size_t ws_bytes, void* stream) {
    ADIL_ENTER();
    if (!d || !norms || !ws || P <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (ws_bytes < adil_atom_workspace_bytes(P, K)) return ADIL_EWORKSPACE;
    const int rows_per_block = (P + kAtomBlocks - 1) / kAtomBlocks;
    const int nblocks = (P + rows_per_block - 1) / rows_per_block;
    const int KT = pow2_at_least(K);
    hipLaunchKernelGGL(atom_sumsq_partial_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, d, P, K, KT,
                       rows_per_block, (float*)ws);
    ADIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(atom_norm_finish_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, (const float*)ws, nblocks, K,
                       norms);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_atom_scale(float* d, int P, int K, const float* norms, int sphere, void* stream) {
    ADIL_ENTER();
    if (!d || !norms || P <= 0 || K <= 0) return ADIL_EINVAL;
    const size_t n = (size_t)P * K;
    hipLaunchKernelGGL(atom_scale_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, d, n, K, norms,
                       sphere);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_atom_l1ball_project(float* d, int C, int HW, int K, float radius, void* stream) {
    ADIL_ENTER();
    if (!d || C <= 0 || HW <= 0 || K <= 0 || radius < 0.0f) return ADIL_EINVAL;
    hipLaunchKernelGGL(atom_l1ball_kernel, dim3(C * K), dim3(1024), 0, (hipStream_t)stream, d, HW, K, radius);
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_image_metrics(const void* adv, const void* x, int B, int P, int dtype, float* sq_err,
                                  float* sq_norm, void* stream) {
    ADIL_ENTER();
    if (!adv || !x || !sq_err || !sq_norm || B <= 0 || P <= 0) return ADIL_EINVAL;
    const int esz = dtype == ADIL_F32 ? 4 : 2;
    // 16-byte loads: every image row must start on a 16-byte boundary and hold a whole number of vectors
    const bool vec = (((uintptr_t)adv | (uintptr_t)x) % 16 == 0) && (((size_t)P * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == ADIL_F32) {
        if (vec) hipLaunchKernelGGL((image_metrics_kernel<float, true>), dim3(B), dim3(1024), 0, st, (const float*)adv, (const float*)x, P, sq_err, sq_norm);
        else hipLaunchKernelGGL((image_metrics_kernel<float, false>), dim3(B), dim3(1024), 0, st, (const float*)adv, (const float*)x, P, sq_err, sq_norm);
    } else if (dtype == ADIL_BF16) {
        if (vec) hipLaunchKernelGGL((image_metrics_kernel<bf16_t, true>), dim3(B), dim3(1024), 0, st, (const bf16_t*)adv, (const bf16_t*)x, P, sq_err, sq_norm);
        else hipLaunchKernelGGL((image_metrics_kernel<bf16_t, false>), dim3(B), dim3(1024), 0, st, (const bf16_t*)adv, (const bf16_t*)x, P, sq_err, sq_norm);
    } else {
        return ADIL_EINVAL;
    }
    ADIL_CHECK_LAUNCH();
    return 0;
}

// =========================================================================== //
// Frozen-classifier epilogues: eval-mode BatchNorm (per-channel scale/shift) + optional residual + optional ReLU
// in ONE pass, forward and input-gradient backward.  Not part of the ADiL maths — they only remove elementwise
// passes PyTorch would run as separate kernels around every convolution of the frozen network (bias/BN, add, ReLU).
// Channel of flat element i is (i / inner) % C: inner = 1 for channels_last storage, H*W for contiguous NCHW.
// =========================================================================== //
// LAYOUT 0: channels_last (channel = i % C, the VEC elements of a thread are VEC consecutive channels, C % VEC == 0)
// LAYOUT 1: NCHW          (channel = (i / inner) % C, constant over the VEC elements, inner % VEC == 0)
// LAYOUT 2: anything else (per-element index arithmetic)
template <int VEC, int LAYOUT>
__device__ __forceinline__ void channel_params(const float* __restrict__ tab, size_t base, int C, int inner, float (&o)[VEC]) {
    if (LAYOUT == 0) {
        const float4* p = reinterpret_cast<const float4*>(tab + (base % (size_t)C));
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) { const float4 t = p[q]; o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w; }
    } else if (LAYOUT == 1) {
        const float t = tab[(base / (size_t)inner) % (size_t)C];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = t;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = tab[((base + j) / (size_t)inner) % (size_t)C];
    }
}

template <typename T, int VEC> struct ActVec;
template <> struct ActVec<float, 4> {
    static __device__ __forceinline__ void load(const float* p, size_t i, float (&o)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p + i); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, size_t i, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct ActVec<bf16_t, 8> {
    static __device__ __forceinline__ void load(const bf16_t* p, size_t i, float (&o)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + i);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] = __uint_as_float(w[q] << 16); o[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, size_t i, const float (&o)[8]) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = pack2_bf16(o[2 * q], o[2 * q + 1]);
        *reinterpret_cast<uint4*>(p + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

template <typename T, int VEC, int LAYOUT>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y,
                                                             size_t n, int C, int inner, int relu) {
    const size_t nv = n / VEC;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const size_t base = i * VEC;
        float v[VEC], sc[VEC], sh[VEC];
        ActVec<T, VEC>::load(x, base, v);
        channel_params<VEC, LAYOUT>(scale, base, C, inner, sc);
        channel_params<VEC, LAYOUT>(shift, base, C, inner, sh);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
        if (res != nullptr) {                                    // uniform
            float r[VEC];
            ActVec<T, VEC>::load(res, base, r);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] += r[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = fmaxf(v[j], 0.0f);
        }
        ActVec<T, VEC>::store(y, base, v);
    }
}

// gx = mask * g * scale[c];  gres (optional) = mask * g;  mask = (y > 0) if relu else 1
template <typename T, int VEC, int LAYOUT>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                             const float* __restrict__ scale, T* __restrict__ gx,
                                                             T* __restrict__ gres, size_t n, int C, int inner, int relu) {
    const size_t nv = n / VEC;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const size_t base = i * VEC;
        float gv[VEC], sc[VEC];
        ActVec<T, VEC>::load(g, base, gv);
        channel_params<VEC, LAYOUT>(scale, base, C, inner, sc);
        if (relu) {                                              // uniform
            float yv[VEC];
            ActVec<T, VEC>::load(y, base, yv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) gv[j] = (yv[j] > 0.0f) ? gv[j] : 0.0f;
        }
        if (gres != nullptr) ActVec<T, VEC>::store(gres, base, gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) gv[j] *= sc[j];
        ActVec<T, VEC>::store(gx, base, gv);
    }
}

static inline int act_layout(size_t n, int C, int inner, int vec) {
    if (inner == 1 && C % vec == 0) return 0;
    if (inner % vec == 0) return 1;
    return 2;
}

#define ADIL_ACT_DISPATCH(KERNEL, T, VEC, ...)                                                                      \
    do {                                                                                                            \
        const int lay = act_layout(n, C, inner, VEC);                                                               \
        const dim3 grid(stream_grid(n / VEC, 256)), block(256);                                                     \
        if (lay == 0) hipLaunchKernelGGL((KERNEL<T, VEC, 0>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);     \
        else if (lay == 1) hipLaunchKernelGGL((KERNEL<T, VEC, 1>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, VEC, 2>), grid, block, 0, (hipStream_t)stream, __VA_ARGS__);              \
    } while (0)

extern "C" int adil_affine_act_fwd(const void* x, const void* res, const float* scale, const float* shift, void* y,
                                   size_t n, int C, int inner, int relu, int dtype, void* stream) {
    ADIL_ENTER();
    if (!x || !scale || !shift || !y || n == 0 || C <= 0 || inner <= 0 || (n % 8) != 0) return ADIL_EINVAL;
    if (dtype == ADIL_F32)
        ADIL_ACT_DISPATCH(affine_act_fwd_kernel, float, 4, (const float*)x, (const float*)res, scale, shift, (float*)y, n, C, inner, relu);
    else if (dtype == ADIL_BF16)
        ADIL_ACT_DISPATCH(affine_act_fwd_kernel, bf16_t, 8, (const bf16_t*)x, (const bf16_t*)res, scale, shift, (bf16_t*)y, n, C, inner, relu);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}

extern "C" int adil_affine_act_bwd(const void* g, const void* y, const float* scale, void* gx, void* gres, size_t n,
                                   int C, int inner, int relu, int dtype, void* stream) {
    ADIL_ENTER();
    if (!g || !scale || !gx || n == 0 || C <= 0 || inner <= 0 || (n % 8) != 0 || (relu && !y)) return ADIL_EINVAL;
    if (dtype == ADIL_F32)
        ADIL_ACT_DISPATCH(affine_act_bwd_kernel, float, 4, (const float*)g, (const float*)y, scale, (float*)gx, (float*)gres, n, C, inner, relu);
    else if (dtype == ADIL_BF16)
        ADIL_ACT_DISPATCH(affine_act_bwd_kernel, bf16_t, 8, (const bf16_t*)g, (const bf16_t*)y, scale, (bf16_t*)gx, (bf16_t*)gres, n, C, inner, relu);
    else
        return ADIL_EINVAL;
    ADIL_CHECK_LAUNCH();
    return 0;
}
