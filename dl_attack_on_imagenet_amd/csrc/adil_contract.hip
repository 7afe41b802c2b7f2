// ADiL dictionary contractions for gfx950.
//   synth :  out = x + vp D^T                 (B x Kp)(Kp x P), fused add / clamps
//   grad  :  grad_d = g^T vp (P x K) ,  grad_vb = g D (B x K)
// D is P x K row-major (atom innermost), image streams are B x P row-major.
#include "adil_common.h"

#define TILE_PX 64     // pixels per workgroup tile
#define TILE_B 32      // batch rows per step

// --------------------------------------------------------------------------- //
// K1 (+K9): synthesis.  One workgroup owns a 64-pixel slice of D (staged once in
// LDS, row stride odd -> conflict-free column walk) and sweeps the whole batch.
// --------------------------------------------------------------------------- //
template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(const T* __restrict__ x, const float* __restrict__ d,
                                                    const float* __restrict__ vp, T* __restrict__ out, int B, int P,
                                                    int K, int Kp, float delta_clamp, int pixel_clamp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ks = Kp + 1;
    float* sd = smem;                       // [TILE_PX][Ks]
    float* sv = smem + TILE_PX * Ks;        // [TILE_B][Kp]  (16-B aligned: TILE_PX*Ks*4 is a multiple of 16)
    const int p0 = blockIdx.x * TILE_PX;
    const int npx = min(TILE_PX, P - p0);
    // stage the D slice: rows p0..p0+npx are one contiguous run of npx*K floats
    for (int i = threadIdx.x; i < TILE_PX * Kp; i += 256) {
        const int r = i / Kp, k = i - r * Kp;
        sd[r * Ks + k] = (r < npx && k < K) ? d[(size_t)(p0 + r) * K + k] : 0.0f;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int p = p0 + tx;
    const int Bp = (B + TILE_B - 1) / TILE_B * TILE_B;
    for (int b0 = 0; b0 < Bp; b0 += TILE_B) {
        __syncthreads();
        for (int i = threadIdx.x; i < TILE_B * Kp / 4; i += 256)
            reinterpret_cast<float4*>(sv)[i] = reinterpret_cast<const float4*>(vp + (size_t)b0 * Kp)[i];
        __syncthreads();
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        for (int k = 0; k < Kp; k += 4) {
            const float d0 = sd[tx * Ks + k], d1 = sd[tx * Ks + k + 1], d2 = sd[tx * Ks + k + 2], d3 = sd[tx * Ks + k + 3];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(&sv[(ty * 8 + i) * Kp + k]);
                acc[i] = fmaf(a.x, d0, acc[i]);
                acc[i] = fmaf(a.y, d1, acc[i]);
                acc[i] = fmaf(a.z, d2, acc[i]);
                acc[i] = fmaf(a.w, d3, acc[i]);
            }
        }
        if (tx < npx) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = b0 + ty * 8 + i;
                if (b < B) {
                    float dv = acc[i];
                    if (delta_clamp >= 0.0f) dv = fminf(fmaxf(dv, -delta_clamp), delta_clamp);
                    const size_t o = (size_t)b * P + p;
                    float r = dv;
                    if (x != nullptr) r += Elem<T>::load(x, o);
                    if (pixel_clamp) r = fminf(fmaxf(r, 0.0f), 1.0f);
                    Elem<T>::store(out, o, r);
                }
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// K3: grad_d tile (64 px x K) = sum over the whole batch of g^T vp.
// thread (tx = pixel, ty) owns atoms k = ty, ty+4, ...
// --------------------------------------------------------------------------- //
template <typename T, int KC>
__global__ __launch_bounds__(256) void grad_d_kernel(const T* __restrict__ g, const float* __restrict__ vp,
                                                     float* __restrict__ grad_d, int B, int P, int K, int Kp,
                                                     int accumulate) {
    __shared__ float sg[TILE_B][TILE_PX + 1];
    extern __shared__ __attribute__((aligned(16))) float sv[];     // [TILE_B][Kp]
    const int p0 = blockIdx.x * TILE_PX;
    const int npx = min(TILE_PX, P - p0);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float acc[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) acc[i] = 0.0f;
    const int Bp = (B + TILE_B - 1) / TILE_B * TILE_B;
    for (int b0 = 0; b0 < Bp; b0 += TILE_B) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = ty * 8 + i, b = b0 + r;
            sg[r][tx] = (b < B && tx < npx) ? Elem<T>::load(g, (size_t)b * P + p0 + tx) : 0.0f;
        }
        for (int i = threadIdx.x; i < TILE_B * Kp / 4; i += 256)
            reinterpret_cast<float4*>(sv)[i] = reinterpret_cast<const float4*>(vp + (size_t)b0 * Kp)[i];
        __syncthreads();
        for (int r = 0; r < TILE_B; ++r) {
            const float gv = sg[r][tx];
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
                if (4 * kk < Kp) acc[kk] = fmaf(gv, sv[r * Kp + ty + 4 * kk], acc[kk]);
        }
    }
    if (tx < npx) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = ty + 4 * kk;
            if (k < K) {
                const size_t o = (size_t)(p0 + tx) * K + k;
                grad_d[o] = accumulate ? grad_d[o] + acc[kk] : acc[kk];
            }
        }
    }
}

// --------------------------------------------------------------------------- //
// K2: grad_vb partials.  Workgroup (chunk c of the pixel axis, batch block bb);
// thread (b = tid&31, kq = tid>>5) owns atoms k = kq, kq+8, ...
// --------------------------------------------------------------------------- //
template <typename T, int KC>
__global__ __launch_bounds__(256) void grad_v_partial_kernel(const T* __restrict__ g, const float* __restrict__ d,
                                                             float* __restrict__ partial, int B, int P, int K, int Kp,
                                                             int tiles_per_chunk) {
    __shared__ float sg[TILE_B][TILE_PX + 1];
    extern __shared__ __attribute__((aligned(16))) float sd[];     // [TILE_PX][Kp]
    const int b0 = blockIdx.y * TILE_B;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int bl = threadIdx.x & 31, kq = threadIdx.x >> 5;
    float acc[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) acc[i] = 0.0f;
    const int ntiles = (P + TILE_PX - 1) / TILE_PX;
    const int t_begin = blockIdx.x * tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + tiles_per_chunk);
    for (int t = t_begin; t < t_end; ++t) {
        const int p0 = t * TILE_PX;
        const int npx = min(TILE_PX, P - p0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = ty * 8 + i, b = b0 + r;
            sg[r][tx] = (b < B && tx < npx) ? Elem<T>::load(g, (size_t)b * P + p0 + tx) : 0.0f;
        }
        for (int i = threadIdx.x; i < TILE_PX * Kp; i += 256) {
            const int r = i / Kp, k = i - r * Kp;
            sd[i] = (r < npx && k < K) ? d[(size_t)(p0 + r) * K + k] : 0.0f;
        }
        __syncthreads();
        for (int px = 0; px < TILE_PX; ++px) {
            const float gv = sg[bl][px];
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
                if (8 * kk < Kp) acc[kk] = fmaf(gv, sd[px * Kp + kq + 8 * kk], acc[kk]);
        }
    }
    const int Bp = gridDim.y * TILE_B;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const int k = kq + 8 * kk;
        if (k < Kp) partial[((size_t)blockIdx.x * Bp + b0 + bl) * Kp + k] = acc[kk];
    }
}

__global__ __launch_bounds__(256) void grad_v_reduce_kernel(const float* __restrict__ partial, int nchunks, int Bp,
                                                            int Kp, int B, int K, float* __restrict__ grad_vb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i - b * K;
    float acc = 0.0f;
    for (int c = 0; c < nchunks; ++c) acc += partial[((size_t)c * Bp + b) * Kp + k];
    grad_vb[i] = acc;
}

// =========================================================================== //
// C ABI
// =========================================================================== //
static const int kGradChunks = 64;

static int set_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

extern "C" size_t adil_grad_workspace_bytes(int B, int P, int K) {
    (void)P;
    const size_t Bp = round_up(B, TILE_B), Kp = round_up(K, 16);
    return (size_t)kGradChunks * Bp * Kp * sizeof(float);
}

template <typename T>
static int launch_synth(const void* x, const float* d, const float* vp, void* out, int B, int P, int K,
                        float delta_clamp, int pixel_clamp, hipStream_t st) {
    const int Kp = round_up(K, 16);
    const size_t lds = ((size_t)TILE_PX * (Kp + 1) + (size_t)TILE_B * Kp) * sizeof(float);
    int rc = set_lds((const void*)synth_kernel<T>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(synth_kernel<T>, dim3((P + TILE_PX - 1) / TILE_PX), dim3(256), lds, st, (const T*)x, d, vp,
                       (T*)out, B, P, K, Kp, delta_clamp, pixel_clamp);
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

template <typename T, int KC>
static int launch_grad(const void* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                       int K, int accumulate_d, void* ws, hipStream_t st) {
    const int Kp = round_up(K, 16), Bp = round_up(B, TILE_B);
    const int ntiles = (P + TILE_PX - 1) / TILE_PX;
    if (grad_d != nullptr) {
        const size_t lds = (size_t)TILE_B * Kp * sizeof(float);
        hipLaunchKernelGGL((grad_d_kernel<T, KC>), dim3(ntiles), dim3(256), lds, st, (const T*)g, vp, grad_d, B, P, K, Kp,
                           accumulate_d);
        ADIL_CHECK_LAUNCH();
    }
    if (grad_vb != nullptr) {
        const int tiles_per_chunk = (ntiles + kGradChunks - 1) / kGradChunks;
        const int nchunks = (ntiles + tiles_per_chunk - 1) / tiles_per_chunk;
        const size_t lds = (size_t)TILE_PX * Kp * sizeof(float);
        hipLaunchKernelGGL((grad_v_partial_kernel<T, (KC + 1) / 2>), dim3(nchunks, Bp / TILE_B), dim3(256), lds, st,
                           (const T*)g, d, (float*)ws, B, P, K, Kp, tiles_per_chunk);
        ADIL_CHECK_LAUNCH();
        hipLaunchKernelGGL(grad_v_reduce_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, (const float*)ws, nchunks,
                           Bp, Kp, B, K, grad_vb);
        ADIL_CHECK_LAUNCH();
    }
    return 0;
}

template <typename T>
static int dispatch_grad(const void* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                         int K, int accumulate_d, void* ws, hipStream_t st) {
    const int Kp = round_up(K, 16);
    if (Kp <= 16) return launch_grad<T, 4>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
    if (Kp <= 32) return launch_grad<T, 8>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
    if (Kp <= 64) return launch_grad<T, 16>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
    return launch_grad<T, 32>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, st);
}

extern "C" int adil_grad(const void* g, const float* d, const float* vp, float* grad_d, float* grad_vb, int B, int P,
                         int K, int dtype, int accumulate_d, void* ws, size_t ws_bytes, void* stream) {
    ADIL_ENTER();
    if (!g || B <= 0 || P <= 0 || K <= 0 || K > ADIL_MAX_ATOMS) return ADIL_EINVAL;
    if (grad_d == nullptr && grad_vb == nullptr) return ADIL_EINVAL;
    if (grad_d != nullptr && !vp) return ADIL_EINVAL;
    if (grad_vb != nullptr) {
        if (!d || !ws) return ADIL_EINVAL;
        if (ws_bytes < adil_grad_workspace_bytes(B, P, K)) return ADIL_EWORKSPACE;
    }
    if (dtype == ADIL_F32) return dispatch_grad<float>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, (hipStream_t)stream);
    if (dtype == ADIL_BF16) return dispatch_grad<bf16_t>(g, d, vp, grad_d, grad_vb, B, P, K, accumulate_d, ws, (hipStream_t)stream);
    return ADIL_EINVAL;
}
