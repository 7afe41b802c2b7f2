/*
 * adil_hip.h — C ABI of libadil_hip.so, the MI355X (gfx950) kernels of the ADiL
 * (Adversarial Dictionary Learning) attack hot path.
 *
 * The upstream reference (flavie-yuan-liu/DL_attack_on_ImageNet) is pure Python on
 * PyTorch: it has no FFI.  Each entry point below replaces a *sequence of torch
 * ops* on the reference's hot path; the reference interface it replaces is cited
 * as file:line.  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library keeps no
 *     state and allocates nothing persistent; scratch comes in through `ws`.
 *   - `stream` is the caller's hipStream_t (as void*; 0 = default stream).
 *     Kernels are launched asynchronously on it; nothing synchronises.
 *   - return value: 0 on success, otherwise a hipError_t value or one of the
 *     ADIL_E* codes (< 0).  Nothing throws across the boundary.
 *   - the dictionary D is the reference's (C,H,W,K) fp32 tensor, i.e. a
 *     row-major P x K matrix, P = C*H*W, atom index innermost (adil.py:20,25).
 *   - image-shaped streams (x, x_adv, g, z) are row-major B x P, element type
 *     chosen by `dtype` (ADIL_F32 or ADIL_BF16); codes V are fp32 N x K.
 *   - "packed codes": vp is [Bp][Kp] fp32 with Bp = roundup(B,32),
 *     Kp = roundup(K,16), zero padded (see adil_pack_codes).
 */
#ifndef ADIL_HIP_H
#define ADIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADIL_F32 0
#define ADIL_BF16 1

#define ADIL_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported K) */
#define ADIL_EWORKSPACE (-2) /* workspace too small */

/* ABI version of this header; bumped on any signature change.  2: frozen-classifier entry points.  3: batch-slot table
 * written by adil_pack_codes and consumed + reset by adil_adamw_l1ball, adil_gather_images, adil_spd_inverse,
 * adil_synth_fp8.  4: device-side stop test arguments of adil_zstep / adil_adamw_l1ball.  5: dyn_scalars of the AdamW
 * entry points incl. adil_zstep (hipGraph replay of the learning step and of the inference iterations).  6: the two
 * helper launches of a gradient pass moved into their neighbours — adil_pack_codes can emit the transposed codes that
 * adil_grad needs (`vpt`) and can take its rows from the grad_v partial sums ("slabs") of a preceding adil_grad;
 * adil_grad can leave the reduction of those slabs to its consumer (`nslabs_out`); adil_adamw_l1ball consumes them.
 * 7: adil_zstep_codes — the z-step of a DDrague iteration also produces the next iteration's codes (as slabs); the
 * persistent fp8 copy of the dictionary (adil_dict_to_fp8, adil_adamw_clamp_fp8, adil_synth_fp8_packed). */
int adil_abi_version(void);

/* Largest K (atoms) the kernels support. */
int adil_max_atoms(void);

/* Bytes of scratch adil_grad needs for this problem size. */
size_t adil_grad_workspace_bytes(int B, int P, int K);

/* Rows A of the transposed code matrix vpt [A][roundup(B,32)] that adil_grad contracts against (K rounded up to the
 * atom tiling of the kernels: 32, 64 or 128); 0 for an unsupported K. */
int adil_grad_code_rows(int K);

/* Where, inside adil_grad's workspace, the grad_v partial sums live when their reduction is left to the consumer
 * (adil_grad's nslabs_out): `*nslabs_out` slabs of [roundup(B,32)][K] fp32 each, starting this many bytes into ws. */
size_t adil_grad_slab_offset(int B, int P, int K);

/* Gather + pad the batch's code rows:  vp[b][k] = v[index[b]][k] (0 for k>=K, b>=B).
 * Replaces the advanced-indexing `self.v[index, :]` of Attack_dict_model.forward
 * (adil.py:25).  index may be NULL (rows 0..B-1, as adil.py:600 `range(n_img)`).
 * pos (optional, one int32 per row of v, all -1 on entry): pos[index[b]] = b — the batch-slot table that
 * adil_adamw_l1ball consumes; it is what autograd's scatter of the batch gradient into a dense (N,K) grad does.
 * vpt (optional): also the transposed copy vpt[a][b] = vp[b][a], [adil_grad_code_rows(K)][roundup(B,32)] in element
 * type vpt_dtype (ADIL_F32 / ADIL_BF16 = the dtype of the g stream adil_grad will be called with), zero padded:
 * handing it to adil_grad saves that call a launch of its own.
 * slabs / nslabs / slab_rows (nslabs > 0): the rows come from the partial sums a preceding adil_grad left in its
 * workspace instead of from v (v, index ignored; pos must be NULL): row b = sum over the nslabs slabs, in the fixed
 * order of the library's own reduction (bitwise the same values) — the codes `z D_dagger^T` and the code gradient
 * `g D` of a DDrague iteration (adil.py:542, :551) reach the next kernel without a reduction launch in between. */
int adil_pack_codes(const float* v, const int64_t* index, int B, int K, float* vp, int32_t* pos, void* vpt,
                    int vpt_dtype, const float* slabs, int nslabs, int slab_rows, void* stream);

/* Batched image gather, the data step in front of the path (the DataLoader's per-item fetch + torch.stack + .to(device)
 * of adil.py:130-133,168-170 on a dataset that is resident in HBM): dst[b][:] = convert(src[index[b]][:]).
 * src is R x P (src_dtype), dst is B x P (dst_dtype), index may be NULL (rows 0..B-1: a pure cast).  P % 8 == 0. */
int adil_gather_images(const void* src, int src_dtype, const int64_t* index, void* dst, int dst_dtype, int B, int P,
                       void* stream);

/* Perturbation synthesis, fused with the add and the optional clamps:
 *     delta = vp D^T ;  delta = clamp(delta, -delta_clamp, +delta_clamp)   if delta_clamp >= 0
 *     out   = x + delta ;  out = clamp(out, 0, 1)                          if pixel_clamp
 * x may be NULL (out = delta).  Replaces `tensordot(v[index,:], d, ([1],[3]))` + `x + dv`
 * (adil.py:25-26, :543-544, :617), the per-image loop of forward_unsupervised with
 * its +-eps clamp (adil.py:480-484) and the final [0,1] clamp (adil.py:567, :623). */
int adil_synth(const void* x, const float* d, const float* vp, void* out, int B, int P, int K, int dtype,
               float delta_clamp, int pixel_clamp, void* stream);

/* Precision variant of adil_synth for BASELINE.json configs[4] ("fp8 D.V MFMA on CDNA4"; no reference counterpart —
 * the reference contracts in fp32, adil.py:25): both operands of the contraction are converted to fp8 (OCP e4m3) on the
 * fly, the codes scaled by 384 / v_absmax (every |vp| must be <= v_absmax: after update_v the l1 radius eps is such a
 * bound, adil.py:29-31) and the dictionary by 256 (|D| <= 1 after update_d, adil.py:33-35; larger values saturate);
 * products accumulate in fp32 and are scaled back before the add / clamps.  Same arguments and fusion as adil_synth. */
int adil_synth_fp8(const void* x, const float* d, const float* vp, void* out, int B, int P, int K, int dtype,
                   float v_absmax, float delta_clamp, int pixel_clamp, void* stream);

/* The fp8 variant with a PERSISTENT fp8 copy of the dictionary (round 4): adil_synth_fp8 converts the fp32 master on the
 * fly, so it still reads P K 4 bytes of D per launch; here D comes in as the bytes e4m3(256 d) — a quarter of that —
 * produced once by adil_dict_to_fp8 and kept current by adil_adamw_clamp_fp8 (the AdamW + clamp launch of the learning
 * step writes them next to the fp32 master, 1 byte per element on top of its 28).  Same results as adil_synth_fp8, bit
 * for bit (the same encoding of the same values).  P a multiple of 128, K a multiple of 4, 16-byte aligned streams;
 * ADIL_EINVAL otherwise (use adil_synth_fp8).  n must be a multiple of 4 in the two helpers. */
int adil_dict_to_fp8(const float* d, size_t n, void* d_fp8, void* stream);
int adil_adamw_clamp_fp8(float* p, const void* g, int g_dtype, float* m, float* s, size_t n, float decay, float b1,
                         float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi, float* max_abs_delta,
                         const float* dyn_scalars, void* p_fp8, void* stream);
int adil_synth_fp8_packed(const void* x, const void* d_fp8, const float* vp, void* out, int B, int P, int K, int dtype,
                          float v_absmax, float delta_clamp, int pixel_clamp, void* stream);

/* Adjoint of the synthesis in ONE pass over the upstream gradient g = dLoss/d(x+delta):
 *     grad_d  (P x K)  = g^T vp          if grad_d  != NULL   (accumulated into grad_d when accumulate_d)
 *     grad_vb (B x K)  = g D             if grad_vb != NULL
 * Replaces autograd's backward of the tensordot at adil.py:25 (loss.backward(), adil.py:185,
 * :281, :308, :606) and the forward contraction `tensordot(z, d_drg, ([1,2,3],[1,2,3]))`
 * (adil.py:542, :563) when called with g := z and d := D_dagger^T (grad_vb only).
 * vpt (optional): the transposed codes written by adil_pack_codes (element type = dtype); NULL = made here.
 * nslabs_out (optional, HOST int): the caller will consume grad_v through adil_adamw_l1ball / adil_pack_codes, which
 * can sum the per-workgroup partial sums themselves.  When one row chunk covers the batch the reduction launch is
 * skipped, *nslabs_out = number of slabs (at ws + adil_grad_slab_offset, valid until ws is next used) and grad_vb is
 * left untouched; otherwise *nslabs_out = 0 and grad_vb is written as usual (grad_vb must be given either way). */
int adil_grad(const void* g, const float* d, const float* vp, const void* vpt, float* grad_d, float* grad_vb, int B, int P,
              int K, int dtype, int accumulate_d, void* ws, size_t ws_bytes, int* nslabs_out, void* stream);

/* Fused AdamW step + box projection on a flat fp32 parameter (torch.optim.AdamW semantics):
 *     p *= decay ; m += (1-b1)(g-m) ; s = b2 s + (1-b2) g^2 ;
 *     p -= step_size * m / (sqrt(s)/bc2_sqrt + eps) ;  p = clamp(p, lo, hi)
 * decay = 1-lr*wd, step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) are computed by the caller in
 * double, exactly as torch does.  g has element type g_dtype.  If max_abs_delta != NULL,
 * atomically maxes |p_new - p_old| into it (a float the caller zeroed).
 * Replaces optimise.step() + update_d (adil.py:186,188 with :33-35; clamp [-1,1]) and
 * optimise.step() + the z clamp (adil.py:554-555, :559; clamp [-eps,eps]).
 * dyn_scalars (optional, 2 device floats {step_size, bc2_sqrt}): when given they override the by-value arguments, so a
 * launch recorded in a hipGraph can be replayed with the scalars of a later step (the caller refreshes the two floats with
 * a stream-ordered copy before each replay). */
int adil_adamw_clamp(float* p, const void* g, int g_dtype, float* m, float* s, size_t n, float decay, float b1,
                     float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi,
                     float* max_abs_delta, const float* dyn_scalars, void* stream);

/* Fused inference step of forward_supervised_DDrague (adil.py:551-559): the gradient wrt z,
 *     gz = gvp D_dagger   (gvp = packed dLoss/dv [Bp][Kp], dpinv_t = D_dagger^T stored P x K like D),
 * is formed in the MFMA accumulators and consumed on the spot by AdamW(z) + clamp[lo,hi] + max|dz|; it never
 * touches HBM.  z, m, s are fp32 B x P.  Replaces `loss.backward()` through the two tensordots (adil.py:542-543),
 * `optimise.step()`, the clamp (adil.py:555) and the stop test (adil.py:559).
 * Device-side stop test (all three optional, adil.py:559 `if max|z - z_old| < 1e-6: break`): if skip_if_below is
 * given and *skip_if_below < skip_threshold the call changes nothing and sets *max_abs_delta = 0 (so its successor skips
 * as well); otherwise `clear` (a float) is set to 0.  With three floats
 * s[0..2] = {0, 0, +big} and iteration t passing max_abs_delta = &s[t%3], skip_if_below = &s[(t+2)%3], clear =
 * &s[(t+1)%3], every launch after the converged one is a no-op, so the host may read s[t%3] only every few iterations
 * and still end on exactly the iterate the reference breaks at.  dyn_scalars as in adil_adamw_clamp. */
int adil_zstep(float* z, float* m, float* s, const float* dpinv_t, const float* gvp, int B, int P, int K, float decay,
               float b1, float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi, float* max_abs_delta,
               const float* skip_if_below, float skip_threshold, float* clear, const float* dyn_scalars, void* stream);

/* adil_zstep that ALSO leaves the codes of the next iteration, v' = z_new D_dagger^T (adil.py:542, recomputed by the
 * reference from the z that adil.py:554-555 just wrote), as per-workgroup partial sums in `code_slabs`: *nslabs_out slabs
 * of [roundup(B,32)][K] fp32 — the layout adil_grad leaves behind, summed by adil_pack_codes(slabs, nslabs, slab_rows =
 * roundup(B,32)).  The separate contraction launch of a DDrague iteration (adil_grad with g := z: a second pass over z and
 * D_dagger) disappears; the z tile is contracted while it is still in registers, against the D_dagger slice the z-step
 * holds in LDS anyway.  Same arithmetic for z, m, s and the stop test as adil_zstep (bitwise).  A launch skipped by the
 * device-side stop test leaves code_slabs untouched: they keep the codes of the converged z.
 * Shapes: P a multiple of 128 (and < 2^23), K <= 112, 16-byte aligned pointers; adil_zstep_codes_slab_bytes returns the
 * size code_slabs must have, 0 when the shape is not supported (use adil_zstep + adil_grad then). */
size_t adil_zstep_codes_slab_bytes(int B, int P, int K);
int adil_zstep_codes(float* z, float* m, float* s, const float* dpinv_t, const float* gvp, int B, int P, int K, float decay,
                     float b1, float b2, float eps, float step_size, float bc2_sqrt, float lo, float hi, float* max_abs_delta,
                     const float* skip_if_below, float skip_threshold, float* clear, const float* dyn_scalars,
                     float* code_slabs, size_t code_slab_bytes, int* nslabs_out, void* stream);

/* Fused AdamW step on ALL N rows of the code matrix + row-wise l1-ball projection.
 * The gradient is non-zero only for the rows of the current batch: pos[n] = b if row n is
 * batch slot b (gradient row grad_vb[b]), -1 otherwise (zero gradient; the row still
 * moves through momentum and weight decay — reference quirk Q3).  pos may be NULL
 * (row n <-> grad_vb[n], N == B).  With reset_pos every consumed slot is written back to -1, so the table
 * filled by adil_pack_codes is all -1 again for the next batch (no fill / scatter launches in between).
 * grad_vb may be NULL when pos is given and holds no slot (a rank with an empty shard of the batch).
 * radius < 0 skips the projection.
 * Replaces optimise.step() + update_v (adil.py:186-187 with :29-31 and utils.py:21-41),
 * and the same pair in forward_supervised_AdamW (adil.py:609-610, :614); skip_if_below / skip_threshold / clear are the
 * device-side stop test described at adil_zstep (adil.py:614); dyn_scalars as in adil_adamw_clamp.
 * slabs / nslabs / slab_rows (nslabs > 0): the batch gradient is still in the partial sums of the preceding adil_grad
 * (its nslabs_out): gradient row b = sum over the slabs, formed inside this launch; grad_vb is then ignored. */
int adil_adamw_l1ball(float* v, const float* grad_vb, int32_t* pos, int reset_pos, float* m, float* s, int N, int K,
                      float decay, float b1, float b2, float eps, float step_size, float bc2_sqrt, float radius,
                      float* max_abs_delta, const float* skip_if_below, float skip_threshold, float* clear,
                      const float* dyn_scalars, const float* slabs, int nslabs, int slab_rows, void* stream);

/* constraint_dict's third branch (utils.py:55-56: `project_onto_l1_ball(d[:, :, :, ind], eps=1)` per atom): every
 * (channel, atom) row of HW pixels of the (C,H,W,K) dictionary onto the l1 ball of `radius`, in place.  Rows of any
 * length (sort-free threshold search); no caller upstream. */
int adil_atom_l1ball_project(float* d, int C, int HW, int K, float radius, void* stream);

/* Row-wise Euclidean projection onto the l1 ball, in place: project_onto_l1_ball (utils.py:21-41). */
int adil_l1ball_project(float* x, int N, int K, float radius, void* stream);

/* Row-wise projection onto the l2 ball: x_i *= radius / max(||x_i||_2, radius) (adil.py:626-629). */
int adil_l2ball_project(float* x, int N, int K, float radius, void* stream);

/* ISTA step: v = softshrink(v - step * g, lam)   (g may be NULL: plain softshrink).
 * Replaces Softshrink(step*lambda)(v - step*grad_v) (adil_regularized.py:141-144, :304,
 * :414-416, :570-573; utils.py:159-161). */
int adil_ista_step(float* v, const float* g, size_t n, float step, float lam, void* stream);

/* Per-atom Frobenius norms of D: norms[k] = ||D[:,k]||_2   (utils.py:48). ws: adil_atom_workspace_bytes. */
size_t adil_atom_workspace_bytes(int P, int K);
int adil_atom_norms(const float* d, int P, int K, float* norms, void* ws, size_t ws_bytes, void* stream);

/* Per-atom scaling: d[:,k] /= (sphere ? norms[k] : max(norms[k], 1))   (utils.py:49-54). */
int adil_atom_scale(float* d, int P, int K, const float* norms, int sphere, void* stream);

/* Gram matrix  gram (K x K) = D^T D   (adil.py:523), fp32-grade on the matrix pipe (split bf16 MFMAs), bitwise
 * reproducible.  ws: adil_gram_workspace_bytes(P, K) bytes of device scratch (per-workgroup partial sums). */
size_t adil_gram_workspace_bytes(int P, int K);
int adil_gram(const float* d, int P, int K, float* gram, void* ws, size_t ws_bytes, void* stream);

/* out (K x K) = a^-1 for a symmetric positive definite K x K matrix (the Gram matrix; `dtd.inverse()`, adil.py:524).
 * One workgroup, fp64 Gauss-Jordan in LDS; no host round trip. */
int adil_spd_inverse(const float* a, int K, float* out, void* stream);

/* out (P x K) = D M^T  with M (K x K):  D_dagger^T = D (DtD^-1)^T   (adil.py:525, stored P x K); fp32-grade on the
 * matrix pipe (split bf16 MFMAs). */
int adil_dict_rightmul(const float* d, const float* mat, int P, int K, float* out, void* stream);

/* Per-image evaluation sums (performance.py:249-266): sq_err[b] = sum_p (adv-x)^2, sq_norm[b] = sum_p x^2. */
int adil_image_metrics(const void* adv, const void* x, int B, int P, int dtype, float* sq_err, float* sq_norm,
                       void* stream);

/* Frozen-classifier epilogues (NOT part of the ADiL maths; no reference counterpart — they replace the separate
 * BatchNorm(eval) / add / ReLU kernels PyTorch launches around each convolution of the frozen network, e.g. the
 * torchvision ResNet blocks the reference builds at demo_dL_attack.py:41-59):
 *     y = act( x * scale[c] + shift[c] (+ res) ),  act = ReLU if relu else identity
 *     gx = mask * g * scale[c],  gres = mask * g  (mask = y > 0 if relu else 1)     [input gradient only: frozen net]
 * channel of flat element i is (i / inner) % C  (inner = 1 for channels_last storage, H*W for NCHW); n % 8 == 0. */
int adil_affine_act_fwd(const void* x, const void* res, const float* scale, const float* shift, void* y, size_t n, int C,
                        int inner, int relu, int dtype, void* stream);
int adil_affine_act_bwd(const void* g, const void* y, const float* scale, void* gx, void* gres, size_t n, int C,
                        int inner, int relu, int dtype, void* stream);

/* Frozen-classifier stem, the stage on either side of the hot path for the torchvision ResNets the reference attacks
 * (demo_dL_attack.py:41-59; NOT part of the ADiL maths): Normalize -> conv 7x7/2 (3->64) -> BatchNorm(eval) -> ReLU ->
 * maxpool 3x3/2, forward and input gradient, on the attack's own layouts (x_adv / dLoss/dx_adv are B x 3 x H x W in the
 * stream dtype; activations are NHWC bf16).  H and W even.  Weights are pre-packed bf16:
 *     w_fwd [64][7][8][4]  = w[co][ci][kh][kw] at [co][kh][kw][ci], zero for kw = 7 / ci = 3
 *     w_bwd [4][49][64]    = w[co][ci][kh][kw] at [ci][kh*7+kw][co], zero for ci = 3
 *   adil_stem_conv_fwd : y1 = relu((conv((x - mean) * inv_std)) * scale[co] + shift[co]),  y1 is B x H/2 x W/2 x 64
 *   adil_maxpool_fwd   : p = maxpool3x3/2/pad1(y) (B x OH x OW x C -> B x PH x PW x C, PH = (OH-1)/2+1), idx = position
 *                        kh*3+kw of the first maximum (torch.nn.functional.max_pool2d's rule); C % 8 == 0
 *   adil_stem_pool_bwd : gy = route(g; idx) * [p > 0] * scale[c]   (maxpool + ReLU + BatchNorm backward, B x OH x OW x C)
 *   adil_stem_conv_bwd : gx = inv_std[ci] * conv7x7/2 input gradient of gy   (B x 3 x H x W, gx_dtype) */
int adil_stem_conv_fwd(const void* x, int x_dtype, const void* w_fwd, float mean0, float mean1, float mean2, float inv_std0,
                       float inv_std1, float inv_std2, const float* scale, const float* shift, void* y, int B, int H, int W,
                       void* stream);
int adil_maxpool_fwd(const void* y, void* p, uint8_t* idx, int B, int OH, int OW, int C, void* stream);
int adil_stem_pool_bwd(const void* g, const uint8_t* idx, const void* p, const float* scale, void* gy, int B, int OH, int OW,
                       int C, void* stream);
int adil_stem_conv_bwd(const void* gy, const void* w_bwd, float inv_std0, float inv_std1, float inv_std2, void* gx,
                       int gx_dtype, int B, int H, int W, void* stream);

/* Pointwise (1x1) convolution of the frozen network on channels_last storage, with the eval-BatchNorm / residual /
 * ReLU epilogue applied to the accumulators (bf16 in/out, fp32 accumulate):
 *     y[M][N] = act( (x'[M][K] . w[N][K]^T) * scale[n] + shift[n] (+ res[M][N]) ),  M = output pixels, K = Cin, N = Cout
 * K % 64 == 0, N % 64 == 0; res may be NULL; relu = 0/1.
 * Stride 2 (the ResNet downsample convolutions): sub_w = OW > 0, sub_hw = OH*OW: output pixel (n,oh,ow) reads input
 * pixel (n,2oh,2ow) of the B x 2OH x 2OW x K tensor x points to (gathered, nothing copied); sub_w = 0: stride 1.
 * Optional prologue (pscale, pshift both non-NULL, K <= 512): x is the RAW output of the previous convolution and
 * x' = relu(x * pscale[k] + pshift[k]) is formed on the operand path (its BatchNorm + ReLU never makes an HBM pass);
 * otherwise x' = x. */
int adil_pw_conv_fwd(const void* x, const void* w, const float* scale, const float* shift, const void* res, void* y, int M,
                     int K, int N, int relu, const float* pscale, const float* pshift, int sub_w, int sub_hw, void* stream);
/* Input gradient of adil_pw_conv_fwd (of the stride-1 form; a stride-2 layer calls it with M = its output pixels and
 * gets the gradient on its own stride-2 grid).  v = g (+ g2) (+ up2(g3)), mask = [y > 0] if relu else 1:
 *     gres[M][N] = v * mask (optional),   gx[M][K] = (v * mask * scale[n]) . w,   weight given transposed, wt[K][N].
 * g2 (optional): the second gradient meeting at a residual join.  g3 (optional, with sub_w = OW, sub_hw = OH*OW,
 * M = B*4*OH*OW): a gradient living on the stride-2 grid, [M/4][N]; pixel (n,h,w) receives g3[(n,h/2,w/2)] for even
 * h and w (the zero-upsampled tensor is never materialised).  N % 64 == 0, K % 64 == 0, N <= 2048.
 * With the forward's prologue (xin = the raw x, pscale, pshift) the result is the gradient wrt xin:
 *     gx *= [xin * pscale + pshift > 0] * pscale. */
int adil_pw_conv_bwd(const void* g, const void* g2, const void* y, const float* scale, const void* wt, void* gx, void* gres,
                     int M, int K, int N, int relu, const void* xin, const float* pscale, const float* pshift, const void* g3,
                     int sub_w, int sub_hw, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of the frozen network on channels_last storage, raw bf16 output (its BatchNorm +
 * ReLU run in the next pointwise kernel's prologue):  y[B][H][W][N] = conv3x3(x[B][H][W][C]; wp), weights packed
 * wp[N][9][C] = w[n][c][kh][kw] at [n][kh*3+kw][c].  The input gradient is the same call on wp' [C][9][N] =
 * w[n][c][2-kh][2-kw] at [c][kh*3+kw][n].  C % 64 == 0, N % 64 == 0, W <= 63. */
int adil_conv3x3(const void* x, const void* wp, void* y, int B, int H, int W, int C, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADIL_HIP_H */
