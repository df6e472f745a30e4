/*
 * vqhip.h — C ABI of the MI355X (gfx950) VQ codebook-lookup library (libvqhip.so).
 *
 * Drop-in boundary for ONE path of magic-research/vector_quantization: the `vq.algorithms` quantizer
 * forward (distance over the whole codebook → row argmin → gather/STE/loss → codebook update).  The
 * reference has no FFI of its own (it is pure Python on ATen ops); each entry point below replaces the
 * ATen call(s) named next to it (paths relative to the reference root), and is what the reference-side
 * binding in INTEGRATION.md (a ctypes stub inside the quantizer modules) calls.
 *
 * Conventions: every pointer is a DEVICE pointer unless noted; tensors are dense row-major; `stream`
 * is a hipStream_t passed as void*; functions never allocate, never synchronise, never touch another
 * stream; they return 0 or a negative VQHIP_E* code.  Two documented exceptions to "never synchronise", both host-side waits the
 * CALLER arms: the vqhip_rccl_* set-up calls (they block inside RCCL), and vqhip_cvq_forward with cap < 0, which waits on the
 * caller's own event (hipEventSynchronize(count_event)) for a count the previous call queued a whole step earlier and then reads
 * the caller's pinned host word — that word must be host-coherent memory (hipHostMalloc's default; not under HIP_HOST_COHERENT=0:
 * pass cap >= 0 there).  x_dtype selects the latent storage type
 * (fp32, or bf16 as produced under autocast); codebooks are fp32 like nn.Embedding.weight.
 * Every caller-owned scratch buffer travels with its size (`ws_bytes`, `cb_bytes`): a buffer smaller than the matching
 * *_bytes function asks for is refused with VQHIP_EINVAL before anything is launched (vqhip_last_error names both numbers).
 *
 * LIMITS (checked; VQHIP_EINVAL beyond them)
 *   N, K                 < 2^31 (row lists and candidate codes are 32-bit)
 *   D                    >= 1; the fp16-MFMA proposal pass exists for D <= 1024 with D % 8 == 0 (every shipped config: 8, 32,
 *                        256, 768) — other D take the all-fp32 route transparently (vqhip_argmin, vqhip_col_argmin) or are
 *                        refused where only the proposal form exists (vqhip_col_argmin_rows, VQHIP_METRIC_COS_BF16 encode)
 *   ordered sums         K <= 32768, D % 4 == 0, ceil(N/1024) * K < 2^31 (vqhip_token_order and the two sums built on it)
 *   packed exchange      <= 256 ranks (count pieces stay exact in fp32), per-rank counts < 2^31, token count < 2^48
 *   vqhip_transpose      B <= 65535 per launch
 *   alignment            every buffer pointer 16-byte aligned (hipMalloc / torch allocations are); rows dense, no padding
 *   device               all pointers belong to the device `stream` was created on, which is the current HIP device
 */
#ifndef VQHIP_H_
#define VQHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQHIP_VERSION 600   /* round 6: + NearestAnchor(sync=True) across ranks by a key exchange (vqhip_cvq_col_keys, vqhip_cvq_pack_sync,
                              * vqhip_allreduce_min_i64, vqhip_cvq_forward_t.anchor_sync / rank / keys); vqhip_cvq_apply takes the capacity */

#define VQHIP_METRIC_L2 0   /* L2Distance      vq/algorithms/vq/distances.py:28-32 */
#define VQHIP_METRIC_COS 1  /* CosineDistance  vq/algorithms/vq/distances.py:35-46 */
/* CosineDistance as the reference's GPU runs evaluate it under bf16 autocast (vq/runners/base.py:30-48: normalize is on
 * autocast's fp32 list, the einsum on its bf16 list): operands rounded to bf16 after the fp32 normalisation, products
 * summed in fp32 (here: the k-ordered fma chain of the fp32 definition), the sum rounded to bf16, 1 - s rounded to bf16,
 * lowest index among equal bf16 distances.  This is what the reference returns inside its autocast region: a caller of
 * this ABI that replaces distances.py:39-46 there must pass THIS metric (check torch.is_autocast_enabled('cuda') and
 * torch.get_autocast_dtype('cuda') == torch.bfloat16, as the Python CosineDistance of this package does by itself;
 * INTEGRATION.md, binding B) — VQHIP_METRIC_COS is the fp32 definition, the reference's result OUTSIDE autocast.  Exists
 * where the proposal image does (D <= 1024, D % 8 == 0). */
#define VQHIP_METRIC_COS_BF16 5

#define VQHIP_DTYPE_F32 0
#define VQHIP_DTYPE_BF16 1

#define VQHIP_OK 0
#define VQHIP_EINVAL (-22)      /* bad argument (null pointer, unsupported D, ...) */
#define VQHIP_ELAUNCH (-5)      /* hip launch error */
#define VQHIP_ERCCL (-71)       /* librccl.so not loadable, or an RCCL call failed (vqhip_last_error carries RCCL's text) */

int vqhip_version(void);
const char *vqhip_last_error(void); /* host string describing the last non-zero return on this thread */

/* ---- sizes of caller-owned buffers -------------------------------------------------------------- */
/* bytes of the prepared-codebook image produced by vqhip_codebook_prepare for a [K,D] codebook */
int64_t vqhip_codebook_bytes(int64_t K, int D);
/* vqhip_codebook_prepare + (cosine: vqhip_normalize_rows of x) + vqhip_argmin in one call with two launches less on the
 * critical path: the codebook statistics and the whole token side (normalisation included) are independent and run as ONE
 * launch (cosine: the whole codebook preparation — statistics AND image — rides in that launch).  This is the training-time shape of vq/algorithms/vq/quantizers.py:92-100, where the codebook changes every
 * step; with a frozen codebook prepare once and call vqhip_argmin.
 *   x [N,D] fp32|bf16: the latents as the quantizer receives them (NOT normalised, also for cosine);
 *   cb: vqhip_codebook_bytes(K,D), written; idx [N] int64; hist [K] int32 or NULL (counts are ADDED);
 *   xq [N,D] fp32: cosine only — receives F.normalize(x, dim=1) (bit-identical to vqhip_normalize_rows; under
 *   VQHIP_METRIC_COS_BF16 each element additionally rounded to the nearest bf16: the operand of THAT metric), which is also the
 *   operand of the exact re-rank: keep it alive until the call has completed on the stream; NULL for L2;
 *   ws: vqhip_workspace_bytes(N,K,D).  Results are those of the separate calls, bit for bit. */
int vqhip_encode(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, void *cb, int64_t cb_bytes,
                 int64_t *idx, int32_t *hist, float *xq, void *ws, int64_t ws_bytes, void *stream);
/* The same with flags: VQHIP_ENCODE_ZERO_HIST — `hist` is zeroed by the call's first launch (no separate fill), so the
 * histogram on return is exactly this call's code counts. */
#define VQHIP_ENCODE_ZERO_HIST 1
int vqhip_encode_ex(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, void *cb, int64_t cb_bytes,
                    int64_t *idx, int32_t *hist, float *xq, void *ws, int64_t ws_bytes, int flags, void *stream);

/* vqhip_encode_ex for latents that arrive as the FEATURE MAP [B, D, HW] the encoder / post_encode connector produced: the
 * 'b c h w -> (b h w) c' rearrangement of vq/tasks/image_tokenization/models/base.py:124,140 is folded into the call's first
 * launch (64-dim x 32-token tiles are read with the tokens along the lanes and turned through LDS), so no transpose kernel
 * runs and the map is read once.  N = B*HW tokens, token n = (b, p) with n = b*HW + p.
 *   xrows [N, D] in x_dtype: receives the token-major copy of the latents the rest of the step works on (the exact re-rank of
 *   this call for L2, vqhip_gather_ste_*, vqhip_vq_backward); xq [N, D] fp32: cosine only — F.normalize(x), as in vqhip_encode
 *   (NULL for L2).  Keep both alive until the call has completed.  D <= 1024, D % 8 == 0.  Same results as transposing and
 *   calling vqhip_encode_ex, bit for bit. */
int vqhip_encode_map(const void *x_map, int x_dtype, const float *e, int64_t B, int64_t HW, int64_t K, int D, int metric, void *cb,
                     int64_t cb_bytes, int64_t *idx, int32_t *hist, void *xrows, float *xq, void *ws, int64_t ws_bytes, int flags,
                     void *stream);

/* Byte offset, inside an image prepared with VQHIP_METRIC_COS, of the fp32 [K, D] rows F.normalize(e, dim=1) the exact
 * definition consumes (bit-identical to vqhip_normalize_rows(e)); 256-byte aligned.  Lets a caller that needs the
 * normalised codebook again in the same step (NearestAnchor's column argmin, vq/algorithms/cvqvae/anchors.py:83-84)
 * read it instead of normalising twice.  Not written for the L2 metric. */
int64_t vqhip_codebook_exact_offset(int64_t K, int D);
/* bytes of per-call scratch for vqhip_argmin / vqhip_argmin_exact / vqhip_distance over N rows.
 * (vqhip_col_argmin needs the LARGER vqhip_col_workspace_bytes, declared next to it below.)
 * (Alignment, density and device preconditions: the LIMITS block at the top of this header.) */
int64_t vqhip_workspace_bytes(int64_t N, int64_t K, int D);

/* ---- codebook preparation --------------------------------------------------------------------------
 * Reads e[K,D] fp32 and writes into `cb` (vqhip_codebook_bytes): the oracle-order row norms |e_k|^2,
 * for COS the normalised codebook F.normalize(e) (distances.py:41), a power-of-two-scaled fp16 copy in
 * MFMA-fragment order, and the error bounds the exact re-rank needs.  Must be re-run whenever e changes
 * (callbacks rebind weight.data every forward: vq/algorithms/vq/callbacks/update.py:56). */
int vqhip_codebook_prepare(const float *e, int64_t K, int D, int metric, void *cb, int64_t cb_bytes, void *stream);

/* ---- fused distance + argmin  (replaces quantizers.py:97-99: distance(x, W) → torch.argmin(d, -1)) ---
 * idx[n] = argmin_k d(x_n, e_k), lowest k on ties, bit-identical to the fp32 definition
 *   L2 : sqrt(clamp_min((sum_d(-2 x_d e_kd) + |x|^2) + |e_k|^2, 0))   (torch.cdist mm path)
 *   COS: 1 - sum_d xh_d eh_kd, xh/eh = F.normalize(.)                  (x must already be normalised:
 *        pass the output of vqhip_normalize_rows; the codebook is normalised by codebook_prepare)
 * evaluated with k-ordered fp32 fma chains (see DESIGN.md "Arithmetic contract").  A fp16 MFMA pass
 * proposes candidates under a rigorous error bound and an exact fp32 re-rank decides; rows the bound
 * cannot settle are re-evaluated over the whole codebook in fp32.  The proposal pass exists for D <= 1024 with
 * D % 8 == 0 (every shipped config: 8, 32, 256, 768); any other D takes the all-fp32 route of vqhip_argmin_exact.
 * Optional outputs (nullable):
 *   hist[K] int32 += code-hit histogram (quant.bincount, vq/algorithms/vq/utils.py:42);
 * `ws` = vqhip_workspace_bytes(N,K,D) of scratch. */
int vqhip_argmin(const void *x, int x_dtype, const float *e, const void *cb, int64_t cb_bytes, int64_t N, int64_t K, int D,
                 int metric, int64_t *idx, int32_t *hist, void *ws, int64_t ws_bytes, void *stream);

/* Same result computed entirely in fp32 (v_mfma_f32_32x32x2_f32) without the fp16 proposal pass.
 * dmin (nullable) receives the winning distance.  For COS `e` must be the normalised codebook. */
int vqhip_argmin_exact(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric,
                       int64_t *idx, float *dmin, int32_t *hist, void *ws, int64_t ws_bytes, void *stream);

/* Materialise d[N,K] fp32 (memo['distance'], quantizers.py:98) for consumers that need the matrix
 * (EntropyLoss losses.py:143, MultinomialAnchor anchors.py:100).  COS: x and e already normalised. */
int vqhip_distance(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric,
                   float *d, void *ws, int64_t ws_bytes, void *stream);

/* NearestAnchor: col_idx[k] = argmin_n d[n,k], lowest n on ties (vq/algorithms/cvqvae/anchors.py:83), same arithmetic
 * contract as vqhip_argmin (bit-identical to the fp32 definition, operand order of the reference kept); never
 * materialises d.  Runs the proposal + re-rank pipeline with the roles of latents and codes swapped.
 * COS: x and e already normalised.  `ws` = vqhip_col_workspace_bytes(N, K, D). */
int64_t vqhip_col_workspace_bytes(int64_t N, int64_t K, int D);
int vqhip_col_argmin(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric,
                     int64_t *col_idx, void *ws, int64_t ws_bytes, void *stream);

/* ---- row kernels ------------------------------------------------------------------------------------ */
/* out[r] = |v_r|^2 in the oracle's order (64 interleaved fma partials + halving tree) */
int vqhip_row_sqnorm(const void *v, int dtype, int64_t R, int D, float *out, void *stream);
/* out = F.normalize(v, dim=1, eps) (callbacks/normalize.py:24,27; distances.py:40-41) */
int vqhip_normalize_rows(const void *v, int dtype, int64_t R, int D, float eps, float *out, void *stream);

/* ---- decode + straight-through + loss partial sums --------------------------------------------------
 * z[n] = e[idx[n]]                       (nn.Embedding gather, quantizers.py:107)
 * z_ste[n] = x[n] + (z[n] - x[n])        (utils/ste.py:10)        — either output may be NULL
 * sse[0] += sum (z - x)^2 in double      (both MSE terms of losses.py:50,62 share this sum; the host
 *                                         divides by N*D).  sse may be NULL. */
int vqhip_gather_ste_loss(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D,
                          float *z, float *z_ste, double *sse, void *stream);
/* The same pass with the mean finished on the device: mse[0] = mse[1] = mean((z - x)^2) as fp32 (double sum, one division,
 * one cast: what `mse_loss` of losses.py:50,62 returns for both terms), mse[2] = mse[0] + beta*mse[1] (VQGANLoss.forward,
 * losses.py:119-127: a product and a sum, each rounded, like the reference's two ops), mse[3] = 0 — mse is fp32[4].
 * `scratch16`: 16 bytes of device memory that are ZERO on entry and are left zero on return (one scratch per stream in
 * flight); N > 0. */
int vqhip_gather_ste_mse(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D,
                         float *z, float *z_ste, float *mse, float beta, void *scratch16, void *stream);

/* The same with the OUTPUT written as the feature map [B, D, HW] ('(b h w) c -> b c h w' + .contiguous(), models/base.py:126-127,
 * folded into the gather: 64 x 64 tiles turned through LDS, 256 contiguous bytes per channel and wave-instruction):
 *   x_rows != NULL: out_map = x + (e[idx] - x) (straight-through output), mse[4] as vqhip_gather_ste_mse, scratch16 as there;
 *   x_rows == NULL: out_map = e[idx] (decode_from_quant, image_reconstruction/models.py:97-106); mse, scratch16 unused. */
int vqhip_gather_ste_map(const void *x_rows, int x_dtype, const float *e, const int64_t *idx, int64_t B, int64_t HW, int D,
                         float *out_map, float *mse, float beta, void *scratch16, void *stream);

/* hist[K] int32 += bincount(idx) (utils.py:42; runners/metrics.py:40-44) */
int vqhip_hist(const int64_t *idx, int64_t N, int64_t K, int32_t *hist, void *stream);

/* dst[idx[n], :] += src[n, :]  (centroids.scatter_add_, vqkd/quantizers/callbacks.py:60-62; also the
 * dense embedding backward).  fp32 atomics. */
int vqhip_scatter_add_rows(const float *src, const int64_t *idx, int64_t N, int64_t K, int D, float *dst,
                           void *stream);

/* ---- codebook updates ------------------------------------------------------------------------------- */
/* VQKDCallback._kmeans tail + after_encode (callbacks.py:66-70,126-128,73-75):
 *   c = where(hist>0, sums/max(hist,1), w); c = normalize(c); c = w*decay + c*(1-decay); w = normalize(c)
 * hist int64[K] / sums[K,D] are the (all-reduced) statistics.  In place on w.  centroid_only != 0 stops after the
 * first line (VQKDCallback._kmeans as used by the k-means lazy init, callbacks.py:104-107). */
int vqhip_vqkd_update(float *w, const int64_t *hist, const float *sums, int64_t K, int D, float decay,
                      int centroid_only, void *stream);
/* CVQVAECallback.after_encode (quantizer_callback.py:94-102):
 *   p = p*ema_decay + (hist/numel)*(1-ema_decay);  decay_k = 1 - exp(-p_k*K*10/(1-ema_decay) - eps)
 *   w_k = w_k*decay_k + anchors_k*(1-decay_k)      — in place on p[K] and w[K,D].
 * numel_dev (nullable DEVICE int64) overrides numel: the all-reduced token count can stay on the device.
 * stage: 1 = update p only (the anchor sampler runs between the two halves, quantizer_callback.py:94-96),
 *        2 = update w only from the current p, 3 = both. */
int vqhip_cvq_update(float *w, float *p, const int64_t *hist, int64_t numel, const int64_t *numel_dev,
                     const float *anchors, int64_t K, int D, float ema_decay, float eps, int stage, void *stream);
/* The whole single-rank update with NearestAnchor in ONE launch (quantizer_callback.py:89-103 + anchors.py:83-84):
 *   p_out[k] = p_in[k]*g + (hist[k]/numel)*(1-g);  decay_k = 1 - exp(-p_out[k]*K*10/(1-g) - eps)
 *   w_out[k] = w_in[k]*decay_k + x[col_idx[k]]*(1-decay_k)
 * hist is the int32 histogram of the argmin epilogue, col_idx the output of vqhip_col_argmin, x the latents [N,D]
 * (fp32 or bf16).  Same expressions in the same order as stage 1 + vqhip_gather_rows + stage 2 (bit-identical);
 * w_out may alias w_in, p_out may alias p_in.  With more than one rank the histogram and the anchors are all-reduced
 * between the stages, so the staged form above is the one to use. */
int vqhip_cvq_step(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist, int64_t numel,
                   const void *x, int x_dtype, const int64_t *col_idx, int64_t K, int D, float ema_decay, float eps,
                   void *stream);
/* The same update for a subset of the codes.  decay_k == 1.0f (every code with p_k above ~1e-6 at K = 16384) multiplies
 * the code's anchor by exactly 0, so only the codes with decay_k < 1 need an anchor at all: vqhip_cvq_decay writes
 * decay[K] with the update's own expression (bit for bit), the caller selects rows = {k : decay_k < 1}, computes /
 * all-reduces anchors for those M codes only, and vqhip_cvq_update_rows applies
 *   w[rows[i]] = w[rows[i]]*decay + anchors_sub[i]*(1-decay).
 * Identical to stage 2 of vqhip_cvq_update on every finite input (a skipped code keeps w_k instead of w_k*1 + a*0: only a
 * negative-zero weight or a non-finite anchor could tell the difference). */
int vqhip_cvq_decay(const float *p, int64_t K, float ema_decay, float eps, float *decay, void *stream);
int vqhip_cvq_update_rows(float *w, const float *p, const int64_t *rows, const float *anchors_sub, int64_t M, int64_t K, int D,
                          float ema_decay, float eps, void *stream);
/* ---- the ONE exchange step of a training forward --------------------------------------------------
 * Packed fp32 buffer = [counts: low 16 bits, K floats][counts: bits above, K floats][token count in 16-bit pieces, 3 floats]
 * [0][payload rows M x D], vqhip_pack_floats(K, M, D) floats in all.  Each count piece is an integer < 2^16, so a SUM
 * all-reduce over up to 256 ranks adds them EXACTLY in fp32 whatever the order: the code-hit histogram and the token
 * count (QuantStatistics' two int64 all-reduces, vq/algorithms/vq/utils.py:34-35) travel in the same collective as the fp32
 * payload (VQ-KD centroid sums, vqkd/quantizers/callbacks.py:63-64; CVQ-VAE anchors, cvqvae/anchors.py:65-67).
 * vqhip_pack_counts writes the header from an int32 (hist_is_int64 = 0) or int64 histogram; vqhip_unpack_counts turns an
 * all-reduced header back into int64 out[K+1] = counts ‖ token count. */
int64_t vqhip_pack_floats(int64_t K, int64_t M, int D);
int vqhip_pack_counts(const void *hist, int hist_is_int64, int64_t numel, int64_t K, float *packed, void *stream);
int vqhip_unpack_counts(const float *packed, int64_t K, int64_t *out, void *stream);

/* ---- the collective of that exchange step, on the CALLER'S stream (vqhip_allreduce_packed) ------------
 * Replaces the reference's per-quantity torch.distributed.all_reduce calls (vq/algorithms/vq/utils.py:34-35,
 * vqkd/quantizers/callbacks.py:63-64, cvqvae/anchors.py:65-67) by ONE in-place fp32 SUM over the packed buffer, enqueued
 * by RCCL on the very stream the pack / apply kernels run on: no side stream, no event hop either side, capturable into a
 * HIP graph with the rest of the step.  libvqhip does not link RCCL: the symbols are resolved at run time from the
 * librccl.so the process already has (PyTorch-ROCm ships one; two RCCL copies in one process must be avoided) or from
 * `path`.  These four are HOST-side set-up calls and the only entry points that block or allocate (inside RCCL):
 *   vqhip_rccl_load(path)            path NULL or "": the librccl.so already mapped into the process (never a fresh copy from the
 *                                    loader's search path: VQHIP_ERCCL if none is mapped under that name); idempotent.
 *   vqhip_rccl_unique_id(id)         HOST buffer of VQHIP_RCCL_ID_BYTES; called on ONE rank, the bytes are handed to the
 *                                    others by whatever the application has (torch.distributed's store here).
 *   vqhip_rccl_comm_init(&comm, nranks, id, rank)   collective over the ranks; binds the CURRENT HIP device.
 *   vqhip_rccl_comm_destroy(comm)
 * vqhip_allreduce_packed: buf[0..floats) += the same range of every other rank, result on every rank; the first
 *   vqhip_pack_floats(K, M, D) floats of a packed buffer.  Never synchronises. */
#define VQHIP_RCCL_ID_BYTES 128
int vqhip_rccl_load(const char *path);
int vqhip_rccl_unique_id(void *id_host);
int vqhip_rccl_comm_init(void **comm, int nranks, const void *id_host, int rank);
int vqhip_rccl_comm_destroy(void *comm);
int vqhip_allreduce_packed(float *buf, int64_t floats, void *comm, void *stream);

/* CVQ-VAE with anchors for the codes that can need one (quantizer_callback.py:85-103, NearestAnchor anchors.py:83-84).
 * decay_k == 1.0f — every code in regular use — multiplies the code's anchor by exactly 0.  vqhip_cvq_rows lists, from the
 * probabilities BEFORE this step's update, the codes whose decay can still come out below 1 (the coming p is >= p*ema_decay
 * and decay is monotone in p; a safety factor of 14 on the rounding boundary: vqhip_exchange_kernels.h): rows[0..count)
 * ascending, slot[k] = position of code k in rows or -1, count[0] — all DEVICE int32 (rows, slot: K entries).  The set
 * depends only on the synchronised p, so every rank derives the same one before anything is exchanged.
 * vqhip_col_argmin_rows: col_idx[i] = nearest latent of code rows[i] for i < count — vqhip_col_argmin's pipeline and
 *   arithmetic on the listed codes; the launches are sized for `cap` (>= the count; the host need not know it: a HIP graph
 *   captures cap = K), rows past the count cost an early exit; ws = vqhip_col_rows_workspace_bytes(N, cap, D).  A SHORT list
 *   (ceil(cap / 32) * ceil(N / 128) * D <= 2^18; fp32 latents) skips the proposal pipeline: the whole-batch fp32 pass of the
 *   definition itself over the listed codes, one launch behind a tiny one (two more for the L2 norms) instead of five — the
 *   same indices (tuning key 15 = 0 sends short lists through the pipeline as well: A/B, tests).
 * vqhip_cvq_pack: this rank's packed buffer — header from the int32 epilogue histogram, payload row i = x[col_idx[i]]
 *   for i < count, zeros up to cap.  All-reduce the first vqhip_pack_floats(K, cap, D) floats.
 * vqhip_cvq_apply: p_out = p_in*g + (hist/numel)*(1-g); decay as above; w_out[k] = w_in[k]*decay + a*(1-decay) for listed
 *   codes, w_in[k]*decay (= w_in[k]) for the others.  packed != NULL: counts and anchor sums from the all-reduced buffer,
 *   a = sum/world; packed == NULL (one rank): counts from hist/numel, a = x[col_idx[slot[k]]].  Outputs may alias inputs.
 *   cap: the capacity the column pass / pack were sized for; a code whose slot lies at or beyond it has no anchor in
 *   col_idx / packed (a list longer than the capacity the caller chose is the caller's error): it keeps w_k * decay_k and
 *   nothing outside the buffers is read.
 *   Bit-identical to vqhip_cvq_update / vqhip_cvq_step on finite data (a non-listed code keeps w_k instead of w_k*1 + a*0:
 *   only a negative-zero weight or a non-finite anchor could tell the difference). */
int vqhip_cvq_rows(const float *p, int64_t K, float ema_decay, float eps, int32_t *rows, int32_t *slot, int32_t *count,
                   void *stream);
int64_t vqhip_col_rows_workspace_bytes(int64_t N, int64_t cap, int D);
int vqhip_col_argmin_rows(const void *x, int x_dtype, const float *e, const int32_t *rows, const int32_t *count, int64_t cap,
                          int64_t N, int64_t K, int D, int metric, int64_t *col_idx, void *ws, int64_t ws_bytes, void *stream);
int vqhip_cvq_pack(const int32_t *hist, int64_t numel, const void *x, int x_dtype, const int64_t *col_idx, const int32_t *count,
                   int64_t cap, int64_t K, int D, float *packed, void *stream);
int vqhip_cvq_apply(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist, int64_t numel,
                    const void *x, int x_dtype, const int64_t *col_idx, const float *packed, int world, const int32_t *slot,
                    int64_t cap, int64_t K, int D, float ema_decay, float eps, void *stream);
/* NearestAnchor(sync=True) over more than one rank (vq/algorithms/cvqvae/anchors.py:50-57,83-84; configs/cluster/model.py:28).
 * The reference all-gathers the latents and the whole [N, K] matrix and takes the column argmin of the concatenation — the
 * nearest latent of a code over ALL ranks' tokens, the lowest (rank, row) among equal distances; the anchor is NOT averaged.
 * Here every rank runs vqhip_col_argmin_rows over its own tokens, then:
 * vqhip_cvq_col_keys: keys[i] = (distance of listed code rows[i] to its local winner col_idx[i] — the fp32 definition's own
 *   value, ordered as torch.argmin orders it: NaN first, -0 == +0 — : rank, 8 bits : row, 24 bits) for i < count, INT64_MAX up
 *   to cap; stored as int64 with the top bit flipped, so that a SIGNED MIN all-reduce over keys[0 .. cap) (8 cap bytes; gloo
 *   and RCCL both have it) leaves on every rank the key of the global winner.  x / e are the operands the column pass was
 *   given (cosine: both normalised).  N <= VQHIP_SYNC_MAX_ROWS tokens per rank, rank < 256.
 * vqhip_cvq_pack_sync: vqhip_cvq_pack with payload row i = x[row] on the rank the reduced key names and -0.0f elsewhere
 *   (x + (-0) == x for every x), so the SUM all-reduce of the packed buffer delivers the winner's row bit for bit whatever
 *   the order of the sum; vqhip_cvq_apply is then called with world = 1 (no averaging: anchors.py:59-63).
 * Exchange per step: 8 cap + 4 (2K + 4 + cap D) bytes, against the reference's all-gather of world * N * (K + D) floats.
 * vqhip_allreduce_min_i64: the MIN on a vqhip_rccl_comm_init communicator, on the caller's stream (as vqhip_allreduce_packed). */
#define VQHIP_SYNC_MAX_ROWS (1 << 24)
int vqhip_cvq_col_keys(const void *x, int x_dtype, const float *e, const int32_t *rows, const int32_t *count, int64_t cap,
                       const int64_t *col_idx, int64_t N, int64_t K, int D, int metric, int rank, int64_t *keys, void *stream);
int vqhip_cvq_pack_sync(const int32_t *hist, int64_t numel, const void *x, int x_dtype, const int64_t *keys, const int32_t *count,
                        int64_t cap, int rank, int64_t K, int D, float *packed, void *stream);
int vqhip_allreduce_min_i64(int64_t *buf, int64_t n, void *comm, void *stream);
/* ---- ONE host call per training forward (round 5) -------------------------------------------------------------------------
 * The reference's training step runs the quantizer's forward as ~20 ATen calls; the entry points above
 * replace them one for one, which leaves an eager nn.Module step with ~10 host calls — at the reference's per-rank batches
 * (3 072-12 544 tokens) the HOST is then the bound.  The two entry points below enqueue a whole training forward from one
 * call: encode -> the codebook update of the callback, exchange included -> decode / straight-through / loss.  Every launch
 * is one the separate entry points would have made, in the same order on the same stream: the results are those of the
 * chain, bit for bit.  The argument block is a plain struct of pointers and sizes (struct_bytes = sizeof, checked).
 *
 * `phases`: VQHIP_STEP_ALL — everything, with the all-reduce issued by the library on `comm` (a vqhip_rccl_comm_init
 *   communicator) when `exchange` != 0; comm may be NULL only for world == 1 (a one-rank SUM is the identity).
 *   VQHIP_STEP_BEFORE_EXCHANGE / VQHIP_STEP_AFTER_EXCHANGE — the two halves around a collective the CALLER issues on
 *   packed[0 .. exchange_floats) (torch.distributed.all_reduce: the default route of the Python callbacks); the same struct,
 *   untouched, goes into both calls.
 * `exchange` == 0: one rank, no packed buffer (counts from the epilogue histogram).
 *
 * vqhip_cvq_forward — VQGANQuantizer + CVQVAECallback(NearestAnchor) in train mode (vq/algorithms/vq/quantizers.py:92-117,
 *   vq/algorithms/cvqvae/quantizer_callback.py:75-105, anchors.py:41-85; sparse anchors as vqhip_cvq_rows describes):
 *   vqhip_encode_ex(x, w_in; ZERO_HIST) -> [vqhip_cvq_rows(p_in) unless list_ready] -> vqhip_col_argmin_rows(cap) ->
 *   [vqhip_cvq_pack -> all-reduce] -> vqhip_cvq_apply(w_in, p_in -> w_out, p_out) -> [prefetch: vqhip_cvq_rows(p_out) into
 *   rows/slot/count for the NEXT step, count copied to the pinned HOST word count_host, count_event recorded] ->
 *   vqhip_gather_ste_mse(x, w_out, idx).
 *   cap >= 0: the capacity the listed-code launches are sized for (>= the count; K under HIP-graph capture).
 *   cap <  0: the call reads it from *count_host after hipEventSynchronize(count_event) — the copy the PREVIOUS call's
 *   prefetch queued a whole step earlier; the wait sits behind the enqueue of the encode, so the GPU has work while the host
 *   looks.  This is the ONE place a compute entry point of this library may block the host, and only on the caller's event.
 *   rows / slot / count: K, K, 1 int32, caller-owned and persistent across steps.  cap_used / exchange_floats: written by the
 *   BEFORE phase (host fields).  ws: vqhip_cvq_forward_ws_bytes(N, K, D, cap_max) with cap_max >= cap; packed: at least
 *   vqhip_pack_floats(K, cap, D) floats (exchange != 0).  xq: cosine only (as vqhip_encode).  z_ste / mse nullable together
 *   (no decode tail).  w_out / p_out may alias w_in / p_in.  early_word_host / early_seq_dev: see the struct.
 *   anchor_sync != 0 (with exchange != 0): NearestAnchor(sync=True) — BEFORE ends with vqhip_cvq_col_keys into `keys` instead
 *   of the pack; the caller MIN-all-reduces keys[0 .. cap_used) as int64, calls the VQHIP_STEP_PACK_SYNC phase
 *   (vqhip_cvq_pack_sync), SUM-all-reduces packed[0 .. exchange_floats) and calls AFTER (anchors not averaged).
 *   VQHIP_STEP_ALL does all of it on `comm` (vqhip_allreduce_min_i64 + vqhip_allreduce_packed).
 *
 * vqhip_vqkd_forward — VQKDQuantizer + VQKDCallback in train mode (vq/algorithms/vq/callbacks/normalize.py:22-29,
 *   vq/algorithms/vqkd/quantizers/callbacks.py:44-75,114-129, vq/algorithms/vq/losses.py:53-62 with mse norm=True):
 *   w_mid = normalize(normalize(w_in)), xn = normalize(x), payload zeroed (ONE launch) -> vqhip_encode_ex(xn, w_mid, cosine)
 *   -> histogram header + centroid sums of normalize(xn) scattered straight into the packed buffer (ONE launch; `ordered`
 *   != 0: vqhip_token_order + vqhip_segsum_rows instead, bit-reproducible) -> [all-reduce] -> the EMA update read straight
 *   from the packed buffer (w_mid -> w_out) -> tail (tail != 0): z_ste = xn + (w_out[idx] - xn), mse[0] = mse[1] =
 *   mean((normalize(w_out[idx]) - normalize(xn))^2), mse[2] = mse[3] = 0 — a double-precision sum of per-workgroup partials
 *   added in a fixed order (they are parked in the record area of `ws`, free by then): the same step gives the same bits.
 *   xq: the encode's by-product (as vqhip_encode).  packed: vqhip_pack_floats(K, K, D) floats, always used.
 *   At D <= 32 the front, the tail and the backward give a row 8 / 16 / 32 lanes instead of a wave (same values).
 *   metric: VQHIP_METRIC_COS or VQHIP_METRIC_COS_BF16.  ws: vqhip_vqkd_forward_ws_bytes(N, K, D).
 * vqhip_vqkd_backward — gradient of that tail with respect to x: grad_x = normalize_bwd(x; g_zste + normalize_bwd(xn;
 *   g_loss * 2/(N D) * (normalize(xn) - normalize(w[idx])))); g_zste [N, D] / g_loss (DEVICE scalar) nullable. */
#define VQHIP_STEP_BEFORE_EXCHANGE 1
#define VQHIP_STEP_AFTER_EXCHANGE 2
#define VQHIP_STEP_ALL 3
#define VQHIP_STEP_PACK_SYNC 4     /* anchor_sync only: between the caller's MIN all-reduce of the keys and its SUM all-reduce of packed */
typedef struct vqhip_cvq_forward_t {
    int64_t struct_bytes;
    int64_t N, K;
    int32_t D, x_dtype, metric, world;
    float ema_decay, eps, beta;
    int32_t phases, exchange, list_ready, prefetch;
    int32_t anchor_sync, rank;      /* NearestAnchor(sync=True) with exchange != 0: the key exchange of vqhip_cvq_col_keys; this rank's number */
    int64_t cap;
    const void *x;
    const float *w_in, *p_in;
    float *w_out, *p_out;
    int32_t *rows, *slot, *count;
    int32_t *count_host;            /* pinned HOST word (nullable) */
    void *count_event;              /* hipEvent_t of the caller (nullable) */
    void *comm;
    void *cb; int64_t cb_bytes;
    int64_t *idx; int32_t *hist; float *xq;
    float *packed; int64_t packed_floats;
    float *z_ste, *mse; void *scratch16;
    void *ws; int64_t ws_bytes;
    int64_t cap_used, exchange_floats;      /* OUT (host), written by the BEFORE phase */
    /* early count (nullable pair): as soon as this step's histogram is final (behind the encode at one rank, behind the exchange
     * otherwise) one small launch writes {sequence number << 32 | length of the NEXT step's list} to the pinned HOST word
     * early_word_host and advances the DEVICE counter early_seq_dev by one — for a caller that replays this call from a HIP graph
     * and must choose the next replay's capacity without an event in the middle of the graph (it polls the word) */
    unsigned long long *early_word_host;
    int32_t *early_seq_dev;
    int64_t *keys;                  /* anchor_sync: cap int64 (DEVICE), MIN-all-reduced between BEFORE and PACK_SYNC */
} vqhip_cvq_forward_t;
int64_t vqhip_cvq_forward_ws_bytes(int64_t N, int64_t K, int D, int64_t cap_max);
int vqhip_cvq_forward(vqhip_cvq_forward_t *args, void *stream);

typedef struct vqhip_vqkd_forward_t {
    int64_t struct_bytes;
    int64_t N, K;
    int32_t D, x_dtype, metric, world;
    float ema_decay;
    int32_t phases, exchange, ordered, tail, reserved0;
    const void *x;
    const float *w_in;
    float *w_mid, *w_out;           /* w_out may alias w_mid */
    float *xn, *xq;                 /* [N, D] fp32 each */
    void *comm;
    void *cb; int64_t cb_bytes;
    int64_t *idx; int32_t *hist;
    float *packed; int64_t packed_floats;
    float *z_ste, *mse; void *scratch16;
    void *ws; int64_t ws_bytes;
    int64_t exchange_floats;        /* OUT (host), written by the BEFORE phase */
} vqhip_vqkd_forward_t;
int64_t vqhip_vqkd_forward_ws_bytes(int64_t N, int64_t K, int D);
int vqhip_vqkd_forward(vqhip_vqkd_forward_t *args, void *stream);
int vqhip_vqkd_backward(const void *x, int x_dtype, const float *xn, const float *w, const int64_t *idx, int64_t N, int D,
                        const float *g_zste, const float *g_loss, float *grad_x, void *stream);

/* vqhip_vq_forward — the forward of a quantizer WITHOUT an update callback, or with NormalizeCallback alone (the LlamaGen
 * tokenizer, configs/llamagen/vqgan.py:18-20), as one host call (vq/algorithms/vq/quantizers.py:92-117, callbacks/normalize.py:22-29,
 * losses.py:41-127): [w_out = normalize(w_in) and xn = normalize(x): ONE launch] -> vqhip_encode_ex(xn | x, w_out | w_in) ->
 * vqhip_gather_ste_mse on the same operands.  normalize != 0: both normalisations (w_out [K, D] and xn [N, D] fp32 are written; w_out
 * may alias w_in); normalize == 0: w_out / xn unused.  hist nullable; xq: cosine only; z_ste / mse nullable together.
 * ws: vqhip_workspace_bytes(N, K, D). */
typedef struct vqhip_vq_forward_t {
    int64_t struct_bytes;
    int64_t N, K;
    int32_t D, x_dtype, metric, normalize;
    float beta;
    int32_t reserved0;
    const void *x;
    const float *w_in;
    float *w_out, *xn;
    void *cb; int64_t cb_bytes;
    int64_t *idx; int32_t *hist; float *xq;
    float *z_ste, *mse; void *scratch16;
    void *ws; int64_t ws_bytes;
} vqhip_vq_forward_t;
int vqhip_vq_forward(vqhip_vq_forward_t *args, void *stream);

/* anchors[k] = x[col_idx[k]] (anchors.py:84) as fp32 */
int vqhip_gather_rows(const void *x, int x_dtype, const int64_t *row_idx, int64_t K, int D, float *out,
                      void *stream);

/* ---- elementwise pieces of the backward / unfused path ------------------------------------------------
 * vqhip_diff: sse[0] += sum (a-b)^2 in double (MSELoss forward, losses.py:50,62) and/or out = (a-b)*scale
 *             (its backward); a, b may each be fp32 or bf16; out / sse nullable (not both).
 * vqhip_ste:  out = x + (z - x)                                      (utils/ste.py:10)
 * vqhip_normalize_rows_bwd: gradient of F.normalize(v, dim=1, eps) given the output gradient g. */
int vqhip_diff(const void *a, int a_dtype, const void *b, int b_dtype, int64_t n, float scale,
               const float *scale_dev /* nullable DEVICE scalar multiplied into scale */, float *out, double *sse,
               void *stream);
/* Fused backward of the quantizer forward (embedding_dense_backward + both MSE gradients + STE):
 *   z = W[idx], z_ste = x + sg(z - x), m_cb = mse(z, sg x) (losses.py:50), m_cm = mse(sg z, x) (losses.py:62)
 *   grad_x = g_zste + g_cm*(2/ND)*(x - z);   grad_w[idx] += g_cb*(2/ND)*(z - x)  (fp32 atomics)
 * g_cb, g_cm: DEVICE scalars (upstream gradients of the two MSE values, nullable = 0); g_zste, grad_x, grad_w
 * nullable (grad_w must be zero-initialised by the caller). */
int vqhip_vq_backward(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D,
                      const float *g_zste, const float *g_cb, const float *g_cm, float *grad_x, float *grad_w,
                      void *stream);
/* ... with the upstream gradient g_comb (nullable DEVICE scalar) of the combined value mse[2] = m_cb + beta*m_cm of
 * vqhip_gather_ste_mse: the effective gradients are g_cb + g_comb and g_cm + beta*g_comb (no scalar kernels in between). */
int vqhip_vq_backward_ex(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D,
                         const float *g_zste, const float *g_cb, const float *g_cm, const float *g_comb, float beta,
                         float *grad_x, float *grad_w, void *stream);
/* grad_x of vqhip_vq_backward_ex for a quantizer call on the NCHW feature map (vq/tasks/image_tokenization/models/base.py:116-128),
 * with both rearrangements folded in: g_map [B, D, HW] fp32 is the upstream gradient of the straight-through output AS the map it
 * arrives in (nullable = 0), grad_map [B, D, HW] receives the gradient of the latents as the map the encoder's backward consumes,
 * in grad_dtype (VQHIP_DTYPE_F32 / _BF16: the map's own dtype; bf16 rounds to nearest even):
 *   grad_map[b, d, p] = g_map[b, d, p] + (g_cm + beta*g_comb) * 2/(N D) * (x_rows[n][d] - e[idx[n]][d]),  n = b*HW + p.
 * No transpose launch and no cast around it.  HW % 256 == 0 and D % 32 == 0 (16 x 16 and larger power-of-two maps; other shapes:
 * vqhip_transpose + vqhip_vq_backward_ex).  The codebook gradient is formed by vqhip_vq_backward_ex with grad_x = NULL (or the
 * ordered route). */
int vqhip_vq_backward_map(const void *x_rows, int x_dtype, const float *e, const int64_t *idx, int64_t B, int64_t HW, int D,
                          const float *g_map, const float *g_cm, const float *g_comb, float beta, void *grad_map, int grad_dtype,
                          void *stream);
int vqhip_ste(const void *x, int x_dtype, const float *z, int64_t n, float *out, void *stream);
int vqhip_normalize_rows_bwd(const void *v, int dtype, const float *g, int64_t R, int D, float eps, float *gv,
                             void *stream);

/* ---- deterministic (ordered) codebook-side sums  (SURVEY.md §7 hard part 9) ---------------------------------------
 * vqhip_scatter_add_rows and the grad_w leg of vqhip_vq_backward add with floating-point atomics, i.e. in arrival
 * order.  The ordered route fixes the order instead: vqhip_token_order sorts the token ids by code (stable counting
 * sort, integer arithmetic only): counts[K] = bincount, offsets[K+1] = its exclusive scan, order[N] = token ids, code
 * by code, ascending within a code.  K <= 32768.  `ws` = vqhip_order_workspace_bytes(N, K).
 * Sums: the sorted order is cut into ranges of 64 positions; rows are added in order inside a range; a code whose
 * tokens span several ranges is the sum of its range pieces in range order — an association fixed by the counts alone.
 * vqhip_segsum_rows: dst[k] = that sum of src[order[p]], p in [offsets[k], offsets[k+1]) (all K rows are written, zero
 * for unused codes) — the centroid sums of callbacks.py:60-64.  vqhip_vq_backward_w_ordered: grad_w[k] = the same sum
 * of g_cb * 2/(N*D) * (e_k - x_n) — the codebook gradient of vqhip_vq_backward (call that one with grad_w = NULL).
 * `ws` of the two sums = vqhip_segsum_workspace_bytes(N, D). */
int64_t vqhip_order_workspace_bytes(int64_t N, int64_t K);
int64_t vqhip_segsum_workspace_bytes(int64_t N, int D);
int vqhip_token_order(const int64_t *idx, int64_t N, int64_t K, int32_t *counts, int32_t *offsets, int32_t *order, void *ws,
                      int64_t ws_bytes, void *stream);
int vqhip_segsum_rows(const float *src, const int64_t *idx, const int32_t *order, const int32_t *offsets, int64_t N, int64_t K,
                      int D, float *dst, void *ws, int64_t ws_bytes, void *stream);
int vqhip_vq_backward_w_ordered(const void *x, int x_dtype, const float *e, const int64_t *idx, const int32_t *order,
                                const int32_t *offsets, int64_t N, int64_t K, int D, const float *g_cb, float *grad_w, void *ws,
                                int64_t ws_bytes, void *stream);

/* ---- callers either side of the path ------------------------------------------------------
 * vqhip_transpose: in[B][R][C] -> out[B][C][R] for 2- or 4-byte elements.  With R = channels, C = h*w it is
 *   'b c h w -> (b h w) c' (vq/tasks/image_tokenization/models/base.py:124,140); with R = h*w, C = channels the
 *   inverse '(b h w) c -> b c h w' (base.py:126).
 * vqhip_codebook_metrics: out[0] = nonzero(counts)/K (CodebookUsageMetric, runners/metrics.py:58-62),
 *   out[1] = entropy of counts/sum(counts) in nats (CodebookPPLMetric, :65-73); counts int64[K], out double[2]. */
int vqhip_transpose(const void *in, void *out, int elem_bytes, int64_t B, int R, int C, void *stream);
int vqhip_codebook_metrics(const int64_t *counts, int64_t K, double *out, void *stream);

/* ---- diagnostics ------------------------------------------------------------------------------------
 * Copies the counters of the last vqhip_argmin on `ws` to out[4] (DEVICE int32): rows given a second proposal
 * pass, rows with more than one identified candidate (re-ranked exactly), rows sent to the whole-codebook fp32
 * pass, reserved. */
int vqhip_argmin_stats(const void *ws, int32_t *out, void *stream);

/* Verification aid: scores[N*K] = the fp16-MFMA proposal score of every (row, code) pair (same operands and
 * instruction sequence as the production kernels), margin[N] = the per-row margin the decision step uses (scaled
 * score units, negative = no usable bound), scale[1] = the power-of-two codebook scale.  A test checks the error
 * bound |score - exact score| <= margin/2 against float64. */
int vqhip_debug_proposal_scores(const void *x, int x_dtype, const void *cb, int64_t N, int64_t K, int D, int metric,
                                float *scores, float *margin, float *scale, void *ws, int64_t ws_bytes, void *stream);

/* Per-launch timing of the proposal (distance+argmin) kernel with HIP events recorded on the caller's stream
 * around that launch (bench.py's roofline leg).  enable(1) starts collecting, collect() synchronises on the
 * recorded events, returns the summed kernel milliseconds and launch count (HOST pointers) and resets. */
int vqhip_profile_enable(int on);
/* Knobs for A/B measurements (results never change): key 2 = number of codebook slices (1,2,4,8,16; 0 = automatic);
 * key 3 = workgroup cap of the gather kernel (0 = automatic); key 4 = its streaming mode (1 on, 2 off, 0 = automatic:
 * on when the outputs exceed 192 MiB); key 5 = filtered epilogue of the D <= 128 proposal kernels (default 1); key 6 = decision
 * stage inside the proposal kernel (0 never, 1 always, 2 = where one slice covers the codebook: default); key 8 = no aux reads
 * for cosine / dot codebooks at D <= 32 (1); key 9 = group records at D <= 32, identified by identify32_kernel (32x32x16 form: K <= 131 072, N < 2^30) or by an in-kernel replay (16x16x32 form) (1); key 10 = balanced
 * tiles per workgroup (1); key 11 = the 32x32x16 proposal kernel at D <= 16 (1); key 13 = whole-image tiles in vqhip_gather_ste_map
 * for maps of 256-position images (1); key 15 = the direct fp32 form of vqhip_col_argmin_rows for short lists (1); key 17 = D = 256 batches of
 * more than 16 384 bf16 rows make their token fragments in the proposal kernel's prologue instead of writing and reading a token
 * image (1; 2 = fp32 rows too, a measurement aid); key 18 = the streamed form of the whole-batch fp32 pass (vqhip_argmin_exact,
 * vqhip_distance, the column fallback: exact_stream_kernel) where D % 4 == 0 (bf16 rows: D % 8 == 0) (1; 0 = the register form).
 * Key 12 is a verification aid, not an A/B knob: value V > 0 sends rows 0 .. min(V, N, 1024) - 1 of every vqhip_argmin batch
 * through the last-resort whole-codebook fp32 pass as well (its indices replace the ones the earlier stages wrote — the same
 * ones; a histogram requested from the call counts those rows twice); 0 = off (default).  Any other key: VQHIP_EINVAL. */
int vqhip_set_tuning(int key, int value);
int vqhip_profile_collect(double *ms_sum, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* VQHIP_H_ */
