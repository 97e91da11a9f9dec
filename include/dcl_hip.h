/*
 * dcl_hip.h -- C ABI of libdcl_hip.so: the MI355X (gfx950) implementation of the
 * multi-scale / cross-scale dense pixel-contrastive loss hot path.
 *
 * The reference (RViMLab/ECCV2022-multi-scale-and-cross-scale-contrastive-segmentation)
 * is pure Python; it has no FFI.  Its boundary for this path is the Python call
 * surface losses/LossWrapper.py:40 -> losses/DenseContrastiveLossV2_ms.py:44 ->
 * losses/DenseContrastiveLossV2.py:44.  The entry points below are the operators
 * those three functions decompose into; each one cites the reference lines it
 * replaces.  The host-side mirror of the reference classes (mscs_amd.losses.*)
 * binds them with ctypes; the signatures use only plain
 * pointers and sizes -- no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing synchronises the host;
 *   - return value: 0 on success, otherwise a hipError_t (or a negative DCL_E*
 *     code for argument errors); dcl_last_error() gives the message of the last
 *     failure on the calling thread;
 *   - bank matrices are row-major [Npad, DCL_CP] f32, Npad = N rounded up to
 *     DCL_ROW_TILE, rows >= N and channels >= C are zero.
 */
#ifndef DCL_HIP_H
#define DCL_HIP_H

#include <stdint.h>

#define DCL_AMAX_SLOTS 64   /* partial absmax values a fused BN kernel emits (a power of two) */

#ifdef __cplusplus
extern "C" {
#endif

#define DCL_CP 256        /* padded embedding width handled by the sweep kernels   */
#define DCL_ROW_TILE 128  /* bank rows per workgroup; Npad granularity             */
#define DCL_SEG 256       /* label pixels per histogram segment                    */
#define DCL_MAX_CLASSES 255

#define DCL_EINVAL (-1)
#define DCL_EUNSUPPORTED (-2)

const char *dcl_last_error(void);
int dcl_version(void);
/* Measurement aid (bench.py's in-step kernel timer): with dcl_trace_kernels(1) the convolution / weight-gradient entry
 * points record the symbol of the kernel they chose (tile and variant are picked inside the library);
 * dcl_last_kernel() returns the calling thread's last one ("" if none).  No effect on results; off by default. */
int dcl_trace_kernels(int on);
const char *dcl_last_kernel(void);

/* ---- K1 ---------------------------------------------------------------------------------
 * Nearest down-sample of the label map to (h, w) = (H / scale, W / scale) and per-image class
 * histogram.  Replaces F.interpolate(label.float(), mode='nearest') (DenseContrastiveLossV2.py:205)
 * and `compare = lbl == arange(K); compare.sum(1)` (:100-103).
 *   label     int64 [n, H, W] (class ids already remapped to network ids)
 *   lbl_s     uint8 [n, h*w]           down-sampled ids; ids outside [0,K) -> 255
 *   seg_hist  int32 [n, nseg, K]       per DCL_SEG-pixel segment histogram, nseg = ceil(h*w/DCL_SEG)
 *   counts    int32 [n, K]             must be zero on entry; accumulated with integer atomics
 */
int dcl_label_hist(const int64_t *label, int n, int H, int W, int scale, int K,
                   uint8_t *lbl_s, int32_t *seg_hist, int32_t *counts, void *stream);

/* ---- host: permutation draws -------------------------------------------------------------
 * Native replacement of the T host-side `torch.randperm(count)[:V]` calls
 * (DenseContrastiveLossV2.py:121-122).  `rng_state_host` is the byte image of PyTorch's CPU
 * generator (torch.get_rng_state(), 5056 bytes, at::mt19937): it is advanced IN PLACE by exactly the
 * draws the reference would consume (count - 1 per pair, pairs in order), so writing it back with
 * torch.set_rng_state() leaves the global RNG stream identical to the reference's.
 *   counts_host int32 [T] (HOST), sel_host int32 [T, V] (HOST, out)
 */
int dcl_host_randperm_select(uint8_t *rng_state_host, int64_t state_bytes,
                             const int32_t *counts_host, int T, int V, int32_t *sel_host);

/* ---- K2 ---------------------------------------------------------------------------------
 * Rank-select: pix[t, v] = the sel[t, v]-th pixel (ascending flat index) of class pair_k[t] in
 * image pair_b[t].  Replaces the per-pair `compare[b,:,k].nonzero()` + `idx[perm[:V]]`
 * (DenseContrastiveLossV2.py:117-122); sel holds the first V entries of the host-drawn
 * torch.randperm(count) so indices are bit-identical to the reference.
 *   pair_b, pair_k int32 [T];  sel int32 [T, V];  pix int32 [T, V]
 */
int dcl_rank_select(const uint8_t *lbl_s, const int32_t *seg_hist, int n, int hw, int K,
                    const int32_t *pair_b, const int32_t *pair_k, int T, int V,
                    const int32_t *sel, int32_t *pix, void *stream);

/* ---- K3 ---------------------------------------------------------------------------------
 * Gather + L2-normalise into a bank.  Replaces `features[b, :, idx]` (:123),
 * F.normalize(dim=1) (:138) and the transpose/contiguous/view (:139-149).
 * Bank row (u*V + v) holds pixel pix[slot_pair[u], v] of image pair_b[slot_pair[u]]; slot order
 * is chosen by the host (class-major).  feat element (b, c, p) is at
 * feat[b*stride_n + c*stride_c + p*stride_p] (NCHW: stride_c = h*w, stride_p = 1; NHWC: 1, C).
 *   bank f32 [Npad, DCL_CP] (fully written, padding zeroed);  nrm f32 [Npad] raw L2 norms
 */
int dcl_gather_normalize(const float *feat, int64_t stride_n, int64_t stride_c, int64_t stride_p,
                         int C, const int32_t *pix, const int32_t *pair_b,
                         const int32_t *slot_pair, int T, int V, float *bank, float *nrm,
                         void *bank_h /* f16 [Npad, 2*DCL_CP] = (hi | lo) halves of bank * 2^10, or NULL */,
                         void *stream);

/* Reference-layout raw bank X[t, c, v] = features[b_t, c, pix[t, v]] (the `sampled_features`
 * member of the 4-tuple returned by DenseContrastiveLossV2.forward, :61). */
int dcl_gather_raw(const float *feat, int64_t stride_n, int64_t stride_c, int64_t stride_p, int C,
                   const int32_t *pix, const int32_t *pair_b, int T, int V, float *X,
                   void *stream);
/* Its adjoint: dfeat[b_t, c, pix[t, v]] = dX[t, c, v] into a ZERO-FILLED map with the given element strides (every
 * address written once: sampled pixels are unique per image within a scale) -- autograd's index backward of
 * `features[b, :, idx]` (losses/DenseContrastiveLossV2.py:123) for the `sampled_features` a bare DenseContrastiveLossV2
 * with cross_scale_contrast returns (:58-61). */
int dcl_scatter_raw(const float *dX, int64_t stride_n, int64_t stride_c, int64_t stride_p, int C,
                    const int32_t *pix, const int32_t *pair_b, int T, int V, float *dfeat, void *stream);

/* ---- K4 ---------------------------------------------------------------------------------
 * Fused similarity + masked InfoNCE forward; the N1 x N2 matrix is never materialised.
 * Replaces matmul/div (:150, ms:114), get_masks2 (:154-171, ms:118-130) and get_loss /
 * InfoNce_loss (:173-192, ms:132-161).
 *
 * Rows of A are anchors, rows of B are contrast samples.  Positives of anchor row i are the
 * contiguous column range [rng_lo[i / V1], rng_hi[i / V1]) of B (banks are class-sorted), minus
 * column i itself when `intra` != 0; every other column < N2 is a negative.
 *
 * dcl_infonce_fwd runs both forward sweeps:
 *   Z[i]       = sum_neg exp(s_ij),               s_ij = <A_i, B_j> / tau
 *   rowloss[i] = sum_pos (s_ij - log(exp(s_ij) + Z[i]))
 *   W[i]       = sum_pos 1 / (exp(s_ij) + Z[i])
 * and loss[0] = -(1/N1) sum_i rowloss[i] / P_i with P_i = #positives (intra: 0/0 = NaN like the
 * reference; cross-scale: P_i -> max(P_i, 1), ms:148-152).
 *   zpart  f32 workspace [nsplit * N1pad];  Z, rowloss, W  f32 [N1pad];  loss f32 [1]
 */
int dcl_infonce_fwd(const float *A, int N1, int V1, const float *B, int N2,
                    const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                    int nsplit, float *zpart, float *Z, float *rowloss, float *W, float *loss,
                    void *stream);

/* The three stages of dcl_infonce_fwd, exposed separately so that the contrast bank may consist of
 * several SEGMENTS (the per-rank banks of the all-gathered negative bank, SURVEY.md section 8 row e):
 *   zsweep    one segment's negatives: zpart slice [nsplit * N1pad] (slices of all segments are summed)
 *   possweep  one segment's positives; reduces ALL zsplits slices into Z, then writes (accumulate = 0)
 *             or adds (accumulate = 1) rowloss / W.  Call in a fixed segment order.
 *   loss      -(1/N1) sum_i rowloss_i / P_i with P_i = pcount[i / V1] if pcount != NULL (positives over
 *             all segments, self already excluded), else derived from rng_lo/hi as in dcl_infonce_fwd.
 *
 * Ah / Bh (both or neither): the banks in f16x3 format (dcl_gather_normalize's bank_h).  When given, the
 * similarity product <A_i, B_j> runs as three f16 MFMA passes (hi.hi + hi.lo + lo.hi, f32 accumulation,
 * fp32-equivalent accuracy) instead of f32 MFMA; everything downstream is unchanged.
 */
int dcl_infonce_zsweep(const float *A, int N1, int V1, const float *B, int N2,
                       const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int nsplit,
                       float *zpart, const void *Ah, const void *Bh, void *stream);
int dcl_infonce_possweep(const float *A, int N1, int V1, const float *B, int N2,
                         const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                         const float *zpart, int zsplits, int accumulate, float *Z, float *rowloss,
                         float *W, const void *Ah, const void *Bh, void *stream);
int dcl_infonce_loss(const float *rowloss, const int32_t *rng_lo, const int32_t *rng_hi,
                     const int32_t *pcount, int N1, int V1, int intra, float *loss, void *stream);
/* One pass over the bank instead of two (single-segment terms): dcl_infonce_zsweep_keep = dcl_infonce_zsweep that also KEEPS the
 * raw similarities of every row's positive columns, spos f32 [N1pad][spos_ld], spos_ld >= max_u (rng_hi[u] - rng_lo[u])
 * (column j of row i at spos[i * spos_ld + j - rng_lo[i / V1]]; the positive range of a row is contiguous in the class-major
 * bank); dcl_infonce_pos_finish then produces dcl_infonce_possweep's Z / rowloss / W from them (same per-element arithmetic;
 * f16x3 = the sweep ran with Ah / Bh).  Reference: DenseContrastiveLossV2.py:173-192, _ms.py:132-161. */
int dcl_infonce_zsweep_keep(const float *A, int N1, int V1, const float *B, int N2, const int32_t *rng_lo,
                            const int32_t *rng_hi, float inv_tau, int nsplit, float *zpart, const void *Ah, const void *Bh,
                            float *spos, int spos_ld, void *stream);
int dcl_infonce_pos_finish(const float *spos, int spos_ld, int N1, int V1, const int32_t *rng_lo, const int32_t *rng_hi,
                           float inv_tau, int intra, int f16x3, const float *zpart, int zsplits, float *Z, float *rowloss,
                           float *W, void *stream);

/* ---- K5 ---------------------------------------------------------------------------------
 * InfoNCE backward.  With G_ij = dL/ds_ij:
 *   dcl_infonce_prep_stats   packs per-row statistics {Z, gs*W/(N1*P), gs*Z/(N1*P), 0} where
 *                            gs = wscale * (*grad_out) * inv_tau  (grad_out: device scalar, may be NULL = 1)
 *   dcl_infonce_bwd          dpart[split][a, :] = sum_b H_ab * B_b  over that split's columns, where
 *                            H_ab = use_row * G(a -> b; rstat[a]) + use_col * G(b -> a; cstat[b]).
 *      intra-scale:  A = B, rstat = cstat, use_row = use_col = 1      (dF = (G + G^T) F / tau)
 *      cross, dF1:   A = F1, B = F2, use_row = 1, use_col = 0         (G F2 / tau)
 *      cross, dF2:   A = F2, B = F1, use_row = 0, use_col = 1, cstat = stats of F1's rows
 *                    (rng_* are then F2's slot ranges into F1)        (G^T F1 / tau)
 *   stat  f32 [N1pad + 1, 4]: rows 0..N1pad-1 as above, row N1pad = {max_i max(e^{1/tau} |cW_i|, |coef_i|), 0, 0, 0}
 *         (the bound the f16x3 backward uses to scale dL/ds into f16 range);   dpart f32 [nsplit, N1pad, DCL_CP]
 */
int dcl_infonce_prep_stats(const float *Z, const float *W, const int32_t *rng_lo,
                           const int32_t *rng_hi, const int32_t *pcount /* may be NULL */, int N1,
                           int V1, int intra, float wscale, float inv_tau, const float *grad_out,
                           float *stat, void *stream);

int dcl_infonce_bwd(const float *A, int N1, int V1, const float *B, int N2,
                    const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                    int use_row, int use_col, const float *rstat, const float *cstat, int nsplit,
                    float *dpart, const void *Ah /* f16x3 banks, see dcl_infonce_zsweep */,
                    const void *Bh, void *stream);

/* Stream-K form of dcl_infonce_bwd for f16x3 banks (Ah, Bh required): same sum, but dout f32 [N1pad, DCL_CP] is the
 * FINISHED gradient tile of every row block instead of nsplit partial slabs.  G = dcl_infonce_bwd_streamk_workgroups(N1,
 * N2) persistent workgroups (one per CU; 0 = switched off with dcl_infonce_set_streamk(0), use dcl_infonce_bwd) share the
 * (row block, 32-column chunk) sequence in equal contiguous ranges; the workgroup that reaches the end of a row block
 * adds the partial tiles of the (lower-numbered) workgroups that covered its earlier chunks in ascending order --
 * bitwise reproducible -- and writes the tile.  ws f32 [G, DCL_ROW_TILE, DCL_CP] partial tiles; flags int32 [1 + G],
 * zero-initialised ONCE by the caller (one pair per stream serves every launch of it, also launches with a smaller G):
 * flags[0] is the error word (below), flags[1 + g] holds the number of the launch whose partial tile g is valid (the library numbers the launches of a flags pointer), so nothing is reset between
 * launches and an aborted launch leaves nothing a later one could mistake.  G = min(units, CUs of the device).  The
 * owner's wait for a contributor is BOUNDED (dcl_infonce_set_streamk_timeout_ms, default 2000 ms per hand-over): when it
 * expires -- a contributor never became resident: CU masks, other persistent kernels, several ranks on one device --
 * flags[0] is incremented and the kernel finishes with an INVALID gradient instead of hanging; the caller must read
 * flags[0] (it only ever grows) and fall back to dcl_infonce_bwd (dcl_infonce_set_streamk(0)).
 * Replaces the same reference lines as dcl_infonce_bwd (autograd of losses/DenseContrastiveLossV2.py:150-192 and
 * losses/DenseContrastiveLossV2_ms.py:84-161). */
int dcl_infonce_bwd_streamk_workgroups(int N1, int N2);
/* Column slices (round 4): the chunk axis of the contrast bank is cut into 4 (default; 1 | 4 | 8:
 * dcl_infonce_set_streamk_slices) slices, each swept -- as a stream-K problem of its own -- by the workgroups of one pair of
 * XCDs (one XCD for 8), so that a slice of the (hi | lo) bank stays in those XCDs' L2 instead of every workgroup streaming the
 * whole bank through the fabric.  Every slice leaves its own finished slab: dout is f32 [slabs][N1pad][DCL_CP] with slabs =
 * dcl_infonce_bwd_streamk_slabs(N1, N2) (1 when the grid is not a whole number of XCD rounds or the bank is short), to be
 * summed in slab order (dcl_normalize_bwd_scatter does). */
int dcl_infonce_bwd_streamk_slabs(int N1, int N2);
int dcl_infonce_set_streamk_slices(int n);
int dcl_infonce_set_streamk(int on);
int dcl_infonce_set_streamk_timeout_ms(int ms);
int dcl_infonce_bwd_streamk(const float *A, int N1, int V1, const float *B, int N2,
                            const int32_t *rng_lo, const int32_t *rng_hi, float inv_tau, int intra,
                            int use_row, int use_col, const float *rstat, const float *cstat, float *dout,
                            float *ws, int32_t *flags, const void *Ah, const void *Bh, void *stream);

/* ---- K6 ---------------------------------------------------------------------------------
 * Sum the partial dF slabs of a bank in a fixed order, apply the VJP of F.normalize
 * (dx = (dF - f (f.dF)) / max(|x|, 1e-12)) and scatter into the dense feature gradient
 * (autograd's IndexBackward x T + add_ x T in the reference).  dfeat must be zero on entry; every
 * sampled pixel is written exactly once (pixels are unique within a scale), so no atomics.
 *   slabs_host   host array of `nslab` device pointers, each f32 [Npad, DCL_CP]
 */
int dcl_normalize_bwd_scatter(const float *const *slabs_host, int nslab, const float *bank,
                              const float *nrm, const int32_t *pix, const int32_t *pair_b,
                              const int32_t *slot_pair, int T, int V, int C, float *dfeat,
                              int64_t stride_n, int64_t stride_c, int64_t stride_p,
                              float *amax /* [DCL_AMAX_SLOTS] zero-initialised: max|dfeat| max-ed in; or NULL */,
                              void *stream);

/* ---- fused BatchNorm2d (+ residual) (+ ReLU), training mode, NCHW f32 ----
 * Replaces nn.BatchNorm2d -> (out += identity) -> nn.ReLU chains of the models (reference
 * models/HRNet.py:77-93, 118-137, 270-285; models/Projector.py:59-63) and their autograd backward.
 *   forward : dcl_bn_stats -> [all-reduce of sums across ranks = SyncBatchNorm] -> dcl_bn_finalize ->
 *             dcl_bn_apply:      y = relu(gamma * (x - mean) * invstd + beta + res)
 *   backward: dcl_bn_bwd_reduce -> [all-reduce of sums] -> dcl_bn_bwd_apply:
 *             g = dy * (y > 0),  dx = gamma * invstd * (g - sum_g/count - xhat * sum_gx/count),  dres = g
 *   sums f32 [C, 2];  part f32 workspace [C * dcl_bn_num_slices(N, C) * 2];  count = elements per channel
 *   over ALL ranks;  res / dres / gamma / beta / running_* may be NULL.  dgamma = sums[:,1], dbeta = sums[:,0].
 */
int dcl_bn_num_slices(int N, int C);
int dcl_bn_stats(const float *x, int N, int C, int HW, float *part, float *sums, void *stream);
int dcl_bn_stats_finalize(const float *x, int N, int C, int HW, float eps, float momentum, float *part,
                          float *sums, float *mean, float *invstd, float *running_mean,
                          float *running_var, int64_t *batches_tracked /* += 1, or NULL */,
                          void *stream);   /* single-rank: stats + finalize */
int dcl_bn_finalize(const float *sums, int C, double count, float eps, float momentum, float *mean,
                    float *invstd, float *running_mean, float *running_var, void *stream);
int dcl_bn_apply(const float *x, const float *res, const float *mean, const float *invstd,
                 const float *gamma, const float *beta, int N, int C, int HW, int relu, float *y,
                 float *amax /* [DCL_AMAX_SLOTS] zero-initialised: max|y| of plane p is max-ed into slot p % DCL_AMAX_SLOTS; or NULL */, void *stream);
int dcl_bn_bwd_reduce(const float *dy, const float *x, const float *y /* NULL with relu and no residual: the mask
                      y > 0 is recomputed from x, gamma, beta */, const float *mean, const float *invstd,
                      const float *gamma, const float *beta, int N, int C, int HW, int relu, float *part,
                      float *sums, float *dbeta /* [C] or NULL */, float *dgamma /* [C] or NULL */, void *stream);
int dcl_bn_bwd_apply(const float *dy, const float *x, const float *y /* as above */, const float *mean,
                     const float *invstd, const float *gamma, const float *beta, const float *sums, double count, int N,
                     int C, int HW, int relu, float *dx, float *dres,
                     float *amax /* [DCL_AMAX_SLOTS] zero-initialised: max|dx|, same slots; or NULL */, void *stream);

/* Fused forms (what FusedBatchNorm2d uses): the per-slice partial sums `part` are combined in the prologue of the
 * apply kernels instead of by a separate launch.  Forward: dcl_bn_stats_part -> [all-reduce of part] ->
 * dcl_bn_apply_fused (also writes mean / invstd, updates the running statistics and num_batches_tracked).
 * Backward: dcl_bn_bwd_reduce_part -> [part_global = all-reduce of a copy] -> dcl_bn_bwd_apply_fused (dx from
 * part_global; dbeta / dgamma from this rank's part_local).  part: f32 [C * dcl_bn_num_slices(N, C) * 2]. */
int dcl_bn_stats_part(const float *x, int N, int C, int HW, float *part,
                      const float *pivot_src /* [C] or NULL: shift of the sums (the running mean: equal on all ranks);
                                                part then holds sum (x - p), sum (x - p)^2 -- no cancellation in
                                                E[x^2] - mean^2 when |mean| >> std */,
                      float *pivot_out /* [C]: the pivot used, handed to dcl_bn_apply_fused (NULL iff pivot_src is) */,
                      void *stream);
int dcl_bn_apply_fused(const float *x, const float *res, const float *part, double count, float eps, float momentum,
                       const float *gamma, const float *beta, int N, int C, int HW, int relu, float *y, float *mean,
                       float *invstd, float *running_mean, float *running_var, int64_t *batches_tracked,
                       float *amax, const float *pivot /* pivot_out of dcl_bn_stats_part, or NULL */,
                       void *relu_mask /* optional (relu, HW % 256 == 0): N * C * HW / 8 bytes, the sign bits of y packed
                                          for the backward -- see below */,
                       void *stream);
/* The HRNet head's norm folded into its classifier (reference models/HRNet.py:596-600: conv3x3 -> BatchNorm2d -> conv1x1, no
 * activation in between): conv1x1(bn(z), W) = (W diag(sc)) z + W sh, so the forward is dcl_bn_stats_part + dcl_bn_finalize_pre (mm,
 * amax = NULL) and a GEMM on z.  Backward: with G = dl z^T and s = sum dl (the classifier's weight-gradient products) the caller
 * forms c1 = -sc invstd dgamma / n, c0 = -sc dbeta / n - c1 mean and this call writes
 *     dz[n, c, p] = sum_k wt[c][k] dl[n, k, p] + c1[c] z[n, c, p] + c0[c]
 * in one pass -- the classifier's data-gradient GEMM, the norm's backward reduce and its backward apply in one HBM-bound kernel.
 * dl [N, K, HW] with K <= 32, wt [C][4 ceil(K / 4)] = (W diag(sc))^T zero-padded, HW % 4 == 0, 16-byte aligned tensors;
 * amax: DCL_AMAX_SLOTS zero-initialised slots receiving max|dz|, or NULL. */
int dcl_head_norm_dz(const float *dl, const float *z, const float *wt, const float *c0, const float *c1, int N, int K, int C,
                     int HW, float *dz, float *amax, void *stream);
/* relu = 2 in the two backward calls: `y` is the packed mask dcl_bn_apply_fused wrote (1/32 of y's size), not y
 * itself -- what the backward of a norm + residual + ReLU needs from y is only y > 0.
 * dcl_bn_bwd_apply_fused only, relu + 4: the norm's INPUT x is the output of a ReLU (conv -> ReLU -> norm, the projection heads,
 * reference models/Projector.py:46-51): dx is zeroed where x <= 0 -- the ReLU's backward inside this kernel, which reads x anyway. */
/* dcl_bn_apply_fused with an explicit number `ns` (1 .. 64) of partial sums per channel, part f32 [C][ns][2], from any producer
 * (reference models/HRNet.py:77-93 conv -> bn: the norm after a convolution). */
int dcl_bn_apply_parts(const float *x, const float *res, const float *part, int ns, double count, float eps, float momentum,
                       const float *gamma, const float *beta, int N, int C, int HW, int relu, float *y, float *mean,
                       float *invstd, float *running_mean, float *running_var, int64_t *batches_tracked,
                       float *amax, const float *pivot, void *relu_mask, void *stream);
int dcl_bn_bwd_reduce_part(const float *dy, const float *x, const float *y, const float *mean, const float *invstd,
                           const float *gamma, const float *beta, int N, int C, int HW, int relu, float *part,
                           void *stream);
int dcl_bn_bwd_apply_fused(const float *dy, const float *x, const float *y, const float *mean, const float *invstd,
                           const float *gamma, const float *beta, const float *part, const float *part_local,
                           double count, int N, int C, int HW, int relu, float *dx, float *dres, float *dbeta,
                           float *dgamma, float *amax, void *stream);
/* A norm whose normalised output is never written: conv1 -> bn1 -> relu -> conv2 of the reference's residual blocks
 * (models/HRNet.py:77-93 BasicBlock, :117-137 Bottleneck conv1 -> bn1 -> relu -> conv2) with bn1's apply pass folded into the
 * operand staging of conv2's forward (dcl_conv3x3_pre_f16x3) and weight gradient (dcl_wgrad3x3_pre_f16x3).
 *   dcl_bn_stats_minmax_part : dcl_bn_stats_part + mm f32 [C * ns * 2], the per-slice {min, max} of x
 *   [SyncBatchNorm: all-reduce of part, as before; mm stays local]
 *   dcl_bn_finalize_pre      : mean / invstd / running statistics / num_batches_tracked exactly as dcl_bn_apply_parts computes
 *                              them, pre_sc[c] = invstd gamma, pre_sh[c] = beta - mean pre_sc (the consumer forms
 *                              relu(fma(x, pre_sc, pre_sh)): bitwise what dcl_bn_apply_parts(relu = 1) would have written), and
 *                              amax [DCL_AMAX_SLOTS] (zero-initialised; or NULL) = max of that tensor, from the extrema.
 * The backward is the unchanged pair dcl_bn_bwd_reduce_part / dcl_bn_bwd_apply_fused with y = NULL, relu = 1 (mask from x). */
int dcl_bn_stats_minmax_part(const float *x, int N, int C, int HW, float *part, float *mm, const float *pivot_src,
                             float *pivot_out, void *stream);
int dcl_bn_finalize_pre(const float *part, const float *mm, int ns, double count, float eps, float momentum,
                        const float *gamma, const float *beta, int C, float *mean, float *invstd, float *running_mean,
                        float *running_var, int64_t *batches_tracked, const float *pivot, float *pre_sc, float *pre_sh,
                        float *amax, void *stream);
/* The two calls above in ONE launch for one rank (nothing to exchange in between): the workgroup that finishes a channel's
 * statistics last finalises the channel (agent-scope hand-over of the partial results, same sums in the same order: bitwise the
 * outputs of the two-call form).  tickets: C zero-initialised 32-bit words; the pivot of the sums is running_mean (NULL: 0). */
int dcl_bn_stats_pre(const float *x, int N, int C, int HW, float *part, float *mm, void *tickets, double count, float eps,
                     float momentum, const float *gamma, const float *beta, float *mean, float *invstd, float *running_mean,
                     float *running_var, int64_t *batches_tracked, float *pre_sc, float *pre_sh, float *amax, void *stream);


/* ---- bilinear up-sampling, NCHW f32 (planes = N * C), ATen index arithmetic ----------------------------
 * Replaces F.interpolate(mode='bilinear') in the models (reference models/HRNet.py:279-282, 549-551, 638)
 * and its autograd backward (gather form: deterministic, no atomics). */
int dcl_upsample_bilinear_fwd(const float *x, const float *addend /* [planes,H,W] or NULL: y = addend + up(x) */,
                              int planes, int h, int w, int H, int W, int align_corners,
                              int relu /* y = max(y, 0): the ReLU that ends an exchange-module sum, HRNet.py:285 */,
                              float *y, void *stream);
int dcl_upsample_bilinear_bwd(const float *dy, int planes, int h, int w, int H, int W, int align_corners,
                              float *dx, void *stream);
/* The same into / out of a channel slice [c0, c0 + C) of a wider [N, ctot, H, W] tensor: the concatenation of the four
 * HRNet branches (reference models/HRNet.py:549-553) is written once, without separate up-sampled maps and a cat copy,
 * and its gradient is read in place. */
int dcl_upsample_bilinear_fwd_slice(const float *x, int N, int C, int h, int w, int H, int W, int align_corners,
                                    float *y_wide, int ctot, int c0, void *stream);
int dcl_upsample_bilinear_bwd_slice(const float *dy_wide, int ctot, int c0, int N, int C, int h, int w, int H, int W,
                                    int align_corners, float *dx, void *stream);

/* ---- 3x3 convolution of bilinearly up-sampled maps without the up-sampled maps (head of HRNet: reference
 * models/HRNet.py:549-553 concat of the up-sampled branches, :596-600 the 3x3 head convolution).  With z[tap] = W_tap x, a 1x1
 * convolution at LOW resolution with 9 * Co output maps (map tap * Co + co, tap = ky * 3 + kx),
 *   conv3x3(up(x), W, padding 1)[n, co, Y, X] = sum_tap interp(z[n, tap * Co + co], (Y + ky - 1, X + kx - 1))   (0 outside the image)
 * dcl_tapup_fwd  y f32 [N, Co, H, W] = (accumulate ? y : 0) + that sum over one or two sources (z1 may be NULL);
 * dcl_tapup_bwd  dz = its adjoint applied to dy f32 [N, Co, H, W], one source per call.
 * z / dz layout: channel_major = 0: [N, 9 * Co, h, w]; 1: [9 * Co, N, h, w] (what ONE GEMM over all images produces).
 * Bilinear index arithmetic as dcl_upsample_bilinear_*; gather form, deterministic. */
int dcl_tapup_fwd(const float *z0, int h0, int w0, const float *z1 /* may be NULL */, int h1, int w1, int N, int Co, int H,
                  int W, int align_corners, int channel_major, float *y, int accumulate, void *stream);
int dcl_tapup_bwd(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners, int channel_major,
                  float *dz, void *stream);
/* dcl_tapup_bwd that also max-es max|dz| into dz_amax[0] (zero-initialised by the caller; integer atomic max): the operand scale of
 * the backward GEMMs that consume dz (an a-priori bound from max|dy| is 16-256x too large: bits of the f16 split given away). */
int dcl_tapup_bwd_amax(const float *dy, int N, int Co, int H, int W, int h, int w, int align_corners, int channel_major,
                       float *dz, float *dz_amax, void *stream);
/* tuning hook: 2 (default) = windowed horizontal pass (k_tapup_bwd_w), 1 = the first form (also the fallback for column
 * windows beyond 8 float4s). */
int dcl_tapup_set_bwd_form(int form);
/* 1 when dcl_tapup_fwd on the sources [h0, w0] (+ [h1, w1]; h1 = 0: one source) and dcl_tapup_bwd of each fit their LDS tiles for
 * an [H, W] output, else 0: asked before a step commits to the split head (no failure mid-step / only in the backward). */
int dcl_tapup_supported(int h0, int w0, int h1, int w1, int H, int W, int align_corners);

/* ---- split-f16 GEMM, f32 in / out (csrc/dcl_gemm.hip) ------------------------------------------------------------
 * C[b][m][n] = (accumulate ? C : 0) + bias[n] + sum_k A[b](m, k) * B[b](n, k),  b < batch, at fp32-equivalent accuracy
 * (hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16, f32 accumulation).  Drop-in for the fp32 library GEMMs behind
 * nn.Linear on token-major rows (reference models/Swin.py:62-76, :198-230, :357-362: forward x W^T + b, data gradient
 * dy W, weight gradient dy^T x), the large 1x1 convolutions (models/HRNet.py:63-100, :236-262, models/UPerNet.py) and
 * the tap products of the HRNet head (dcl_tapup_*).
 *   operand X in {A, B} is addressed as (row, k): x_kmajor = 1: element at X[row * ldx + k] (the contraction index is
 *   contiguous); 0: X[k * ldx + row] (the row index is contiguous, rows % 4 == 0).  K % 32 == 0 (any K >= 32 when BOTH operands are row-contiguous), ldx % 4 == 0, 16-byte
 *   aligned bases and batch strides (dcl_gemm_supported says whether a shape qualifies; 0 -> use the library).
 *   a_amax / b_amax: a_count / b_count device floats whose maximum is an upper bound of max|A| / max|B| (dcl_absmax,
 *   or a producer's partial maxima): the operand scales are powers of two derived from them on the device.  c_amax (optional, zero-initialised by the caller)
 *   receives max|C| through an integer atomic max -- the tag of the next consumer.
 *   splitk > 1: the contraction is cut into splitk ranges, each writes a slab to ws (dcl_gemm_workspace_floats floats)
 *   and a second kernel sums the slabs in ascending order (+ bias, + C): deterministic; for outputs too small to fill
 *   256 CUs (weight gradients).  dcl_gemm_suggest_splitk returns 1 when the tiles alone fill the chip.
 *   dcl_gemm_set_tile: 0 automatic, 1..5 = 256x256 / 256x128 / 128x256 / 128x128 / 256x192 workgroup tile (A/B runs). */
int dcl_gemm_supported(int M, int N, int K, int64_t lda, int a_kmajor, int64_t ldb, int b_kmajor);
int64_t dcl_gemm_workspace_floats(int M, int N, int batch, int splitk);
int dcl_gemm_suggest_splitk(int M, int N, int K, int batch);
int dcl_gemm_set_tile(int tile);
int dcl_gemm_f16x3(const float *A, int64_t lda, int a_kmajor, int64_t strideA, const float *B, int64_t ldb, int b_kmajor,
                   int64_t strideB, int M, int N, int K, int batch, const float *a_amax, int a_count,
                   const float *b_amax, int b_count,
                   const float *bias /* [N] or NULL */, float *C, int64_t ldc, int64_t strideC, int accumulate,
                   float *c_amax /* or NULL */, int splitk, float *ws /* or NULL */,
                   float *a_rowsum /* or NULL; [M] = sum_k A(m, k), row-contiguous A and batch 1 only: the bias gradient of a
                                      Linear (column sums of dy) rides on its weight-gradient GEMM, whose A operand is dy^T */,
                   void *stream);

/* The same product (batch 1, no k-split) with a fused epilogue, v = the product (+ bias); C2 / aux are [M, N] with C's ldc:
 *   ep 1  GELU forward   C = v (the pre-activation, kept for the backward), C2 = gelu(v): Mlp.fc1 + act in one launch (reference
 *                        models/Swin.py:62-76; nn.GELU(), erf form).  Both operands k-major (a Linear's forward).
 *   ep 2  GELU backward  C = v * gelu'(aux), aux = the pre-activation: the data gradient of fc2 handed to fc1 as its dy.
 *                        A k-major, B row-contiguous (a Linear's data gradient).
 *   ep 3  residual       C = aux + rowscale[row / rows_per_scale] * v (rowscale NULL: 1): shortcut + drop_path(branch) with the
 *                        per-sample factor of DropPath (Swin.py:318-321).  Both operands k-major.
 * c_amax (optional) receives max|C|. */
/* dcl_gemm_f16x3 (batch 1) with a per-token factor on the A operand: element (row, k) of A times a_scale[t / a_scale_group], t = the
 * element's row for a k-major A (a Linear's data gradient, A = dy [tokens, N]) or its k for a row-contiguous A (the weight
 * gradient, A = dy^T; a_scale_group % 32 == 0).  DropPath's per-sample factor on a branch's gradient (reference
 * models/Swin.py:318-321) costs no pass over dy: it is folded into the scale of the f16 split.  B row-contiguous.  a_rowsum (weight
 * gradient: the bias gradient) sums the SCALED rows.  ep 0, or 2 = the GELU-backward epilogue of dcl_gemm_f16x3_ep (aux). */
int dcl_gemm_f16x3_ascaled(const float *A, int64_t lda, int a_kmajor, const float *B, int64_t ldb, int b_kmajor, int M, int N, int K,
                           const float *a_amax, int a_count, const float *b_amax, int b_count, float *C, int64_t ldc,
                           float *c_amax /* or NULL */, int splitk, float *ws /* or NULL */, float *a_rowsum /* or NULL */,
                           const float *a_scale, int a_scale_group, int ep, const float *aux /* ep 2 */, void *stream);
int dcl_gemm_f16x3_ep(const float *A, int64_t lda, int a_kmajor, const float *B, int64_t ldb, int b_kmajor, int M, int N, int K,
                      const float *a_amax, int a_count, const float *b_amax, int b_count, const float *bias /* or NULL */,
                      float *C, int64_t ldc, float *c_amax /* or NULL */, int ep, float *C2 /* ep 1 */,
                      const float *aux /* ep 2, 3 */, const float *rowscale /* ep 3, or NULL */, int rows_per_scale,
                      void *stream);

/* out = a + b (+ c) (+ d), n floats: the gradient of a tensor with several consumers in one pass (HRNet exchange
 * modules: every branch output feeds all fuse rows, reference models/HRNet.py:264-287). */
int dcl_add_n(const float *a, const float *b, const float *c /* or NULL */, const float *d /* or NULL */, int64_t n,
              float *out, void *stream);

/* ---- direct f16x3 3x3 convolution (stride 1, pad 1, NCHW f32 in / out) --------------------------------------
 * Replaces the nn.Conv2d(C, C, 3, 1, 1, bias=False) of the reference's BasicBlock / Bottleneck
 * (models/HRNet.py:32-60, 63-100) -- ~80 % of HRNet-W48's FLOPs -- and its data gradient (same kernel on the
 * transposed, tap-flipped weights).  Arithmetic: both operands split into f16 (hi, lo) pairs after a
 * power-of-two scaling, products hi.hi + hi.lo + lo.hi on the f16 MFMA with f32 accumulation.
 * Operand scales are derived ON THE DEVICE from absmax values (s = 2^floor(log2(2^14 / max|v|))): the fused BN
 * kernels emit partial maxima of their outputs (dcl_bn_apply / dcl_bn_bwd_apply `amax`, DCL_AMAX_SLOTS values: one
 * load per lane in the consumer's prologue instead of a loop over N * C per-plane values), dcl_absmax covers
 * every other tensor.
 *   dcl_absmax      : out[0] = max(out[0], max|x|)  (out zero-initialised by the caller)
 *   dcl_conv3x3_pack: w [M][K][3][3] (transposed = 0) or [K][M][3][3] read as its data-gradient kernel
 *                    (transposed = 1) -> MFMA fragment order, ceil(M/32) * ceil(K/16) * 9 * 2 KiB at wp;
 *                    wamax = max|w| (1 float)
 *   dcl_conv3x3_f16x3: y[N,Cout,H,W] = conv(x[N,Cin,H,W], packed weights); xamax = xcount partial maxima of |x|;
 *                    tile_r / tile_p select the workgroup tile (channel tiles per wave / rows per wave),
 *                    0 = automatic */
int dcl_absmax(const float *x, int64_t n, float *out, void *stream);
/* out[0] = max(a[0..na)) + max(b[0..nb)): bound of max|A + B| from the absmax partials of A and B (the sum of an
 * up-sampled map and its addend: interpolation is a convex combination, a ReLU on top only shrinks) -- saves the
 * dcl_absmax pass over the sum. */
int dcl_amax_sum2(const float *a, int na, const float *b, int nb, float *out, void *stream);
/* Multi-tensor forms: ONE launch for all weights of a model.  Job tables live in device memory:
 *   absmax job {const float *x; float *out; int64 n; int32 first_block; int32 pad}   (4096 elements / workgroup)
 *   pack job   {const float *w; void *wp; const float *amax; int32 M, K, transposed, first_block}  (256 items / wg)
 * blk2job[b] = index of the job workgroup b works on; job.first_block = its first workgroup. */
int dcl_absmax_multi(const void *jobs, const int32_t *blk2job, int nblocks, void *stream);
int dcl_conv3x3_pack_multi(const void *jobs, const int32_t *blk2job, int nblocks, void *stream);
int dcl_conv3x3_pack(const float *w, int M, int K, int transposed, const float *wamax, void *wp, void *stream);
int dcl_conv3x3_f16x3(const float *x, int N, int Cin, int H, int W /* stored input size */, const void *wp, int Cout,
                      const float *xamax, int xcount, const float *wamax,
                      const float *addend /* [N,Cout,Hout,Wout] added to the result, or NULL */,
                      const float *bias /* [Cout] or NULL */, float *y,
                      int stride /* 1 | 2; Hout = (H - 1) / 2 + 1 for 2 */,
                      int in_up /* 1, or 2: x holds the even samples of a zero-inserted [Hout, Wout] input --
                                   the data gradient of a stride-2 convolution (stride must be 1) */,
                      int Hout, int Wout /* output size (checked; required for in_up = 2, may be 0 otherwise) */,
                      int tile_r, int tile_p, void *stream);
/* 3x3 / stride 2 / pad 1 convolution with 1 .. 4 input channels on plain fp32 FMAs (the stem's conv1 on the image, reference
 * models/HRNet.py:404-405): x [N, Cin, H, W], w [Cout, Cin, 3, 3] (unpacked), bias [Cout] or NULL, y [N, Cout, (H - 1) / 2 + 1,
 * (W - 1) / 2 + 1].  The tile kernels would pad the contraction to 16 channels. */
int dcl_conv3x3_s2_smallcin(const float *x, int N, int Cin, int H, int W, const float *w, int Cout, const float *bias,
                            float *y, void *stream);
/* Weight gradient of that layer (the backward of nn.Conv2d(3, 64, 3, 2, 1) at reference models/HRNet.py:404-405; replaces
 * aten::convolution_backward, whose kernel for this shape adds its pixel splits with atomics): x [N, Cin, H, W] with Cin * 9 <= 32,
 * gy [N, Cout, (H - 1) / 2 + 1, (W - 1) / 2 + 1] with Cout <= 64, dw [Cout, Cin, 3, 3]; fp32 products on the matrix pipe
 * (v_mfma_f32_32x32x2f32), a FIXED summation order (bitwise reproducible).  part: workspace of
 * dcl_wgrad3x3_s2_smallcin_workspace(Cout) floats. */
int dcl_wgrad3x3_s2_smallcin_workspace(int Cout);
int dcl_wgrad3x3_s2_smallcin(const float *x, int N, int Cin, int H, int W, const float *gy, int Cout, float *part, float *dw,
                             void *stream);

/* tuning hook: in_up = 2 (data gradient of a stride-2 convolution) -- 2 (default, round 4): one workgroup stages a patch of the
 * stored gradient once and computes all four parity classes of its output tile (9 taps per staged patch, k_conv3x3_pm); 1: every
 * workgroup computes one parity class with the 1, 2 or 4 taps that class sees; 0: stride-1 tile over the zero-inserted input */
int dcl_conv3x3_set_up2_phases(int on);
/* tuning hook (automatic tile choice): the rows per wave are halved while a launch would have fewer workgroups than this
 * (default 96 since round 4; 192 = one round on 256 CUs three quarters full). */
int dcl_conv3x3_set_min_workgroups(int n);
/* tuning hook: 2 (default) = stride-1 3x3 tiles stage the next chunks interleaved with the MFMAs (k_conv3x3_il) and the
 * (2, 2) tile splits its waves 2 (rows) x 2 (channel tiles) (k_conv3x3_il_ws2); 1 = interleaved, no wave split; 0 = staging
 * in fenced blocks between the MFMAs (k_conv3x3) */
int dcl_conv3x3_set_interleave(int on);

/* Weight gradient of the same convolution, dw[Cout,Cin,3,3] = sum_n dy (*) x, on the f16x3 MFMA path with no LDS
 * staging (csrc/dcl_wgrad3x3.hip).  Cin % 16 == 0, Cout % 16 == 0, W % 8 == 0.  part: workspace of
 * dcl_wgrad3x3_splits(...) * 9 * Cout * Cin floats (one partial slab per workgroup, summed in fixed order). */
int dcl_wgrad3x3_splits(int N, int Cin, int Cout, int H, int W /* of x */, int stride);
/* tuning hook: force the (co tiles, ci tiles) per wave of the weight-gradient kernel (0, 0 = automatic choice) */
int dcl_wgrad3x3_set_tile(int nco, int nci);
/* tuning hook: -1 (default) = 2 = per-wave kernel with LDS-DMA operand staging (csrc/dcl_wgrad3x3d.hip), 0 = every wave
 * loads both operands itself in MFMA order (csrc/dcl_wgrad3x3.hip; also the fallback for tensors beyond 4 GiB per
 * image and for the zero-inserted stride-2 form).  Changes dcl_wgrad3x3_splits(). */
int dcl_wgrad3x3_set_variant(int variant);
/* tuning hook (stride 2): 1 (default) = GEMM over the output pixels (csrc/dcl_wgrad3x3_s2.hip, needs W % 16 == 0), the x rows
 * staged by LDS-DMA (k_wgrad3x3_s2d, round 5; one ci tile per wave); 2 = the same GEMM with every operand loaded in MFMA order
 * (k_wgrad3x3_s2, round 4: bitwise the same result); 0 = the stride-1 kernels on a zero-inserted dy.  Changes
 * dcl_wgrad3x3_splits(). */
int dcl_wgrad3x3_set_stride2(int native);
/* tuning hook (per-wave kernels): pixel splits per tile pair, 0 = automatic.  Changes dcl_wgrad3x3_splits(). */
int dcl_wgrad3x3_set_splits(int nx);
/* tuning hook (per-wave kernels): workgroups a launch aims at, pixel splits = target / tile pairs (default 256 = one per CU). */
int dcl_wgrad3x3_set_workgroup_target(int n);
/* tuning hook (stride 1, LDS-DMA kernel, (3, 1) tile): 2 (default) / 1 = with 129 .. 256 tile pairs (one workgroup per pair
 * would leave CUs empty: the head's 144 -> 720 launch has 135) the pixel splits go to single waves, 128 / ceil(pairs / 8) per
 * pair, each XCD owning a contiguous run of pairs (2: a workgroup's four waves take one split of four pairs, 1: four splits of
 * one pair); 0 = one workgroup per pair.  Changes dcl_wgrad3x3_splits(). */
int dcl_wgrad3x3_set_wave_mode(int on);
/* tuning hook (stride 1, LDS-DMA kernel, workgroup form): 1 (default) = the four waves of a workgroup walk four (two) ADJACENT
 * 32-pixel strips over the same rows, so that the halo lines of an x row are its neighbours' own lines in the same CU's L1 / L2;
 * 0 = four row ranges of one strip. */
int dcl_wgrad3x3_set_strip_group(int on);
/* tuning hook (wave form): rows per column of the traversal (a divisor of H; default 0 = whole strips): with bands the strips of
 * an image follow each other band by band, so that the halo lines of an x row are still in L2 when the neighbouring strip comes
 * (measured: no gain, a column prologue every `rows` rows costs more). */
int dcl_wgrad3x3_set_wave_band(int rows);
int dcl_wgrad3x3_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W /* of x */,
                       const float *xamax, int xcount, const float *gamax, int gcount,
                       int stride /* 1 | 2: dy is [N, Cout, (H - 1) / 2 + 1, W / 2] for 2 */, float *part, float *dw,
                       void *stream);
/* The same two directions of conv2d(relu(x * pre_sc[c] + pre_sh[c]), w, padding = 1) where x is the RAW tensor in front of a
 * training-mode norm (reference models/HRNet.py:77-93: conv2 of a BasicBlock reads relu(bn1(conv1(x)))), the map applied while
 * the operand is staged; xamax = the absmax slots dcl_bn_finalize_pre wrote.  Bitwise dcl_bn_apply_parts (relu) followed by
 * dcl_conv3x3_f16x3 / dcl_wgrad3x3_f16x3.  Forward: stride 1 | 2, Cin % 16 == 0; weight gradient: stride 1 | 2, slabs as
 * dcl_wgrad3x3_splits(.., stride).  *_supported: 1 when the automatic tile has the form (else the caller writes the tensor). */
int dcl_conv3x3_pre_supported(int N, int Cin, int Cout, int H, int W, int stride);
int dcl_conv3x3_pre_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout, const float *xamax,
                          int xcount, const float *wamax, const float *pre_sc, const float *pre_sh,
                          const float *bias /* [Cout] or NULL */, float *y, int stride,
                          int tile_r, int tile_p /* 0, 0 = automatic (what *_supported answers for) */, void *stream);
int dcl_wgrad3x3_pre_supported(int N, int Cin, int Cout, int H, int W, int stride);
int dcl_wgrad3x3_pre_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax,
                           int xcount, const float *gamax, int gcount, const float *pre_sc, const float *pre_sh,
                           int stride /* 1 | 2, as dcl_wgrad3x3_f16x3 */, float *part, float *dw, void *stream);

/* ---- 1x1 convolution on the same kernels ----------------------------------------------------------------------------
 * Replaces nn.Conv2d(C_in, C_out, 1) forward / data gradient / weight gradient (reference models/HRNet.py:63-100
 * Bottleneck, :236-262 fuse layers, models/Projector.py:46-51) at fp32-equivalent accuracy (f16x3).
 * Weights: dcl_conv3x3_pack(w, M, K, transposed | 2, ...) -- bit 1 = 1x1 kernel, w is [M][K] (data gradient: [K][M],
 * bit 0), M x ceil(K / 16) fragments of 2 KiB (a ninth of the 3x3 size). */
int dcl_conv1x1_f16x3(const float *x, int N, int Cin, int H, int W, const void *wp, int Cout, const float *xamax,
                      int xcount, const float *wamax, const float *addend /* optional, y's shape */,
                      const float *bias /* optional [Cout] */, float *y, int tile_r, int tile_p, void *stream);
/* dw[Cout,Cin] = sum_n dy_n x_n^T.  Cin % 16 == 0, Cout % 16 == 0, W % 8 == 0; part: workspace of
 * dcl_wgrad1x1_splits(...) * Cout * Cin floats (slabs summed in fixed order: deterministic). */
int dcl_wgrad1x1_splits(int N, int Cin, int Cout, int H, int W);
int dcl_wgrad1x1_f16x3(const float *x, const float *dy, int N, int Cin, int Cout, int H, int W, const float *xamax,
                       int xcount, const float *gamax, int gcount, float *part, float *dw, void *stream);

/* ---- per-step metrics -----------------------------------------------------
 * Confusion matrix of argmax(logits, dim=1) against the target, rows = predicted class, columns = target class:
 * replaces t_get_confusion_matrix (reference utils/torch_utils.py:157-183: transpose copy + argmax + two one-hot
 * matrices + float matmul, called after every training step, managers/HRNet_Manager.py:117-121) by one pass over
 * the logits (csrc/dcl_metrics.hip).  Integer arithmetic: bit-exact.
 *   logits  f32 [N, C, HW] (NCHW contiguous)       target  int64 | int32 | uint8 [N, HW] (target_bytes = 8 | 4 | 1)
 *   cols    C, or C + 1 when the dataset's experiment has an ignore id (the reference one-hots the target with
 *           C + 1 classes and drops the last column afterwards, :172-175)
 *   cm      int32 [C, cols], ACCUMULATED into (zero it for a fresh matrix; pass a running one for existing_matrix)
 *   oob     int32 [1], accumulated count of targets outside [0, cols) (the reference's one_hot raises on them)
 * argmax follows torch: first maximal index, NaN is maximal. */
int dcl_confusion_matrix(const float *logits, int N, int C, int HW, const void *target, int target_bytes,
                         int cols, int32_t *cm, int32_t *oob, void *stream);

/* The same matrix from an arg-max map (uint8 [total], e.g. the `pred` output of dcl_upsample_ce_fwd). */
int dcl_confusion_matrix_pred(const uint8_t *pred, int64_t total, const void *target, int target_bytes, int C, int cols,
                              int32_t *cm, int32_t *oob, void *stream);

/* Pixel accuracy, mean per-class accuracy and mean IoU of a confusion matrix in one launch: replaces
 * t_get_pixel_accuracy (utils/torch_utils.py:201-213) and t_get_miou over all classes (:253-283).
 *   cm int32 [C, ld] (ld >= C: row stride; ld = C + 1 for the uncropped matrix of dcl_confusion_matrix)
 *   out3 f32 {pa, pac, miou} */
int dcl_metrics_from_cm(const int32_t *cm, int C, int ld, float *out3, void *stream);

/* ---- Swin window attention, fp32, window 7 x 7, head_dim 32 -----------------------
 * Replaces WindowAttention.forward (reference models/Swin.py:198-230) and the pad / roll / window_partition /
 * window_reverse / roll / crop data movement of SwinTransformerBlock.forward around it (:286-318): tokens are read
 * from and written to their natural [B, H*W, .] order (csrc/dcl_winattn.hip).
 *   qkv       f32 [B, H*W, 3*C]   output of the qkv projection applied to the un-permuted tokens (C = 32 * heads)
 *   qkv_bias  f32 [3*C]           qkv of the tokens in the zero-padded border (= the projection's bias; zeros if none)
 *   bias      f32 [heads, 49, 49] relative-position bias gathered from the table (:216-219)
 *   shift     0 or window // 2; the shifted-window mask (:448-466) is computed from the coordinates
 *   out       f32 [B, H*W, C]     lse f32 [B, nW, heads, 49] (log-sum-exp of every score row, for the backward)
 * backward:
 *   dqkv       f32 [B, H*W, 3*C]
 *   dpad       f32 [B, npad, 3*C], npad = dcl_winattn_npad(H, W): gradient of the padded tokens' qkv (it belongs to
 *              the projection's bias: the caller adds its sum over the first two axes to the bias gradient)
 *   dbias_part f32 [nwaves, 49, 49], nwaves = dcl_winattn_bwd_waves(...): partial bias gradients, wave w -> head
 *              w % heads; the caller sums them in index order (deterministic) */
int dcl_winattn_npad(int H, int W);
int dcl_winattn_fwd(const float *qkv, const float *qkv_bias, const float *bias, int B, int H, int W, int C, int heads,
                    int shift, float scale, float *out, float *lse, void *stream);
int dcl_winattn_bwd_waves(int B, int H, int W, int heads);
/* kernel variant: bit 0 = forward on the f16 matrix cores (split-f16, fp32-equivalent; csrc/dcl_winattn_mfma.hip; default on),
 * bit 1 = backward; 0 = the fp32 vector-ALU kernels (A/B runs, tests) */
int dcl_winattn_set_mfma(int mask);
int dcl_winattn_bwd(const float *qkv, const float *qkv_bias, const float *bias, const float *lse, const float *dout,
                    int B, int H, int W, int C, int heads, int shift, float scale, float *dqkv, float *dpad,
                    float *dbias_part, float *dqkv_amax /* DCL_AMAX_SLOTS partial maxima of |dqkv| and |dpad| (caller zero-
                    initialises), or NULL: the operand scale of the qkv projection's backward GEMMs */, void *stream);

/* ---- LayerNorm over the channel axis of token-major rows (the Swin backbone) --------
 * Replaces nn.LayerNorm and its autograd in the Swin port (reference models/Swin.py:251-332 norm1 / norm2, :357-362
 * PatchMerging.norm, :452-455 PatchEmbed.norm, :560-565 the per-stage output norms); csrc/dcl_layernorm.hip.
 *   x, y, gy, gx f32 [M, C]      gamma, beta f32 [C]      mean, rstd f32 [M] (forward outputs, backward inputs)
 *   supported row lengths: C = 4 V G with G a power of two in 8 .. 64 and V in {1, 2, 3, 4, 6, 8}
 *   (dcl_layernorm_supported; Swin: 96 .. 1536)
 *   yamax        optional DCL_AMAX_SLOTS partial maxima of |y| (zero-initialised by the caller), NULL = not wanted
 * backward: parts f32 [dcl_layernorm_bwd_parts(M, C), 2, C] is workspace (one partial {dgamma, dbeta} row per
 * workgroup); dgamma_dbeta f32 [2, C] receives their fixed-order sums (deterministic). */
int dcl_layernorm_supported(int C);
int dcl_layernorm_bwd_parts(long long M, int C);
int dcl_layernorm_fwd(const float *x, const float *gamma, const float *beta, long long M, int C, float eps, float *y,
                      float *mean, float *rstd, float *yamax, void *stream);
int dcl_layernorm_bwd(const float *gy, const float *x, const float *gamma, const float *mean, const float *rstd,
                      long long M, int C, float *gx, float *parts, float *dgamma_dbeta,
                      const float *addend /* [M, C] added to gx, or NULL: the gradient of the residual connection around the
                                             norm (x feeds norm AND shortcut; reference models/Swin.py:286-321) */,
                      float *gxamax /* DCL_AMAX_SLOTS partial maxima of |gx| (caller zero-initialises), or NULL */,
                      void *stream);

/* ---- fused bilinear up-sampling + class-weighted cross-entropy --------------------
 * loss = CrossEntropyLoss(weight, ignore_index)(F.interpolate(z, (H, W), 'bilinear', align_corners), target) without
 * materialising the up-sampled logits (reference models/HRNet.py:638 + losses/LossWrapper.py:26-30, :82); PyTorch's
 * weighted-mean reduction: sum_p w[t_p] (lse_p - v_{p,t_p}) / sum_p w[t_p] over the non-ignored pixels.
 *   z f32 [N, C, h, w] (C <= 255)      target int64 [N, H, W]      weight f32 [C] or NULL
 *   lse f32 [N, H, W]   (out; input of the backward)       pred uint8 [N, H, W] or NULL (out: argmax class, torch's
 *   tie / NaN rule -- what the per-step confusion matrix needs)
 *   partial f32 [N * H, 2] workspace          out2 f32 {loss, sum of weights}
 * backward: dz = d loss / d z for the upstream gradient folded into gscale = grad_out / out2[1] (device scalar);
 * gather form, deterministic. */
int dcl_upsample_ce_fwd(const float *z, int N, int C, int h, int w, int H, int W, int align_corners,
                        const int64_t *target, const float *weight, int ignore_index, float *lse, uint8_t *pred,
                        float *partial, float *out2, void *stream);
int dcl_upsample_ce_bwd(const float *z, int N, int C, int h, int w, int H, int W, int align_corners,
                        const int64_t *target, const float *weight, int ignore_index, const float *lse,
                        const float *gscale, float *dz, void *stream);
/* tuning hook: class chunk of dcl_upsample_ce_bwd (0 = automatic: as many classes as fit half a CU's LDS) */
int dcl_upsample_ce_set_bwd_chunk(int cc);
/* tuning hook: KiB of LDS the forward stages a class chunk in (0 = default) */
int dcl_upsample_ce_set_fwd_lds(int kib);

/* Number of column splits the sweep kernels should use for an (N1 x N2) problem so that the
 * grid fills the 256 CUs evenly (host helper, no device work). */
int dcl_suggest_nsplit(int N1, int N2);

#ifdef __cplusplus
}
#endif
#endif /* DCL_HIP_H */
