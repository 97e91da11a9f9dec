/* ---- one-kernel batch-norm backward (csrc/dcl_bn_onepass.hip) -------------------------------------------------------
 * dcl_bn_bwd_reduce_part + dcl_bn_bwd_apply_fused of a single-rank norm in ONE launch that reads dy and x once: 256 persistent
 * workgroups keep their share of a channel in registers across the per-channel statistics exchange (teams of H W / 1024
 * workgroups of one XCD, relaxed agent-scope counters).  Shapes: H W % 1024 == 0, H W / 1024 in {1, 2, 4, 8, 16, 32}, N <= 12
 * (dcl_bn_bwd_onepass_supported); relu: 0 none, 1 the mask is recomputed from x (y = NULL), 2 y is the packed sign mask.
 * ws: dcl_bn_onepass_workspace_bytes() bytes, every byte 0xFF ONCE by the caller; seq = 0, 1, 2, ... the launch number on that
 * workspace.  The teams wait for all their members: launch it on ONE stream only (never two instances in flight). */
int dcl_bn_bwd_onepass_supported(int N, int C, int HW, int relu);
int64_t dcl_bn_onepass_workspace_bytes(void);
int dcl_bn_bwd_onepass(const float *dy, const float *x, const void *y_or_mask, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, double count, int N, int C, int HW, int relu, float *dx,
                       float *dres /* or NULL */, float *dbeta /* or NULL */, float *dgamma /* or NULL */,
                       float *amax /* DCL_AMAX_SLOTS or NULL */, void *ws, int64_t seq, void *stream);

