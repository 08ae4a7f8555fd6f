/*
 * maskplanner_hip.h -- C ABI of libmaskplanner_hip.so: the MI355X (gfx950 / CDNA4) kernels of the
 * MaskPlanner hot path (PointNet++ set abstraction + set losses).
 *
 * The reference (gabrieletiboni/MaskPlanner) has no FFI layer: its hot path is PyTorch tensor algebra
 * plus pytorch3d.ops.knn.knn_points and scipy.optimize.linear_sum_assignment.  Each entry point below
 * replaces one of those Python-level operators; the `replaces:` line cites the reference file:line.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points)
 *   - plain C: raw DEVICE pointers + int64 sizes; no torch / C++ types in any signature;
 *   - layouts are contiguous row-major; coordinates points-major [B,N,C]; indices int64 (reference dtype);
 *   - the caller owns every buffer (inputs, outputs, workspace).  The COMPUTE entry points never allocate, free or retain
 *     device memory and keep no mutable global state: re-entrant, thread-safe.  Three bookkeeping facilities do hold
 *     process-wide state, each behind a mutex and each off unless the caller turns it on: the zero arena (a table of
 *     caller-owned ranges keyed by stream: mp_zero_arena_*; no device memory), the kernel profiler (host-side event records:
 *     mp_profiler_enable / _collect) and the profiler's marks (mp_profiler_mark: the ONE allocation the library ever makes --
 *     a 33 KB device ring for in-graph time stamps, allocated at the first mark request and kept for the process; a process
 *     that never calls mp_profiler_mark never triggers it).  The per-kernel dynamic-LDS opt-in (hipFuncSetAttribute, once per
 *     kernel and device) is cached in atomics;
 *   - enqueue-only: work is queued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no hipDeviceSynchronize, no hidden device->host copies => safe under hipGraph capture;
 *   - return value: MP_OK (0) or a negative MP_E* code.  No exceptions, no abort().
 *   - fp32 arithmetic follows the exact rounding sequence of the reference's CPU torch path (see each
 *     entry); index outputs are bit-exact, fp32 outputs agree within 1e-5.
 */
#ifndef MASKPLANNER_HIP_H
#define MASKPLANNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_EINVAL (-1)        /* null pointer / negative or inconsistent dimension            */
#define MP_EUNSUPPORTED (-2)  /* size outside what the kernels are built for (see each entry) */
#define MP_EWORKSPACE (-3)    /* workspace too small                                          */
#define MP_ELAUNCH (-4)       /* hipGetLastError() != hipSuccess after the launch             */

#define MP_ABI_VERSION 1

typedef void* mp_stream_t; /* hipStream_t */

int mp_abi_version(void);
const char* mp_error_string(int code);

/* ---- farthest point sampling -------------------------------------------------------------------
 * replaces: models/pointnet2_utils.py:65-86 farthest_point_sample(xyz, npoint)
 *   xyz [B,N,3] f32; start_idx [B] i64 = the reference's torch.randint draw (:77), passed in so the
 *   caller keeps seed compatibility; out_idx [B,S] i64; out_xyz [B,S,3] f32 or NULL (fused
 *   index_points(xyz, fps_idx), :131).  dist = (dx*dx + dy*dy) + dz*dz without FMA; running min;
 *   argmax with lowest index on ties.  N <= 13312 (cloud resident in LDS), else MP_EUNSUPPORTED. */
int mp_fps_f32(const float* xyz, int64_t B, int64_t N, int64_t S, const int64_t* start_idx,
               int64_t* out_idx, float* out_xyz, mp_stream_t stream);
/* Measurement aid: the latency floor of mp_fps_f32 for the same arguments -- the kernel it would launch with the per-point
 * distance arithmetic removed (same launch shape, same dependent reduce + barrier chain per step; SURVEY 8d "S * t_iter").
 * Outputs are written but meaningless.  Supported for N <= 10240. */
int mp_fps_floor_f32(const float* xyz, int64_t B, int64_t N, int64_t S, const int64_t* start_idx,
                     int64_t* out_idx, float* out_xyz, mp_stream_t stream);

/* ---- ball query ------------------------------------------------------------------------------
 * replaces: models/pointnet2_utils.py:89-109 query_ball_point(radius, nsample, xyz, new_xyz)
 *   xyz [B,N,3], new_xyz [B,S,3] f32 -> out_idx [B,S,K] i64: the first K indices (ascending) with
 *   !(d > (float)(radius*radius)), d in the reference's expanded form ((-2*dot)+|q|^2)+|p|^2 (:39-41),
 *   dot = fma(qz,pz, fma(qy,py, qx*px)); empty slots take the first hit (a query with no hit yields N
 *   in every slot, as the reference's intermediate does).  K <= 1024.  N <= 13312. */
int mp_ball_query_f32(const float* xyz, const float* new_xyz, int64_t B, int64_t N, int64_t S,
                      double radius, int64_t K, int64_t* out_idx, mp_stream_t stream);
/* replaces: the loop over `radius_list` of PointNetSetAbstractionMsg (models/pointnet2_utils.py:255-258): n_radii (<= 3) ball queries of ONE
 *   (cloud, query set) pair in one scan of the cloud -- one distance per pair, one membership test and one hit list per radius.
 *   radii [n_radii], K [n_radii] on the HOST; out_idx: host array of n_radii device pointers, out_idx[r] is [B,S,K[r]] i64.  Each list equals
 *   mp_ball_query_f32(radii[r], K[r]) bit for bit. */
int mp_ball_query_multi_f32(const float* xyz, const float* new_xyz, int64_t B, int64_t N, int64_t S, int64_t n_radii,
                            const double* radii, const int64_t* K, int64_t* const* out_idx, mp_stream_t stream);

/* ---- square_distance -------------------------------------------------------------------------
 * replaces: models/pointnet2_utils.py:21-42 square_distance(src, dst) -> [B,S,N] f32 (expanded form) */
int mp_square_distance_f32(const float* src, const float* dst, int64_t B, int64_t S, int64_t N,
                           float* out, mp_stream_t stream);

/* ---- index_points (row gather) and its backward ------------------------------------------------
 * replaces: models/pointnet2_utils.py:45-62 index_points(points, idx)
 *   points [B,N,C] f32, idx [B,M] i64 (M = product of idx's trailing dims) -> out [B,M,C].
 *   backward: grad_points [B,N,C] = scatter-add of grad_out rows (overwritten, not accumulated).
 *   deterministic != 0 selects the fixed-order (bitwise reproducible) variant instead of atomics. */
int mp_index_points_f32(const float* points, const int64_t* idx, int64_t B, int64_t N, int64_t C,
                        int64_t M, float* out, mp_stream_t stream);
int mp_index_points_bwd_f32(const float* grad_out, const int64_t* idx, int64_t B, int64_t N, int64_t C,
                            int64_t M, float* grad_points, int deterministic, mp_stream_t stream);

/* ---- feature propagation: 3 nearest neighbours + inverse-distance interpolation -------------------
 * replaces: models/pointnet2_utils.py:310-317 (PointNetFeaturePropagation.forward: square_distance, sort, [:3],
 *           1/(d+1e-8) weights, index_points + weighted sum)
 *   three_nn: xyz1 [B,N,3], xyz2 [B,S,3] (S >= 3) -> dist [B,N,3] (ascending, may be NULL), idx [B,N,3] i64,
 *             weight [B,N,3] (may be NULL).  Expanded-form distances, bit-exact; lowest index first on ties
 *             (the reference's sort is not stable, so its order on exact ties is unspecified).
 *   three_interpolate: points2 [B,S,D] -> out [B,N,D] = (p[i0]*w0 + p[i1]*w1) + p[i2]*w2.
 *   three_interpolate_bwd: grad_points2 [B,S,D] = scatter of grad_out [B,N,D] * weight (overwritten);
 *             deterministic != 0 sums in ascending (n,k) order instead of using atomics. */
int mp_three_nn_f32(const float* xyz1, const float* xyz2, int64_t B, int64_t N, int64_t S, float* dist,
                    int64_t* idx, float* weight, mp_stream_t stream);
int mp_three_interpolate_f32(const float* points2, const int64_t* idx, const float* weight, int64_t B, int64_t N,
                             int64_t S, int64_t D, float* out, mp_stream_t stream);
int mp_three_interpolate_bwd_f32(const float* grad_out, const int64_t* idx, const float* weight, int64_t B,
                                 int64_t N, int64_t S, int64_t D, float* grad_points2, int deterministic,
                                 mp_stream_t stream);

/* ---- column permutation with zero fill (host-side helper of the fused MLP: first-layer weight in the internal column order)
 *   dst [R,Cd] : dst[r,c] = src[r, perm[c]] for perm[c] >= 0, else 0;  src [R,Cs], perm i32 [Cd]. */
int mp_permute_cols_f32(const float* src, const int32_t* perm, int64_t R, int64_t Cs, int64_t Cd, float* dst,
                        mp_stream_t stream);
/* up to 8 of those in one launch (HOST arrays of device pointers and shapes; they travel in the kernel arguments) */
int mp_permute_cols_multi_f32(int64_t count, const float* const* src, const int32_t* const* perm, const int64_t* R,
                              const int64_t* Cs, const int64_t* Cd, float* const* dst, mp_stream_t stream);

/* ---- grouping (gather + centre + concat) -----------------------------------------------------
 * replaces: models/pointnet2_utils.py:133-143 (sample_and_group tail) and :258-262 (MSG variant)
 *   out [B,S,K,3+D] = cat(xyz[idx] - new_xyz, feats[idx])   (xyz_last == 0, SSG order, :138)
 *                   = cat(feats[idx], xyz[idx] - new_xyz)   (xyz_last != 0, MSG order, :262)
 *   feats [B,N,D] may be NULL with D == 0.  out_stride (floats, 0 = 3+D): distance between output rows; columns
 *   beyond 3+D are zero-filled (the fused MLP wants rows of a multiple of 4 floats).
 *   backward: grad_feats [B,N,D] = scatter-add of the feature channels of grad_out (rows grad_stride apart;
 *   overwritten). */
int mp_group_f32(const float* xyz, const float* feats, const float* new_xyz, const int64_t* idx,
                 int64_t B, int64_t N, int64_t S, int64_t K, int64_t D, int xyz_last, int64_t out_stride,
                 float* out, mp_stream_t stream);
int mp_group_bwd_f32(const float* grad_out, const int64_t* idx, int64_t B, int64_t N, int64_t S,
                     int64_t K, int64_t D, int xyz_last, int64_t grad_stride, float* grad_feats,
                     int deterministic, mp_stream_t stream);

/* ---- K nearest neighbours (brute force) and backward ------------------------------------------------
 * replaces: pytorch3d.ops.knn.knn_points (third party; call sites pytorch3d_chamfer.py:182-183,
 *           205-206, 257-258) and its autograd backward.
 *   p1 [B,P1,D], p2 [B,P2,D] f32; len1/len2 [B] i64 or NULL (= full);  K in {1,2,...,8}.
 *   dists [B,P1,K] f32 (squared L2, ascending), idx [B,P1,K] i64.  dist = fma chain over d of
 *   (p1[d]-p2[d])^2; strict < keeps the first index on ties; rows >= len1 and slots >= len2 hold 0.
 *   backward: grad_p1 [B,P1,D] / grad_p2 [B,P2,D] (either may be NULL = not needed; overwritten):
 *     g = 2*grad_dists[i,k]*(p1[i]-p2[idx]) ; grad_p1[i] += g ; grad_p2[idx] -= g.
 *   workspace: mp_knn_workspace_bytes(B,P1,K) bytes (device). */
size_t mp_knn_workspace_bytes(int64_t B, int64_t P1, int64_t K);
int mp_knn_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B,
               int64_t P1, int64_t P2, int64_t D, int64_t K, float* dists, int64_t* idx,
               void* workspace, size_t workspace_bytes, mp_stream_t stream);
int mp_knn_bwd_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2,
                   const int64_t* idx, const float* grad_dists, int64_t B, int64_t P1, int64_t P2,
                   int64_t D, int64_t K, float* grad_p1, float* grad_p2, int deterministic,
                   mp_stream_t stream);
/* The two K = 1 searches behind mp_knn_f32, by name: screened != 0 -- bf16 matrix cores screen the pairs (three-plane split dot products,
 * |y|^2 - 2 x.y), the reference's arithmetic is evaluated only on the blocks inside the error window of the minimum; 0 -- the direct scan.
 * D in {3, 6, 12, 24}.  Identical outputs (distance bits and indices, first index on ties). */
size_t mp_knn1_workspace_bytes(int64_t B, int64_t P2, int64_t D);   /* 16-byte aligned device bytes for the screened search; 0 = not available */
int mp_knn1_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B, int64_t P1, int64_t P2,
                int64_t D, float* dists, int64_t* idx, int screened, void* workspace, size_t workspace_bytes, mp_stream_t stream);
/* [r4] the screened search in two calls, for references that many searches share (a batch's ground-truth segments: prepared once per
 * batch, off the step's stream): mp_knn1_prepare_f32 fills the workspace (planes + norms of p2 under len2), mp_knn1_prepared_f32 searches
 * against it -- p2 / len2 unchanged in between.  Outputs identical to mp_knn_f32. */
int mp_knn1_prepare_f32(const float* p2, const int64_t* len2, int64_t B, int64_t P2, int64_t D, void* workspace, size_t workspace_bytes,
                        mp_stream_t stream);
int mp_knn1_prepared_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B, int64_t P1, int64_t P2,
                         int64_t D, float* dists, int64_t* idx, const void* workspace, size_t workspace_bytes, mp_stream_t stream);
/* The backward of K = 1 distances that went straight into mp_chamfer_reduce_f32 (one loss term = nearest neighbours + reduction,
 * pytorch3d_chamfer.py:257-334): grad_out is the gradient of the REDUCED value ([1], or [B] when batch_mode == 0) and the per-row
 * factor scale / div / len1[b] is applied inside the scatter -- no [B,P1] gradient tensor, no mp_chamfer_reduce_bwd_f32 launch.
 * `deterministic` is a flag word: bit 0 = ordered (bit-reproducible) grad_p2; bit 1 [r4] = grad_p1 is ADDED to (the terms of a composite
 * loss on the same prediction accumulate into one buffer: loss_handler.py:660-664 without autograd's fan-out adds). */
int mp_knn_bwd_reduced_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2,
                           const int64_t* idx, const float* grad_out, int point_mean, int batch_mode, double div,
                           double scale, int64_t B, int64_t P1, int64_t P2, int64_t D, float* grad_p1,
                           float* grad_p2, int deterministic, mp_stream_t stream);

/* ---- padded-length detection -------------------------------------------------------------------
 * replaces: pytorch3d_chamfer.py:138-149 (`padded=True`): lengths[b] = first column c with
 *   y[b,c,0] == -100, else P2.  Removes the reference's B host syncs. */
int mp_padded_lengths_f32(const float* y, int64_t B, int64_t P2, int64_t D, int64_t* lengths,
                          mp_stream_t stream);

/* ---- chamfer reductions -----------------------------------------------------------------------------------------
 * replaces: pytorch3d_chamfer.py:295-326 (cham.sum(1) [/ lengths], then .sum() [/ N]) and its autograd backward
 *   cham [N,P] f32 nearest-neighbour distances, rows >= lengths[n] already zero (mp_knn_f32 output).
 *   point_mean != 0: per-cloud sums are divided by lengths[n] (i64 [N], required then).
 *   batch_mode 0: out [N] per-cloud values; 1: out [1] = their sum; 2: out [1] = sum / div (per_cloud [N]: scratch for modes
 *   1 and 2).  Every output is multiplied by
 *   `scale` (the loss's constant factors, e.g. 100 * term weight, folded in instead of separate scalar launches).
 *   Deterministic (fixed summation order).  The backward writes grad_cham [N,P] (zero at p >= lengths[n] when lengths
 *   is given) from grad_out ([N] for batch_mode 0, [1] otherwise).
 *   add_to (modes 1, 2; NULL otherwise): a device scalar added to the reduced value -- the composite losses
 *   (loss_handler.py:660-664) chain their weighted terms through it instead of launching an elementwise add per term. */
int mp_chamfer_reduce_f32(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean, int batch_mode,
                          double div, double scale, float* per_cloud, float* out, const float* add_to, mp_stream_t stream);
/* the same reduction in one launch: `counter` = device uint32, zero on entry and zero again on exit (the workgroup that finishes last
 * combines the clouds in cloud order -- identical result bits); one counter per stream that runs reductions */
int mp_chamfer_reduce1_f32(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                           int batch_mode, double div, double scale, float* per_cloud, float* out, const float* add_to,
                           uint32_t* counter, mp_stream_t stream);
int mp_chamfer_reduce_bwd_f32(const float* grad_out, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                              int batch_mode, double div, double scale, float* grad_cham, mp_stream_t stream);

/* ---- stroke-mask matching: BCE cost + rectangular LAP, on device ------------------------------------
 * replaces: loss_handler.py:838-875 (target ids -> binary masks -> BCE cost -> scipy
 *           linear_sum_assignment, in a Python loop over the batch with one host sync per sample)
 *   pred_masks [B,M,S] f32 logits; target_ids [B,S] f32 = stroke_ids.gather(1, idx_x) (:838).
 *   Per sample: unique ids ascending, -1 skipped (:938-967) -> Kb masks; cost[m,k] = sum_s
 *   BCEWithLogits(pred[m,s], mask[k,s]); LAP solved with scipy's algorithm and tie-breaking.
 *   target_value [B,S] f32 or NULL: when given (`smooth_target_stroke_masks`, :830,:841-844,:959-964) the masks hold
 *   target_value[b,s] instead of 1 and the cost is the MSE sum_s (pred[m,s] - mask[k,s])^2 (:810-811).
 *   Outputs: match_col [B,M] i64 = matched mask index k or -1; uniq_ids [B,M_cap] f32 (ascending,
 *   first n_targets[b] valid); n_targets [B] i64; cost [B,M,M_cap] f32 or NULL.  M_cap = 64.
 *   status [B] i32: 0, or a bit set of the conditions the reference asserts on / raises for (match_col = -1 then):
 *   MP_MATCH_TOO_MANY_IDS (M > 64 or Kb > 64), MP_MATCH_PADDING_ID (a target id is the padding id -1: the assertion of
 *   loss_handler.py:852), MP_MATCH_INFEASIBLE (non-finite costs: scipy's ValueError at :875; any NaN or infinite logit of
 *   the sample gives such a cost here, because the masked sums are a product with the one-hot ranks).  mp_mask_loss_f32 takes this
 *   array and turns the loss into NaN when any entry is non-zero, so a bad batch cannot pass silently. */
#define MP_MASK_CAP 64
#define MP_MATCH_TOO_MANY_IDS 1
#define MP_MATCH_PADDING_ID 2
#define MP_MATCH_INFEASIBLE 4
int mp_mask_match_f32(const float* pred_masks, const float* target_ids, const float* target_value, int64_t B,
                      int64_t M, int64_t S, int64_t* match_col, float* uniq_ids, int64_t* n_targets, float* cost,
                      int32_t* status, mp_stream_t stream);

/* ---- rectangular linear sum assignment, batched (segment matching) -------------------------------------------
 * replaces: scipy.optimize.linear_sum_assignment as called by models/hungarianMatcher.py:58-61 (and loss_handler.py:
 *           990-1009 `emd`): one host round trip + ~0.2 s of one core per 999 x ~900 sample, serial over the batch.
 *   cost [B, Rmax, ld] f32 (sample b at cost + b*batch_stride, row stride ld >= Cmax); n_rows[b] <= n_cols[b] <= Cmax
 *   (the caller passes the transposed matrix when a sample has more rows than columns, as scipy does internally).
 *   Same algorithm and tie-breaking as scipy's rectangular_lsap, fp64 on the fp32 costs; one wave per sample.
 *   Outputs: col4row [B, Rmax] i64 (column assigned to each row < n_rows[b], -1 elsewhere); status [B] i32:
 *   0, MP_EINVAL (bad sizes) or MP_EUNSUPPORTED (infeasible: non-finite costs).  Cmax, Rmax <= 2048. */
int mp_lsap_f32(const float* cost, int64_t B, int64_t Rmax, int64_t Cmax, int64_t ld, int64_t batch_stride,
                const int32_t* n_rows, const int32_t* n_cols, int64_t* col4row, int32_t* status, mp_stream_t stream);
/* The matcher's cost matrices, all samples in one launch: replaces the torch.cdist of models/hungarianMatcher.py:51 (one
 * [B*S, sum Sgt] matrix of which only the diagonal blocks are used).  outputs [B, S, D]; targets [T, D]: the samples' targets
 * back to back, sample b owning rows offsets[b] .. offsets[b+1]; cost [B, Rmax, Cmax] f32 = ||x - y||_2 (direct differences),
 * written in the layout mp_lsap_f32 takes: sample b's block is S x T_b, or its transpose when S > T_b, zero padded; n_rows /
 * n_cols [B] i32 receive the block's shape.  Rmax >= max_b min(S, T_b), Cmax >= max_b max(S, T_b). */
int mp_cdist_batch_f32(const float* outputs, const float* targets, const int64_t* offsets, int64_t B, int64_t S, int64_t D,
                       int64_t Rmax, int64_t Cmax, float* cost, int32_t* n_rows, int32_t* n_cols, mp_stream_t stream);

/* ---- fused tails: pose output and stroke-mask loss ---------------------------------------------------------------
 * mp_pose_output: replaces models/pointnet2_cls_ssg.py:332-339 (tanh -> view(B,-1,3) -> F.normalize * weight_orient, cat with
 *   the positions): pos, raw [n_pose*3] -> out [n_pose, 6]; the backward writes grad_pos / grad_raw (either may be NULL).
 * mp_mask_loss: replaces loss_handler.py:877-934 for binary targets, after mp_mask_match_f32: matched BCE-with-logits
 *   .sum(-1).mean() + weighted confidence BCE .mean(): out [1] = w_masks*mask_loss + w_conf*conf_loss.  per_mask [B*M] and
 *   n_matched [1] are scratch / saved for the backward, which writes grad_masks [B,M,S] and grad_scores [B,M] (or NULL).
 *   status: mp_mask_match_f32's per-sample status (or NULL): any non-zero entry makes out NaN.
 *   All sums run in a fixed order. */
int mp_pose_output_f32(const float* pos, const float* raw, int64_t n_pose, double weight_orient, float* out, mp_stream_t stream);
int mp_pose_output_bwd_f32(const float* grad_out, const float* raw, int64_t n_pose, double weight_orient, float* grad_pos,
                           float* grad_raw, mp_stream_t stream);
int mp_mask_loss_f32(const float* pred_masks, const float* scores, const float* target_ids, const int64_t* match_col,
                     const float* uniq_ids, int64_t B, int64_t M, int64_t S, double w_masks, double w_conf,
                     double no_stroke_weight, float* per_mask, float* out, float* n_matched, const int32_t* status /* [B] or NULL */,
                     const float* add_to /* device scalar added to out, or NULL */, mp_stream_t stream);
int mp_mask_loss_bwd_f32(const float* grad_out, const float* pred_masks, const float* scores, const float* target_ids,
                         const int64_t* match_col, const float* uniq_ids, const float* n_matched, int64_t B, int64_t M, int64_t S,
                         double w_masks, double w_conf, double no_stroke_weight, float* grad_masks, float* grad_scores,
                         mp_stream_t stream);

/* ---- BatchNorm1d + ReLU over a skinny batch (regression heads) -------------------------------------------------------
 * replaces: F.relu(bn(x)) of models/pointnet2_cls_ssg.py:309-327 for x [B, C] with a small B (nn.BatchNorm1d semantics:
 *   biased batch variance normalises, running_var gets the unbiased one, momentum update; eval uses the running stats).
 *   Forward: y [B,C], save_mean / save_rstd [C].  Backward: grad_x [B,C], grad_gamma / grad_beta [C] (any may be NULL). */
int mp_bn_relu_rows_f32(const float* x, int64_t B, int64_t C, int training, double momentum, double eps, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float* y, float* save_mean,
                        float* save_rstd, mp_stream_t stream);
int mp_bn_relu_rows_bwd_f32(const float* grad_y, const float* y, const float* x, int64_t B, int64_t C, int training,
                            const float* gamma, const float* save_mean, const float* save_rstd, float* grad_x,
                            float* grad_gamma, float* grad_beta, mp_stream_t stream);
/* The same block with the nn.Dropout(p) behind it (models/pointnet2_cls_ssg.py:309-327: self.dropout(F.relu(self.bn1(.)))) in the same
 * launch.  rng: device int64 [2] = (seed, step), advanced by the caller once per training step; the keep mask is a counter-based hash of
 * (seed, step, layer, element) -- Bernoulli(1 - p) like torch's, not the same draws.  Backward: the mask is `y > 0`. */
int mp_bn_relu_drop_rows_f32(const float* x, int64_t B, int64_t C, int training, double momentum, double eps,
                             const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                             float* save_mean, float* save_rstd, double drop_p, const int64_t* rng, int layer, mp_stream_t stream);
int mp_bn_relu_drop_rows_bwd_f32(const float* grad_y, const float* y, const float* x, int64_t B, int64_t C, int training,
                                 const float* gamma, const float* save_mean, const float* save_rstd, float* grad_x,
                                 float* grad_gamma, float* grad_beta, double drop_p, mp_stream_t stream);


/* ---- set-abstraction shared MLP: (1x1 conv -> BatchNorm -> ReLU) x L -> max over the K group members ----------
 * replaces: models/pointnet2_utils.py:208-214 (PointNetSetAbstraction.forward tail; :264-269 for MSG) and the
 *           autograd graph torch builds for it.
 *   x0 [P, c_0] f32: grouped input, positions-major (P = B*S*K rows, the K members of a group consecutive).
 *   Layer l: weight [c_out, c_in] (Conv2d weight [c_out,c_in,1,1] as is), bias [c_out] or NULL, BatchNorm
 *   gamma/beta, running_mean/var (updated in place when training; read when not).  The library writes the raw
 *   pre-BN activations z [P, c_out] and the folded affine (scale, shift) + (mean, rstd) per layer: these are the
 *   tensors backward needs, owned by the caller.
 *   out [P/K, c_L] = max_k relu(bn(z_L)); argk i32 [P/K, c_L] = arg-max member (first wins); zmax = raw z there.
 *   GEMMs run on v_mfma_f32_32x32x2_f32 (exact fp32); BatchNorm sums are reduced in fp64.
 *   backward: grad_out [P/K, c_L] -> d_weight [c_out,c_in], d_bias (zeros in training: the bias cancels inside
 *   BN), d_gamma, d_beta per layer, and grad_x0 [P, c_0] (NULL = not needed).  d_weight is accumulated with fp32
 *   atomics over position slices (not bitwise reproducible).
 *   Every channel count must be a multiple of 4 (zero-pad the input columns / first weight otherwise:
 *   maskplanner_amd/sa_mlp.py does) and P * max(c) < 2^31, else MP_EUNSUPPORTED.
 *   workspace: mp_sa_mlp_workspace_bytes(P, K, L, channels[L+1], backward). */
typedef struct {
    const float* weight;
    const float* bias;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t c_in;
    int64_t c_out;
    float* z;     /* [P, c_out] */
    float* mean;  /* [c_out] */
    float* rstd;
    float* scale;
    float* shift;
    double* bn_state; /* optional (NULL: as before).  MP_BN_STATE_DOUBLES(c_out) doubles of PERSISTENT, zero-initialised device memory owned by
                       * the caller and used by one call at a time (keep it with the BatchNorm module).  With every layer's bn_state given (and
                       * n_layers >= 2), a train-mode call without the SyncBN hook runs (almost) no BatchNorm finalize launches: the producing kernel
                       * adds its fp64 sums into slot rows of this buffer and the first kernel that consumes the constants derives them in its
                       * prologue (it also updates the running statistics / writes d_gamma, d_beta).  Rows are zeroed again by the kernels that
                       * follow their consumer; between calls only rows that were already consumed can be non-zero, and every call starts by
                       * zeroing those.  After a FAILED call zero the buffers yourself.  Same statistics up to the order of fp64 additions. */
} mp_mlp_layer_t;
#define MP_BN_SLOTS 8
#define MP_BN_STATE_DOUBLES(c_out) ((size_t)4 * MP_BN_SLOTS * (size_t)(c_out))

typedef struct {
    float* d_weight;
    float* d_bias; /* may be NULL */
    float* d_gamma;
    float* d_beta;
} mp_mlp_grads_t;

size_t mp_sa_mlp_workspace_bytes(int64_t P, int64_t K, int n_layers, const int64_t* channels, int backward);
/* 1 if the first layer of the chain can be RECOMPUTED instead of stored: pass layers[0].z = NULL to mp_sa_mlp_fwd_f32 and
 * mp_sa_mlp_bwd_f32 (no grad_x0 then) and Z_0 [P, 64] is never written -- a 4-channel input (xyz + pad) makes it four FMAs per
 * element.  channels[n_layers + 1] as for the workspace query.  (Reference: the first Conv2d of sa1, pointnet2_utils.py:208-213;
 * the reference stores every activation for autograd.) */
int mp_sa_mlp_recompute_first(int n_layers, const int64_t* channels, int64_t K);
/* bf16 variant only (mp_sa_mlp_{fwd,bwd}_bf16, _gather_bf16, the _ex forms with bf16 = 1): 1 if this chain keeps its raw activations in
 * memory as bf16 -- every layers[l].z the caller passes is then a bf16 [P, c_out] buffer (2 bytes per element) although the struct
 * field is typed float*, and the gradient buffers inside the workspace are bf16 too.  first_layer: 1 = recomputed (layers[0].z NULL),
 * 2 = factorised (gather forms), 0 = stored grouped input.  Chains that qualify run entirely on the position-stream kernels (first layer
 * recomputed or factorised, later layers 64 / 128 -> 64 / 128 / 256, K in {32, 64, 128}); BatchNorm statistics are taken before the
 * rounding, everything downstream sees the rounded value.  Other bf16 chains keep fp32 storage (and cannot recompute their first layer).
 * (Reference: the activations autograd stores for models/pointnet2_utils.py:208-214; BASELINE configs[4] "bf16 MFMA grouped-MLP".) */
int mp_sa_mlp_bf16_storage(int n_layers, const int64_t* channels, int64_t K, int first_layer);
/* FACTORISED first layer of a level with input features: the gather forms (mp_sa_mlp_{fwd,bwd}_gather_*) take this descriptor instead
 * of x0 -- the grouped input [feats[b, idx[p]] | xyz[b, idx[p]] - new_xyz[b, s]] (models/pointnet2_utils.py:133-143) is never materialised.
 * Shapes: layers[0].c_in == 4 and CF == layers[0].c_out in {64, 128, 256}.  The first Conv2d is linear in [f ; x - c] (pointnet2_utils.py:138 / :262 + :208-213), so the caller computes A = F W_f^T once per
 * SOURCE point and passes it as `feats` [B, N, Co]; layers[0].weight is the coordinate part (W_x | 0) [Co, 4].  The library forms
 * Z_0[p] = A[b, idx[p]] + W_x (xyz[b, idx[p]] - new_xyz[b, s]) and its BatchNorm statistics, then the ordinary chain; backward it writes
 * dZ_0 into grad_x0 [P, Co + 4] (grad_x0_cols = Co; mp_group_bwd_f32 over the gathering rows turns it into dA [B, N, Co], from which dW_f
 * and dF follow by two small GEMMs on the caller's side) and dW_x into grads[0].d_weight [Co, 4].  With grad_x0_cols = 0 instead, grad_x0
 * IS dA [B, N, Co]: the library sorts each cloud's rows by source point and reduces dZ_0 on the fly (no [P, Co + 4] buffer; fp32 atomics,
 * summation order not fixed; N <= 15000).  layers[0].z must be given. */
typedef struct {
    const float* feats;    /* [B, N, CF] */
    const float* xyz;      /* [B, N, 3] */
    const float* new_xyz;  /* [B, S, 3] */
    const int64_t* idx;    /* [B, S, K] */
    int64_t N, S, CF;
    const int32_t* rows;   /* [r4] NULL, or the sorted row lists of mp_csr_rows_i64(idx): the backward then sorts nothing itself */
} mp_gather_t;
/* rows [2][B][S*K] int32: per cloud its rows sorted by the source point they gathered, and that point -- a function of idx alone (a
 * training harness prepares it with the sampling plan, off the step's stream); N <= 15000, S*K < 2^24 */
int mp_csr_rows_i64(const int64_t* idx, int64_t B, int64_t N, int64_t M, int32_t* rows, mp_stream_t stream);
int mp_sa_mlp_fwd_f32(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                      double momentum, double eps, float* out, int32_t* argk, float* zmax, void* workspace,
                      size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_bwd_f32(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                      const float* grad_out, const float* out, const int32_t* argk, const float* zmax,
                      const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols, void* workspace,
                      size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_fwd_gather_f32(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                             int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                             void* workspace, size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_bwd_gather_f32(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                             int training, const float* grad_out, const float* out, const int32_t* argk,
                             const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                             void* workspace, size_t workspace_bytes, mp_stream_t stream);
/* The same two calls with every contraction on the bf16 matrix cores (BASELINE configs[4]: containers, N = 10240, MSG
 * encoder, "bf16 MFMA grouped-MLP"): both operands of each GEMM -- act(Z_{l-1}) and W_l forward; dZ_l, W_l and
 * act(Z_{l-1}) backward -- are rounded to bf16 (round-to-nearest-even) as they are staged, v_mfma_f32_32x32x16_bf16
 * accumulates in fp32, and everything else stays fp32: stored raw activations, BatchNorm statistics and affine folding,
 * ReLU masks, pooling, dW accumulation, all outputs.  Same arguments, workspace and layouts as the _f32 calls.
 * Reference: models/pointnet2_utils.py:208-214, 265-271 (the reference has no reduced-precision mode; this is the build's option
 * for config 5).
 * [r3] the bf16 calls take the same fast paths as the fp32 ones: the position-stream kernels with ONE bf16 operand plane, the recomputed
 * first layer (layers[0].z = NULL: pass x0 and layers[0].weight already rounded to bf16 values -- the recomputation is then an exact
 * product of rounded operands) and, through the _gather_bf16 pair, the factorised first layer (`feats` = A computed from rounded
 * operands; the library rounds W_x, the centred coordinates and dZ_0 itself). */
int mp_sa_mlp_fwd_gather_bf16(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                              int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                              void* workspace, size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_bwd_gather_bf16(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                              int training, const float* grad_out, const float* out, const int32_t* argk,
                              const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                              void* workspace, size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_fwd_bf16(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                       double momentum, double eps, float* out, int32_t* argk, float* zmax, void* workspace,
                       size_t workspace_bytes, mp_stream_t stream);
int mp_sa_mlp_bwd_bf16(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                       const float* grad_out, const float* out, const int32_t* argk, const float* zmax,
                       const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols, void* workspace,
                       size_t workspace_bytes, mp_stream_t stream);

/* ---- the same calls with options: operand type and SyncBN --------------------------------------------------------------
 * Data-parallel training shards the batch over the GPUs of a node; the reference's train-mode BatchNorm2d
 * (models/pointnet2_utils.py:208-213) then sees per-replica statistics.  With `sync` non-NULL every train-mode BatchNorm of the
 * chain uses GLOBAL-batch statistics instead: per layer the library writes this rank's fp64 sums (sum z, sum z^2 forward;
 * sum dy, sum dy*z backward: 2 * c_out doubles) into `exchange` and calls `allreduce(user, exchange, 2 * c_out, stream)`,
 * which must SUM the buffer over all ranks in place, ordered on `stream` (under PyTorch: torch.distributed.all_reduce on a
 * tensor wrapping the buffer; in a C host: ncclAllReduce on `stream`) and return 0.  Statistics, running-stat updates and the
 * dZ constants then use the global sums and the global count P * world; d_gamma / d_beta stay this rank's contribution (the
 * gradient exchange averages them).  `exchange`: caller-owned device buffer of >= 4 * max(c_out) doubles.  One small
 * collective per BatchNorm layer and pass: 2 x 9 of them per step for the SSG encoder.  Eval mode ignores `sync`.
 * bf16 != 0: the _bf16 variant above. */
typedef int (*mp_allreduce_f64_fn)(void* user, double* device_buffer, int64_t count, mp_stream_t stream);
typedef struct {
    mp_allreduce_f64_fn allreduce;
    void* user;
    int64_t world;      /* number of ranks, each with the same P */
    double* exchange;   /* device, >= 4 * max(c_out) doubles */
} mp_syncbn_t;
int mp_sa_mlp_fwd_ex(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                     double momentum, double eps, float* out, int32_t* argk, float* zmax, void* workspace,
                     size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream);
int mp_sa_mlp_bwd_ex(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, int training,
                     const float* grad_out, const float* out, const int32_t* argk, const float* zmax,
                     const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols, void* workspace,
                     size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream);
/* [r3] The gather forms (mp_sa_mlp_{fwd,bwd}_gather_f32 / _bf16: factorised or gathered first layer) with the same hook: a data-parallel
 * run with SyncBN keeps the factorised first layer (the level's BatchNorm sums are exchanged exactly as in the _ex calls above). */
int mp_sa_mlp_fwd_gather_ex(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                            int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                            void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream);
int mp_sa_mlp_bwd_gather_ex(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                            int training, const float* grad_out, const float* out, const int32_t* argk,
                            const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                            void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream);

/* ---- fused "gradient from rank-B factors + Adam" for the head matrices (scope table row f1, "next") -----------------
 * replaces, for a Linear weight W [O, I] whose gradient is dW = g^T x (g = dLoss/dy [Bg,O], x = input [Bg,I]):
 *   the dW GEMM of autograd (models/pointnet2_cls_ssg.py:311,336,327 fc3 / fc_normals / sm_fc3) + the Adam update of
 *   torch.optim.Adam (train_maskplanner.py:159, defaults: no amsgrad / weight decay).  dW is never materialised; under
 *   data parallelism the factors (x, g) are gathered instead of all-reducing dW (grad_scale = 1/world).
 *   param / exp_avg / exp_avg_sq [O, I] updated in place; step = 1-based update count, or -- for launches recorded into a
 *   hipGraph, where a host-side count would be frozen -- step_dev: device pointer to the count as one f32 (step ignored). */
int mp_adam_lowrank_f32(float* param, float* exp_avg, float* exp_avg_sq, const float* x, const float* g, int64_t Bg,
                        int64_t O, int64_t I, double grad_scale, double lr, double beta1, double beta2, double eps,
                        int64_t step, const float* step_dev, mp_stream_t stream);

/* input gradient of the same layers: grad_x [B, I] = g [B, O] * W [O, I] for B <= 32 (one streaming read of W; the
 * library GEMM rocBLAS selects for this skinny shape reaches ~0.6 TB/s).  I % 4 == 0.  grad_x is overwritten. */
/* the same product on the matrix cores (fp32 operands as three bf16 planes, fp32 accumulation: fp32-accurate): B <= 32, I % 128 == 0,
 * no workspace; the K slices meet in grad_x through fp32 atomics, so the summation order is not fixed (mp_linear_dx_skinny_f32 is the
 * bit-reproducible form).  ~4 TB/s of weight stream against ~1.5 TB/s. */
int mp_linear_dx_mfma_f32(const float* g, const float* weight, int64_t B, int64_t O, int64_t I, float* grad_x, mp_stream_t stream);
/* grad_x = g1 W1 + g2 W2 for two Linears fed by the same activation, one launch (the K slices of both add into grad_x): no fan-out add */
int mp_linear_dx_mfma2_f32(const float* g1, const float* w1, int64_t O1, const float* g2, const float* w2, int64_t O2, int64_t B, int64_t I,
                           float* grad_x, mp_stream_t stream);
/* weight gradient of the same layers, materialised: dW [O, I] = g^T x (B <= 32 rows of factors, I % 4 == 0), b-ordered fma chains, one
 * streaming write of dW -- for loops that keep torch.optim.Adam on dense gradients (train_maskplanner.py:159) */
int mp_linear_dw_outer_f32(const float* g, const float* x, int64_t B, int64_t O, int64_t I, float* dW, mp_stream_t stream);
size_t mp_linear_dx_skinny_workspace_bytes(int64_t B, int64_t O, int64_t I);
int mp_linear_dx_skinny_f32(const float* g, const float* weight, int64_t B, int64_t O, int64_t I, float* grad_x,
                            void* workspace, size_t workspace_bytes, mp_stream_t stream);

/* dW [Co, Ci] = dz^T x over P rows (dz [P, Co], x [P, Ci], row-major; Co, Ci multiples of 4, <= 1024): the weight gradient of the
 * per-source-point half of a factorised first layer (models/pointnet2_utils.py:208-213 applied before the grouping);
 * row slices add with atomics (summation order not fixed) into a dW the call clears first (unless it lies in the armed zero arena). */
int mp_dw_gemm_f32(const float* dz, const float* x, int64_t P, int64_t Co, int64_t Ci, float* dW, mp_stream_t stream);

/* ---- head blocks: Linear (+ BatchNorm1d + ReLU + Dropout) over a skinny batch, one launch each way [r4] ------------------------------
 * replaces: models/pointnet2_cls_ssg.py:309-327 `self.dropout(F.relu(self.bn1(self.fc1(x))))` (bn != 0) and the plain nn.Linear of
 *           :311, :327, :336 (bn == 0), with their autograd.  csrc/head_linear.hip.
 *   x [B <= 32, I] f32, weight [O, I], bias [O] or NULL; I in {128, 256, 512, 1024, 2048} (mp_head_block_supported).
 *   forward: y [B, O]; with bn: z [B, O] = the Linear's output (kept for the backward), save_mean / save_rstd [O], running statistics
 *   updated like nn.BatchNorm1d (training) or used (eval), rng / drop_p / layer as mp_bn_relu_drop_rows_f32 (rng NULL: no dropout).
 *   backward (bn blocks): dz [B, O] = gradient at the Linear's output (the factor of dW = dz^T x), grad_gamma / grad_beta [O] (or NULL),
 *   grad_x [B, I] = dz W; with mp_head_block_bwd_slices(O) > 1 the row slices of W add their tiles with atomics (summation order not
 *   fixed) into a grad_x the call clears first (unless it lies in the armed zero arena).  O <= 4096, I % 64 == 0. */
typedef struct {
    /* forward */
    const float* x;          /* [B, I] */
    const float* weight;     /* [O, I] */
    const float* bias;       /* [O] or NULL */
    int64_t O;
    int bn, training;        /* bn == 0: plain Linear (y only) */
    double momentum, eps;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* z;                /* [B, O] the Linear's output (bn blocks; kept for the backward) */
    float* y;                /* [B, O] */
    float* save_mean;        /* [O] */
    float* save_rstd;        /* [O] */
    double drop_p;
    const int64_t* rng;      /* device (seed, step) or NULL */
    int layer;
    /* backward (bn blocks) */
    const float* grad_y;     /* [B, O] */
    float* dz;               /* [B, O] */
    float* grad_gamma;       /* [O] or NULL */
    float* grad_beta;        /* [O] or NULL */
    float* grad_x;           /* [B, I]; two blocks of one call may name the SAME buffer: their contributions add (no fan-out add) */
} mp_head_block_t;
/* n = 1 or 2 blocks (same B and I) in ONE launch: the two branches of the heads advance side by side (fc1 / sm_fc1, fc2 / sm_fc2) */
int mp_head_blocks_fwd_f32(int n, const mp_head_block_t* blocks, int64_t B, int64_t I, mp_stream_t stream);
int mp_head_blocks_bwd_f32(int n, const mp_head_block_t* blocks, int64_t B, int64_t I, mp_stream_t stream);
int mp_head_block_supported(int64_t B, int64_t I, int64_t O);
int mp_head_block_fwd_f32(const float* x, const float* weight, const float* bias, int64_t B, int64_t I, int64_t O, int bn, int training,
                          double momentum, double eps, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float* z, float* y, float* save_mean, float* save_rstd, double drop_p, const int64_t* rng, int layer,
                          mp_stream_t stream);
/* one or two plain Linears fed by the same x (w2 NULL: one) in one launch (models/pointnet2_cls_ssg.py:311 + :327: fc3 and fc_normals
 * both read the last hidden activation); their input gradient in one launch: mp_linear_dx_mfma2_f32 below */
int mp_head_linear2_fwd_f32(const float* x, int64_t B, int64_t I, const float* w1, const float* b1, int64_t O1, float* y1,
                            const float* w2, const float* b2, int64_t O2, float* y2, mp_stream_t stream);
int mp_head_block_bwd_slices(int64_t O);
int mp_head_block_bwd_f32(const float* grad_y, const float* y, const float* z, const float* weight, int64_t B, int64_t I, int64_t O,
                          int training, const float* gamma, const float* save_mean, const float* save_rstd, double drop_p, float* dz,
                          float* grad_gamma, float* grad_beta, float* grad_x, mp_stream_t stream);

/* ---- zero arena (launch count) -------------------------------------------------------------------------------------------
 * Outputs the library accumulates with atomics start from zero; by default each call clears its own (one small launch each).
 * mp_zero_arena_arm clears [base, base + bytes) with ONE launch on `stream` and remembers the range: until it is armed again or
 * disarmed, a call on the same stream whose zero-initialised output lies inside the range skips its own clear.  The caller hands
 * out every part of the range at most once per arming.  [r5] One armed range PER STREAM (a mutex-protected table keyed by the stream
 * handle -- bookkeeping of caller-owned memory, the library holds no device memory): callers on different streams do not interact.
 * mp_zero_arena_disarm_stream ends the arming of one stream, mp_zero_arena_disarm of every stream.
 * replaces: the implicit zero-initialisation of autograd's scatter / index_add / matmul outputs (models/pointnet2_utils.py:45-62,
 * pytorch3d knn_points backward) -- launch bookkeeping only, no arithmetic. */
int mp_zero_arena_arm(void* base, size_t bytes, mp_stream_t stream);
int mp_zero_arena_disarm(void);
int mp_zero_arena_disarm_stream(mp_stream_t stream);
/* [r4] mp_zero_arena_arm that also adds 1 to up to 40 int64 and 8 float32 device counters in the same launch (the num_batches_tracked of
 * the step's train-mode BatchNorm layers, models/pointnet2_utils.py:208-213 / pointnet2_cls_ssg.py:309-327; a dropout step; an
 * optimizer's device-side update count): bytes a non-zero multiple of 16. */
int mp_zero_arena_arm_ticks(void* base, size_t bytes, int n_i64, int64_t* const* counters_i64, int n_f32, float* const* counters_f32,
                            mp_stream_t stream);

/* ---- optional per-kernel device timing (bench / profiling aid; off by default) ------------------------------------
 * No counterpart in the reference (its only timing is wall-clock prints: train_maskplanner.py:236-239).
 * When enabled, launches inside the library are bracketed by HIP events on the launch stream.  collect() waits for
 * them and writes one line per kernel: "name\tcalls\ttotal_ms\talgorithmic_flops\talgorithmic_bytes\n";
 * returns the text length, 0 if nothing was recorded, MP_EWORKSPACE if `cap` is too small. */
int mp_profiler_enable(int on);
int mp_profiler_collect(char* buf, size_t cap);
/* [r5] Marks: launches whose tag contains `tag_substr` (several substrings: separated by '|'; at most 8 launches per recording) are bracketed
 * by two one-thread kernels that append the device's wall clock to a ring while a stream records a hipGraph, so that every replay timestamps
 * them; mp_profiler_read_marks returns their durations summed over the last `last_n` executions (mp_profiler_collect's line format, calls =
 * samples).  bench.py: the roofline kernel's average duration over the replayed, timed steps.  NULL / "": no marks. */
int mp_profiler_mark(const char* tag_substr);
int mp_profiler_read_marks(char* buf, size_t cap, int last_n);

/* Adam (torch defaults, no weight decay / amsgrad) for `count` dense tensors in as few launches as their pointers fit into kernel
 * arguments (48 per launch).  params / grads / exp_avg / exp_avg_sq: HOST arrays of device pointers, numels their lengths; grad_scale
 * multiplies every gradient (1/world under data parallelism when the sum was not averaged).  step > 0: host-side step count; step
 * <= 0: the count is read from step_dev (device float, graph-capturable).  Replaces torch.optim.Adam.step() on the non-factored
 * parameters (train_maskplanner.py:159). */
int mp_adam_multi_f32(int64_t count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* numels, double grad_scale, double lr, double beta1, double beta2, double eps, int64_t step,
                      const float* step_dev, mp_stream_t stream);
/* Column sums of `count` skinny matrices g[i] [rows, cols[i]] -> out[i] [cols[i]] in one launch: the bias gradients of the
 * head Linears (db = dy.sum(0); models/pointnet2_cls_ssg.py:270-295), which autograd would produce with one reduce launch
 * each.  g / out / cols are HOST arrays (device pointers and widths travel in the kernel arguments). */
int mp_colsum_multi_f32(int64_t count, const void* const* g, void* const* out, const int64_t* cols, int64_t rows,
                        mp_stream_t stream);

/* Batch collation: out[b, r, :] = r < len_b ? flat[offsets[b] + r, :] : fill, len_b = offsets[b+1] - offsets[b]; flat
 * [offsets[B], D], offsets i64 [B+1] on the device, out [B, R, D] (rows beyond R are dropped).  Replaces the per-sample numpy
 * concatenate + torch.stack of utils/dataset/paintnet_ODv1.py:738-748 (add_fake_vectors_v2 :887-904: fill -100;
 * add_fake_values_v2 :907-925: fill -1, D = 1). */
int mp_pad_ragged_f32(const float* flat, const int64_t* offsets, int64_t B, int64_t R, int64_t D, float fill, float* out,
                      mp_stream_t stream);


/* ---- lambda-segments of a batch on the device ---------------------------------------------------------------------------
 * replaces: utils/pointcloud.py:294-413 get_sequences_of_lambda_points (+ add_padding :98-105) as the dataset calls it per
 *           sample (utils/dataset/paintnet_ODv1.py:294) followed by the collate function's padding (:738-748).
 *   poses [T, D] f32: the poses of all samples back to back; stroke_ids [T] f32 (per sample ascending 0, 0, .., 1, ..);
 *   offsets [B+1] i64: sample b owns rows offsets[b] .. offsets[b+1].  Per stroke of L >= lambda poses: with overlapping > 0
 *   (L - lambda) / (lambda - overlapping) + 1 windows of lambda consecutive poses, stride lambda - overlapping; with
 *   overlapping = 0, L / lambda windows starting at pose (L % lambda) / 2.  Shorter strokes are dropped and the others
 *   renumbered.  out_traj [B, R, lambda*D] (rows behind a sample's last window: -100), out_ids [B, R] (-1); the caller picks
 *   R >= the largest window count (the reference pads each sample to (n - lambda) / (lambda - overlapping) + 1 rows, resp.
 *   n / lambda, and the batch to the largest of those).  status [B] i32: MP_OK, MP_EINVAL (ids not ascending / not contiguous,
 *   or more windows than R), MP_EUNSUPPORTED (more than 1024 strokes in a sample). */
int mp_lambda_segments_f32(const float* poses, const float* stroke_ids, const int64_t* offsets, int64_t B, int64_t D,
                           int64_t lambda, int64_t overlapping, int64_t R, float* out_traj, float* out_ids, int32_t* status,
                           mp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MASKPLANNER_HIP_H */
