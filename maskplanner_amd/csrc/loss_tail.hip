// Small fused kernels for the tail of the model and of the loss (gfx950): each replaces a chain of 20-40 tiny
// elementwise / reduction launches (and as many again in autograd's backward) that cost device time, not bandwidth.
//
//  * pose output   -- models/pointnet2_cls_ssg.py:332-339: tanh -> view(B,-1,3) -> F.normalize(dim=-1) * weight_orient,
//                     interleaved with the positions into [B, S, lambda*6].
//  * stroke-mask loss (binary targets) -- loss_handler.py:877-934: BCE-with-logits of the matched (pred mask, target mask)
//                     pairs, .sum(-1).mean() over the matched pairs, plus the weighted confidence BCE over all masks.
// All reductions run in a fixed order (deterministic).
#include "common.h"

namespace {

// ---- pose output ---------------------------------------------------------------------------------------------
// pos [B, n_pose*3], raw [B, n_pose*3] -> out [B, n_pose, 6] = (pos, normalize(tanh(raw)) * w)
__global__ __launch_bounds__(256) void pose_output_kernel(const float* __restrict__ pos, const float* __restrict__ raw,
                                                          int64_t n, float w, float* __restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float t0 = tanhf(raw[3 * p]), t1 = tanhf(raw[3 * p + 1]), t2 = tanhf(raw[3 * p + 2]);
    const float nrm = sqrtf((t0 * t0 + t1 * t1) + t2 * t2);
    const float d = fmaxf(nrm, 1e-12f);                           // F.normalize: v / max(||v||, eps)
    float* o = out + 6 * p;
    o[0] = pos[3 * p]; o[1] = pos[3 * p + 1]; o[2] = pos[3 * p + 2];
    o[3] = (t0 / d) * w; o[4] = (t1 / d) * w; o[5] = (t2 / d) * w;
}

__global__ __launch_bounds__(256) void pose_output_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ raw,
                                                              int64_t n, float w, float* __restrict__ grad_pos,
                                                              float* __restrict__ grad_raw)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float* g = grad_out + 6 * p;
    if (grad_pos) { grad_pos[3 * p] = g[0]; grad_pos[3 * p + 1] = g[1]; grad_pos[3 * p + 2] = g[2]; }
    if (!grad_raw) return;
    const float t0 = tanhf(raw[3 * p]), t1 = tanhf(raw[3 * p + 1]), t2 = tanhf(raw[3 * p + 2]);
    const float nrm = sqrtf((t0 * t0 + t1 * t1) + t2 * t2);
    const float d = fmaxf(nrm, 1e-12f);
    const float g0 = g[3] * w, g1 = g[4] * w, g2 = g[5] * w;      // gradient w.r.t. the unit vector
    // y = t / d, d = clamp_min(||t||, eps): dy/dt = I/d - (t t^T) / (d^2 ||t||) where the clamp is inactive, I/d where active
    float a0 = g0 / d, a1 = g1 / d, a2 = g2 / d;
    if (nrm > 1e-12f) {
        const float dot = (g0 * t0 + g1 * t1) + g2 * t2;
        const float c = dot / (d * d * nrm);
        a0 -= t0 * c; a1 -= t1 * c; a2 -= t2 * c;
    }
    grad_raw[3 * p] = a0 * (1.0f - t0 * t0);
    grad_raw[3 * p + 1] = a1 * (1.0f - t1 * t1);
    grad_raw[3 * p + 2] = a2 * (1.0f - t2 * t2);
}

// ---- stroke-mask loss, binary targets ---------------------------------------------------------------------------------
__device__ __forceinline__ float bce_logits(float x, float t) { return (fmaxf(x, 0.0f) - x * t) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one workgroup per (sample, predicted mask): sum_s BCE(pred[b,m,s], [ids[b,s] == uid]) of a matched mask (0 otherwise)
__global__ __launch_bounds__(256) void mask_loss_rows_kernel(const float* __restrict__ pred, const float* __restrict__ ids,
                                                             const int64_t* __restrict__ match, const float* __restrict__ uniq,
                                                             int M, int S, int cap, float* __restrict__ per_mask)
{
    __shared__ float red[4];
    const int bm = blockIdx.x, b = bm / M;
    const int64_t k = match[bm];
    float s = 0.0f;
    if (k >= 0) {
        const float uid = uniq[(size_t)b * cap + k];
        const float* x = pred + (size_t)bm * S;
        const float* id = ids + (size_t)b * S;
        for (int i = threadIdx.x; i < S; i += 256) s += bce_logits(x[i], id[i] == uid ? 1.0f : 0.0f);
    }
    s = mp::wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) per_mask[bm] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = w_masks * sum(per_mask) / n_matched + w_conf * mean_{b,m}( weight * BCE(score, matched) ); stats = (n_matched)
__global__ __launch_bounds__(256) void mask_loss_final_kernel(const float* __restrict__ per_mask, const float* __restrict__ scores,
                                                              const int64_t* __restrict__ match, int BM, float w_masks, float w_conf,
                                                              float no_stroke_weight, float* __restrict__ out,
                                                              float* __restrict__ n_matched_out)
{
    __shared__ float r0[4], r1[4], r2[4];
    float sm = 0.0f, sc = 0.0f, nm = 0.0f;
    for (int i = threadIdx.x; i < BM; i += 256) {
        const bool matched = match[i] >= 0;
        sm += per_mask[i];
        nm += matched ? 1.0f : 0.0f;
        sc += (matched ? 1.0f : no_stroke_weight) * bce_logits(scores[i], matched ? 1.0f : 0.0f);
    }
    sm = mp::wave_sum_f32(sm); sc = mp::wave_sum_f32(sc); nm = mp::wave_sum_f32(nm);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = sm; r1[threadIdx.x >> 6] = sc; r2[threadIdx.x >> 6] = nm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tm = (r0[0] + r0[1]) + (r0[2] + r0[3]);
        const float tc = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        const float tn = (r2[0] + r2[1]) + (r2[2] + r2[3]);
        out[0] = w_masks * (tm / tn) + w_conf * (tc / (float)BM);
        n_matched_out[0] = tn;
    }
}

__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ pred,
                                                            const float* __restrict__ scores, const float* __restrict__ ids,
                                                            const int64_t* __restrict__ match, const float* __restrict__ uniq,
                                                            const float* __restrict__ n_matched, int M, int S, int cap, int BM,
                                                            float w_masks, float w_conf, float no_stroke_weight,
                                                            float* __restrict__ grad_pred, float* __restrict__ grad_scores)
{
    const int bm = blockIdx.x, b = bm / M;
    const int64_t k = match[bm];
    const float g = grad_out[0];
    float* gp = grad_pred + (size_t)bm * S;
    if (k >= 0) {
        const float c = g * w_masks / n_matched[0];
        const float uid = uniq[(size_t)b * cap + k];
        const float* x = pred + (size_t)bm * S;
        const float* id = ids + (size_t)b * S;
        for (int i = threadIdx.x; i < S; i += 256) gp[i] = c * (sigmoidf(x[i]) - (id[i] == uid ? 1.0f : 0.0f));
    } else {
        for (int i = threadIdx.x; i < S; i += 256) gp[i] = 0.0f;
    }
    if (threadIdx.x == 0 && grad_scores) {
        const bool matched = k >= 0;
        grad_scores[bm] = g * w_conf / (float)BM * (matched ? 1.0f : no_stroke_weight) * (sigmoidf(scores[bm]) - (matched ? 1.0f : 0.0f));
    }
}

}  // namespace

extern "C" int mp_pose_output_f32(const float* pos, const float* raw, int64_t n_pose, double weight_orient, float* out,
                                  mp_stream_t stream_)
{
    if (n_pose < 0) return MP_EINVAL;
    if (n_pose == 0) return MP_OK;
    if (!pos || !raw || !out) return MP_EINVAL;
    hipLaunchKernelGGL(pose_output_kernel, dim3((unsigned)((n_pose + 255) / 256)), dim3(256), 0, mp_stream(stream_), pos, raw, n_pose,
                       (float)weight_orient, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_pose_output_bwd_f32(const float* grad_out, const float* raw, int64_t n_pose, double weight_orient,
                                      float* grad_pos, float* grad_raw, mp_stream_t stream_)
{
    if (n_pose < 0) return MP_EINVAL;
    if (n_pose == 0) return MP_OK;
    if (!grad_out || !raw) return MP_EINVAL;
    hipLaunchKernelGGL(pose_output_bwd_kernel, dim3((unsigned)((n_pose + 255) / 256)), dim3(256), 0, mp_stream(stream_), grad_out, raw,
                       n_pose, (float)weight_orient, grad_pos, grad_raw);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_mask_loss_f32(const float* pred_masks, const float* scores, const float* target_ids, const int64_t* match_col,
                                const float* uniq_ids, int64_t B, int64_t M, int64_t S, double w_masks, double w_conf,
                                double no_stroke_weight, float* per_mask, float* out, float* n_matched, mp_stream_t stream_)
{
    if (B < 0 || M < 0 || S < 0) return MP_EINVAL;
    if (B * M == 0) return MP_EINVAL;
    if (!pred_masks || !scores || !target_ids || !match_col || !uniq_ids || !per_mask || !out || !n_matched) return MP_EINVAL;
    if (B * M > (1 << 24) || S > (1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    hipLaunchKernelGGL(mask_loss_rows_kernel, dim3((unsigned)(B * M)), dim3(256), 0, stream, pred_masks, target_ids, match_col, uniq_ids,
                       (int)M, (int)S, MP_MASK_CAP, per_mask);
    MP_CHECK_LAUNCH();
    hipLaunchKernelGGL(mask_loss_final_kernel, dim3(1), dim3(256), 0, stream, per_mask, scores, match_col, (int)(B * M), (float)w_masks,
                       (float)w_conf, (float)no_stroke_weight, out, n_matched);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_mask_loss_bwd_f32(const float* grad_out, const float* pred_masks, const float* scores, const float* target_ids,
                                    const int64_t* match_col, const float* uniq_ids, const float* n_matched, int64_t B, int64_t M,
                                    int64_t S, double w_masks, double w_conf, double no_stroke_weight, float* grad_masks,
                                    float* grad_scores, mp_stream_t stream_)
{
    if (B < 0 || M < 0 || S < 0) return MP_EINVAL;
    if (B * M == 0) return MP_OK;
    if (!grad_out || !pred_masks || !scores || !target_ids || !match_col || !uniq_ids || !n_matched || !grad_masks) return MP_EINVAL;
    hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3((unsigned)(B * M)), dim3(256), 0, mp_stream(stream_), grad_out, pred_masks, scores,
                       target_ids, match_col, uniq_ids, n_matched, (int)M, (int)S, MP_MASK_CAP, (int)(B * M), (float)w_masks, (float)w_conf,
                       (float)no_stroke_weight, grad_masks, grad_scores);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
