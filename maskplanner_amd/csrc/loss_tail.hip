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
                                                              float* __restrict__ n_matched_out,
                                                              const int32_t* __restrict__ status, int B,
                                                              const float* __restrict__ add_to)
{
    __shared__ float r0[4], r1[4], r2[4];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    // a sample whose matching failed (mp_mask_match_f32 status != 0) poisons the loss: the reference asserts / raises there
    // (loss_handler.py:852-854, scipy's ValueError), this path has no host in the loop to raise
    if (status) for (int i = threadIdx.x; i < B; i += 256) if (status[i] != 0) s_bad = 1;
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
        out[0] = s_bad ? __builtin_nanf("") : (w_masks * (tm / tn) + w_conf * (tc / (float)BM)) + (add_to ? add_to[0] : 0.0f);
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
                                double no_stroke_weight, float* per_mask, float* out, float* n_matched, const int32_t* status,
                                const float* add_to, mp_stream_t stream_)
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
                       (float)w_conf, (float)no_stroke_weight, out, n_matched, status, (int)B, add_to);
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

// ---- BatchNorm1d + ReLU over a skinny batch (the regression heads: [B=32, 1024] activations) ------------------------------
// torch runs this as collect-statistics + transform + running-stat update + clamp_min (4 launches, 3 more in backward) on
// 128 KB of data.  One thread per channel walks the B rows (coalesced across channels): statistics, normalisation, ReLU and
// the running-stat update in one launch; the backward (ReLU mask, dgamma / dbeta, dx) in another.
namespace {

// Workgroup = 64 channels x 4 row groups (256 threads): thread (c, g) keeps rows g, g+4, ... of channel c in registers (RPT of
// them, all fetched with independent loads: one memory latency), the four row groups are combined through LDS.
// RPT == 0: any B, rows re-read per pass.
template <int RPT>
__global__ __launch_bounds__(256) void bn_relu_rows_kernel(const float* __restrict__ x, int B, int C, int training,
                                                           float momentum, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ y,
                                                           float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                           float drop_p, const long long* __restrict__ rng, int layer)
{   // rng != NULL: nn.Dropout(p = drop_p) of the result in the same pass -- element (r, c) is kept when a counter-based hash of
    // (rng[0] = seed, rng[1] = step, layer, r * C + c) maps to [drop_p, 1), and scaled by 1 / (1 - drop_p).  A dropped element is an
    // exact 0, a kept one is positive iff the ReLU passed it: the backward kernel needs the scale only, not the mask.
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool live = c < C;
    constexpr int NR = RPT > 0 ? RPT : 1;
    float xv[NR];
    if constexpr (RPT > 0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const int r = g + 4 * k; xv[k] = (live && r < B) ? x[(size_t)r * C + c] : 0.0f; }
    }
    float mean = 0.0f, rstd = 0.0f;
    if (training) {
        float s = 0.0f;
        if constexpr (RPT > 0) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) s += xv[k];          // rows beyond B hold 0
        } else {
            if (live) for (int r = g; r < B; r += 4) s += x[(size_t)r * C + c];
        }
        red[g][cl] = s;
        __syncthreads();
        mean = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)B;
        __syncthreads();
        float v = 0.0f;
        if constexpr (RPT > 0) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) { const float d = xv[k] - mean; v += (g + 4 * k < B) ? d * d : 0.0f; }
        } else {
            if (live) for (int r = g; r < B; r += 4) { const float d = x[(size_t)r * C + c] - mean; v += d * d; }
        }
        red[g][cl] = v;
        __syncthreads();
        v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        const float var = v / (float)B;                       // biased: what normalises
        rstd = 1.0f / sqrtf(var + eps);
        if (live && g == 0 && running_mean) {
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (B > 1 ? v / (float)(B - 1) : var);   // unbiased
        }
    } else if (live) {
        mean = running_mean[c];
        rstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    if (!live) return;
    if (g == 0) { save_mean[c] = mean; save_rstd[c] = rstd; }
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const bool drop = rng != nullptr && drop_p > 0.0f;
    const unsigned long long key = drop ? (unsigned long long)rng[0] + 0xD1B54A32D192ED03ull * (unsigned long long)rng[1] +
                                          ((unsigned long long)(unsigned)layer << 48) : 0ull;
    const float keep_scale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
    auto finish_row = [&](int r, float xin) {
        float v = (xin - mean) * rstd * ga + be;
        v = v > 0.0f ? v : 0.0f;
        if (drop) {
            unsigned long long z = key + 0x9E3779B97F4A7C15ull * ((unsigned long long)r * (unsigned long long)C + (unsigned long long)c + 1ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;          // splitmix64 finaliser: 24 uniform bits per element
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
            v = u >= drop_p ? v * keep_scale : 0.0f;
        }
        y[(size_t)r * C + c] = v;
    };
    if constexpr (RPT > 0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = g + 4 * k;
            if (r < B) finish_row(r, xv[k]);
        }
    } else {
        for (int r = g; r < B; r += 4) finish_row(r, x[(size_t)r * C + c]);
    }
}

template <int RPT>
__global__ __launch_bounds__(256) void bn_relu_rows_bwd_kernel(const float* __restrict__ grad_y, const float* __restrict__ y,
                                                               const float* __restrict__ x, int B, int C, int training,
                                                               const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                               const float* __restrict__ save_rstd, float* __restrict__ grad_x,
                                                               float* __restrict__ grad_gamma, float* __restrict__ grad_beta, float keep_scale)
{   // keep_scale = 1 / (1 - p) of a dropout fused into the forward (1 without): y > 0 exactly where the element passed ReLU AND dropout
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool live = c < C;
    const float mean = live ? save_mean[c] : 0.0f, rstd = live ? save_rstd[c] : 0.0f;
    const float ga = (live && gamma) ? gamma[c] : 1.0f;
    constexpr int NR = RPT > 0 ? RPT : 1;
    float dyv[NR], xh[NR];
    float db = 0.0f, dg = 0.0f;
    if constexpr (RPT > 0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = g + 4 * k;
            const bool ok = live && r < B;
            const size_t o = ok ? (size_t)r * C + c : 0;
            const float yy = y[o], gy = grad_y[o], xx = x[o];
            dyv[k] = (ok && yy > 0.0f) ? gy * keep_scale : 0.0f;
            xh[k] = ok ? (xx - mean) * rstd : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) { db += dyv[k]; dg += dyv[k] * xh[k]; }
    } else if (live) {
        for (int r = g; r < B; r += 4) {
            const size_t o = (size_t)r * C + c;
            const float dy = y[o] > 0.0f ? grad_y[o] * keep_scale : 0.0f;
            db += dy;
            dg += dy * ((x[o] - mean) * rstd);
        }
    }
    red[0][g][cl] = db;
    red[1][g][cl] = dg;
    __syncthreads();
    db = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    dg = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    if (!live) return;
    if (g == 0) {
        if (grad_beta) grad_beta[c] = db;
        if (grad_gamma) grad_gamma[c] = dg;
    }
    if (!grad_x) return;
    const float inv = 1.0f / (float)B;
    if constexpr (RPT > 0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = g + 4 * k;
            if (r < B) grad_x[(size_t)r * C + c] = training ? ga * rstd * (dyv[k] - db * inv - xh[k] * dg * inv) : ga * rstd * dyv[k];
        }
    } else {
        for (int r = g; r < B; r += 4) {
            const size_t o = (size_t)r * C + c;
            const float dy = y[o] > 0.0f ? grad_y[o] * keep_scale : 0.0f;
            const float xhat = (x[o] - mean) * rstd;
            grad_x[o] = training ? ga * rstd * (dy - db * inv - xhat * dg * inv) : ga * rstd * dy;
        }
    }
}

}  // namespace

static int bn_relu_rows_fwd(const float* x, int64_t B, int64_t C, int training, double momentum, double eps,
                            const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                            float* save_mean, float* save_rstd, double drop_p, const int64_t* rng, int layer, mp_stream_t stream_)
{
    if (B <= 0 || C < 0 || drop_p < 0.0 || drop_p >= 1.0) return MP_EINVAL;
    if (C == 0) return MP_OK;
    if (!x || !y || !save_mean || !save_rstd || (!training && (!running_mean || !running_var))) return MP_EINVAL;
    if (B > 4096 || C > (1 << 24)) return MP_EUNSUPPORTED;   // a thread walks the rows: made for skinny batches
    const dim3 grid((unsigned)((C + 63) / 64));   // 64 channels per workgroup: 16 workgroups for the 1024-wide heads
    const long long* r = reinterpret_cast<const long long*>(rng);
    if (B <= 32)
        hipLaunchKernelGGL(bn_relu_rows_kernel<8>, grid, dim3(256), 0, mp_stream(stream_), x, (int)B, (int)C, training, (float)momentum,
                           (float)eps, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, (float)drop_p, r, layer);
    else if (B <= 64)
        hipLaunchKernelGGL(bn_relu_rows_kernel<16>, grid, dim3(256), 0, mp_stream(stream_), x, (int)B, (int)C, training, (float)momentum,
                           (float)eps, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, (float)drop_p, r, layer);
    else
        hipLaunchKernelGGL(bn_relu_rows_kernel<0>, grid, dim3(256), 0, mp_stream(stream_), x, (int)B, (int)C, training, (float)momentum,
                           (float)eps, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, (float)drop_p, r, layer);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

static int bn_relu_rows_bwd(const float* grad_y, const float* y, const float* x, int64_t B, int64_t C, int training,
                            const float* gamma, const float* save_mean, const float* save_rstd, float* grad_x,
                            float* grad_gamma, float* grad_beta, double keep_scale, mp_stream_t stream_)
{
    if (B <= 0 || C < 0) return MP_EINVAL;
    if (C == 0) return MP_OK;
    if (!grad_y || !y || !x || !save_mean || !save_rstd) return MP_EINVAL;
    if (B > 4096 || C > (1 << 24)) return MP_EUNSUPPORTED;
    const dim3 grid((unsigned)((C + 63) / 64));
    if (B <= 32)
        hipLaunchKernelGGL(bn_relu_rows_bwd_kernel<8>, grid, dim3(256), 0, mp_stream(stream_), grad_y, y, x, (int)B, (int)C, training, gamma,
                           save_mean, save_rstd, grad_x, grad_gamma, grad_beta, (float)keep_scale);
    else if (B <= 64)
        hipLaunchKernelGGL(bn_relu_rows_bwd_kernel<16>, grid, dim3(256), 0, mp_stream(stream_), grad_y, y, x, (int)B, (int)C, training, gamma,
                           save_mean, save_rstd, grad_x, grad_gamma, grad_beta, (float)keep_scale);
    else
        hipLaunchKernelGGL(bn_relu_rows_bwd_kernel<0>, grid, dim3(256), 0, mp_stream(stream_), grad_y, y, x, (int)B, (int)C, training, gamma,
                           save_mean, save_rstd, grad_x, grad_gamma, grad_beta, (float)keep_scale);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_bn_relu_rows_f32(const float* x, int64_t B, int64_t C, int training, double momentum, double eps,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                                   float* save_mean, float* save_rstd, mp_stream_t stream_)
{
    return bn_relu_rows_fwd(x, B, C, training, momentum, eps, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, 0.0, nullptr, 0,
                            stream_);
}

extern "C" int mp_bn_relu_rows_bwd_f32(const float* grad_y, const float* y, const float* x, int64_t B, int64_t C, int training,
                                       const float* gamma, const float* save_mean, const float* save_rstd, float* grad_x,
                                       float* grad_gamma, float* grad_beta, mp_stream_t stream_)
{
    return bn_relu_rows_bwd(grad_y, y, x, B, C, training, gamma, save_mean, save_rstd, grad_x, grad_gamma, grad_beta, 1.0, stream_);
}

// BatchNorm1d + ReLU + Dropout(p) in one launch (models/pointnet2_cls_ssg.py:309-327: `self.dropout(F.relu(self.bn1(...)))`).  rng: device
// int64 [2] = (seed, step); the caller advances the step once per training step.  The mask is a counter-based hash of (seed, step, layer,
// element), not torch's Philox stream: same distribution, different draws for a given seed.
extern "C" int mp_bn_relu_drop_rows_f32(const float* x, int64_t B, int64_t C, int training, double momentum, double eps,
                                        const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                                        float* save_mean, float* save_rstd, double drop_p, const int64_t* rng, int layer,
                                        mp_stream_t stream_)
{
    if (!rng) return MP_EINVAL;
    return bn_relu_rows_fwd(x, B, C, training, momentum, eps, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, drop_p, rng, layer,
                            stream_);
}

extern "C" int mp_bn_relu_drop_rows_bwd_f32(const float* grad_y, const float* y, const float* x, int64_t B, int64_t C, int training,
                                            const float* gamma, const float* save_mean, const float* save_rstd, float* grad_x,
                                            float* grad_gamma, float* grad_beta, double drop_p, mp_stream_t stream_)
{
    if (drop_p < 0.0 || drop_p >= 1.0) return MP_EINVAL;
    return bn_relu_rows_bwd(grad_y, y, x, B, C, training, gamma, save_mean, save_rstd, grad_x, grad_gamma, grad_beta, 1.0 / (1.0 - drop_p), stream_);
}
