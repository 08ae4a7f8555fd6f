// Adam for the dense (non-factored) parameters as ONE small family of launches (gfx950).
//
// Reference: train_maskplanner.py:159 (torch.optim.Adam(model.parameters(), lr), torch defaults: betas (0.9, 0.999), eps 1e-8, no
// weight decay / amsgrad).  The encoder, the BatchNorm affine parameters and the biases are ~150 tensors with 0.9 M elements
// in all: 22 MB of traffic, a few microseconds of HBM time -- but torch's capturable fused Adam spends 60-90 us per step on them
// (per-tensor device step counters, multi_tensor_apply chunking).  Here up to AM_MAX tensors travel in the kernel arguments
// (pointers by value: nothing to upload, graph-capturable as is), a block owns one 4096-element chunk of one tensor, and the
// step count lives in one device float shared with the factor Adam (adam_lowrank.hip).  Same update algebra as torch's fused
// kernel: lerp form of the first moment, sqrt(v) / sqrt(1 - beta2^t) + eps in the denominator.
#include "common.h"

namespace {

constexpr int AM_MAX = 48;       // tensors per launch: 48 x (4 pointers + 2 ints) = 1.9 KB of kernel arguments
constexpr int AM_CHUNK = 4096;   // elements per block: 256 threads x 4 float4

struct AdamMultiArgs {
    float* p[AM_MAX];
    const float* g[AM_MAX];
    float* m[AM_MAX];
    float* v[AM_MAX];
    int n[AM_MAX];
    int blk0[AM_MAX + 1];        // first block of every tensor; blk0[count] = grid size
    int count;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMultiArgs a, float gscale, float lr, float beta1, float beta2, float eps,
                                                         float lr_c1, float inv_sqrt_c2, const float* __restrict__ step_dev)
{
    __shared__ float s_c[2];
    if (step_dev) {   // graph-capturable form: bias corrections from the device-side step count, once per block, in fp64
        if (threadIdx.x == 0) {
            const double st = (double)step_dev[0];
            s_c[0] = (float)((double)lr / (1.0 - pow((double)beta1, st)));
            s_c[1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, st)));
        }
        __syncthreads();
        lr_c1 = s_c[0];
        inv_sqrt_c2 = s_c[1];
    }
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;   // uniform: scalar loop over <= 48 entries
    const int n = a.n[t];
    const int e0 = ((int)blockIdx.x - a.blk0[t]) * AM_CHUNK;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    auto upd = [&](float grad, float& pp, float& mm, float& vv) {
        grad *= gscale;
        mm = mm + (1.0f - beta1) * (grad - mm);
        vv = beta2 * vv + (1.0f - beta2) * grad * grad;
        pp -= lr_c1 * (mm / (sqrtf(vv) * inv_sqrt_c2 + eps));
    };
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
#pragma unroll
    for (int it = 0; it < AM_CHUNK / 1024; ++it) {
        const int e = e0 + it * 1024 + threadIdx.x * 4;
        if (e >= n) break;
        if (vec && e + 3 < n) {
            float4 p4 = *reinterpret_cast<float4*>(p + e), m4 = *reinterpret_cast<float4*>(m + e), v4 = *reinterpret_cast<float4*>(v + e);
            const float4 g4 = *reinterpret_cast<const float4*>(g + e);
            upd(g4.x, p4.x, m4.x, v4.x);
            upd(g4.y, p4.y, m4.y, v4.y);
            upd(g4.z, p4.z, m4.z, v4.z);
            upd(g4.w, p4.w, m4.w, v4.w);
            *reinterpret_cast<float4*>(p + e) = p4;
            *reinterpret_cast<float4*>(m + e) = m4;
            *reinterpret_cast<float4*>(v + e) = v4;
        } else {
            for (int c = 0; c < 4 && e + c < n; ++c) upd(g[e + c], p[e + c], m[e + c], v[e + c]);
        }
    }
}

}  // namespace

extern "C" int mp_adam_multi_f32(int64_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const int64_t* numels, double grad_scale, double lr, double beta1,
                                 double beta2, double eps, int64_t step, const float* step_dev, mp_stream_t stream_)
{
    if (count < 0 || (step <= 0 && !step_dev)) return MP_EINVAL;
    if (count == 0) return MP_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numels) return MP_EINVAL;
    const double sh = step > 0 ? (double)step : 1.0;   // placeholders when the device-side count is used
    const double c1 = 1.0 - pow(beta1, sh), c2 = 1.0 - pow(beta2, sh);
    hipStream_t stream = mp_stream(stream_);
    int64_t i = 0;
    while (i < count) {
        AdamMultiArgs a;
        a.count = 0;
        int blocks = 0;
        double elems = 0.0;
        for (; i < count && a.count < AM_MAX; ++i) {
            const int64_t n = numels[i];
            if (n < 0 || n >= ((int64_t)1 << 31)) return MP_EUNSUPPORTED;
            if (n == 0) continue;
            if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return MP_EINVAL;
            const int k = a.count++;
            a.p[k] = params[i]; a.g[k] = grads[i]; a.m[k] = exp_avg[i]; a.v[k] = exp_avg_sq[i];
            a.n[k] = (int)n;
            a.blk0[k] = blocks;
            blocks += (int)((n + AM_CHUNK - 1) / AM_CHUNK);
            elems += (double)n;
        }
        if (a.count == 0) break;
        for (int k = a.count; k <= AM_MAX; ++k) a.blk0[k] = blocks;
        MP_LAUNCH("adam_multi_kernel", 12.0 * elems, 28.0 * elems, adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a,
                  (float)grad_scale, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)(lr / c1), (float)(1.0 / sqrt(c2)), step_dev);
        MP_CHECK_LAUNCH();
    }
    return MP_OK;
}

// ---- column sums of several skinny matrices in one launch: the bias gradients of the head Linears ------------------------------
// db = g.sum(0) for g [rows, cols] (rows = batch): seven heads would be seven reduce launches of a few microseconds each
// (models/pointnet2_cls_ssg.py:270-295: fc1/fc2/fc3/fc_normals and the sm_ twins + mask_conf_out).  Pointers travel in the kernel
// arguments like adam_multi's; a thread owns one column and walks the rows (coalesced across columns, fixed order).
namespace {
constexpr int CS_MAX = 16;
struct ColsumArgs {
    const float* g[CS_MAX];
    float* out[CS_MAX];
    int cols[CS_MAX];
    int blk0[CS_MAX + 1];
    int count, rows;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumArgs a)
{
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
    const int c = ((int)blockIdx.x - a.blk0[t]) * 256 + threadIdx.x;
    const int C = a.cols[t];
    if (c >= C) return;
    const float* g = a.g[t] + c;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int r = 0;
    for (; r + 3 < a.rows; r += 4) {      // four independent loads in flight
        s0 += g[(size_t)r * C]; s1 += g[(size_t)(r + 1) * C]; s2 += g[(size_t)(r + 2) * C]; s3 += g[(size_t)(r + 3) * C];
    }
    for (; r < a.rows; ++r) s0 += g[(size_t)r * C];
    a.out[t][c] = (s0 + s1) + (s2 + s3);
}
}  // namespace

extern "C" int mp_colsum_multi_f32(int64_t count, const void* const* g, void* const* out, const int64_t* cols, int64_t rows,
                                   mp_stream_t stream_)
{
    if (count < 0 || rows < 0) return MP_EINVAL;
    if (count == 0) return MP_OK;
    if (!g || !out || !cols) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    for (int64_t base = 0; base < count; base += CS_MAX) {
        ColsumArgs a;
        a.count = (int)((count - base) < CS_MAX ? (count - base) : CS_MAX);
        a.rows = (int)rows;
        int blocks = 0;
        for (int i = 0; i < a.count; ++i) {
            if (!g[base + i] || !out[base + i] || cols[base + i] < 0 || cols[base + i] > (1 << 30)) return MP_EINVAL;
            a.g[i] = static_cast<const float*>(g[base + i]);
            a.out[i] = static_cast<float*>(out[base + i]);
            a.cols[i] = (int)cols[base + i];
            a.blk0[i] = blocks;
            blocks += (int)((cols[base + i] + 255) / 256);
        }
        a.blk0[a.count] = blocks;
        if (blocks == 0) continue;
        hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        MP_CHECK_LAUNCH();
    }
    return MP_OK;
}

