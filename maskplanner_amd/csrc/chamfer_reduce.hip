// Reductions of the chamfer wrapper (gfx950): per-cloud sum or mean over the points, then sum or mean over the batch.
//
// Reference: pytorch3d_chamfer.py:295-326 -- cham.sum(1), / lengths, .sum(), / N: four tiny launches forward and as many
// again in autograd's backward, three times per training step.  Here: one workgroup forward (deterministic: every cloud
// is summed by one wave in a fixed lane order, the clouds are combined in index order), one elementwise kernel backward.
// The nearest-neighbour distances come from knn.hip with rows at or beyond a cloud's length already zero
// (pytorch3d_chamfer.py:263-266), so the forward needs no mask; the backward writes zeros there.
#include "common.h"

namespace {

constexpr int CR_THREADS = 1024;

__global__ __launch_bounds__(CR_THREADS) void chamfer_reduce_kernel(const float* __restrict__ cham,
                                                                    const int64_t* __restrict__ lengths, int N, int P,
                                                                    int point_mean, int batch_mode, float div, float scale,
                                                                    float* __restrict__ out)
{
    __shared__ float per_cloud[CR_THREADS / 64];
    __shared__ float total;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) total = 0.0f;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += CR_THREADS / 64) {
        const int n = n0 + wave;
        float s = 0.0f;
        if (n < N) {
            const float* row = cham + (size_t)n * P;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            int p = lane;
            for (; p + 192 < P; p += 256) { a0 += row[p]; a1 += row[p + 64]; a2 += row[p + 128]; a3 += row[p + 192]; }
            for (; p < P; p += 64) a0 += row[p];
            s = mp::wave_sum_f32((a0 + a1) + (a2 + a3));
            if (point_mean) s = s / (float)lengths[n];
            if (batch_mode == 0 && lane == 0) out[n] = s * scale;
        }
        if (batch_mode != 0) {
            if (lane == 0) per_cloud[wave] = (n < N) ? s : 0.0f;
            __syncthreads();
            if (threadIdx.x == 0) {
                float t = total;
                for (int w = 0; w < CR_THREADS / 64; ++w) t += per_cloud[w];   // clouds in index order
                total = t;
            }
            __syncthreads();
        }
    }
    if (batch_mode != 0 && threadIdx.x == 0) out[0] = ((batch_mode == 2) ? total / div : total) * scale;
}

// grad_cham[n,p] = g(n) / (len_n if point mean) / (div if batch mean) for p < len_n (when lengths are given), else 0
__global__ __launch_bounds__(256) void chamfer_reduce_bwd_kernel(const float* __restrict__ grad_out,
                                                                 const int64_t* __restrict__ lengths, int N, int P,
                                                                 int point_mean, int batch_mode, float div, float scale,
                                                                 float* __restrict__ grad_cham)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)N * P) return;
    const int n = (int)(e / P), p = (int)(e - (int64_t)n * P);
    float g = (batch_mode == 0 ? grad_out[n] : grad_out[0]) * scale;
    if (batch_mode == 2) g = g / div;
    int64_t len = P;
    if (lengths) len = lengths[n];
    if (point_mean) g = g / (float)len;
    grad_cham[e] = p < len ? g : 0.0f;
}

}  // namespace

extern "C" int mp_chamfer_reduce_f32(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                                     int batch_mode, double div, double scale, float* out, mp_stream_t stream_)
{
    if (N < 0 || P < 0 || batch_mode < 0 || batch_mode > 2) return MP_EINVAL;
    if (N == 0) return MP_OK;
    if (!out || (P > 0 && !cham) || (point_mean && !lengths)) return MP_EINVAL;
    if (N > (1 << 24) || P > (1 << 30)) return MP_EUNSUPPORTED;
    MP_LAUNCH("chamfer_reduce_kernel", (double)(N * P), 4.0 * (double)(N * P), chamfer_reduce_kernel, dim3(1), dim3(CR_THREADS), 0,
              mp_stream(stream_), cham, lengths, (int)N, (int)P, point_mean, batch_mode, (float)div, (float)scale, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_chamfer_reduce_bwd_f32(const float* grad_out, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                                         int batch_mode, double div, double scale, float* grad_cham, mp_stream_t stream_)
{
    if (N < 0 || P < 0 || batch_mode < 0 || batch_mode > 2) return MP_EINVAL;
    if (N * P == 0) return MP_OK;
    if (!grad_out || !grad_cham || (point_mean && !lengths)) return MP_EINVAL;
    if (N > (1 << 24) || P > (1 << 30)) return MP_EUNSUPPORTED;
    MP_LAUNCH("chamfer_reduce_bwd_kernel", (double)(N * P), 4.0 * (double)(N * P), chamfer_reduce_bwd_kernel,
              dim3((unsigned)((N * P + 255) / 256)), dim3(256), 0, mp_stream(stream_), grad_out, lengths, (int)N, (int)P, point_mean,
              batch_mode, (float)div, (float)scale, grad_cham);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
