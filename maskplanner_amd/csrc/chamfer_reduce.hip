// Reductions of the chamfer wrapper (gfx950): per-cloud sum or mean over the points, then sum or mean over the batch.
//
// Reference: pytorch3d_chamfer.py:295-326 -- cham.sum(1), / lengths, .sum(), / N: four tiny launches forward and as many
// again in autograd's backward, three times per training step.  Here: one workgroup per cloud plus a one-wave combine
// forward (deterministic: fixed lane / wave / cloud order), one elementwise kernel backward.
// The nearest-neighbour distances come from knn.hip with rows at or beyond a cloud's length already zero
// (pytorch3d_chamfer.py:263-266), so the forward needs no mask; the backward writes zeros there.
#include "common.h"

namespace {

// one workgroup per cloud: sum of its P distances (fixed lane / wave order), optionally divided by its length
// counter != NULL: the batch combine rides in the same launch -- the workgroup that finishes last (a device counter, left at zero again)
// adds the per-cloud values in cloud order exactly like chamfer_batch_kernel: same result bits, one launch less per loss term.
__global__ __launch_bounds__(256) void chamfer_rows_kernel(const float* __restrict__ cham, const int64_t* __restrict__ lengths,
                                                           int P, int point_mean, float scale, float* __restrict__ per_cloud,
                                                           unsigned* __restrict__ counter, int N, int batch_mode, float div,
                                                           float* __restrict__ out, const float* __restrict__ add_to)
{
    __shared__ float red[4];
    __shared__ bool last;
    const int n = blockIdx.x;
    const float* row = cham + (size_t)n * P;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int p = threadIdx.x;
    for (; p + 768 < P; p += 1024) { a0 += row[p]; a1 += row[p + 256]; a2 += row[p + 512]; a3 += row[p + 768]; }
    for (; p < P; p += 256) a0 += row[p];
    const float s = mp::wave_sum_f32((a0 + a1) + (a2 + a3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = (red[0] + red[1]) + (red[2] + red[3]);
        if (point_mean) t = t / (float)lengths[n];
        per_cloud[n] = t * scale;
        if (counter) {
            __threadfence();                                            // the value above is visible before the count
            last = atomicAdd(counter, 1u) == (unsigned)(N - 1);
        }
    }
    if (!counter) return;
    __syncthreads();
    if (last && threadIdx.x < 64) {
        float s = 0.0f;
        for (int c = threadIdx.x; c < N; c += 64) s += __hip_atomic_load(per_cloud + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = mp::wave_sum_f32(s);
        if (threadIdx.x == 0) {
            out[0] = (batch_mode == 2 ? s / div : s) + (add_to ? add_to[0] : 0.0f);
            *counter = 0u;                                              // ready for the next launch on this stream
        }
    }
}

// clouds combined in index order by one wave (lane l takes clouds l, l+64, ...; lanes then summed in a fixed tree)
__global__ __launch_bounds__(64) void chamfer_batch_kernel(const float* __restrict__ per_cloud, int N, int batch_mode, float div,
                                                           float* __restrict__ out, const float* __restrict__ add_to)
{
    float s = 0.0f;
    for (int n = threadIdx.x; n < N; n += 64) s += per_cloud[n];
    s = mp::wave_sum_f32(s);
    // add_to: a running total of loss terms (device scalar) the reduced value is added to -- the composite losses chain their
    // terms through it instead of launching one elementwise add per term
    if (threadIdx.x == 0) out[0] = (batch_mode == 2 ? s / div : s) + (add_to ? add_to[0] : 0.0f);
}

// grad_cham[n,p] = g(n) * scale / (len_n if point mean) / (div if batch mean) for p < len_n (when lengths are given), else 0
__global__ __launch_bounds__(256) void chamfer_reduce_bwd_kernel(const float* __restrict__ grad_out,
                                                                 const int64_t* __restrict__ lengths, int P,
                                                                 int point_mean, int batch_mode, float div, float scale,
                                                                 float* __restrict__ grad_cham)
{
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float g = (batch_mode == 0 ? grad_out[n] : grad_out[0]) * scale;
    if (batch_mode == 2) g = g / div;
    int64_t len = P;
    if (lengths) len = lengths[n];
    if (point_mean) g = g / (float)len;
    grad_cham[(size_t)n * P + p] = p < len ? g : 0.0f;
}

}  // namespace

static int chamfer_reduce(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean, int batch_mode, double div,
                          double scale, float* per_cloud, float* out, const float* add_to, unsigned* counter, mp_stream_t stream_)
{
    if (N < 0 || P < 0 || batch_mode < 0 || batch_mode > 2) return MP_EINVAL;
    if (N == 0) return MP_OK;
    if (!out || (P > 0 && !cham) || (point_mean && !lengths) || (batch_mode != 0 && !per_cloud)) return MP_EINVAL;
    if (N > 65535 * 64 || P > (1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    MP_LAUNCH("chamfer_rows_kernel", (double)(N * P), 4.0 * (double)(N * P), chamfer_rows_kernel, dim3((unsigned)N), dim3(256), 0, stream, cham,
              lengths, (int)P, point_mean, (float)scale, batch_mode == 0 ? out : per_cloud, (batch_mode != 0 ? counter : nullptr), (int)N,
              batch_mode, (float)div, out, add_to);
    MP_CHECK_LAUNCH();
    if (batch_mode != 0 && !counter) {
        hipLaunchKernelGGL(chamfer_batch_kernel, dim3(1), dim3(64), 0, stream, per_cloud, (int)N, batch_mode, (float)div, out, add_to);
        MP_CHECK_LAUNCH();
    }
    return MP_OK;
}

extern "C" int mp_chamfer_reduce_f32(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                                     int batch_mode, double div, double scale, float* per_cloud, float* out, const float* add_to, mp_stream_t stream_)
{
    return chamfer_reduce(cham, lengths, N, P, point_mean, batch_mode, div, scale, per_cloud, out, add_to, nullptr, stream_);
}

// The same reduction in ONE launch: `counter` is a device uint32 that is zero on entry and zero again on exit (the last workgroup to
// finish combines the clouds, in cloud order: the same bits as the two-launch form).  One counter per stream that runs reductions.
extern "C" int mp_chamfer_reduce1_f32(const float* cham, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                                      int batch_mode, double div, double scale, float* per_cloud, float* out, const float* add_to,
                                      uint32_t* counter, mp_stream_t stream_)
{
    if (batch_mode != 0 && !counter) return MP_EINVAL;
    return chamfer_reduce(cham, lengths, N, P, point_mean, batch_mode, div, scale, per_cloud, out, add_to, counter, stream_);
}

extern "C" int mp_chamfer_reduce_bwd_f32(const float* grad_out, const int64_t* lengths, int64_t N, int64_t P, int point_mean,
                                         int batch_mode, double div, double scale, float* grad_cham, mp_stream_t stream_)
{
    if (N < 0 || P < 0 || batch_mode < 0 || batch_mode > 2) return MP_EINVAL;
    if (N * P == 0) return MP_OK;
    if (!grad_out || !grad_cham || (point_mean && !lengths)) return MP_EINVAL;
    if (N > 65535 || P > (1 << 30)) return MP_EUNSUPPORTED;
    MP_LAUNCH("chamfer_reduce_bwd_kernel", (double)(N * P), 4.0 * (double)(N * P), chamfer_reduce_bwd_kernel,
              dim3((unsigned)((P + 255) / 256), (unsigned)N), dim3(256), 0, mp_stream(stream_), grad_out, lengths, (int)P, point_mean,
              batch_mode, (float)div, (float)scale, grad_cham);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
