// Row gather (index_points), grouping (gather + centre + concat) and their scatter-add backwards.
//
// Reference: models/pointnet2_utils.py:45-62 (index_points: advanced-index gather; backward = index_put
// accumulate), :133-143 (sample_and_group tail) and :258-262 (MSG variant, features first).
// These are HBM-bound copy kernels: lanes run along the channel axis of one row so that every wave
// instruction reads / writes one contiguous segment.  Rows repeat (ball query pads with its first hit), so
// the backward is a scatter-add with guaranteed collisions: the default uses global float atomics, the
// `deterministic` variant gives each destination row to one wave that scans the index list in order and
// adds the matching source rows in a fixed order (bitwise reproducible, no atomics).
#include "common.h"

namespace {

// out[b,m,:] = points[b, idx[b,m], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ points,
                                                          const int64_t* __restrict__ idx, int64_t N, int64_t C,
                                                          int64_t M, int64_t total, float* __restrict__ out)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / C;
        const int64_t c = e - row * C;
        const int64_t b = row / M;
        int64_t i = idx[row];
        i = i < 0 ? 0 : (i >= N ? N - 1 : i);
        out[e] = points[(b * N + i) * C + c];
    }
}

__global__ __launch_bounds__(256) void scatter_rows_atomic_kernel(const float* __restrict__ grad_out,
                                                                  const int64_t* __restrict__ idx, int64_t N,
                                                                  int64_t C, int64_t M, int64_t total,
                                                                  float* __restrict__ grad_points)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / C;
        const int64_t c = e - row * C;
        const int64_t b = row / M;
        const int64_t i = idx[row];
        if (i >= 0 && i < N) atomicAdd(grad_points + (b * N + i) * C + c, grad_out[e]);
    }
}

// Deterministic scatter: one wave per destination row (b, n).  Source rows are [M] per batch with row
// stride `src_stride` floats and the wanted channels at offset `src_off`; lanes scan idx 64 at a time.
__global__ __launch_bounds__(256) void scatter_rows_ordered_kernel(const float* __restrict__ grad_out,
                                                                   const int64_t* __restrict__ idx, int64_t N,
                                                                   int64_t C, int64_t M, int64_t src_stride,
                                                                   int64_t src_off, int64_t rows_total,
                                                                   float* __restrict__ grad_points)
{
    const int lane = threadIdx.x & 63;
    const int64_t dest = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (dest >= rows_total) return;
    const int64_t b = dest / N;
    const int64_t n = dest - b * N;
    const int64_t* bi = idx + b * M;
    const float* src = grad_out + b * M * src_stride + src_off;
    for (int64_t c0 = 0; c0 < C; c0 += 64) {
        const int64_t c = c0 + lane;
        float acc = 0.0f;
        for (int64_t m0 = 0; m0 < M; m0 += 64) {
            const int64_t m = m0 + lane;
            unsigned long long hit = __ballot(m < M && bi[m] == n);
            while (hit) {
                const int j = __builtin_ctzll(hit);
                hit &= hit - 1;
                if (c < C) acc += src[(m0 + j) * src_stride + c];
            }
        }
        if (c < C) grad_points[dest * C + c] = acc;
    }
}

// out[b,s,k,:] = cat(xyz[idx]-new_xyz, feats[idx])  (or feats first when xyz_last)
__global__ __launch_bounds__(256) void group_kernel(const float* __restrict__ xyz, const float* __restrict__ feats,
                                                    const float* __restrict__ new_xyz,
                                                    const int64_t* __restrict__ idx, int64_t N, int64_t S, int64_t K,
                                                    int64_t D, int xyz_last, int64_t Cs, int64_t total,
                                                    float* __restrict__ out)
{
    const int64_t C = D + 3;  // logical channels; rows are Cs >= C floats apart, the tail zero-filled
    const int64_t xoff = xyz_last ? D : 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / Cs;  // (b*S + s)*K + k
        const int64_t c = e - row * Cs;
        if (c >= C) { out[e] = 0.0f; continue; }
        const int64_t bs = row / K;
        const int64_t b = bs / S;
        int64_t i = idx[row];
        i = i < 0 ? 0 : (i >= N ? N - 1 : i);
        const int64_t cx = c - xoff;
        float v;
        if (cx >= 0 && cx < 3)
            v = xyz[(b * N + i) * 3 + cx] - new_xyz[bs * 3 + cx];
        else
            v = feats[(b * N + i) * D + (xyz_last ? c : c - 3)];
        out[e] = v;
    }
}

__global__ __launch_bounds__(256) void group_bwd_atomic_kernel(const float* __restrict__ grad_out,
                                                               const int64_t* __restrict__ idx, int64_t N, int64_t SK,
                                                               int64_t D, int xyz_last, int64_t Cs, int64_t total,
                                                               float* __restrict__ grad_feats)
{
    const int64_t C = Cs;
    const int64_t foff = xyz_last ? 0 : 3;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / D;
        const int64_t c = e - row * D;
        const int64_t b = row / SK;
        const int64_t i = idx[row];
        if (i >= 0 && i < N) atomicAdd(grad_feats + (b * N + i) * D + c, grad_out[row * C + foff + c]);
    }
}

inline unsigned grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

// dst[r, c] = src[r, perm[c]] (perm[c] < 0: zero).  Used to hand the first set-abstraction weight to the fused MLP in its
// internal column order (features first, xyz last, padded to a multiple of 4) in ONE launch instead of cat + pad; the
// backward is the same kernel with the inverse permutation.
__global__ __launch_bounds__(256) void permute_cols_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm,
                                                           int Cs, int Cd, int64_t total, float* __restrict__ dst)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int64_t r = e / Cd;
    const int c = (int)(e - r * Cd);
    const int j = perm[c];
    dst[e] = j >= 0 ? src[r * Cs + j] : 0.0f;
}

extern "C" int mp_permute_cols_f32(const float* src, const int32_t* perm, int64_t R, int64_t Cs, int64_t Cd, float* dst,
                                   mp_stream_t stream_)
{
    if (R < 0 || Cs < 0 || Cd < 0) return MP_EINVAL;
    const int64_t total = R * Cd;
    if (total == 0) return MP_OK;
    if (!src || !perm || !dst) return MP_EINVAL;
    if (Cs > (1 << 30) || Cd > (1 << 30)) return MP_EUNSUPPORTED;
    hipLaunchKernelGGL(permute_cols_kernel, dim3(grid_for(total)), dim3(256), 0, mp_stream(stream_), src, perm, (int)Cs, (int)Cd,
                       total, dst);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_index_points_f32(const float* points, const int64_t* idx, int64_t B, int64_t N, int64_t C,
                                   int64_t M, float* out, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || C < 0 || M < 0) return MP_EINVAL;
    const int64_t total = B * M * C;
    if (total == 0) return MP_OK;
    if (!points || !idx || !out || N == 0) return MP_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, mp_stream(stream_), points, idx, N, C,
                       M, total, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_index_points_bwd_f32(const float* grad_out, const int64_t* idx, int64_t B, int64_t N, int64_t C,
                                       int64_t M, float* grad_points, int deterministic, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || C < 0 || M < 0) return MP_EINVAL;
    if (B * N * C == 0) return MP_OK;
    if (!grad_points || (B * M * C > 0 && (!grad_out || !idx))) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    if (deterministic) {
        const int64_t rows = B * N;
        hipLaunchKernelGGL(scatter_rows_ordered_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
                           grad_out, idx, N, C, M, C, (int64_t)0, rows, grad_points);
    } else {
        if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)(B * N * C), stream) != hipSuccess) return MP_ELAUNCH;
        const int64_t total = B * M * C;
        if (total > 0)
            hipLaunchKernelGGL(scatter_rows_atomic_kernel, dim3(grid_for(total)), dim3(256), 0, stream, grad_out, idx,
                               N, C, M, total, grad_points);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_group_f32(const float* xyz, const float* feats, const float* new_xyz, const int64_t* idx,
                            int64_t B, int64_t N, int64_t S, int64_t K, int64_t D, int xyz_last, int64_t out_stride,
                            float* out, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || S < 0 || K < 0 || D < 0) return MP_EINVAL;
    if (out_stride == 0) out_stride = D + 3;
    if (out_stride < D + 3) return MP_EINVAL;
    const int64_t total = B * S * K * out_stride;
    if (total == 0) return MP_OK;
    if (!xyz || !new_xyz || !idx || !out || (D > 0 && !feats) || N == 0) return MP_EINVAL;
    MP_LAUNCH("group_kernel", 0.0, 4.0 * (double)total + 8.0 * (double)(B * S * K) + 4.0 * (double)(B * N * (D + 3)), group_kernel,
              dim3(grid_for(total)), dim3(256), 0, mp_stream(stream_), xyz, feats, new_xyz, idx, N, S, K, D, xyz_last, out_stride, total, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_group_bwd_f32(const float* grad_out, const int64_t* idx, int64_t B, int64_t N, int64_t S,
                                int64_t K, int64_t D, int xyz_last, int64_t grad_stride, float* grad_feats,
                                int deterministic, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || S < 0 || K < 0 || D < 0) return MP_EINVAL;
    if (grad_stride == 0) grad_stride = D + 3;
    if (grad_stride < D + 3) return MP_EINVAL;
    if (B * N * D == 0) return MP_OK;
    if (!grad_feats || (B * S * K > 0 && (!grad_out || !idx))) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    if (deterministic) {
        const int64_t rows = B * N;
        hipLaunchKernelGGL(scatter_rows_ordered_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
                           grad_out, idx, N, D, S * K, grad_stride, (int64_t)(xyz_last ? 0 : 3), rows, grad_feats);
    } else {
        if (hipMemsetAsync(grad_feats, 0, sizeof(float) * (size_t)(B * N * D), stream) != hipSuccess) return MP_ELAUNCH;
        const int64_t total = B * S * K * D;
        if (total > 0)
            MP_LAUNCH("group_bwd_atomic_kernel", 0.0, 8.0 * (double)total + 8.0 * (double)(B * S * K), group_bwd_atomic_kernel,
                      dim3(grid_for(total)), dim3(256), 0, stream, grad_out, idx, N, S * K, D, xyz_last, grad_stride, total, grad_feats);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}
