// Feature propagation, interpolation half (gfx950): 3 nearest neighbours + inverse-distance interpolation.
//
// Reference: models/pointnet2_utils.py:305-317 (PointNetFeaturePropagation.forward) materialises the [B,N,S]
// all-pairs matrix (square_distance, :21-42), SORTS every row and keeps 3 columns, then gathers [B,N,3,D] rows
// and reduces them.  Here: the S source points are staged through LDS once per workgroup, each thread keeps the
// running 3 smallest distances of its query (lowest index first on ties), and the interpolation reads the three
// source rows directly.
//
// Arithmetic: the distance is the reference's expanded form, bit-exact (same helper as the ball query):
//   d = ((-2*dot) + |q|^2) + |p|^2, dot = fma(qz,pz, fma(qy,py, qx*px)); it goes slightly negative for coincident
// points, and 1/(d + 1e-8) is then huge or negative exactly as in the reference.  weight_k = r_k / ((r0+r1)+r2),
// out = (p0*w0 + p1*w1) + p2*w2 with separately rounded products (the build uses -ffp-contract=off).
#include "common.h"

namespace {

__device__ __forceinline__ float norm3(float x, float y, float z) { return (x * x + y * y) + z * z; }

constexpr int NN_CHUNK = 1024;   // source points per LDS stage: 4 x 4 KB

__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                       int N, int S, float* __restrict__ dist,
                                                       int64_t* __restrict__ idx, float* __restrict__ weight)
{
    __shared__ float sx[NN_CHUNK], sy[NN_CHUNK], sz[NN_CHUNK], sn[NN_CHUNK];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool live = n < N;
    const float* q = xyz1 + ((size_t)b * N + (live ? n : 0)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float qn = norm3(qx, qy, qz);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    const float* p = xyz2 + (size_t)b * S * 3;
    for (int base = 0; base < S; base += NN_CHUNK) {
        const int cnt = min(NN_CHUNK, S - base);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) {
            const float x = p[(size_t)(base + j) * 3], y = p[(size_t)(base + j) * 3 + 1], z = p[(size_t)(base + j) * 3 + 2];
            sx[j] = x; sy[j] = y; sz[j] = z; sn[j] = norm3(x, y, z);
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {   // every lane reads the same LDS word: broadcast, conflict-free
            const float dot = __builtin_fmaf(qz, sz[j], __builtin_fmaf(qy, sy[j], qx * sx[j]));
            const float d = ((-2.0f * dot) + qn) + sn[j];
            if (d < d2) {
                const int s = base + j;
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = s; }
                    else { d1 = d; i1 = s; }
                } else { d2 = d; i2 = s; }
            }
        }
    }
    if (!live) return;
    const size_t o = ((size_t)b * N + n) * 3;
    idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
    if (dist) { dist[o] = d0; dist[o + 1] = d1; dist[o + 2] = d2; }
    if (weight) {
        const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
        const float nrm = (r0 + r1) + r2;
        weight[o] = r0 / nrm; weight[o + 1] = r1 / nrm; weight[o + 2] = r2 / nrm;
    }
}

// out[b,n,:] = (p2[b,i0,:]*w0 + p2[b,i1,:]*w1) + p2[b,i2,:]*w2 ; one thread per output element, channels fastest.
__global__ __launch_bounds__(256) void three_interp_kernel(const float* __restrict__ points2, const int64_t* __restrict__ idx,
                                                           const float* __restrict__ weight, int N, int S, int D,
                                                           int64_t total, float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int64_t row = e / D;           // b*N + n
    const int d = (int)(e - row * D);
    const int64_t b = row / N;
    const int64_t* ip = idx + row * 3;
    const float* wp = weight + row * 3;
    const float* src = points2 + b * S * D + d;
    out[e] = (src[ip[0] * D] * wp[0] + src[ip[1] * D] * wp[1]) + src[ip[2] * D] * wp[2];
}

__global__ __launch_bounds__(256) void three_interp_bwd_atomic_kernel(const float* __restrict__ grad_out,
                                                                      const int64_t* __restrict__ idx,
                                                                      const float* __restrict__ weight, int N, int S, int D,
                                                                      int64_t total, float* __restrict__ grad_points2)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int64_t row = e / D;
    const int d = (int)(e - row * D);
    const int64_t b = row / N;
    const float g = grad_out[e];
    float* dst = grad_points2 + b * S * D + d;
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(dst + idx[row * 3 + k] * D, g * weight[row * 3 + k]);
}

// Deterministic variant: one workgroup per source row (b, s).  The B*N*3 index entries of cloud b are scanned in
// order, 256 per step; a ballot turns the matches of a step into an ordered list and every thread (one channel each)
// walks it, so each element is summed in ascending (n, k) order -- the order of the CPU oracle, bit for bit.
__global__ __launch_bounds__(256) void three_interp_bwd_ordered_kernel(const float* __restrict__ grad_out,
                                                                       const int64_t* __restrict__ idx,
                                                                       const float* __restrict__ weight, int N, int S, int D,
                                                                       float* __restrict__ grad_points2)
{
    __shared__ int hit[256];
    __shared__ int nhit_w[4];
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t* ib = idx + (size_t)b * N * 3;
    const float* wb = weight + (size_t)b * N * 3;
    const float* gb = grad_out + (size_t)b * N * D;
    float* dst = grad_points2 + ((size_t)b * S + s) * D;
    constexpr int DPT = 4;                   // channels per thread per pass: 1024 channels per pass
    for (int dbase = 0; dbase < D; dbase += 256 * DPT) {
        float acc[DPT];
#pragma unroll
        for (int u = 0; u < DPT; ++u) acc[u] = 0.0f;
        for (int e0 = 0; e0 < 3 * N; e0 += 256) {
            const int e = e0 + tid;
            const bool m = e < 3 * N && ib[e] == s;
            const unsigned long long bal = __ballot(m);
            if (lane == 0) nhit_w[wave] = __popcll(bal);
            __syncthreads();
            int off = 0;
            for (int w = 0; w < wave; ++w) off += nhit_w[w];
            const int total = nhit_w[0] + nhit_w[1] + nhit_w[2] + nhit_w[3];
            if (m) hit[off + mp::prefix_popc(bal)] = e;
            __syncthreads();
            for (int h = 0; h < total; ++h) {
                const int en = hit[h];
                const float wv = wb[en];
                const float* gr = gb + (size_t)(en / 3) * D;
#pragma unroll
                for (int u = 0; u < DPT; ++u) {
                    const int d = dbase + u * 256 + tid;
                    if (d < D) acc[u] += gr[d] * wv;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int d = dbase + u * 256 + tid;
            if (d < D) dst[d] = acc[u];
        }
    }
}

inline unsigned grid_for(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

extern "C" int mp_three_nn_f32(const float* xyz1, const float* xyz2, int64_t B, int64_t N, int64_t S, float* dist,
                               int64_t* idx, float* weight, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || S < 0) return MP_EINVAL;
    if (B * N == 0) return MP_OK;
    if (S < 3) return MP_EINVAL;   // S == 1 is the reference's repeat branch (host side), S == 2 has no third neighbour
    if (!xyz1 || !xyz2 || !idx) return MP_EINVAL;
    if (N > (int64_t)1 << 30 || S > (int64_t)1 << 30 || B > 65535) return MP_EUNSUPPORTED;
    MP_LAUNCH("three_nn_kernel", 8.0 * (double)(B * N * S), 12.0 * (double)(B * (N + S)) + 48.0 * (double)(B * N), three_nn_kernel,
              dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, mp_stream(stream_), xyz1, xyz2, (int)N, (int)S, dist, idx,
              weight);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_three_interpolate_f32(const float* points2, const int64_t* idx, const float* weight, int64_t B, int64_t N,
                                        int64_t S, int64_t D, float* out, mp_stream_t stream_)
{
    if (B < 0 || N < 0 || S < 0 || D < 0) return MP_EINVAL;
    const int64_t total = B * N * D;
    if (total == 0) return MP_OK;
    if (!points2 || !idx || !weight || !out || S == 0) return MP_EINVAL;
    if (N > (int64_t)1 << 30 || S > (int64_t)1 << 30 || D > (int64_t)1 << 30) return MP_EUNSUPPORTED;
    MP_LAUNCH("three_interp_kernel", 5.0 * (double)total, 16.0 * (double)total + 4.0 * (double)(B * S * D), three_interp_kernel,
              dim3(grid_for(total)), dim3(256), 0, mp_stream(stream_), points2, idx, weight, (int)N, (int)S, (int)D, total, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_three_interpolate_bwd_f32(const float* grad_out, const int64_t* idx, const float* weight, int64_t B,
                                            int64_t N, int64_t S, int64_t D, float* grad_points2, int deterministic,
                                            mp_stream_t stream_)
{
    if (B < 0 || N < 0 || S < 0 || D < 0) return MP_EINVAL;
    if (B * S * D == 0) return MP_OK;
    if (!grad_points2 || (B * N * D > 0 && (!grad_out || !idx || !weight))) return MP_EINVAL;
    if (N > (int64_t)1 << 29 || S > (int64_t)1 << 30 || D > (int64_t)1 << 30 || B * S > (int64_t)1 << 31) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    if (deterministic) {
        hipLaunchKernelGGL(three_interp_bwd_ordered_kernel, dim3((unsigned)(B * S)), dim3(256), 0, stream, grad_out, idx, weight, (int)N,
                           (int)S, (int)D, grad_points2);
    } else {
        if (!mp::zero_async(grad_points2, (size_t)(B * S * D), stream)) return MP_ELAUNCH;
        const int64_t total = B * N * D;
        if (total > 0)
            MP_LAUNCH("three_interp_bwd_atomic_kernel", 6.0 * (double)total, 16.0 * (double)total, three_interp_bwd_atomic_kernel,
                      dim3(grid_for(total)), dim3(256), 0, stream, grad_out, idx, weight, (int)N, (int)S, (int)D, total, grad_points2);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}
