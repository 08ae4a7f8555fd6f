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

// No features (the first level groups coordinates only): one thread per output row [dx, dy, dz, 0], 32-bit index arithmetic, one
// 16-byte store.  (The generic kernel spends three 64-bit divides and a 4-byte access per ELEMENT: 30 us for 19 MB.)
__global__ __launch_bounds__(256) void group_xyz_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                        const int64_t* __restrict__ idx, int N, int S, int K, int rows,
                                                        float4* __restrict__ out)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int bs = row / K, b = bs / S;
    const int64_t i64 = idx[row];
    const int i = (int)(i64 < 0 ? 0 : (i64 >= N ? N - 1 : i64));
    const float* p = xyz + ((size_t)b * N + i) * 3;
    const float* c = new_xyz + (size_t)bs * 3;
    out[row] = make_float4(p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.0f);
}

// Fast path of group_kernel for the internal layout of the set-abstraction modules: features first (D % 4 == 0), then the
// centred xyz and one zero pad column (row stride D + 4).  One wave per output row, a float4 per lane: the source row is read
// and the output row written as whole 16-byte pieces (the generic kernel does a 64-bit divide and a 4-byte access per
// element); 32-bit index arithmetic.
__global__ __launch_bounds__(256) void group_rows4_kernel(const float* __restrict__ xyz, const float* __restrict__ feats,
                                                          const float* __restrict__ new_xyz, const int64_t* __restrict__ idx,
                                                          int N, int S, int K, int D, int rows, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int nq = D / 4 + 1;                       // float4 pieces per row
    const int Cs = D + 4;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const int bs = row / K, b = bs / S;
        int64_t i64 = idx[row];
        int i = (int)(i64 < 0 ? 0 : (i64 >= N ? N - 1 : i64));
        const size_t src = (size_t)b * N + i;
        for (int q = lane; q < nq; q += 64) {
            float4 v;
            if (q < D / 4) {
                v = *reinterpret_cast<const float4*>(feats + src * D + 4 * q);
            } else {
                const float* p = xyz + src * 3;
                const float* c = new_xyz + (size_t)bs * 3;
                v = make_float4(p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.0f);
            }
            *reinterpret_cast<float4*>(out + (size_t)row * Cs + 4 * q) = v;
        }
    }
}

// The same for D = 128 (set abstraction 2): a 512-byte feature row is exactly 32 lanes of float4, so a wave takes TWO rows per pass and
// four passes per iteration -- eight rows whose index loads, then feature loads, then stores are issued together (the one-row kernel
// walks a dependent index -> row -> store chain per row with half its lanes idle).  The coordinate quad of row u of a half-wave is
// written by lane u of that half.
__global__ __launch_bounds__(256) void group_rows128_kernel(const float* __restrict__ xyz, const float* __restrict__ feats,
                                                            const float* __restrict__ new_xyz, const int64_t* __restrict__ idx,
                                                            int N, int S, int K, int rows, float* __restrict__ out)
{
    constexpr int D = 128, Cs = D + 4;
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    for (int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; row0 < rows; row0 += gridDim.x * 32) {
        int r[4];
        size_t src[4];
        int bs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            r[u] = row0 + 2 * u + half;
            const int rr = r[u] < rows ? r[u] : rows - 1;
            bs[u] = rr / K;
            const int64_t i64 = idx[rr];
            const int i = (int)(i64 < 0 ? 0 : (i64 >= N ? N - 1 : i64));
            src[u] = (size_t)(bs[u] / S) * N + i;
        }
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(feats + src[u] * D + 4 * l);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        int tr = -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (l == u) {
                const float* p = xyz + src[u] * 3;
                const float* c = new_xyz + (size_t)bs[u] * 3;
                t = make_float4(p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.0f);
                tr = r[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r[u] < rows) *reinterpret_cast<float4*>(out + (size_t)r[u] * Cs + 4 * l) = v[u];
        if (tr >= 0 && tr < rows) *reinterpret_cast<float4*>(out + (size_t)tr * Cs + D) = t;
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

// Backward of the grouping gather without global atomics, for the internal layout (features first, D % 4 == 0, row stride
// D + 4).  A workgroup owns GP consecutive source points of one cloud: it scans the cloud's S*K neighbour indices once,
// collecting for each of its points the rows that gathered it (LDS lists of GCAP rows per point plus a shared overflow list;
// beyond both the workgroup falls back to atomics), then every wave sums the gradient rows of its points with float4 loads -- each 512-byte gradient row is
// read once, coalesced, and every destination row is written once.  (Summation order follows the LDS list order, i.e. it
// is not fixed: this is the non-deterministic variant's replacement; the ordered kernel above stays for deterministic runs.)
constexpr int GP = 16, GCAP = 1024, GOVF = 2048;
// Row widths beyond one wave's reach (D / 4 > 64 float4 pieces: the 320 concatenated features of a multi-scale level) are
// cut into column slabs of `slab` floats along gridDim.z (the last one may be narrower); every slab's workgroup repeats the index
// scan.  Narrow slabs also spread a HOT point (ball-query padding repeats one index up to K times per group, so a few points
// collect thousands of rows and their wave sums them alone) over several workgroups: measured at D = 320, one workgroup looping
// over the slabs from one scan 577 us, slabs of 128 along the grid 537 us.
__global__ __launch_bounds__(256) void group_bwd_gather_kernel(const float* __restrict__ grad_out, const int64_t* __restrict__ idx,
                                                               int N, int M, int D, int slab, float* __restrict__ grad_feats)
{
    __shared__ int cnt[GP];
    __shared__ int novf;
    __shared__ unsigned short lists[GP][GCAP];      // 32 KB
    __shared__ unsigned short ovf_m[GOVF];          // rows beyond a full list (ball-query padding repeats one index up to K
    __shared__ unsigned char ovf_r[GOVF];           // times per group, so a few points collect hundreds of rows)
    const int b = blockIdx.y, n0 = blockIdx.x * GP;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Cs = D + 4;
    const int c0 = blockIdx.z * slab, Dw = min(slab, D - c0);      // this workgroup's columns [c0, c0 + Dw)
    const int64_t* bi = idx + (size_t)b * M;
    const float* gb = grad_out + (size_t)b * M * Cs + c0;
    float* dst = grad_feats + ((size_t)b * N + n0) * D + c0;
    const int npts = min(GP, N - n0);
    if (tid < GP) cnt[tid] = 0;
    if (tid == 0) novf = 0;
    __syncthreads();
    // the scan: eight index loads in flight per thread (one load, one compare and a rare LDS atomic per index is otherwise a chain of
    // L2 latencies -- 32 of them per thread at M = 8192)
    for (int m0 = tid; m0 < M; m0 += 8 * 256) {
        int64_t iv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) iv[u] = (m0 + u * 256 < M) ? bi[m0 + u * 256] : (int64_t)-1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = iv[u];
            const int m = m0 + u * 256;
            const int r = (int)(i - n0);
            if (i >= n0 && r < npts) {
                const int slot = atomicAdd(&cnt[r], 1);
                if (slot < GCAP) lists[r][slot] = (unsigned short)m;
                else {
                    const int o = atomicAdd(&novf, 1);
                    if (o < GOVF) { ovf_m[o] = (unsigned short)m; ovf_r[o] = (unsigned char)r; }
                }
            }
        }
    }
    __syncthreads();
    if (novf > GOVF) {
        // pathological skew (more than GCAP + GOVF rows on this workgroup's points): plain atomics for the whole workgroup
        for (int e = tid; e < npts * Dw; e += 256) dst[(size_t)(e / Dw) * D + e % Dw] = 0.0f;
        __syncthreads();
        for (int m = tid; m < M; m += 256) {
            const int64_t i = bi[m];
            const int r = (int)(i - n0);
            if (i >= n0 && r < npts) {
                const float* g = gb + (size_t)m * Cs;
                for (int c = 0; c < Dw; ++c) atomicAdd(dst + (size_t)r * D + c, g[c]);
            }
        }
        return;
    }
    const int no = novf;
    const int q = Dw / 4;                      // float4 pieces per row of the slab (<= 64)
    const int rpw = 64 / q;                    // rows a wave reads at once (2 for D = 128)
    const int sub = lane / q, ql = lane - sub * q;
    for (int r = wave; r < npts; r += 4) {
        const int n = min(cnt[r], GCAP);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        {   // eight row loads in flight per lane group: a hot point (hundreds of rows) is otherwise one latency per row
            int j = sub;
            for (; j + 7 * rpw < n; j += 8 * rpw) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(gb + (size_t)lists[r][j + u * rpw] * Cs + 4 * ql);
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; j < n; j += rpw) {
                const float4 v = *reinterpret_cast<const float4*>(gb + (size_t)lists[r][j] * Cs + 4 * ql);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        // rows that did not fit the list (rare), still without global atomics
        if (cnt[r] > GCAP)
            for (int o = sub; o < no; o += rpw)
                if (ovf_r[o] == r) {
                    const float4 v = *reinterpret_cast<const float4*>(gb + (size_t)ovf_m[o] * Cs + 4 * ql);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
        // combine the row slots of the wave (lanes ql, ql + q, ...): only rpw in {1, 2, 4} are dispatched
        if (rpw >= 2) {
            acc.x += __shfl_down(acc.x, q, 64); acc.y += __shfl_down(acc.y, q, 64);
            acc.z += __shfl_down(acc.z, q, 64); acc.w += __shfl_down(acc.w, q, 64);
        }
        if (rpw == 4) {
            acc.x += __shfl_down(acc.x, 2 * q, 64); acc.y += __shfl_down(acc.y, 2 * q, 64);
            acc.z += __shfl_down(acc.z, 2 * q, 64); acc.w += __shfl_down(acc.w, 2 * q, 64);
        }
        if (lane < q) reinterpret_cast<float4*>(dst + (size_t)r * D)[ql] = acc;
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

// Several column permutations in one launch (the first-layer weights of all set-abstraction levels at the start of the forward, their
// gradients at the end of the backward): blockIdx.y = tensor, pointers and shapes travel in the kernel arguments.
namespace {
constexpr int PERM_MAX = 8;
struct PermTable {
    const float* src[PERM_MAX];
    const int32_t* perm[PERM_MAX];
    float* dst[PERM_MAX];
    int Cs[PERM_MAX], Cd[PERM_MAX];
    long long total[PERM_MAX];
};
__global__ __launch_bounds__(256) void permute_cols_multi_kernel(PermTable t)
{
    const int k = blockIdx.y;
    const int Cd = t.Cd[k], Cs = t.Cs[k];
    const int32_t* perm = t.perm[k];
    const float* src = t.src[k];
    float* dst = t.dst[k];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < t.total[k]; e += (long long)gridDim.x * 256) {
        const long long r = e / Cd;
        const int c = (int)(e - r * Cd);
        const int j = perm[c];
        dst[e] = j >= 0 ? src[r * Cs + j] : 0.0f;
    }
}
}  // namespace

extern "C" int mp_permute_cols_multi_f32(int64_t count, const float* const* src, const int32_t* const* perm, const int64_t* R,
                                         const int64_t* Cs, const int64_t* Cd, float* const* dst, mp_stream_t stream_)
{
    if (count < 0 || count > PERM_MAX) return MP_EINVAL;
    if (count == 0) return MP_OK;
    if (!src || !perm || !R || !Cs || !Cd || !dst) return MP_EINVAL;
    PermTable t{};
    long long most = 0;
    for (int k = 0; k < (int)count; ++k) {
        if (R[k] < 0 || Cs[k] < 0 || Cd[k] < 0 || Cs[k] > (1 << 30) || Cd[k] > (1 << 30)) return MP_EINVAL;
        t.total[k] = (long long)(R[k] * Cd[k]);
        if (t.total[k] > 0 && (!src[k] || !perm[k] || !dst[k])) return MP_EINVAL;
        t.src[k] = src[k]; t.perm[k] = perm[k]; t.dst[k] = dst[k];
        t.Cs[k] = (int)Cs[k]; t.Cd[k] = Cd[k] > 0 ? (int)Cd[k] : 1;
        most = t.total[k] > most ? t.total[k] : most;
    }
    if (most == 0) return MP_OK;
    hipLaunchKernelGGL(permute_cols_multi_kernel, dim3(grid_for(most), (unsigned)count), dim3(256), 0, mp_stream(stream_), t);
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
        if (!mp::zero_async(grad_points, (size_t)(B * N * C), stream)) return MP_ELAUNCH;
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
    const double bytes = 4.0 * (double)total + 8.0 * (double)(B * S * K) + 4.0 * (double)(B * N * (D + 3));
    if (xyz_last && D > 0 && (D & 3) == 0 && out_stride == D + 4 && B * S * K < ((int64_t)1 << 31) && N < ((int64_t)1 << 31) &&
        (reinterpret_cast<uintptr_t>(feats) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int64_t rows = B * S * K;
        if (D == 128 && (reinterpret_cast<uintptr_t>(feats) & 15) == 0) {
            int64_t g8 = (rows + 31) / 32;
            if (g8 > 256 * 32) g8 = 256 * 32;
            MP_LAUNCH("group_kernel", 0.0, bytes, group_rows128_kernel, dim3((unsigned)g8), dim3(256), 0, mp_stream(stream_), xyz, feats, new_xyz, idx,
                      (int)N, (int)S, (int)K, (int)rows, out);
            MP_CHECK_LAUNCH();
            return MP_OK;
        }
        int64_t g = (rows + 3) / 4;
        if (g > 256 * 64) g = 256 * 64;
        MP_LAUNCH("group_kernel", 0.0, bytes, group_rows4_kernel, dim3((unsigned)g), dim3(256), 0, mp_stream(stream_), xyz, feats, new_xyz, idx,
                  (int)N, (int)S, (int)K, (int)D, (int)rows, out);
        MP_CHECK_LAUNCH();
        return MP_OK;
    }
    if (D == 0 && out_stride == 4 && B * S * K < ((int64_t)1 << 31) && N < ((int64_t)1 << 31) && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int64_t rows = B * S * K;
        MP_LAUNCH("group_kernel", 0.0, bytes, group_xyz_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, mp_stream(stream_), xyz, new_xyz, idx,
                  (int)N, (int)S, (int)K, (int)rows, reinterpret_cast<float4*>(out));
        MP_CHECK_LAUNCH();
        return MP_OK;
    }
    MP_LAUNCH("group_kernel", 0.0, bytes, group_kernel,
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
    } else if (xyz_last && D > 0 && (D == 64 || D == 128 || D == 256 || (D % 64 == 0 && D <= 1024)) && grad_stride == D + 4 && S * K < 65536 && B < 65536 &&
               N < ((int64_t)1 << 30) && (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(grad_feats) & 15) == 0) {
        // rpw = 64 / (D/4) in {4, 2, 1}: the shuffle combine of the gather kernel covers exactly these
        // one slab for the widths a wave covers by itself, else slabs of 128 columns (+ a 64-wide remainder): every slab width is in {64, 128, 256}
        const int slab = (D == 64 || D == 128 || D == 256) ? (int)D : 128;
        MP_LAUNCH("group_bwd_gather_kernel", 0.0, 4.0 * (double)(B * S * K) * (D + 2) + 4.0 * (double)(B * N * D), group_bwd_gather_kernel,
                  dim3((unsigned)((N + GP - 1) / GP), (unsigned)B, (unsigned)((D + slab - 1) / slab)), dim3(256), 0, stream, grad_out, idx, (int)N,
                  (int)(S * K), (int)D, slab, grad_feats);
    } else {
        if (!mp::zero_async(grad_feats, (size_t)(B * N * D), stream)) return MP_ELAUNCH;
        const int64_t total = B * S * K * D;
        if (total > 0)
            MP_LAUNCH("group_bwd_atomic_kernel", 0.0, 8.0 * (double)total + 8.0 * (double)(B * S * K), group_bwd_atomic_kernel,
                      dim3(grid_for(total)), dim3(256), 0, stream, grad_out, idx, N, S * K, D, xyz_last, grad_stride, total, grad_feats);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}
