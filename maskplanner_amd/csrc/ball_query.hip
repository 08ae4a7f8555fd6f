// Ball query + square_distance for gfx950.
//
// Reference: models/pointnet2_utils.py:89-109 (query_ball_point) builds a [B,S,N] int64 index tensor,
// masks it with the all-pairs distance matrix (:21-42) and SORTS it to take the first `nsample` in-ball
// indices.  Equivalent, and what runs here: scan the cloud in index order, keep the first K hits, stop.
//
// Design: the cloud (SoA, 12 B/point) is staged once per workgroup in LDS; one wave per query scans 64
// points per step: lane-parallel membership test, one ballot, a popcount prefix gives every hit its output
// slot in index order; the scan stops as soon as K hits are found (balls saturate on surface clouds).  The
// K slots are buffered per wave in LDS and written as one coalesced row, padding included.
//
// Arithmetic (bit-exact membership is the contract): d = ((-2*dot) + |q|^2) + |p|^2 with
// dot = fma(qz,pz, fma(qy,py, qx*px)), norms (x*x + y*y) + z*z, test !(d > (float)(radius*radius)).
// -2*dot is exact, so fma(-2, dot, |q|^2) rounds once exactly like the reference's separate add.
#include <cstdlib>

#include <cstdio>

#include "common.h"

namespace {

__device__ __forceinline__ float norm3(float x, float y, float z) { return (x * x + y * y) + z * z; }

__device__ __forceinline__ float sqdist_expanded(float qx, float qy, float qz, float qn, float x, float y, float z)
{
    const float dot = __builtin_fmaf(qz, z, __builtin_fmaf(qy, y, qx * x));
    return ((-2.0f * dot) + qn) + norm3(x, y, z);
}

constexpr int BQ_WAVES = 8;
constexpr int BQ_THREADS = BQ_WAVES * MP_WAVE;
constexpr int BQ_UNROLL = 4;

__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float* __restrict__ xyz,
                                                                const float* __restrict__ new_xyz, int N, int S,
                                                                float r2, int K, int qpb,
                                                                int64_t* __restrict__ out_idx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npad = (N + 255) & ~255;
    float* sx = reinterpret_cast<float*>(smem_raw);
    float* sy = sx + npad;
    float* sz = sy + npad;
    int* hits = reinterpret_cast<int*>(sz + npad);  // [BQ_WAVES][K]

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    for (int i = tid; i < 3 * N; i += BQ_THREADS) {
        const float v = p[i];
        const int pt = i / 3;
        const int c = i - 3 * pt;
        (c == 0 ? sx : (c == 1 ? sy : sz))[pt] = v;
    }
    // padding far outside any ball; it is also masked by the i < N test
    for (int i = N + tid; i < npad; i += BQ_THREADS) { sx[i] = 3.0e18f; sy[i] = 3.0e18f; sz[i] = 3.0e18f; }
    __syncthreads();

    int* my = hits + wave * K;
    const int q0 = blockIdx.x * qpb;
    const int q1 = min(S, q0 + qpb);
    for (int q = q0 + wave; q < q1; q += BQ_WAVES) {
        const float* qp = new_xyz + ((size_t)b * S + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const float qn = norm3(qx, qy, qz);
        int cnt = 0;
        for (int base = 0; base < N && cnt < K; base += 64 * BQ_UNROLL) {
            unsigned long long masks[BQ_UNROLL];
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                const int i = base + u * 64 + lane;  // < npad: the LDS image is padded to 256
                const float d = sqdist_expanded(qx, qy, qz, qn, sx[i], sy[i], sz[i]);
                masks[u] = __ballot(i < N && !(d > r2));
            }
#pragma unroll
            for (int u = 0; u < BQ_UNROLL; ++u) {
                const unsigned long long m = masks[u];
                if (m != 0ull && cnt < K) {
                    const int rank = cnt + mp::prefix_popc(m);
                    if (((m >> lane) & 1ull) && rank < K) my[rank] = base + u * 64 + lane;
                    cnt += __popcll(m);
                }
            }
        }
        cnt = min(cnt, K);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int first = cnt > 0 ? my[0] : N;
        int64_t* o = out_idx + ((size_t)b * S + q) * K;
        for (int k = lane; k < K; k += 64) o[k] = (int64_t)(k < cnt ? my[k] : first);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// [r5] Register-tiled scan: Q queries per wave and all radii of a multi-scale level in one pass.
//
// ball_query_kernel above reads every point from LDS once PER QUERY (three ds_read_b32 + the point's norm + the test: ~13 VALU and three LDS
// reads per pair -- ~1 GB of LDS reads per call at one query per wave).  Here a lane loads its point and forms |p|^2 ONCE per step and tests
// it against the wave's Q queries (six VALU per pair: the reference's expanded form, unchanged), and against every radius of a
// multi-scale level (models/pointnet2_utils.py:255-258 loops `query_ball_point` over the radius list on the same query / cloud pair: one
// distance, NR compares).  Hits are buffered per wave in LDS (rank = hits so far + popcount prefix: index order) and leave as coalesced
// rows with their padding (`first hit`, pointnet2_utils.py:106-108).  (First version: hit lanes stored straight to the output rows -- a
// partially active scattered 8-byte store costs the address unit a full pass per instruction: N = 10240, three radii: 297 us against 38 us
// of arithmetic.)  A wave's scan ends when every one of its Q x NR lists is full.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BQM_MAXR = 3;
struct BqRadii {
    float r2[BQM_MAXR];
    int K[BQM_MAXR];
    int64_t* out[BQM_MAXR];
};

template <int Q, int NR, int WAVES>
__global__ __launch_bounds__(WAVES * MP_WAVE) void ball_query_multi_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                                           int N, int S, BqRadii rr, int qpb)
{
    constexpr int T = WAVES * MP_WAVE, U = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npad = (N + 127) & ~127;
    float* sx = reinterpret_cast<float*>(smem_raw);
    float* sy = sx + npad;
    float* sz = sy + npad;
    int ksum = 0, koff[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { koff[r] = ksum; ksum += rr.K[r]; }
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* hits = reinterpret_cast<int*>(sz + npad) + wave * Q * ksum;      // [WAVES][Q][sum K]
    const float* p = xyz + (size_t)b * N * 3;
    // staging: the [N, 3] rows as 16-byte pieces, eight requests per thread in flight (a 4-byte load per trip of a dependent loop cost one
    // memory latency per trip: ~25 us of a 41 us launch at N = 5120), turned into the SoA image with one ds_write_b32 per float
    if (((3 * N) & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const int n4 = (3 * N) >> 2;
        for (int i0 = tid; i0 < n4; i0 += 8 * T) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p4[min(i0 + k * T, n4 - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i4 = i0 + k * T;
                if (i4 < n4) {
                    const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    const int f0 = 4 * i4, pt0 = f0 / 3, c0 = f0 - 3 * pt0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int c = c0 + t, pt = pt0 + (c >= 3 ? 1 : 0) + (c >= 6 ? 1 : 0), cc = c - (c >= 3 ? 3 : 0) - (c >= 6 ? 3 : 0);
                        (cc == 0 ? sx : (cc == 1 ? sy : sz))[pt] = e[t];
                    }
                }
            }
        }
    } else {
        for (int i = tid; i < 3 * N; i += T) {
            const float v = p[i];
            const int pt = i / 3;
            const int c = i - 3 * pt;
            (c == 0 ? sx : (c == 1 ? sy : sz))[pt] = v;
        }
    }
    for (int i = N + tid; i < npad; i += T) { sx[i] = 3.0e18f; sy[i] = 3.0e18f; sz[i] = 3.0e18f; }   // (also masked by i < N)
    __syncthreads();

    const int q0 = blockIdx.x * qpb;
    const int q1 = min(S, q0 + qpb);
    for (int qb = q0 + wave * Q; qb < q1; qb += WAVES * Q) {
        float qx[Q], qy[Q], qz[Q], qn[Q];
        int cnt[Q][NR];
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int q = min(qb + j, q1 - 1);       // (a ragged last tile repeats its last query; nothing is written for the repeats)
            const float* qp = new_xyz + ((size_t)b * S + q) * 3;
            qx[j] = qp[0]; qy[j] = qp[1]; qz[j] = qp[2];
            qn[j] = norm3(qx[j], qy[j], qz[j]);
#pragma unroll
            for (int r = 0; r < NR; ++r) cnt[j][r] = (qb + j < q1) ? 0 : rr.K[r];
        }
        // [second version] every mask is consumed where it is formed (24 live 64-bit masks + their counters spilled the scalar registers:
        // ~1 000 scalar instructions per 128 points).  The radii are ASCENDING (host): a (point group, query) pair without a hit in the
        // largest ball has none in the others -- one scalar test skips its NR lists.
        for (int base = 0; base < N; base += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * 64 + lane;       // < npad
                const float x = sx[i], y = sy[i], z = sz[i];
                const float pn = norm3(x, y, z);
                const bool inb = i < N;
#pragma unroll
                for (int j = 0; j < Q; ++j) {
                    const float dot = __builtin_fmaf(qz[j], z, __builtin_fmaf(qy[j], y, qx[j] * x));
                    const float d = ((-2.0f * dot) + qn[j]) + pn;
                    const unsigned long long mbig = __ballot(inb && !(d > rr.r2[NR - 1]));
                    if (mbig == 0ull) continue;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const unsigned long long mm = (r == NR - 1) ? mbig : __ballot(inb && !(d > rr.r2[r]));
                        const int K = rr.K[r];
                        if (mm != 0ull && cnt[j][r] < K) {
                            const int rank = cnt[j][r] + mp::prefix_popc(mm);
                            if (((mm >> lane) & 1ull) && rank < K) hits[j * ksum + koff[r] + rank] = i;
                            cnt[j][r] += __popcll(mm);
                        }
                    }
                }
            }
            bool open = false;
#pragma unroll
            for (int j = 0; j < Q; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) open = open || cnt[j][r] < rr.K[r];
            if (!open) break;
        }
        // the lists leave as coalesced rows; the slots behind the hits repeat the first hit (N when the ball is empty)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            if (qb + j >= q1) continue;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int K = rr.K[r];
                const int c = min(cnt[j][r], K);
                int64_t* o = rr.out[r] + ((size_t)b * S + (qb + j)) * K;
                const int first = c > 0 ? hits[j * ksum + koff[r]] : N;
                for (int k = lane; k < K; k += 64) o[k] = (int64_t)(k < c ? hits[j * ksum + koff[r] + k] : first);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void square_distance_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ dst, int S, int N,
                                                              float* __restrict__ out)
{
    // grid: (ceil(N/256), S, B).  One query per block row, points coalesced across lanes.
    const int b = blockIdx.z, s = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const float* q = src + ((size_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float qn = norm3(qx, qy, qz);
    if (n < N) {
        const float* p = dst + ((size_t)b * N + n) * 3;
        out[((size_t)b * S + s) * N + n] = sqdist_expanded(qx, qy, qz, qn, p[0], p[1], p[2]);
    }
}

}  // namespace

// [r5] n_radii ball queries of ONE (cloud, query) pair in one scan (PointNetSetAbstractionMsg, models/pointnet2_utils.py:255-258);
// out_idx[r]: [B, S, K[r]].  n_radii = 1 is mp_ball_query_f32.
template <int Q, int NR, int WAVES>
static int launch_bq_multi(const float* xyz, const float* new_xyz, int64_t B, int64_t N, int64_t S, const BqRadii& rr, hipStream_t stream)
{
    const int npad = ((int)N + 127) & ~127;
    size_t ksum = 0;
    for (int r = 0; r < NR; ++r) ksum += (size_t)rr.K[r];
    const size_t smem = (size_t)3 * npad * sizeof(float) + (size_t)WAVES * Q * ksum * sizeof(int);
    if (smem > 160 * 1024) return MP_EUNSUPPORTED;
    // queries per block: whole wave tiles, enough blocks to cover the chip a few times
    int64_t qpb = (B * S + 767) / 768;
    qpb = ((qpb + WAVES * Q - 1) / (WAVES * Q)) * (WAVES * Q);
    const int chunks = (int)((S + qpb - 1) / qpb);
    auto kern = ball_query_multi_kernel<Q, NR, WAVES>;
    static mp::DynLds lds;
    if (!lds.ensure(reinterpret_cast<const void*>(kern), smem)) return MP_ELAUNCH;
    double wr = 0.0;
    for (int r = 0; r < NR; ++r) wr += (double)rr.K[r];
    char tag[64];       // [r6] one profile row per launch SHAPE (the two levels of the encoder used to be averaged into one row)
    snprintf(tag, sizeof tag, "ball_query_kernel[%dx%d]", (int)N, (int)S);
    MP_LAUNCH(tag, 8.0 * B * (double)S * N, (double)B * (N * 12.0 + S * 12.0 + S * wr * 8.0), kern, dim3(chunks, (unsigned)B),
              dim3(WAVES * MP_WAVE), smem, stream, xyz, new_xyz, (int)N, (int)S, rr, (int)qpb);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_ball_query_multi_f32(const float* xyz, const float* new_xyz, int64_t B, int64_t N, int64_t S, int64_t n_radii,
                                       const double* radii, const int64_t* K, int64_t* const* out_idx, mp_stream_t stream_)
{
    if (B < 0 || N <= 0 || S < 0 || n_radii < 1 || !radii || !K || !out_idx) return MP_EINVAL;
    if (n_radii > BQM_MAXR) return MP_EUNSUPPORTED;
    BqRadii rr{};
    int order[BQM_MAXR] = {0, 1, 2};
    for (int r = 0; r < (int)n_radii; ++r) {
        if (K[r] <= 0 || !(radii[r] >= 0.0)) return MP_EINVAL;
        if (K[r] > 1024) return MP_EUNSUPPORTED;
        if ((B > 0 && S > 0) && !out_idx[r]) return MP_EINVAL;
    }
    // the kernel wants the radii ascending (the largest ball's mask decides whether a point group is looked at all); each list still goes
    // to the caller's buffer for its radius
    for (int a = 0; a < (int)n_radii; ++a)
        for (int c = a + 1; c < (int)n_radii; ++c)
            if (radii[order[c]] < radii[order[a]]) { const int t = order[a]; order[a] = order[c]; order[c] = t; }
    for (int r = 0; r < (int)n_radii; ++r) {
        const int o = order[r];
        rr.r2[r] = (float)(radii[o] * radii[o]);      // squared in double, then cast: pointnet2_utils.py:104
        rr.K[r] = (int)K[o];
        rr.out[r] = out_idx[o];
    }
    if (B == 0 || S == 0) return MP_OK;
    if (!xyz || !new_xyz) return MP_EINVAL;
    if (N > 13312 || B > 65535) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    // small clouds (the second level: 512 points, 128 queries): fewer queries per wave, so that the grid still fills the chip
    const bool small = B * S < 8192;
    // [r6] The hit lists take WAVES * Q * sum(K) * 4 bytes of LDS next to the cloud: where four queries per wave do not fit (N = 13312 with
    // K = 64, N = 10240 with K = 512, N = 5120 with K = 1024) the launch steps down to two queries per wave, then to four waves of two --
    // the LDS need of r4's one-query-per-wave kernel, so every shape that kernel took still runs.
#define BQ_TRY(NR_)                                                                                           \
    {                                                                                                         \
        int rc = small ? MP_EUNSUPPORTED : launch_bq_multi<4, NR_, 8>(xyz, new_xyz, B, N, S, rr, stream);     \
        if (rc == MP_EUNSUPPORTED) rc = launch_bq_multi<2, NR_, 8>(xyz, new_xyz, B, N, S, rr, stream);        \
        if (rc == MP_EUNSUPPORTED) rc = launch_bq_multi<2, NR_, 4>(xyz, new_xyz, B, N, S, rr, stream);        \
        return rc;                                                                                            \
    }
    switch ((int)n_radii) {
        case 1: BQ_TRY(1)
        case 2: BQ_TRY(2)
        default: BQ_TRY(3)
    }
#undef BQ_TRY
}

extern "C" int mp_ball_query_f32(const float* xyz, const float* new_xyz, int64_t B, int64_t N, int64_t S,
                                 double radius, int64_t K, int64_t* out_idx, mp_stream_t stream_)
{
    if (B < 0 || N <= 0 || S < 0 || K <= 0) return MP_EINVAL;
    if (B == 0 || S == 0) return MP_OK;
    if (!xyz || !new_xyz || !out_idx) return MP_EINVAL;
    if (N > 13312 || K > 1024 || B > 65535) return MP_EUNSUPPORTED;
    {   // [r5] the register-tiled scan (MP_BQ_LEGACY=1: the one-query-per-wave kernel below, for A/B timing)
        const char* e = getenv("MP_BQ_LEGACY");
        if (!(e && atoi(e) == 1)) {
            int64_t* outs[1] = {out_idx};
            return mp_ball_query_multi_f32(xyz, new_xyz, B, N, S, 1, &radius, &K, outs, stream_);
        }
    }
    const int npad = ((int)N + 255) & ~255;
    const size_t smem = (size_t)3 * npad * sizeof(float) + (size_t)BQ_WAVES * K * sizeof(int);
    if (smem > 160 * 1024) return MP_EUNSUPPORTED;
    // queries per block: enough blocks to fill 256 CUs several times over, whole waves' worth of queries
    int64_t qpb = (B * S + 1023) / 1024;
    qpb = ((qpb + BQ_WAVES - 1) / BQ_WAVES) * BQ_WAVES;
    if (qpb < BQ_WAVES) qpb = BQ_WAVES;
    const int chunks = (int)((S + qpb - 1) / qpb);
    static mp::DynLds lds;   // see common.h
    if (!lds.ensure(reinterpret_cast<const void*>(ball_query_kernel), smem)) return MP_ELAUNCH;
    const float r2 = (float)(radius * radius);  // squared in double, then cast: pointnet2_utils.py:104
    char tag[64];
    snprintf(tag, sizeof tag, "ball_query_kernel[%dx%d]", (int)N, (int)S);
    MP_LAUNCH(tag, 8.0 * B * (double)S * N, (double)B * (N * 12.0 + S * 12.0 + S * K * 8.0), ball_query_kernel,
              dim3(chunks, (unsigned)B), dim3(BQ_THREADS), smem, mp_stream(stream_), xyz, new_xyz, (int)N, (int)S, r2, (int)K,
              (int)qpb, out_idx);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_square_distance_f32(const float* src, const float* dst, int64_t B, int64_t S, int64_t N,
                                      float* out, mp_stream_t stream_)
{
    if (B < 0 || S < 0 || N < 0) return MP_EINVAL;
    if (B == 0 || S == 0 || N == 0) return MP_OK;
    if (!src || !dst || !out) return MP_EINVAL;
    if (B > 65535 || S > 65535) return MP_EUNSUPPORTED;
    hipLaunchKernelGGL(square_distance_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)S, (unsigned)B), dim3(256), 0,
                       mp_stream(stream_), src, dst, (int)S, (int)N, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
