// Brute-force K nearest neighbours (+ backward) and padded-length detection for gfx950.
//
// Replaces pytorch3d.ops.knn.knn_points (third party; call sites pytorch3d_chamfer.py:182-183, 205-206,
// 257-258) -- the only genuine CUDA kernel on the reference's hot path -- and the Python length loop of
// pytorch3d_chamfer.py:138-149.
//
// Forward design (fp32 VALU bound: 2 instructions per (pair, dimension)):
//   - one lane owns one query, its D coordinates live in VGPRs;
//   - the references stream through LDS in contiguous tiles; every lane of a wave reads the SAME
//     reference, so each ds_read_b128 is a broadcast (one bank row) feeding 4 dimensions x 64 queries;
//   - a workgroup is 4 waves on the same 64 queries, each scanning a quarter of every tile, so short
//     query sets (999 segments) still put >= 2048 waves on the chip; the four partial results are merged
//     through LDS with a (distance, index) lexicographic compare == "first index wins" of a serial scan;
//   - per-cloud lengths are read on device (no host sync): rows >= len1 and slots >= len2 produce 0 / 0.
// Arithmetic: dist = fma chain over d of (p1[d]-p2[d])^2, as pytorch3d's kernel (`dist += diff*diff`
// contracted); strict < keeps the first index on ties.
#include <cstdio>

#include "common.h"

namespace {

constexpr int KNN_WAVES = 4;
constexpr int KNN_THREADS = KNN_WAVES * MP_WAVE;
constexpr int KNN_TILE = 256;  // references per LDS tile (64 per wave)

template <int K>
struct KBest {
    float d[K];
    int i[K];
    int have;
    __device__ __forceinline__ void init()
    {
        have = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) { d[k] = __builtin_inff(); i[k] = 0x7fffffff; }
    }
    // insert keeping (d, i) lexicographic order; candidates that tie on both never occur
    __device__ __forceinline__ void push(float nd, int ni)
    {
        if constexpr (K == 1) {
            if (nd < d[0] || (nd == d[0] && ni < i[0])) { d[0] = nd; i[0] = ni; }
        } else {
            if (!(nd < d[K - 1] || (nd == d[K - 1] && ni < i[K - 1]))) return;
            d[K - 1] = nd;
            i[K - 1] = ni;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
                const bool sw = d[k] < d[k - 1] || (d[k] == d[k - 1] && i[k] < i[k - 1]);
                const float td = sw ? d[k - 1] : d[k];
                const int ti = sw ? i[k - 1] : i[k];
                d[k - 1] = sw ? d[k] : d[k - 1];
                i[k - 1] = sw ? i[k] : i[k - 1];
                d[k] = td;
                i[k] = ti;
            }
        }
    }
};

// D > 0: compile-time dimension, query in registers.  D == 0: run-time dimension `Drt`, query re-read from
// global (L1) per reference -- the slow generic path for dimensions outside the dispatch table.
template <int D, int K>
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                          const int64_t* __restrict__ len1,
                                                          const int64_t* __restrict__ len2, int P1, int P2, int Drt,
                                                          int Kout, float* __restrict__ dists,
                                                          int64_t* __restrict__ idx)
{
    const int Dn = D > 0 ? D : Drt;
    const int DP = (Dn + 3) & ~3;  // LDS row stride, 16-byte aligned rows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);          // [KNN_TILE][DP]
    float* md = tile + KNN_TILE * DP;                          // [KNN_WAVES][K][64] merge buffers
    int* mi = reinterpret_cast<int*>(md + KNN_WAVES * K * 64);

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int l1 = len1 ? (int)min((int64_t)P1, len1[b]) : P1;
    const int l2 = len2 ? (int)min((int64_t)P2, len2[b]) : P2;
    if (blockIdx.x * 64 >= l1) {
        // whole block beyond the valid rows: zeros (pytorch3d pads with 0)
        if (wave == 0 && q < P1)
            for (int k = 0; k < Kout; ++k) {
                dists[((size_t)b * P1 + q) * Kout + k] = 0.0f;
                idx[((size_t)b * P1 + q) * Kout + k] = 0;
            }
        return;
    }
    const bool qvalid = q < l1;
    const float* a_ptr = p1 + ((size_t)b * P1 + (qvalid ? q : 0)) * Dn;
    float a[D > 0 ? D : 1];
    if constexpr (D > 0) {
#pragma unroll
        for (int t = 0; t < D; ++t) a[t] = a_ptr[t];
    }
    KBest<K> best;
    best.init();

    const float* refs = p2 + (size_t)b * P2 * Dn;
    for (int base = 0; base < l2; base += KNN_TILE) {
        const int nt = min(KNN_TILE, l2 - base);
        __syncthreads();  // previous tile fully consumed
        if (DP == Dn) {
            const float4* src = reinterpret_cast<const float4*>(refs + (size_t)base * Dn);
            float4* dst = reinterpret_cast<float4*>(tile);
            const int nv = nt * Dn / 4;
            if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                for (int v = tid; v < nv; v += KNN_THREADS) dst[v] = src[v];
            } else {
                for (int v = tid; v < nt * Dn; v += KNN_THREADS) tile[v] = refs[(size_t)base * Dn + v];
            }
        } else {
            for (int v = tid; v < nt * Dn; v += KNN_THREADS) {
                const int r = v / Dn, t = v - r * Dn;
                tile[r * DP + t] = refs[(size_t)base * Dn + v];
            }
        }
        __syncthreads();
        const int j0 = wave * (KNN_TILE / KNN_WAVES);
        const int j1 = min(nt, j0 + KNN_TILE / KNN_WAVES);
        for (int j = j0; j < j1; ++j) {
            float d = 0.0f;
            if constexpr (D > 0) {
                const float* r = tile + j * DP;
#pragma unroll
                for (int t4 = 0; t4 < (D + 3) / 4; ++t4) {
                    const float4 rv = *reinterpret_cast<const float4*>(r + 4 * t4);
                    const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (4 * t4 + u < D) {
                            const float diff = a[4 * t4 + u] - rr[u];
                            d = __builtin_fmaf(diff, diff, d);
                        }
                    }
                }
            } else {
                const float* r = tile + j * DP;
                for (int t = 0; t < Dn; ++t) {
                    const float diff = a_ptr[t] - r[t];
                    d = __builtin_fmaf(diff, diff, d);
                }
            }
            best.push(d, base + j);
        }
    }

    // merge the four waves' K-best lists
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        md[(wave * K + k) * 64 + lane] = best.d[k];
        mi[(wave * K + k) * 64 + lane] = best.i[k];
    }
    __syncthreads();
    if (wave == 0 && q < P1) {
#pragma unroll
        for (int w = 1; w < KNN_WAVES; ++w)
#pragma unroll
            for (int k = 0; k < K; ++k) best.push(md[(w * K + k) * 64 + lane], mi[(w * K + k) * 64 + lane]);
        float* od = dists + ((size_t)b * P1 + q) * Kout;
        int64_t* oi = idx + ((size_t)b * P1 + q) * Kout;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (k < Kout) {
                const bool ok = qvalid && k < l2;
                od[k] = ok ? best.d[k] : 0.0f;
                oi[k] = ok ? (int64_t)best.i[k] : 0;
            }
        }
    }
}


// ---- K = 1 fast path (every chamfer call of the MaskPlanner loss) ---------------------------------------------------
// QPL queries per lane share each broadcast reference read (halves LDS traffic per pair and gives the fma chains
// independent work to interleave); WAVES waves split every tile, so even 999-segment query sets put two waves on each
// SIMD.  Same arithmetic and tie rule as knn_kernel.
template <int D, int QPL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void knn1_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                          const int64_t* __restrict__ len1,
                                                          const int64_t* __restrict__ len2, int P1, int P2,
                                                          float* __restrict__ dists, int64_t* __restrict__ idx)
{
    constexpr int DP = (D + 3) & ~3;
    constexpr int QB = 64 * QPL;                 // queries per block
    constexpr int RPW = KNN_TILE / WAVES;        // references per wave and tile
    __shared__ __attribute__((aligned(16))) float tile[KNN_TILE * DP];
    __shared__ float md[WAVES][QPL][64];
    __shared__ int mi[WAVES][QPL][64];

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int qbase = blockIdx.x * QB;
    const int l1 = len1 ? (int)min((int64_t)P1, len1[b]) : P1;
    const int l2 = len2 ? (int)min((int64_t)P2, len2[b]) : P2;

    float a[QPL][D];
    float best[QPL];
    int bidx[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int q = qbase + u * 64 + lane;
        const float* ap = p1 + ((size_t)b * P1 + (q < l1 ? q : 0)) * D;
#pragma unroll
        for (int t = 0; t < D; ++t) a[u][t] = ap[t];
        best[u] = __builtin_inff();
        bidx[u] = 0x7fffffff;
    }
    const float* refs = p2 + (size_t)b * P2 * D;
    const bool any_valid = qbase < l1;
    for (int base = 0; any_valid && base < l2; base += KNN_TILE) {
        const int nt = min(KNN_TILE, l2 - base);
        __syncthreads();
        if (DP == D && ((reinterpret_cast<uintptr_t>(refs + (size_t)base * D) & 15) == 0)) {
            const float4* src = reinterpret_cast<const float4*>(refs + (size_t)base * D);
            float4* dst = reinterpret_cast<float4*>(tile);
            for (int v = tid; v < nt * D / 4; v += WAVES * 64) dst[v] = src[v];
        } else {
            for (int v = tid; v < nt * D; v += WAVES * 64) {
                const int r = v / D, t = v - r * D;
                tile[r * DP + t] = refs[(size_t)base * D + v];
            }
        }
        __syncthreads();
        const int j0 = wave * RPW;
        const int j1 = min(nt, j0 + RPW);
        for (int j = j0; j < j1; ++j) {
            const float* r = tile + j * DP;
            float d[QPL];
#pragma unroll
            for (int u = 0; u < QPL; ++u) d[u] = 0.0f;
#pragma unroll
            for (int t4 = 0; t4 < DP / 4; ++t4) {
                const float4 rv = *reinterpret_cast<const float4*>(r + 4 * t4);
                const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * t4 + e < D) {
#pragma unroll
                        for (int u = 0; u < QPL; ++u) {
                            const float diff = a[u][4 * t4 + e] - rr[e];
                            d[u] = __builtin_fmaf(diff, diff, d[u]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < QPL; ++u)
                if (d[u] < best[u]) { best[u] = d[u]; bidx[u] = base + j; }   // ascending j inside a wave: first wins
        }
    }
    // merge the waves: lexicographic (distance, index) == first index of a serial scan
#pragma unroll
    for (int u = 0; u < QPL; ++u) { md[wave][u][lane] = best[u]; mi[wave][u][lane] = bidx[u]; }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            float bd = best[u];
            int bi = bidx[u];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const float od = md[w][u][lane];
                const int oi = mi[w][u][lane];
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            const int q = qbase + u * 64 + lane;
            if (q < P1) {
                const bool ok = q < l1 && l2 > 0;
                dists[(size_t)b * P1 + q] = ok ? bd : 0.0f;
                idx[(size_t)b * P1 + q] = ok ? (int64_t)bi : 0;
            }
        }
    }
}

template <int D>
int launch_knn1(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int B, int P1, int P2,
                float* dists, int64_t* idx, hipStream_t stream)
{
    char tag[48];
    snprintf(tag, sizeof tag, "knn1_kernel<%d>", D);
    const double flops = 3.0 * D * (double)B * P1 * P2, bytes = (double)B * ((P1 + P2) * 4.0 * D + P1 * 12.0);
    MP_LAUNCH(tag, flops, bytes, (knn1_kernel<D, 2, 8>), dim3((P1 + 127) / 128, B), dim3(512), 0, stream, p1, p2, len1, len2,
              P1, P2, dists, idx);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

template <int D, int K>
int launch_knn(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int B, int P1, int P2,
               int Drt, int Kout, float* dists, int64_t* idx, hipStream_t stream)
{
    const int Dn = D > 0 ? D : Drt;
    const int DP = (Dn + 3) & ~3;
    const size_t smem = (size_t)KNN_TILE * DP * sizeof(float) + (size_t)KNN_WAVES * K * 64 * 8;
    if (smem > 64 * 1024) return MP_EUNSUPPORTED;
    char tag[48];
    snprintf(tag, sizeof tag, "knn_kernel<%d, %d>", D, K);
    MP_LAUNCH(tag, 3.0 * Dn * (double)B * P1 * P2, (double)B * ((P1 + P2) * 4.0 * Dn + P1 * 12.0 * Kout), (knn_kernel<D, K>),
              dim3((P1 + 63) / 64, B), dim3(KNN_THREADS), smem, stream, p1, p2, len1, len2, P1, P2, Drt, Kout, dists, idx);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

template <int K>
int dispatch_d(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int B, int P1, int P2,
               int D, int Kout, float* dists, int64_t* idx, hipStream_t stream)
{
    switch (D) {
        case 3: return launch_knn<3, K>(p1, p2, len1, len2, B, P1, P2, D, Kout, dists, idx, stream);
        case 6: return launch_knn<6, K>(p1, p2, len1, len2, B, P1, P2, D, Kout, dists, idx, stream);
        case 12: return launch_knn<12, K>(p1, p2, len1, len2, B, P1, P2, D, Kout, dists, idx, stream);
        case 24: return launch_knn<24, K>(p1, p2, len1, len2, B, P1, P2, D, Kout, dists, idx, stream);
        default: return launch_knn<0, K>(p1, p2, len1, len2, B, P1, P2, D, Kout, dists, idx, stream);
    }
}

// grad_p1[b,i,t] = sum_k 2*g[i,k]*(p1[i,t]-p2[idx[i,k],t]); optional atomic scatter of the negative into grad_p2
// Gradient of a REDUCED chamfer term w.r.t. the distance of row (b, i) -- what chamfer_reduce_bwd_kernel (chamfer_reduce.hip) would
// write into grad_dists, evaluated in place (same operations in the same order): the loss terms of a training step then need no
// [B, P1] gradient tensor and no launch of their own between the scalar loss gradient and the scatter below.
struct RowGrad {
    const float* grad_out;   // [1], or [B] when batch_mode == 0
    int point_mean, batch_mode;
    float div, scale;
    __device__ __forceinline__ float at(int64_t b, int64_t len) const
    {
        float g = (batch_mode == 0 ? grad_out[b] : grad_out[0]) * scale;
        if (batch_mode == 2) g = g / div;
        if (point_mean) g = g / (float)len;
        return g;
    }
};

__global__ __launch_bounds__(256) void knn_bwd_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                      const int64_t* __restrict__ len1,
                                                      const int64_t* __restrict__ len2,
                                                      const int64_t* __restrict__ idx,
                                                      const float* __restrict__ grad_dists, int64_t P1, int64_t P2,
                                                      int64_t D, int64_t K, int64_t total,
                                                      float* __restrict__ grad_p1, float* __restrict__ grad_p2_atomic, RowGrad rg)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / D;  // b*P1 + i
        const int64_t t = e - row * D;
        const int64_t b = row / P1;
        const int64_t i = row - b * P1;
        const int64_t l1 = len1 ? len1[b] : P1;
        const int64_t l2 = len2 ? len2[b] : P2;
        float acc = 0.0f;
        if (i < l1) {
            const float a = p1[e];
            for (int64_t k = 0; k < K && k < l2; ++k) {
                const int64_t j = idx[row * K + k];
                const float g = grad_dists ? grad_dists[row * K + k] : rg.at(b, l1);
                const float v = 2.0f * g * (a - p2[(b * P2 + j) * D + t]);
                acc += v;
                if (grad_p2_atomic) atomicAdd(grad_p2_atomic + (b * P2 + j) * D + t, -v);
            }
        }
        if (grad_p1) grad_p1[e] = acc;
    }
}

// deterministic grad_p2: one wave per destination row (b, j); scan idx[b] in (i, k) order
__global__ __launch_bounds__(256) void knn_bwd_p2_ordered_kernel(const float* __restrict__ p1,
                                                                 const float* __restrict__ p2,
                                                                 const int64_t* __restrict__ len1,
                                                                 const int64_t* __restrict__ len2,
                                                                 const int64_t* __restrict__ idx,
                                                                 const float* __restrict__ grad_dists, int64_t P1,
                                                                 int64_t P2, int64_t D, int64_t K, int64_t rows_total,
                                                                 float* __restrict__ grad_p2, RowGrad rg)
{
    const int lane = threadIdx.x & 63;
    const int64_t dest = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (dest >= rows_total) return;
    const int64_t b = dest / P2;
    const int64_t j = dest - b * P2;
    const int64_t l1 = len1 ? min(len1[b], P1) : P1;
    const int64_t l2 = len2 ? min(len2[b], P2) : P2;
    const int64_t kk = K < l2 ? K : l2;
    const int64_t M = l1 * K;  // flattened (i, k), valid rows only
    const int64_t* bi = idx + b * P1 * K;
    const float* bg = grad_dists + b * P1 * K;
    for (int64_t c0 = 0; c0 < D; c0 += 64) {
        const int64_t t = c0 + lane;
        float acc = 0.0f;
        const float pj = t < D ? p2[dest * D + t] : 0.0f;
        for (int64_t m0 = 0; m0 < M; m0 += 64) {
            const int64_t m = m0 + lane;
            unsigned long long hit = __ballot(m < M && (m % K) < kk && bi[m] == j);
            while (hit) {
                const int s = __builtin_ctzll(hit);
                hit &= hit - 1;
                const int64_t mm = m0 + s;
                if (t < D) acc -= 2.0f * (grad_dists ? bg[mm] : rg.at(b, len1 ? len1[b] : P1)) * (p1[(b * P1 + mm / K) * D + t] - pj);
            }
        }
        if (t < D) grad_p2[dest * D + t] = acc;
    }
}

__global__ __launch_bounds__(256) void padded_lengths_kernel(const float* __restrict__ y, int P2, int D,
                                                             int64_t* __restrict__ lengths)
{
    // one block per sample; first column whose leading coordinate equals the -100 sentinel
    __shared__ int smin;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) smin = P2;
    __syncthreads();
    int first = P2;
    for (int c = threadIdx.x; c < P2; c += 256)
        if (y[((size_t)b * P2 + c) * D] == -100.0f) { first = c; break; }
    first = (int)mp::wave_min_u32((unsigned)first);
    if ((threadIdx.x & 63) == 0) atomicMin(&smin, first);
    __syncthreads();
    if (threadIdx.x == 0) lengths[b] = smin;
}

inline unsigned grid_for(int64_t total)
{
    int64_t g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" size_t mp_knn_workspace_bytes(int64_t, int64_t, int64_t) { return 0; }

extern "C" int mp_knn_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B,
                          int64_t P1, int64_t P2, int64_t D, int64_t K, float* dists, int64_t* idx, void*, size_t,
                          mp_stream_t stream_)
{
    if (B < 0 || P1 < 0 || P2 < 0 || D <= 0 || K <= 0) return MP_EINVAL;
    if (B == 0 || P1 == 0) return MP_OK;
    if (!p1 || !dists || !idx || (P2 > 0 && !p2)) return MP_EINVAL;
    if (K > 8 || D > 1024 || B > 65535 || P1 > (1 << 30) || P2 > (1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n1 = (int)P1, n2 = (int)P2, d = (int)D, k = (int)K;
    if (k == 1) {
        switch (d) {
            case 3: return launch_knn1<3>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
            case 6: return launch_knn1<6>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
            case 12: return launch_knn1<12>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
            case 24: return launch_knn1<24>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
            default: return dispatch_d<1>(p1, p2, len1, len2, b, n1, n2, d, k, dists, idx, stream);
        }
    }
    if (k == 2) return dispatch_d<2>(p1, p2, len1, len2, b, n1, n2, d, k, dists, idx, stream);
    if (k <= 4) return dispatch_d<4>(p1, p2, len1, len2, b, n1, n2, d, k, dists, idx, stream);
    return dispatch_d<8>(p1, p2, len1, len2, b, n1, n2, d, k, dists, idx, stream);
}

static int knn_bwd(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, const int64_t* idx,
                   const float* grad_dists, const RowGrad& rg, int64_t B, int64_t P1, int64_t P2, int64_t D, int64_t K,
                   float* grad_p1, float* grad_p2, int deterministic, mp_stream_t stream_)
{
    if (B < 0 || P1 < 0 || P2 < 0 || D <= 0 || K <= 0) return MP_EINVAL;
    if (B == 0) return MP_OK;
    if (P1 > 0 && P2 > 0 && (!p1 || !p2 || !idx || (!grad_dists && !rg.grad_out))) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    if (grad_p2 && P2 > 0 && (P1 == 0 || !deterministic)) {
        if (!mp::zero_async(grad_p2, (size_t)(B * P2 * D), stream)) return MP_ELAUNCH;
    }
    if (P1 == 0 || P2 == 0) {
        if (grad_p1 && P1 > 0 && !mp::zero_async(grad_p1, (size_t)(B * P1 * D), stream))
            return MP_ELAUNCH;
        return MP_OK;
    }
    const int64_t total = B * P1 * D;
    float* atomic_dst = (grad_p2 && !deterministic) ? grad_p2 : nullptr;
    if (grad_p1 || atomic_dst) {
        hipLaunchKernelGGL(knn_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p1, p2, len1, len2, idx,
                           grad_dists, P1, P2, D, K, total, grad_p1, atomic_dst, rg);
        MP_CHECK_LAUNCH();
    }
    if (grad_p2 && deterministic) {
        const int64_t rows = B * P2;
        hipLaunchKernelGGL(knn_bwd_p2_ordered_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, p1, p2,
                           len1, len2, idx, grad_dists, P1, P2, D, K, rows, grad_p2, rg);
        MP_CHECK_LAUNCH();
    }
    return MP_OK;
}

extern "C" int mp_knn_bwd_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2,
                              const int64_t* idx, const float* grad_dists, int64_t B, int64_t P1, int64_t P2,
                              int64_t D, int64_t K, float* grad_p1, float* grad_p2, int deterministic,
                              mp_stream_t stream_)
{
    if (B > 0 && P1 > 0 && P2 > 0 && !grad_dists) return MP_EINVAL;
    return knn_bwd(p1, p2, len1, len2, idx, grad_dists, RowGrad{}, B, P1, P2, D, K, grad_p1, grad_p2, deterministic, stream_);
}

// The same backward for distances that went straight into mp_chamfer_reduce_f32 (K = 1): `grad_out` is the gradient of the reduced
// value ([1], or [B] when batch_mode == 0), and the per-row factor scale / div / len1[b] is applied here (point_mean, batch_mode,
// div, scale as given to the reduction) -- mp_chamfer_reduce_bwd_f32 and its [B, P1] output are not needed.
extern "C" int mp_knn_bwd_reduced_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2,
                                      const int64_t* idx, const float* grad_out, int point_mean, int batch_mode, double div,
                                      double scale, int64_t B, int64_t P1, int64_t P2, int64_t D, float* grad_p1,
                                      float* grad_p2, int deterministic, mp_stream_t stream_)
{
    if (batch_mode < 0 || batch_mode > 2 || (point_mean && !len1)) return MP_EINVAL;
    if (B > 0 && P1 > 0 && P2 > 0 && !grad_out) return MP_EINVAL;
    RowGrad rg{grad_out, point_mean, batch_mode, (float)div, (float)scale};
    return knn_bwd(p1, p2, len1, len2, idx, nullptr, rg, B, P1, P2, D, 1, grad_p1, grad_p2, deterministic, stream_);
}

extern "C" int mp_padded_lengths_f32(const float* y, int64_t B, int64_t P2, int64_t D, int64_t* lengths,
                                     mp_stream_t stream_)
{
    if (B < 0 || P2 < 0 || D <= 0) return MP_EINVAL;
    if (B == 0) return MP_OK;
    if (!lengths || (P2 > 0 && !y)) return MP_EINVAL;
    hipLaunchKernelGGL(padded_lengths_kernel, dim3((unsigned)B), dim3(256), 0, mp_stream(stream_), y, (int)P2, (int)D,
                       lengths);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
