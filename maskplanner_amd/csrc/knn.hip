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
#include <cstdlib>

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

// ---- K = 1, screened on the matrix cores ----------------------------------------------------------------------------
// The direct kernel above spends 2 VALU instructions per (pair, dimension).  Here the bf16 matrix cores SCREEN the pairs and the
// direct arithmetic runs only where it can matter:
//   * every operand is three bf16 planes h + m + l (24 significant bits); the six plane products of order <= 2^-16 give the dot
//     product x.y to fp32 accuracy on v_mfma_f32_32x32x16_bf16 (the planes are laid along K: 6 x D slots, D padded to 8), and
//         t(q, j) = |y_j|^2 - 2 x_q.y_j            (= d(q, j) - |x_q|^2 up to E <= c * 2^-24 * (|x|^2 + |y|^2))
//     is a screening value: a wave takes 32 queries as the MFMA columns, so a lane holds 16 references of ONE query per 32-row block and
//     the block minimum is an in-register min;
//   * a lane records (block, block minimum) whenever the minimum is within the error window of its running minimum -- a superset of
//     the blocks that can hold the exact nearest neighbour; the last three records live in registers;
//   * at the end the recorded blocks still inside the window of the FINAL minimum (one or two per query) are evaluated with the direct
//     kernel's arithmetic -- d = fma chain over the dimensions of (x - y)^2 -- each lane over its own 16 rows, and merged with the same
//     lexicographic (distance, index) rule.  A lane whose window ever held more than three blocks (duplicated references) scans
//     every block exactly instead: slower than the direct kernel there, never different.
// The exact minimiser j* satisfies t(j*) <= t_min + 2E + (rounding of the chain), so it is always among the evaluated pairs and the
// result is bit-identical to knn1_kernel's (tests/test_gpu_ops.py compares them on ties, duplicates and ragged lengths).
typedef __attribute__((ext_vector_type(8))) __bf16 kbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 kbf16x4;
typedef __attribute__((ext_vector_type(16))) float kf32x16;

__device__ __forceinline__ void ksplit3(float x, __bf16& h, __bf16& m, __bf16& l)
{
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

constexpr int KS_RT = 128;      // references per LDS tile (four 32-row MFMA blocks)

// Pre-pass of the screened search: every reference row once as (h, m, l) bf16 planes [row][3][DPAD] + its squared norm, into the
// caller's workspace (clouds padded to a multiple of KS_RT rows; rows at or beyond the cloud's length: zero planes, infinite norm --
// never a candidate).  The main kernel's tiles are then plain 16-byte copies.
template <int D>
__global__ __launch_bounds__(256) void knn_planes_kernel(const float* __restrict__ p2, const int64_t* __restrict__ len2, int P2, int P2pad,
                                                         __bf16* __restrict__ planes, float* __restrict__ norms)
{
    constexpr int DPAD = (D + 7) & ~7, LDH = 3 * DPAD;
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P2pad) return;
    const int l2 = len2 ? (int)min((int64_t)P2, len2[b]) : P2;
    __bf16* out = planes + ((size_t)b * P2pad + j) * LDH;
    float ny = __builtin_inff();
    float y[D];
#pragma unroll
    for (int t = 0; t < D; ++t) y[t] = 0.0f;
    if (j < l2) {
        const float* yr = p2 + ((size_t)b * P2 + j) * D;
        ny = 0.0f;
#pragma unroll
        for (int t = 0; t < D; ++t) { y[t] = yr[t]; ny = __builtin_fmaf(y[t], y[t], ny); }
    }
#pragma unroll
    for (int c = 0; c < DPAD; c += 8) {
        kbf16x8 h, m, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 hh = (__bf16)0.0f, mm = (__bf16)0.0f, ll = (__bf16)0.0f;
            if (c + e < D) ksplit3(y[c + e], hh, mm, ll);
            h[e] = hh; m[e] = mm; l[e] = ll;
        }
        *reinterpret_cast<kbf16x8*>(out + c) = h;
        *reinterpret_cast<kbf16x8*>(out + DPAD + c) = m;
        *reinterpret_cast<kbf16x8*>(out + 2 * DPAD + c) = l;
    }
    norms[(size_t)b * P2pad + j] = ny;
    if constexpr (DPAD > D) {
        // a spare K slot: |y|^2 rides in the MFMA -- its three planes in slot D of the (h, m, l) planes, met by -0.5 on the query side,
        // so that the accumulator is x.y - |y|^2 / 2 = -t / 2 and the screening needs no norm reads and no fma per pair.  Rows that
        // are not there get a huge FINITE norm (an infinity would meet the zeros of the other plane pairs: NaN).
        __bf16 nh, nm, nl;
        ksplit3(j < l2 ? ny : 1e30f, nh, nm, nl);
        out[D] = nh;
        out[DPAD + D] = nm;
        out[2 * DPAD + D] = nl;
    }
}

template <int D>
__global__ __launch_bounds__(512) void knn1_screen_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                          const int64_t* __restrict__ len1, const int64_t* __restrict__ len2,
                                                          int P1, int P2, int P2pad, const __bf16* __restrict__ planes,
                                                          const float* __restrict__ norms, float* __restrict__ dists,
                                                          int64_t* __restrict__ idx)
{
    constexpr int DPAD = (D + 7) & ~7, NB = DPAD / 8;         // dimension blocks of 8 K-slots
    constexpr int NS = 3 * NB;                                 // MFMA k-steps: 6 plane pairs x NB blocks / 2 per step
    constexpr int LDH = 3 * DPAD;                              // halves per reference: planes h | m | l
    constexpr int PCH = KS_RT * LDH / 8;                       // 16-byte chunks of a tile's planes
    constexpr int NCH = PCH + KS_RT / 4;                       // ... plus its norms
    constexpr int CPT = (NCH + 511) / 512;                     // chunks per thread
    constexpr bool FOLD = DPAD > D;                            // |y|^2 folded into the MFMA (knn_planes_kernel): acc = -t / 2
    constexpr bool ROWMASK = true;                             // records keep which rows of a block are inside the window
    // plane pair of slot group g: reference plane, query plane -- (h,h) (m,h) (h,m) (m,m) (l,h) (h,l)
    constexpr int RPL[6] = {0, 1, 0, 1, 2, 0}, QPL[6] = {0, 0, 1, 1, 0, 2};
    __shared__ __attribute__((aligned(16))) __bf16 refH[2][KS_RT * LDH];
    __shared__ __attribute__((aligned(16))) float refN[2][KS_RT];
    __shared__ float xd[4][64];                                // merge of the two waves that share a query group
    __shared__ int xi[4][64];

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qg = wave & 3, hf = wave >> 2;                   // query group (32 queries) and which row blocks of a tile this wave scans
    const int col = lane & 31, kh = lane >> 5;
    const int l1 = len1 ? (int)min((int64_t)P1, len1[b]) : P1;
    const int l2 = len2 ? (int)min((int64_t)P2, len2[b]) : P2;
    const int qbase = blockIdx.x * 128;
    const int q = qbase + qg * 32 + col;
    const bool qvalid = q < l1;

    // this lane's query: fp32 coordinates (kept for the exact evaluation), |x|^2, the B fragments of every k-step
    float a[D];
    {
        const float* ap = p1 + ((size_t)b * P1 + (qvalid ? q : 0)) * D;
#pragma unroll
        for (int t = 0; t < D; ++t) a[t] = ap[t];
    }
    float nx = 0.0f;
#pragma unroll
    for (int t = 0; t < D; ++t) nx = __builtin_fmaf(a[t], a[t], nx);
    kbf16x8 bq[NS];
    {
        __bf16 pl[3][DPAD];
#pragma unroll
        for (int t = 0; t < DPAD; ++t) {
            if (t < D) ksplit3(a[t], pl[0][t], pl[1][t], pl[2][t]);
            else { pl[0][t] = (__bf16)0.0f; pl[1][t] = (__bf16)0.0f; pl[2][t] = (__bf16)0.0f; }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kbf16x8 f0, f1;          // slot groups 2s (kh = 0) and 2s + 1 (kh = 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                auto slot = [&](int g) -> __bf16 {
                    const int pt = g / NB, dim = 8 * (g % NB) + e;
                    if (FOLD && dim == D) return (__bf16)((pt == 0 || pt == 1 || pt == 4) ? -0.5f : 0.0f);   // ref planes h, m, l once each
                    return pl[QPL[pt]][dim];
                };
                f0[e] = slot(2 * s);
                f1[e] = slot(2 * s + 1);
            }
            bq[s] = kh ? f1 : f0;
        }
    }
    // |err of t| <= c 2^-24 (|x|^2 + |y|^2) with |y|^2 <= 2 (|x|^2 + d) for the references that matter; c = 64 covers the dropped
    // plane products (3), the MFMA's fp32 accumulation over <= 144 slots and the norms; the window is twice that plus the chain's own
    // rounding.  wnd(b1): window above a running minimum b1 of t (d = b1 + nx).
    auto wnd = [&](float b1) { const float dd = __builtin_fmaxf(b1 + nx, 0.0f); return 7.7e-6f * (3.0f * nx + 2.0f * dd) + 1e-30f; };

    float b1 = __builtin_inff();
    float rm0 = __builtin_inff(), rm1 = __builtin_inff(), rm2 = __builtin_inff();
    int rb0 = 0, rb1 = 0, rb2 = 0;
    bool overflow = false;
    const float* refs = p2 + (size_t)b * P2 * D;
    const bool any_valid = qbase < l1;
    const int ntiles = any_valid ? (l2 + KS_RT - 1) / KS_RT : 0;
    // tiles: 16-byte copies of the pre-split planes and norms, fetched one tile ahead into registers, two LDS buffers, one barrier per tile
    const uint4* gpl = reinterpret_cast<const uint4*>(planes + (size_t)b * P2pad * LDH);
    const uint4* gno = reinterpret_cast<const uint4*>(norms + (size_t)b * P2pad);
    uint4 pre[CPT];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int e = tid + 512 * u;
            if (e < PCH) pre[u] = gpl[(size_t)t * PCH + e];
            else if (e < NCH) pre[u] = gno[(size_t)t * (KS_RT / 4) + (e - PCH)];
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int e = tid + 512 * u;
            if (e < PCH) reinterpret_cast<uint4*>(refH[buf])[e] = pre[u];
            else if (e < NCH) reinterpret_cast<uint4*>(refN[buf])[e - PCH] = pre[u];
        }
    };
    if (ntiles > 0) { fetch(0); stash(0); }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1, base = t * KS_RT;
        if (t + 1 < ntiles) fetch(t + 1);
#if defined(KS_ABL) && KS_ABL == 3
        const int nrb = 0;
#else
        const int nrb = min(4, (l2 - base + 31) / 32);
#endif
        for (int rb = hf; rb < nrb; rb += 2) {
            kf32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const __bf16* rowp = refH[cur] + (rb * 32 + col) * LDH;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int g0 = 2 * s, g1 = 2 * s + 1;
                const int o0 = RPL[g0 / NB] * DPAD + 8 * (g0 % NB), o1 = RPL[g1 / NB] * DPAD + 8 * (g1 % NB);
                const kbf16x8 ar = *reinterpret_cast<const kbf16x8*>(rowp + (kh ? o1 : o0));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar, bq[s], acc, 0, 0, 0);
            }
            // this lane's 16 references of the block: rows (r & 3) + 8 (r >> 2) + 4 kh.  tv = -t / 2 (FOLD: the accumulator itself)
            float tv[16];
            if constexpr (FOLD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tv[r] = acc[r];
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 ny = *reinterpret_cast<const float4*>(&refN[cur][rb * 32 + 8 * r4 + 4 * kh]);
                    tv[4 * r4 + 0] = __builtin_fmaf(-0.5f, ny.x, acc[4 * r4 + 0]);
                    tv[4 * r4 + 1] = __builtin_fmaf(-0.5f, ny.y, acc[4 * r4 + 1]);
                    tv[4 * r4 + 2] = __builtin_fmaf(-0.5f, ny.z, acc[4 * r4 + 2]);
                    tv[4 * r4 + 3] = __builtin_fmaf(-0.5f, ny.w, acc[4 * r4 + 3]);
                }
            }
            float mx = __builtin_fmaxf(__builtin_fmaxf(tv[0], tv[1]), tv[2]);          // (v_max3_f32)
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, tv[r]), tv[r + 1]);
            mx = __builtin_fmaxf(mx, tv[15]);
            const float m = -2.0f * mx;
            b1 = __builtin_fminf(b1, m);
#if defined(KS_ABL) && KS_ABL == 2
            if (false) {
#else
            if (qvalid && m <= b1 + wnd(b1)) {        // (always true for the block that sets a new minimum)
#endif
                // the last three records of the lane, in registers: (block minimum, block, ROWMASK: which of the 16 rows are inside the
                // window now -- a superset of those inside the final one).  A record that is pushed out while still inside the window of
                // the running minimum (four blocks within ~1e-5 |x|^2 of each other: duplicated references) sends the lane to the full scan.
                const float thr = b1 + wnd(b1);
                if (rm2 <= thr) overflow = true;
                unsigned mask = 0xffffu;
                if constexpr (ROWMASK) {
                    // bit (15 - r): row r is inside the window, i.e. tv[r] >= -thr / 2: the sign bit of (-thr / 2 - tv[r]) shifted in with one
                    // v_alignbit_b32 per row (threshold one notch down so that equality counts)
                    const float ta = -0.5f * thr;
                    const float ta_dn = ta - __builtin_fabsf(ta) * 1.2e-7f - 1e-37f;
                    mask = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(ta_dn - tv[r]), 31);
                }
                rm2 = rm1; rb2 = rb1;
                rm1 = rm0; rb1 = rb0;
                rm0 = m; rb0 = ((base / 32 + rb) << 16) | (int)mask;
            }
        }
        if (t + 1 < ntiles) stash(cur ^ 1);
        __syncthreads();
    }
    // ---- exact evaluation of the surviving blocks: every lane walks its own 16 rows of each block with the direct kernel's chain ----
    float bd = __builtin_inff();
    int bi = 0x7fffffff;
    const float lim = b1 + wnd(b1);
    auto eval_rows = [&](int blk, unsigned mask) {      // rows i of block blk with bit i of mask set, ascending
        while (mask) {           // bit (15 - i) <-> row i: highest bit first == ascending rows
            const int i = __builtin_clz(mask) - 16;
            mask &= ~(0x8000u >> i);
            const int j = blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (j < l2) {
                const float* yr = refs + (size_t)j * D;
                float y[D];           // one row in as few, as wide loads as its alignment allows (every lane reads its own row)
                if constexpr (D % 4 == 0) {
#pragma unroll
                    for (int t = 0; t < D; t += 4) { const float4 v = *reinterpret_cast<const float4*>(yr + t); y[t] = v.x; y[t + 1] = v.y; y[t + 2] = v.z; y[t + 3] = v.w; }
                } else if constexpr (D % 2 == 0) {
#pragma unroll
                    for (int t = 0; t < D; t += 2) { const float2 v = *reinterpret_cast<const float2*>(yr + t); y[t] = v.x; y[t + 1] = v.y; }
                } else {
#pragma unroll
                    for (int t = 0; t < D; ++t) y[t] = yr[t];
                }
                float d = 0.0f;
#pragma unroll
                for (int t = 0; t < D; ++t) { const float diff = a[t] - y[t]; d = __builtin_fmaf(diff, diff, d); }
                if (d < bd || (d == bd && j < bi)) { bd = d; bi = j; }
            }
        }
    };
#if defined(KS_ABL) && KS_ABL >= 1
    if (false) {
#else
    {
#endif
        // the rows inside the final window, of at most three blocks: the first four of a lane are fetched TOGETHER (one global latency
        // for the wave instead of one per row), any further ones -- and the lanes that overflowed -- take the loops
        const bool live = qvalid && !overflow;
        unsigned long long cand = 0;
        if (live && rm0 <= lim) cand |= (unsigned long long)(rb0 & 0xffff);
        if (live && rm1 <= lim) cand |= (unsigned long long)(rb1 & 0xffff) << 16;
        if (live && rm2 <= lim) cand |= (unsigned long long)(rb2 & 0xffff) << 32;
        int cj[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cj[k] = -1;
            if (cand) {
                const int pos = (int)__builtin_ctzll(cand);
                cand &= cand - 1;
                const int rec = pos >> 4, i = 15 - (pos & 15);          // bit (15 - i) <-> row i
                const int blk = (rec == 0 ? rb0 : (rec == 1 ? rb1 : rb2)) >> 16;
                const int j = blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                cj[k] = j < l2 ? j : -1;
            }
        }
        float cy[4][D];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* yr = refs + (size_t)(cj[k] >= 0 ? cj[k] : 0) * D;
            if constexpr (D % 4 == 0) {
#pragma unroll
                for (int t = 0; t < D; t += 4) { const float4 v = *reinterpret_cast<const float4*>(yr + t); cy[k][t] = v.x; cy[k][t + 1] = v.y; cy[k][t + 2] = v.z; cy[k][t + 3] = v.w; }
            } else if constexpr (D % 2 == 0) {
#pragma unroll
                for (int t = 0; t < D; t += 2) { const float2 v = *reinterpret_cast<const float2*>(yr + t); cy[k][t] = v.x; cy[k][t + 1] = v.y; }
            } else {
#pragma unroll
                for (int t = 0; t < D; ++t) cy[k][t] = yr[t];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = 0.0f;
#pragma unroll
            for (int t = 0; t < D; ++t) { const float diff = a[t] - cy[k][t]; d = __builtin_fmaf(diff, diff, d); }
            if (cj[k] >= 0 && (d < bd || (d == bd && cj[k] < bi))) { bd = d; bi = cj[k]; }
        }
        if (__ballot(cand != 0)) {             // more than four rows in the window
            while (cand) {
                const int pos = (int)__builtin_ctzll(cand);
                cand &= cand - 1;
                const int rec = pos >> 4;
                eval_rows((rec == 0 ? rb0 : (rec == 1 ? rb1 : rb2)) >> 16, 0x8000u >> (15 - (pos & 15)));
            }
        }
        if (__ballot(qvalid && overflow)) {      // every block this wave scanned, exactly (as slow as the direct kernel; correct)
            if (qvalid && overflow)
                for (int blk = 0; blk * 32 < l2; ++blk)
                    if (((blk & 3) & 1) == hf) eval_rows(blk, 0xffffu);
        }
    }
    // the two row halves of a query, then the two waves of its group
    {
        const float od = __shfl_xor(bd, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (hf == 1) { xd[qg][lane] = bd; xi[qg][lane] = bi; }
    __syncthreads();
    if (hf == 0) {
        const float od = xd[qg][lane];
        const int oi = xi[qg][lane];
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        if (kh == 0 && q < P1) {
            const bool ok = qvalid && l2 > 0;
            dists[(size_t)b * P1 + q] = ok ? bd : 0.0f;
            idx[(size_t)b * P1 + q] = ok ? (int64_t)bi : 0;
        }
    }
}

inline size_t knn1_screen_ws(int64_t B, int64_t P2, int64_t D)
{
    const int64_t dpad = (D + 7) & ~7, p2pad = (P2 + KS_RT - 1) / KS_RT * KS_RT;
    return (size_t)(B * p2pad * (3 * dpad * 2 + 4));
}

template <int D>
int launch_knn1_screen(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int B, int P1, int P2,
                       float* dists, int64_t* idx, void* ws, size_t ws_bytes, hipStream_t stream, int phases = 3)
{   // phases: 1 = the references' planes + norms into ws only, 2 = the search against a ws prepared by an earlier phase-1 call, 3 = both
    constexpr int DPAD = (D + 7) & ~7;
    if (!ws || ws_bytes < knn1_screen_ws(B, P2, D) || (reinterpret_cast<uintptr_t>(ws) & 15)) return MP_EINVAL;
    const int P2pad = (P2 + KS_RT - 1) / KS_RT * KS_RT;
    __bf16* planes = reinterpret_cast<__bf16*>(ws);
    float* norms = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)B * P2pad * 3 * DPAD * 2);
    if (phases & 1) {
        hipLaunchKernelGGL((knn_planes_kernel<D>), dim3((P2pad + 255) / 256, B), dim3(256), 0, stream, p2, len2, P2, P2pad, planes, norms);
        MP_CHECK_LAUNCH();
    }
    if (!(phases & 2)) return MP_OK;
    char tag[48];
    snprintf(tag, sizeof tag, "knn1_screen_kernel<%d>[%dx%d]", D, (int)P1, (int)P2);      // [r6] queries x references: one profile row per launch shape
    const double flops = 3.0 * D * (double)B * P1 * P2, bytes = (double)B * ((P1 + P2) * 4.0 * D + P1 * 12.0);
    MP_LAUNCH(tag, flops, bytes, (knn1_screen_kernel<D>), dim3((P1 + 127) / 128, B), dim3(512), 0, stream, p1, p2, len1, len2, P1, P2,
              P2pad, planes, norms, dists, idx);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

// MP_KNN_SCREEN=0: the direct kernel for every K = 1 search (A/B timing, and the reference the screened kernel is tested against)
inline bool knn_screen_enabled()
{
    static const bool on = !(getenv("MP_KNN_SCREEN") && atoi(getenv("MP_KNN_SCREEN")) == 0);
    return on;
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
    int accumulate_p1;       // grad_p1 += (a running total shared by several loss terms on the same points) instead of =
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
        if (grad_p1) grad_p1[e] = rg.accumulate_p1 ? grad_p1[e] + acc : acc;
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
    for (int c0 = threadIdx.x; c0 < P2 && first == P2; c0 += 4 * 256) {      // four rows in flight (the one-row loop is a chain of latencies)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (c0 + u * 256 < P2) ? y[((size_t)b * P2 + c0 + u * 256) * D] : 0.0f;
#pragma unroll
        for (int u = 3; u >= 0; --u)
            if (v[u] == -100.0f) first = c0 + u * 256;          // the lowest hit of the batch wins
    }
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
// workspace of the screened K = 1 search (bf16 planes + norms of the references); without it mp_knn_f32 uses the direct scan
extern "C" size_t mp_knn1_workspace_bytes(int64_t B, int64_t P2, int64_t D)
{
    if (B <= 0 || P2 <= 0 || !(D == 3 || D == 6 || D == 12 || D == 24)) return 0;
    return knn1_screen_ws(B, P2, D);
}

extern "C" int mp_knn_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B,
                          int64_t P1, int64_t P2, int64_t D, int64_t K, float* dists, int64_t* idx, void* ws, size_t ws_bytes,
                          mp_stream_t stream_)
{
    if (B < 0 || P1 < 0 || P2 < 0 || D <= 0 || K <= 0) return MP_EINVAL;
    if (B == 0 || P1 == 0) return MP_OK;
    if (!p1 || !dists || !idx || (P2 > 0 && !p2)) return MP_EINVAL;
    if (K > 8 || D > 1024 || B > 65535 || P1 > (1 << 30) || P2 > (1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n1 = (int)P1, n2 = (int)P2, d = (int)D, k = (int)K;
    if (k == 1 && knn_screen_enabled() && n2 >= 256 && ws && ws_bytes >= mp_knn1_workspace_bytes(B, P2, D) && mp_knn1_workspace_bytes(B, P2, D) > 0) {
        switch (d) {
            case 3: return launch_knn1_screen<3>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream);
            case 6: return launch_knn1_screen<6>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream);
            case 12: return launch_knn1_screen<12>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream);
            case 24: return launch_knn1_screen<24>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream);
            default: break;
        }
    }
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

// The screened K = 1 search in two calls, for references that stay the same over many searches (the ground-truth segments of a batch:
// their planes are prepared once per batch, off the training step's own stream): phase 1 = mp_knn1_prepare_f32 fills `workspace` from
// (p2, len2); phase 2 = mp_knn1_prepared_f32 searches against it.  Same outputs as mp_knn_f32 / mp_knn1_f32(screened).
static int knn1_phase(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B, int64_t P1, int64_t P2, int64_t D,
                      float* dists, int64_t* idx, void* ws, size_t ws_bytes, int phases, mp_stream_t stream_)
{
    if (B < 0 || P1 < 0 || P2 <= 0 || D <= 0) return MP_EINVAL;
    if (B == 0 || ((phases & 2) && P1 == 0)) return MP_OK;
    if (!p2 || !ws || ((phases & 2) && (!p1 || !dists || !idx))) return MP_EINVAL;
    if (B > 65535 || P1 > (1 << 30) || P2 > (1 << 30) || mp_knn1_workspace_bytes(B, P2, D) == 0) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n1 = (int)P1, n2 = (int)P2;
    switch ((int)D) {
        case 3: return launch_knn1_screen<3>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream, phases);
        case 6: return launch_knn1_screen<6>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream, phases);
        case 12: return launch_knn1_screen<12>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream, phases);
        case 24: return launch_knn1_screen<24>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream, phases);
        default: return MP_EUNSUPPORTED;
    }
}

extern "C" int mp_knn1_prepare_f32(const float* p2, const int64_t* len2, int64_t B, int64_t P2, int64_t D, void* ws, size_t ws_bytes,
                                   mp_stream_t stream_)
{
    return knn1_phase(nullptr, p2, nullptr, len2, B, 0, P2, D, nullptr, nullptr, ws, ws_bytes, 1, stream_);
}

extern "C" int mp_knn1_prepared_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B, int64_t P1,
                                    int64_t P2, int64_t D, float* dists, int64_t* idx, const void* ws, size_t ws_bytes, mp_stream_t stream_)
{
    return knn1_phase(p1, p2, len1, len2, B, P1, P2, D, dists, idx, const_cast<void*>(ws), ws_bytes, 2, stream_);
}

// The two K = 1 implementations behind mp_knn_f32, callable by name (tests compare them; tools time them): `screened` != 0 is the
// matrix-core screened search (D in {3, 6, 12, 24}), 0 the direct VALU scan.  Same outputs, bit for bit.
extern "C" int mp_knn1_f32(const float* p1, const float* p2, const int64_t* len1, const int64_t* len2, int64_t B, int64_t P1,
                           int64_t P2, int64_t D, float* dists, int64_t* idx, int screened, void* ws, size_t ws_bytes, mp_stream_t stream_)
{
    if (B < 0 || P1 < 0 || P2 < 0 || D <= 0) return MP_EINVAL;
    if (B == 0 || P1 == 0) return MP_OK;
    if (!p1 || !dists || !idx || (P2 > 0 && !p2)) return MP_EINVAL;
    if (B > 65535 || P1 > (1 << 30) || P2 > (1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n1 = (int)P1, n2 = (int)P2;
    switch ((int)D) {
        case 3: return screened ? launch_knn1_screen<3>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream) : launch_knn1<3>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
        case 6: return screened ? launch_knn1_screen<6>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream) : launch_knn1<6>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
        case 12: return screened ? launch_knn1_screen<12>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream) : launch_knn1<12>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
        case 24: return screened ? launch_knn1_screen<24>(p1, p2, len1, len2, b, n1, n2, dists, idx, ws, ws_bytes, stream) : launch_knn1<24>(p1, p2, len1, len2, b, n1, n2, dists, idx, stream);
        default: return MP_EUNSUPPORTED;
    }
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
    // `deterministic` carries flags: bit 0 = the ordered (bit-reproducible) grad_p2, bit 1 = grad_p1 ACCUMULATES ([r4] several loss terms
    // on the same prediction share one gradient buffer: no fan-out adds)
    RowGrad rg{grad_out, point_mean, batch_mode, (float)div, (float)scale, (deterministic & 2) ? 1 : 0};
    return knn_bwd(p1, p2, len1, len2, idx, nullptr, rg, B, P1, P2, D, 1, grad_p1, grad_p2, deterministic & 1, stream_);
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
