// Stroke-mask Hungarian matching on device: unique target ids -> BCE cost matrix -> rectangular LAP.
//
// Reference: loss_handler.py:838-875.  Per sample the reference builds Kb binary target masks from the
// per-segment target stroke ids (:938-967, torch.unique => ascending ids, -1 skipped), evaluates
// BCE-with-logits of every (pred mask, target mask) pair (:863-872), copies the [M,Kb] cost to the host
// and calls scipy.optimize.linear_sum_assignment (:875) -- B host syncs per step.  Here one workgroup per
// sample does all of it on chip and leaves the assignment in device memory.
//
//   cost[m,k] = sum_s BCE(x_ms, t_ks),  t_ks = [id_s == uid_k]
//             = sum_s f0(x_ms) - sum_{s: id_s == uid_k} x_ms,   f0(x) = max(x,0) + log1p(exp(-|x|))
// (BCE(x,0) = f0(x), BCE(x,1) = f0(x) - x: ATen's (1-t)*x - log_sigmoid(x).)  Sums are accumulated in
// fp64 (the masked sums as a one-hot product on the fp64 MFMA) and rounded once to fp32 -- the reference's
// cost dtype -- then the LAP runs in fp64 on those fp32 values exactly like scipy does.
//
// LAP: scipy's rectangular_lsap (shortest augmenting path, Crouse 2016) with its tie-breaking, executed by
// ONE wave: lane j owns column j (dual v, shortest-path cost, predecessor, assignment, scan position),
// lane i owns row i (dual u, assignment, visited flag).  Every "scan the remaining columns" loop of the
// serial algorithm becomes one lane-parallel update + one wave-wide min; the serial scan order that
// decides ties is reproduced through each column's position in scipy's `remaining` array.
#include "common.h"

namespace {

constexpr int MM_THREADS = 1024;
constexpr int CAP = MP_MASK_CAP;
constexpr int IDW = 64;   // bitmap words of the integer-id route of the unique pass (ids < 2048)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = mp::dpp_u32<CTRL>((unsigned)u), hi = mp::dpp_u32<CTRL>((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __uint_as_float(mp::dpp_u32<CTRL>(__float_as_uint(v)));
}

// Minimum over the columns of the LAP (lanes), uniform in every lane: four DPP steps inside the 16-lane rows, then -- unless
// all columns sit in the first row (ONE) -- the four row results by readlane.  No LDS round trips: this is the serial path
// of the LAP.
template <bool ONE>
__device__ __forceinline__ double cols_min_f64(double v)
{
    double t;
    t = dpp_f64<mp::DPP_QUAD_XOR1>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_QUAD_XOR2>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_ROW_HALF_MIRROR>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_ROW_MIRROR>(v); v = t < v ? t : v;
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int i = 0; i < (ONE ? 1 : 4); ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * i);
        r[i] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (ONE) return r[0];
    r[0] = r[1] < r[0] ? r[1] : r[0];
    r[2] = r[3] < r[2] ? r[3] : r[2];
    return r[2] < r[0] ? r[2] : r[0];
}

template <bool ONE>
__device__ __forceinline__ unsigned cols_max_u32(unsigned v)
{
    if (!ONE) return mp::wave_max_u32(v);
    return (unsigned)__builtin_amdgcn_readlane((int)mp::row16_max_u32(v), 0);
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// One wave.  cost: LDS [M][CAP] fp32.  Returns through match[m] (LDS int [CAP]); false when the cost matrix is infeasible
// (non-finite entries: scipy raises "cost matrix is infeasible" / "matrix contains invalid numeric entries").
// ONE: every column fits the first 16 lanes (nc <= 16), which shortens the three reductions of each scan.
template <bool ONE>
__device__ bool lsap_wave(const float* cost, int M, int Kb, int* match)
{
    const int lane = threadIdx.x & 63;
    const bool transpose = Kb < M;  // scipy solves the transposed problem when there are more rows than columns
    const int nr = transpose ? Kb : M;
    const int nc = transpose ? M : Kb;
    const float* ccol = transpose ? cost + lane * CAP : cost + lane;   // cost of (row i, this lane's column): ccol[i * cstep]
    const int cstep = transpose ? 1 : CAP;
    // column state (lane j < nc)
    double v = 0.0, spc = 0.0;
    int path = -1, row4col = -1, pos = 0;
    bool SC = false;
    // row state (lane i < nr)
    double u = 0.0;
    int col4row = -1;
    bool SR = false;
    const double INF = __builtin_inf();

    for (int cur = 0; cur < nr; ++cur) {
        SR = false;
        SC = false;
        spc = INF;
        pos = nc - 1 - lane;  // remaining[it] = nc - it - 1
        int num_remaining = nc;
        int i = cur;
        double minVal = 0.0;
        int sink = -1;
        while (sink == -1) {
            if (lane == i) SR = true;
            const double ui = readlane_f64(u, i);
            const bool active = lane < nc && !SC;
            if (active) {
                const double c = (double)ccol[i * cstep];
                const double r = minVal + c - ui - v;
                if (r < spc) { path = i; spc = r; }
            }
            const double lowest = cols_min_f64<ONE>(active ? spc : INF);
            if (!(lowest < INF)) { sink = -2; break; }   // scipy: "cost matrix is infeasible" (non-finite costs)
            // scipy's scan keeps the first minimum in `remaining` order unless a later minimum is a free column, and then the
            // last of those: one max over (free ? 2^31 + pos : 2^31 - 1 - pos)
            const bool is_min = active && spc == lowest;
            const unsigned key = !is_min ? 0u : row4col == -1 ? 0x80000000u + (unsigned)pos : 0x7fffffffu - (unsigned)pos;
            const unsigned best = cols_max_u32<ONE>(key);
            const int j = __builtin_ctzll(__ballot(key == best));   // is_min holds somewhere: best > 0 and the keys are distinct
            const int chosen_pos = (int)((best & 0x80000000u) ? best - 0x80000000u : 0x7fffffffu - best);
            minVal = lowest;
            const int r4c = __builtin_amdgcn_readlane(row4col, j);
            if (r4c == -1) sink = j; else i = r4c;
            // remaining[index] = remaining[--num_remaining]
            --num_remaining;
            if (active && pos == num_remaining) pos = chosen_pos;
            if (lane == j) SC = true;
        }
        if (sink < 0) {
            if (lane < M) match[lane] = -1;
            return false;
        }
        // dual updates
        const double spc_of_row = __shfl(spc, col4row < 0 ? 0 : col4row, 64);
        if (lane == cur) u += minVal;
        else if (lane < nr && SR) u += minVal - spc_of_row;
        if (lane < nc && SC) v -= minVal - spc;
        // augment along the path
        int j = sink;
        for (;;) {
            const int pi = __builtin_amdgcn_readlane(path, j);
            if (lane == j) row4col = pi;
            const int t = __builtin_amdgcn_readlane(col4row, pi);
            if (lane == pi) col4row = j;
            j = t;
            if (pi == cur) break;
        }
    }
    if (lane < M) match[lane] = transpose ? row4col : col4row;
    return true;
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_min_f32(float v)
{
    v = fminf(v, dpp_f32<mp::DPP_QUAD_XOR1>(v));
    v = fminf(v, dpp_f32<mp::DPP_QUAD_XOR2>(v));
    v = fminf(v, dpp_f32<mp::DPP_ROW_HALF_MIRROR>(v));
    v = fminf(v, dpp_f32<mp::DPP_ROW_MIRROR>(v));
    return v;
}

__device__ __forceinline__ float wave_min_f32(float v)
{
    v = row16_min_f32(v);
    const float a = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    const float b = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    const float c = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    const float d = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}

// f0(x) = BCE-with-logits of x against target 0 = max(x,0) + log1p(exp(-|x|)), on the hardware exp2 / log2 (1 ulp each):
// t = 2^(-|x| log2 e) in (0,1], u = fl(1+t), log1p(t) = ln2*log2(u) + (t-(u-1))/u (the rounding of 1+t put back to first
// order).  Absolute error ~1e-7, the size of one fp32 ulp of the value; the library log1pf/expf pair costs ~190 issue slots
// per element and was two thirds of this kernel.
__device__ __forceinline__ float bce0(float x)
{
    const float t = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(x));
    const float u = 1.0f + t;
    const float l = fmaf(0.693147180559945309f, __builtin_amdgcn_logf(u), (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u));
    return fmaxf(x, 0.0f) + l;
}

// Cost sums.  Binary targets: sum_s BCE(x, t) = sum_s f0(x) - sum_{s in mask k} x.  Smooth targets (target value c_s inside
// the mask, 0 outside): sum_s (x - t)^2 = sum_s x^2 - sum_{s in mask k} c_s*(2*x - c_s).
// The masked sums are the product [M,S] x [S,Kb] of the logits with the one-hot ranks: fp64 MFMA
// (v_mfma_f64_16x16x4_f64; products with 0/1 are exact, the accumulation is fp64).  A work item is one block of 16 pred
// masks over one slice of the segments, against every block of 16 targets (KTM of them at most); lane (i = lane%16,
// q = lane/16) loads four consecutive segments of mask i, which feed the K slot q of four MFMAs.  U such loads are in flight
// together: the slices are short (80 segments at the bench shape) and the pass is bound by the load round trips.
// Non-finite logits: NaN/inf times the zeros of the one-hot operand is NaN, so such a sample reports MP_MATCH_INFEASIBLE
// (the reference: NaN cost -> scipy's ValueError; -inf logits alone would give +inf costs).
template <int KTM, int U>
__device__ __forceinline__ void mask_sums(const float* __restrict__ pm, const float* __restrict__ tv, const int* rank,
                                          double* accA, double* accB, int M, int S, int Kb)
{
    constexpr int NW = MM_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, q = lane >> 4;
    // where the accumulator registers of this lane sit in the 16x16 block: asked of the instruction itself
    int di[4], dj[4];
    {
        const v4f64 z = {0.0, 0.0, 0.0, 0.0};
        const v4f64 pi = __builtin_amdgcn_mfma_f64_16x16x4f64(q == 0 ? (double)li : 0.0, q == 0 ? 1.0 : 0.0, z, 0, 0, 0);
        const v4f64 pj = __builtin_amdgcn_mfma_f64_16x16x4f64(q == 0 ? 1.0 : 0.0, q == 0 ? (double)li : 0.0, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { di[r] = (int)pi[r]; dj[r] = (int)pj[r]; }
    }
    const int MT = max(1, (M + 15) >> 4), KT = (Kb + 15) >> 4;
    const int NS = NW / MT;                                   // slices of the segment axis
    const int SL = (((S + NS - 1) / NS) + 15) & ~15;
    const bool vec = (S & 3) == 0 && (reinterpret_cast<uintptr_t>(pm) & 15) == 0;
    for (int item = wave; item < MT * NS; item += NW) {
        const int mt = item % MT, sl = item / MT;
        const int m = mt * 16 + li;
        const float* row = pm + (size_t)min(m, max(M - 1, 0)) * S;
        const int s_end = min(S, (sl + 1) * SL);
        v4f64 acc[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) acc[kt] = v4f64{0.0, 0.0, 0.0, 0.0};
        double fa = 0.0;
        for (int s0 = sl * SL; s0 < s_end; s0 += 16 * U) {
            float x[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sb = s0 + 16 * u + q * 4;
                if (m < M && vec && sb + 3 < s_end) {
                    const float4 t = *reinterpret_cast<const float4*>(row + sb);
                    x[u][0] = t.x; x[u][1] = t.y; x[u][2] = t.z; x[u][3] = t.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[u][t] = (m < M && sb + t < s_end) ? row[sb + t] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + 16 * u >= s_end) break;
                const int sb = s0 + 16 * u + q * 4;
                const int4 rk = *reinterpret_cast<const int4*>(rank + sb);   // rank[] is padded with -1 to a multiple of 16
                const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
                float c[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (tv) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) c[t] = sb + t < s_end ? tv[sb + t] : 0.0f;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool in = m < M && sb + t < s_end;
                    double a;
                    if (tv) {
                        a = 2.0 * (double)x[u][t] - (double)c[t];
                        if (in) fa += (double)x[u][t] * (double)x[u][t];
                    } else {
                        a = (double)x[u][t];
                        if (in) fa += (double)bce0(x[u][t]);
                    }
                    if (!in) a = 0.0;
#pragma unroll
                    for (int kt = 0; kt < KTM; ++kt) {
                        if (kt < KT) {
                            const double bv = rks[t] == kt * 16 + li ? (double)c[t] : 0.0;
                            acc[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[kt], 0, 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
            if (kt < KT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mm = mt * 16 + di[r], kk = kt * 16 + dj[r];
                    if (mm < M && kk < Kb) atomicAdd(&accB[mm * CAP + kk], acc[kt][r]);
                }
            }
        }
        fa += __shfl_xor(fa, 16, 64);
        fa += __shfl_xor(fa, 32, 64);
        if (q == 0 && m < M) atomicAdd(&accA[m], fa);
    }
}

__global__ __launch_bounds__(MM_THREADS) void mask_match_kernel(const float* __restrict__ pred_masks,
                                                                const float* __restrict__ target_ids,
                                                                const float* __restrict__ target_value, int M, int S,
                                                                int64_t* __restrict__ match_col,
                                                                float* __restrict__ uniq_ids,
                                                                int64_t* __restrict__ n_targets,
                                                                float* __restrict__ cost_out,
                                                                int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* accB = reinterpret_cast<double*>(smem_raw);        // [CAP][CAP]  sum of logits inside each target mask
    double* accA = accB + CAP * CAP;                           // [CAP]       sum of f0 per pred mask
    float* cost = reinterpret_cast<float*>(accA + CAP);        // [CAP][CAP]
    float* uniq = cost + CAP * CAP;                            // [CAP]
    float* red = uniq + CAP;                                   // [2][MM_THREADS/64]
    int* match = reinterpret_cast<int*>(red + 2 * (MM_THREADS / 64));  // [CAP]
    int* rank = match + CAP;                                   // [S rounded up to 16]; holds the ids first
    float* idsL = reinterpret_cast<float*>(rank);
    __shared__ int s_pad, s_nan, s_general;
    __shared__ unsigned bm[IDW + 1];                           // presence bitmap of integer ids 0 .. 32*IDW-1
    __shared__ int pfx[IDW + 1];                               // distinct ids below each bitmap word

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NW = MM_THREADS / 64;
    const float* ids = target_ids + (size_t)b * S;
    const float* pm = pred_masks + (size_t)b * M * S;
    const float* tv = target_value ? target_value + (size_t)b * S : nullptr;   // smooth targets: MSE cost (:830)
    const int Sp = (S + 15) & ~15;
#ifdef MP_MM_TIMING   // phase timestamps of every sample into uniq_ids[b][48..55] (tools/mask_match_time.py --phases)
    long long tq[8];
    int tn = 0;
#define MM_TICK() tq[tn++] = (long long)wall_clock64()
#else
#define MM_TICK()
#endif
    MM_TICK();

    for (int e = tid; e < CAP * CAP + CAP; e += MM_THREADS) accB[e] = 0.0;  // accB and accA are contiguous
    if (tid < IDW + 1) bm[tid] = 0u;
    if (tid == 0) { s_pad = 0; s_nan = 0; s_general = 0; }
    __syncthreads();
    // 1+2. unique ids ascending (torch.unique; -1 skipped) and the rank of every segment's id among them (-1 for the padding
    // id, which the reference asserts is never a target: loss_handler.py:852).  Stroke ids are small whole numbers: when every
    // id is an integer in [-1, 32*IDW) a presence bitmap gives both in two passes; anything else takes the general route.
    for (int s = tid; s < S; s += MM_THREADS) {
        const float x = ids[s];
        idsL[s] = x;
        const int xi = (x >= -1.0f && x < (float)(32 * IDW)) ? (int)x : -2;
        if (xi == -2 || (float)xi != x) s_general = 1;      // benign race: every writer stores 1
        else if (xi >= 0) atomicOr(&bm[xi >> 5], 1u << (xi & 31));
    }
    __syncthreads();
    MM_TICK();
    int n_ids = 0;
    if (s_general == 0) {
        if (wave == 0) {                              // lane w: ids 32w .. 32w+31
            const unsigned word = bm[lane];
            const int cnt = __popc(word);
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            pfx[lane] = incl - cnt;
            if (lane == 63) pfx[64] = incl;
            int k = incl - cnt;
            for (unsigned wv = word; wv != 0u && k < CAP; wv &= wv - 1u, ++k) uniq[k] = (float)(32 * lane + __builtin_ctz(wv));
        }
        __syncthreads();
        n_ids = pfx[64];
        if (n_ids <= CAP) {
            for (int s = tid; s < Sp; s += MM_THREADS) {
                int r = -1;
                if (s < S) {
                    const int xi = (int)idsL[s];
                    if (xi >= 0) r = pfx[xi >> 5] + __popc(bm[xi >> 5] & ((1u << (xi & 31)) - 1u));
                    else s_pad = 1;
                }
                rank[s] = r;
            }
        }
    } else {
        // repeated "smallest value above the last one": every wave combines the 16 wave minima itself (double-buffered, one
        // barrier per id) and so carries the same count
        float last = -__builtin_inff();
        for (int it = 0; it < CAP + 2; ++it) {
            float mn = __builtin_inff();
            for (int s = tid; s < S; s += MM_THREADS) {
                const float x = idsL[s];
                if (x > last && x < mn) mn = x;
            }
            mn = wave_min_f32(mn);
            float* rb = red + (it & 1) * NW;
            if (lane == 0) rb[wave] = mn;
            __syncthreads();
            static_assert(NW == 16, "the wave minima fill one 16-lane row");
            const float g =
                __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(row16_min_f32(rb[lane & 15]))));
            if (g == __builtin_inff()) break;
            last = g;
            if (g != -1.0f) {
                if (n_ids < CAP && tid == 0) uniq[n_ids] = g;
                ++n_ids;
            }
        }
        __syncthreads();
        if (n_ids <= CAP) {
            for (int s = tid; s < Sp; s += MM_THREADS) {
                int r = -1;
                if (s < S) {
                    const float x = idsL[s];
                    for (int k = 0; k < n_ids; ++k) r = (uniq[k] == x) ? k : r;
                    if (r < 0) s_pad = 1;      // benign race: every writer stores 1
                }
                rank[s] = r;
            }
        }
    }
    const int Kb = min(n_ids, CAP);
    if (M > CAP || n_ids > CAP) {
        if (tid == 0) { status[b] = MP_MATCH_TOO_MANY_IDS; n_targets[b] = n_ids; }
        for (int m = tid; m < M; m += MM_THREADS) match_col[(size_t)b * M + m] = -1;
        return;
    }
    __syncthreads();
    MM_TICK();
    // 3. cost sums (mask_sums above)
    {
        // measured at the bench shape (6 masks, 1280 segments): 8.6 us, 2 of them the f64 MFMAs, the rest the per-element f0 /
        // convert / select work of 16 waves on one CU.  A separate f0 pass with the lanes laid flat (60 of 64 at work instead
        // of 6 rows of 16) paid its second load round trip and came out at 9.2 us.
        const int KT = (Kb + 15) >> 4;
        if (KT <= 1) mask_sums<1, 8>(pm, tv, rank, accA, accB, M, S, Kb);
        else mask_sums<4, 2>(pm, tv, rank, accA, accB, M, S, Kb);
    }
    __syncthreads();
    MM_TICK();
    for (int e = tid; e < M * CAP; e += MM_THREADS) {
        const int m = e / CAP, k = e - m * CAP;
        const float c = k < Kb ? (float)(accA[m] - accB[m * CAP + k]) : 0.0f;
        cost[e] = c;
        if (c != c) s_nan = 1;
        if (cost_out) cost_out[((size_t)b * M + m) * CAP + k] = c;
    }
    for (int k = tid; k < CAP; k += MM_THREADS) uniq_ids[(size_t)b * CAP + k] = k < Kb ? uniq[k] : 0.0f;
    __syncthreads();
    MM_TICK();
    // 4. LAP on one wave
    if (wave == 0) {
        bool feasible = s_nan == 0;          // scipy: NaN entries are rejected before the solve
        if (Kb > 0 && feasible) feasible = max(M, Kb) <= 16 ? lsap_wave<true>(cost, M, Kb, match) : lsap_wave<false>(cost, M, Kb, match);
        else if (lane < M) match[lane] = -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < M) match_col[(size_t)b * M + lane] = (int64_t)match[lane];
        if (lane == 0) {
            n_targets[b] = Kb;
            status[b] = (feasible ? 0 : MP_MATCH_INFEASIBLE) | (s_pad ? MP_MATCH_PADDING_ID : 0);
        }
#ifdef MP_MM_TIMING
        MM_TICK();
        if (lane == 0)
            for (int i = 1; i < tn; ++i) uniq_ids[(size_t)b * CAP + 47 + i] = (float)(tq[i] - tq[i - 1]) * 0.01f;
#endif
    }
}

}  // namespace

extern "C" int mp_mask_match_f32(const float* pred_masks, const float* target_ids, const float* target_value, int64_t B,
                                 int64_t M, int64_t S,
                                 int64_t* match_col, float* uniq_ids, int64_t* n_targets, float* cost,
                                 int32_t* status, mp_stream_t stream_)
{
    if (B < 0 || M < 0 || S < 0) return MP_EINVAL;
    if (B == 0) return MP_OK;
    if (!match_col || !uniq_ids || !n_targets || !status || (M * S > 0 && (!pred_masks || !target_ids)))
        return MP_EINVAL;
    if (M > CAP || S > 16384) return MP_EUNSUPPORTED;
    const size_t smem = sizeof(double) * (CAP * CAP + CAP) + sizeof(float) * (CAP * CAP + CAP + 2 * (MM_THREADS / 64)) +
                        sizeof(int) * (CAP + (((size_t)S + 15) & ~(size_t)15));
    static mp::DynLds lds;   // see common.h
    if (!lds.ensure(reinterpret_cast<const void*>(mask_match_kernel), smem)) return MP_ELAUNCH;
    hipLaunchKernelGGL(mask_match_kernel, dim3((unsigned)B), dim3(MM_THREADS), smem, mp_stream(stream_), pred_masks,
                       target_ids, target_value, (int)M, (int)S, match_col, uniq_ids, n_targets, cost, status);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
