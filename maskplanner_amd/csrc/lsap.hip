// Rectangular linear sum assignment for segment matching (up to 2048 x 2048 per sample), batched on device (one wave per sample up to
// 128 columns: lsap_kernel below; four waves beyond: lsap4_kernel).
//
// Reference: models/hungarianMatcher.py:58-61 copies a [S, Sgt] Euclidean cost matrix per sample to the host and calls
// scipy.optimize.linear_sum_assignment (999 x ~900: ~0.2 s per sample on one core, serial over the batch).  Here one
// WAVE per sample runs the same algorithm -- scipy's rectangular_lsap (shortest augmenting paths, Crouse 2016) in
// fp64 on the fp32 costs, including the scan order that decides ties -- with all dual / path / assignment state in LDS
// and the samples of a batch side by side on different CUs.  The small (<= 64 x 64) stroke-mask LAP of the training
// step keeps everything in registers instead (mask_match.hip); this is the large, off-step variant.
//
// Every "scan the remaining columns" loop of the serial algorithm is one lane-parallel pass (a lane owns columns
// lane, lane+64, ...: the cost row is fetched with coalesced loads issued together) followed by a wave-wide
// lexicographic arg-min: lowest shortest-path cost first; among equal costs scipy's scan keeps the LAST free column
// it meets, else the FIRST assigned one, in the order of its `remaining` array -- reproduced through pos[] / rem[].
// A single wave needs no barriers: LDS operations of one wave execute in program order.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int LSAP_MAX = 2048;          // columns per sample: up to 32 per lane

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = mp::dpp_u32<CTRL>((unsigned)u), hi = mp::dpp_u32<CTRL>((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// wave-wide minimum, uniform in every lane: four DPP steps inside the 16-lane rows, then the four row results by readlane -- no LDS
// round trips ([r4]: the xor-shuffle form was twelve dependent ds_bpermute on the serial path of every Dijkstra step)
__device__ __forceinline__ double wave_min_f64(double v)
{
    double t;
    t = dpp_f64<mp::DPP_QUAD_XOR1>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_QUAD_XOR2>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_ROW_HALF_MIRROR>(v); v = t < v ? t : v;
    t = dpp_f64<mp::DPP_ROW_MIRROR>(v); v = t < v ? t : v;
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * i);
        r[i] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    r[0] = r[1] < r[0] ? r[1] : r[0];
    r[2] = r[3] < r[2] ? r[3] : r[2];
    return r[2] < r[0] ? r[2] : r[0];
}

__device__ __forceinline__ void wave_sync()
{   // orders this wave's LDS traffic across divergent single-lane sections (compiler + memory model; no s_barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LSAP_T>   // columns per lane (8 / 16 / 32): Cmax <= 64 * LSAP_T
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost, int64_t batch_stride, int ld,
                                                  const int32_t* __restrict__ nr_, const int32_t* __restrict__ nc_,
                                                  int Rmax, int Cmax, int64_t* __restrict__ col4row_out,
                                                  int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* v = reinterpret_cast<double*>(smem_raw);     // [Cmax] column duals
    double* spc = v + Cmax;                              // [Cmax] shortest path costs
    double* u = spc + Cmax;                              // [Rmax] row duals
    int* path = reinterpret_cast<int*>(u + Rmax);        // [Cmax]
    int* row4col = path + Cmax;                          // [Cmax]
    int* pos = row4col + Cmax;                           // [Cmax] position of a column in `remaining`
    int* rem = pos + Cmax;                               // [Cmax] scipy's `remaining`
    int* SC = rem + Cmax;                                // [Cmax]
    int* col4row = SC + Cmax;                            // [Rmax]
    int* SR = col4row + Rmax;                            // [Rmax]

    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int nr = nr_[b], nc = nc_[b];
    int64_t* out = col4row_out + (size_t)b * Rmax;
    if (nr < 0 || nc < 0 || nr > nc || nc > Cmax || nr > Rmax || nc > 64 * LSAP_T) {
        if (lane == 0) status[b] = MP_EINVAL;
        for (int i = lane; i < Rmax; i += 64) out[i] = -1;
        return;
    }
    const float* C = cost + (size_t)b * batch_stride;
    const double INF = __builtin_inf();
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    wave_sync();

    bool feasible = true;
    for (int cur = 0; cur < nr && feasible; ++cur) {
        for (int j = lane; j < nc; j += 64) {
            spc[j] = INF;
            SC[j] = 0;
            pos[j] = nc - 1 - j;                 // remaining[it] = nc - it - 1
            rem[nc - 1 - j] = j;
        }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        wave_sync();
        int num_remaining = nc;
        int i = cur;
        double minVal = 0.0;
        int sink = -1;
        while (sink == -1) {
            const double ui = u[i];
            const float* row = C + (size_t)i * ld;
            float c[LSAP_T];
#pragma unroll
            for (int t = 0; t < LSAP_T; ++t) {
                const int j = lane + 64 * t;
                c[t] = (j < nc) ? row[j] : 0.0f;     // the whole row in flight at once
            }
            double bv = INF;
            unsigned bkey = 0u;
#pragma unroll
            for (int t = 0; t < LSAP_T; ++t) {
                const int j = lane + 64 * t;
                if (j < nc && !SC[j]) {
                    const double r = minVal + (double)c[t] - ui - v[j];
                    double s = spc[j];
                    if (r < s) { s = r; spc[j] = r; path[j] = i; }
                    const unsigned p = (unsigned)pos[j];
                    const unsigned key = (row4col[j] == -1) ? 0x80000000u + p : 0x7fffffffu - p;
                    if (s < bv || (s == bv && key > bkey)) { bv = s; bkey = key; }
                }
            }
            const double lowest = wave_min_f64(bv);
            const unsigned key = mp::wave_max_u32(bv == lowest ? bkey : 0u);
            if (!(lowest < INF) || key == 0u) { feasible = false; break; }   // infeasible: non-finite costs
            const int chosen_pos = (key & 0x80000000u) ? (int)(key - 0x80000000u) : (int)(0x7fffffffu - key);
            const int j = rem[chosen_pos];
            minVal = lowest;
            const int r4c = row4col[j];
            --num_remaining;
            if (lane == 0) {
                SR[i] = 1;
                SC[j] = 1;
                const int last = rem[num_remaining];     // remaining[index] = remaining[--num_remaining]
                rem[chosen_pos] = last;
                pos[last] = chosen_pos;
            }
            wave_sync();
            if (r4c == -1) sink = j; else i = r4c;
        }
        if (!feasible) break;
        // dual updates (rows in the tree other than cur, columns in the tree)
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        wave_sync();
        // augment along the path (serial chain; every lane walks it, lane 0 writes)
        int j = sink;
        for (;;) {
            const int pi = path[j];
            const int t = col4row[pi];
            wave_sync();                                   // all lanes have read before lane 0 overwrites
            if (lane == 0) { row4col[j] = pi; col4row[pi] = j; }
            wave_sync();
            j = t;
            if (pi == cur) break;
        }
    }
    for (int i = lane; i < Rmax; i += 64) out[i] = (feasible && i < nr) ? (int64_t)col4row[i] : -1;
    if (lane == 0) status[b] = feasible ? 0 : MP_EUNSUPPORTED;
}

// ---- [r4] the same algorithm with FOUR waves per sample --------------------------------------------------------------------
// One wave per sample spends most of a Dijkstra step in its own vector work: 16 columns per lane x ~20 instructions (fp64 adds,
// LDS reads of the column state) on ONE of the CU's four SIMDs.  Here a sample has a 256-thread workgroup, a thread owns the columns
// tid + 256 t with their dual, shortest-path cost, assignment and scan position in REGISTERS (path[], rem[], u[], the row / column
// assignments stay in LDS: they are accessed by index), every wave reduces its own columns to one candidate (value, key, column,
// assigned row) and the four candidates meet in LDS behind one barrier per step.  Dual updates run from the column side:
// u[row4col[j]] += minVal - spc[j] for every scanned column j but the sink -- the same expression as scipy's row loop, since
// col4row[row4col[j]] = j.  The augmentation along the path is wave 0's (serial chain), the other waves reload their assignments
// after it.  Tie-breaking as above (key from the column's position in scipy's `remaining`).
struct LsapSlot { double val; unsigned key; int j; int r4c; int pad; };
#ifndef LSAP_SPEC
#define LSAP_SPEC 1
#endif

template <int T, int NW = 4>   // columns per thread: Cmax <= 64 * NW * T
__global__ __launch_bounds__(64 * NW) void lsap4_kernel(const float* __restrict__ cost, int64_t batch_stride, int ld,
                                                    const int32_t* __restrict__ nr_, const int32_t* __restrict__ nc_,
                                                    int Rmax, int Cmax, int64_t* __restrict__ col4row_out,
                                                    int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* u = reinterpret_cast<double*>(smem_raw);                 // [Rmax] row duals
    LsapSlot* slots = reinterpret_cast<LsapSlot*>(u + Rmax);          // [2][NW] (room for 8)
    int* path = reinterpret_cast<int*>(slots + 16);                    // [Cmax]
    int* row4col = path + Cmax;                                       // [Cmax]
    int* rem = row4col + Cmax;                                        // [Cmax] scipy's `remaining`
    int* col4row = rem + Cmax;                                        // [Rmax]

    constexpr int NT = 64 * NW;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = nr_[b], nc = nc_[b];
    int64_t* out = col4row_out + (size_t)b * Rmax;
    if (nr < 0 || nc < 0 || nr > nc || nc > Cmax || nr > Rmax || nc > NT * T) {
        if (tid == 0) status[b] = MP_EINVAL;
        for (int i = tid; i < Rmax; i += NT) out[i] = -1;
        return;
    }
    const float* C = cost + (size_t)b * batch_stride;
    const double INF = __builtin_inf();
    double v[T], spc[T];
    int r4c[T], pos[T];
    float cs[T];              // a prefetched cost row (this thread's columns) and which row it is
    int spec_row = -1;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        v[t] = 0.0; r4c[t] = -1; spc[t] = INF; pos[t] = 0; cs[t] = 0.0f;
        const int j = tid + NT * t;
        if (j < nc) row4col[j] = -1;
    }
    for (int i = tid; i < nr; i += NT) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();

    bool feasible = true;
    for (int cur = 0; cur < nr && feasible; ++cur) {
        unsigned scanned = 0u;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = tid + NT * t;
            spc[t] = INF;
            pos[t] = nc - 1 - j;                 // remaining[it] = nc - it - 1
            if (j < nc) rem[nc - 1 - j] = j;
        }
        __syncthreads();
        int num_remaining = nc;
        int i = cur;
        double minVal = 0.0;
        int sink = -1;
        int parity = 0;
        while (sink == -1) {
            const double ui = u[i];
            const float* row = C + (size_t)i * ld;
            float c[T];
            if (LSAP_SPEC && i == spec_row) {      // the row was requested a step ago as the runner-up's (below)
#pragma unroll
                for (int t = 0; t < T; ++t) c[t] = cs[t];
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int j = tid + NT * t;
                    c[t] = (j < nc) ? row[j] : 0.0f;
                }
            }
            double bv = INF;
            unsigned bkey = 0u;
            int bj = -1, br = -1;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = tid + NT * t;
                if (j < nc && !((scanned >> t) & 1u)) {
                    const double r = minVal + (double)c[t] - ui - v[t];
                    if (r < spc[t]) { spc[t] = r; path[j] = i; }
                    const double s_ = spc[t];
                    const unsigned p = (unsigned)pos[t];
                    const unsigned key = (r4c[t] == -1) ? 0x80000000u + p : 0x7fffffffu - p;
                    if (s_ < bv || (s_ == bv && key > bkey)) { bv = s_; bkey = key; bj = j; br = r4c[t]; }
                }
            }
            const double wlow = wave_min_f64(bv);
            const unsigned wkey = mp::wave_max_u32(bv == wlow ? bkey : 0u);
            LsapSlot* sl = slots + parity * NW + wave;
            if (wkey == 0u) { if (lane == 0) { sl->val = INF; sl->key = 0u; sl->j = -1; sl->r4c = -1; } }
            else if (bv == wlow && bkey == wkey) { sl->val = bv; sl->key = bkey; sl->j = bj; sl->r4c = br; }     // keys are unique: one lane
            __syncthreads();
            double lowest = INF, second = INF;
            unsigned key = 0u, key2 = 0u;
            int jstar = -1, rstar = -1, r2 = -1;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const LsapSlot q = slots[parity * NW + w];
                if (q.key != 0u && (q.val < lowest || (q.val == lowest && q.key > key))) {
                    second = lowest; key2 = key; r2 = rstar;
                    lowest = q.val; key = q.key; jstar = q.j; rstar = q.r4c;
                } else if (q.key != 0u && (q.val < second || (q.val == second && q.key > key2))) {
                    second = q.val; key2 = q.key; r2 = q.r4c;
                }
            }
            if (!(lowest < INF) || key == 0u) { feasible = false; break; }       // infeasible: non-finite costs
            // the best candidate of the OTHER waves is the likely next minimum (unless the row about to be scanned, or the winner's own
            // wave, beats it): its assigned row is requested now, a step ahead of its use -- the cost matrix never changes, so a
            // prefetched row stays valid for as long as it is kept
            if (LSAP_SPEC && rstar != -1 && r2 != -1 && r2 != spec_row && key2 != 0u) {
                const float* row2 = C + (size_t)r2 * ld;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int j = tid + NT * t;
                    cs[t] = (j < nc) ? row2[j] : 0.0f;
                }
                spec_row = r2;
            }
            const int chosen_pos = (key & 0x80000000u) ? (int)(key - 0x80000000u) : (int)(0x7fffffffu - key);
            --num_remaining;
            const int last = rem[num_remaining];                 // remaining[index] = remaining[--num_remaining]
            if (tid == 0) rem[chosen_pos] = last;                // (chosen_pos == num_remaining: the same value)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = tid + NT * t;
                if (j == last) pos[t] = chosen_pos;
                if (j == jstar) scanned |= 1u << t;
            }
            minVal = lowest;
            if (rstar == -1) sink = jstar; else i = rstar;
            parity ^= 1;
        }
        if (!feasible) break;
        // dual updates from the column side (rows in the tree other than cur <-> scanned columns other than the sink)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if ((scanned >> t) & 1u) {
                const double d = minVal - spc[t];
                v[t] -= d;
                if (r4c[t] != -1) u[r4c[t]] += d;
            }
        }
        if (tid == 0) u[cur] += minVal;
        __syncthreads();
        if (wave == 0) {     // augment along the path: a serial chain, one wave (its LDS traffic is ordered by wave_sync)
            int j = sink;
            for (;;) {
                const int pi = path[j];
                const int t_ = col4row[pi];
                wave_sync();
                if (lane == 0) { row4col[j] = pi; col4row[pi] = j; }
                wave_sync();
                j = t_;
                if (pi == cur) break;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = tid + NT * t;
            if (j < nc) r4c[t] = row4col[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < Rmax; i += NT) out[i] = (feasible && i < nr) ? (int64_t)col4row[i] : -1;
    if (tid == 0) status[b] = feasible ? 0 : MP_EUNSUPPORTED;
}

size_t lsap4_smem(int64_t Rmax, int64_t Cmax)
{
    return sizeof(double) * (size_t)Rmax + sizeof(LsapSlot) * 16 + sizeof(int) * (size_t)(3 * Cmax + Rmax);
}

size_t lsap_smem(int64_t Rmax, int64_t Cmax)
{
    return sizeof(double) * (size_t)(2 * Cmax + Rmax) + sizeof(int) * (size_t)(5 * Cmax + 2 * Rmax);
}

}  // namespace

extern "C" int mp_lsap_f32(const float* cost, int64_t B, int64_t Rmax, int64_t Cmax, int64_t ld, int64_t batch_stride,
                           const int32_t* n_rows, const int32_t* n_cols, int64_t* col4row, int32_t* status,
                           mp_stream_t stream_)
{
    if (B < 0 || Rmax < 0 || Cmax < 0 || ld < Cmax) return MP_EINVAL;
    if (B == 0) return MP_OK;
    if (!n_rows || !n_cols || !col4row || !status || (Rmax * Cmax > 0 && !cost)) return MP_EINVAL;
    if (Cmax > LSAP_MAX || Rmax > LSAP_MAX || B > 65535 * 32) return MP_EUNSUPPORTED;
    const size_t smem = lsap_smem(Rmax, Cmax);
    if (smem > 160 * 1024) return MP_EUNSUPPORTED;
    auto launch = [&](auto kernel, mp::DynLds& lds) -> int {
        if (!lds.ensure(reinterpret_cast<const void*>(kernel), smem)) return MP_ELAUNCH;   // see common.h
        MP_LAUNCH("lsap_kernel", 0.0, 4.0 * (double)(B * Rmax * Cmax), kernel, dim3((unsigned)B), dim3(64), smem, mp_stream(stream_), cost,
                  batch_stride, (int)ld, n_rows, n_cols, (int)Rmax, (int)Cmax, col4row, status);
        MP_CHECK_LAUNCH();
        return MP_OK;
    };
    // [r4] four waves per sample (lsap4_kernel) where a sample has enough columns to feed them; MP_LSAP_WAVES=1: the one-wave kernel
    static const bool four = []() { const char* e = getenv("MP_LSAP_WAVES"); return !(e && e[0] == '1'); }();
    if (four && Cmax > 128) {
        const size_t smem4 = lsap4_smem(Rmax, Cmax);
        auto launch4 = [&](auto kernel, mp::DynLds& lds, int nt) -> int {
            if (!lds.ensure(reinterpret_cast<const void*>(kernel), smem4)) return MP_ELAUNCH;
            MP_LAUNCH("lsap4_kernel", 0.0, 4.0 * (double)(B * Rmax * Cmax), kernel, dim3((unsigned)B), dim3((unsigned)nt), smem4, mp_stream(stream_), cost,
                      batch_stride, (int)ld, n_rows, n_cols, (int)Rmax, (int)Cmax, col4row, status);
            MP_CHECK_LAUNCH();
            return MP_OK;
        };
        static mp::DynLds c2, c4, c8;
        // (measured with 2 / 4 / 8 waves per sample at 999 x ~900: 36.2 / 35.2 / 43.5 ms)
        if (Cmax <= 512) return launch4(lsap4_kernel<2>, c2, 256);
        if (Cmax <= 1024) return launch4(lsap4_kernel<4>, c4, 256);
        return launch4(lsap4_kernel<8>, c8, 256);
    }
    static mp::DynLds conf8, conf16, conf32;
    if (Cmax <= 512) return launch(lsap_kernel<8>, conf8);
    if (Cmax <= 1024) return launch(lsap_kernel<16>, conf16);
    return launch(lsap_kernel<32>, conf32);
}

// ---- batched Euclidean cost for the segment matcher ---------------------------------------------------------------------
// models/hungarianMatcher.py:44-55: the reference builds ONE torch.cdist matrix [B*S, sum Sgt] (every prediction against the
// ground truth of EVERY sample) and slices the diagonal blocks.  Here the B blocks are computed directly, in one launch, into the
// padded layout mp_lsap_f32 reads -- transposed where a sample has more predictions than targets, as scipy transposes then.
// cost = sqrt(sum_d (x_d - y_d)^2), direct differences in fp32 (fma chain over d).
namespace {
__global__ __launch_bounds__(256) void cdist_batch_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const int64_t* __restrict__ offsets, int S, int D, int Rmax, int Cmax,
                                                          float* __restrict__ cost, int32_t* __restrict__ n_rows,
                                                          int32_t* __restrict__ n_cols)
{
    const int b = blockIdx.z;
    const int64_t o0 = offsets[b];
    const int T = (int)(offsets[b + 1] - o0);
    const bool tr = S > T;                       // rows > columns: the solver works on the transpose
    const int nr = tr ? T : S, nc = tr ? S : T;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { n_rows[b] = nr; n_cols[b] = nc; }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= Rmax || c >= Cmax) return;
    float v = 0.0f;                              // padding of the [Rmax, Cmax] slab
    if (r < nr && c < nc) {
        const float* xp = x + ((size_t)b * S + (tr ? c : r)) * D;
        const float* yp = y + (size_t)(o0 + (tr ? r : c)) * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float df = xp[d] - yp[d];
            acc = __builtin_fmaf(df, df, acc);
        }
        v = __builtin_sqrtf(acc);
    }
    cost[((size_t)b * Rmax + r) * Cmax + c] = v;
}
}  // namespace

extern "C" int mp_cdist_batch_f32(const float* outputs, const float* targets, const int64_t* offsets, int64_t B, int64_t S, int64_t D,
                                  int64_t Rmax, int64_t Cmax, float* cost, int32_t* n_rows, int32_t* n_cols, mp_stream_t stream_)
{
    if (B < 0 || S < 0 || D <= 0 || Rmax < 0 || Cmax < 0) return MP_EINVAL;
    if (B == 0 || Rmax == 0 || Cmax == 0) return MP_OK;
    if (!offsets || !cost || !n_rows || !n_cols || (S > 0 && !outputs)) return MP_EINVAL;
    if (B > 65535 || Rmax > (1 << 20) || Cmax > (1 << 20)) return MP_EUNSUPPORTED;
    MP_LAUNCH("cdist_batch_kernel", 3.0 * (double)B * Rmax * Cmax * D, 4.0 * (double)B * Rmax * Cmax, cdist_batch_kernel,
              dim3((unsigned)((Cmax + 63) / 64), (unsigned)((Rmax + 3) / 4), (unsigned)B), dim3(256), 0, mp_stream(stream_), outputs, targets,
              offsets, (int)S, (int)D, (int)Rmax, (int)Cmax, cost, n_rows, n_cols);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

