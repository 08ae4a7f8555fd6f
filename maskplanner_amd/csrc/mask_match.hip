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
// fp64 (LDS atomics) and rounded once to fp32 -- the reference's cost dtype -- then the LAP runs in
// fp64 on those fp32 values exactly like scipy does.
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

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
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
__device__ bool lsap_wave(const float* cost, int M, int Kb, int* match)
{
    const int lane = threadIdx.x & 63;
    const bool transpose = Kb < M;  // scipy solves the transposed problem when there are more rows than columns
    const int nr = transpose ? Kb : M;
    const int nc = transpose ? M : Kb;
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
                const double c = (double)(transpose ? cost[lane * CAP + i] : cost[i * CAP + lane]);
                const double r = minVal + c - ui - v;
                if (r < spc) { path = i; spc = r; }
            }
            const double lowest = wave_min_f64(active ? spc : INF);
            const bool is_min = active && spc == lowest;
            // scipy's scan: first minimum in `remaining` order, replaced by any later minimum that is a free column
            const unsigned fpos = mp::wave_min_u32(is_min ? (unsigned)pos : 0xffffffffu);
            const unsigned lfree = mp::wave_max_u32((is_min && row4col == -1) ? (unsigned)pos + 1u : 0u);
            const int chosen_pos = lfree ? (int)lfree - 1 : (int)fpos;
            const unsigned long long cm = __ballot(active && pos == chosen_pos);
            if (cm == 0ull) { sink = -2; break; }  // infeasible (non-finite costs): leave unmatched
            const int j = __builtin_ctzll(cm);
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
    float* red = uniq + CAP;                                   // [MM_THREADS/64]
    int* match = reinterpret_cast<int*>(red + MM_THREADS / 64);  // [CAP]
    int* rank = match + CAP;                                   // [S]
    __shared__ float s_last;
    __shared__ int s_n, s_bad, s_pad, s_nan;

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* ids = target_ids + (size_t)b * S;
    const float* pm = pred_masks + (size_t)b * M * S;
    const float* tv = target_value ? target_value + (size_t)b * S : nullptr;   // smooth targets: MSE cost (:830)

    for (int e = tid; e < CAP * CAP + CAP; e += MM_THREADS) accB[e] = 0.0;  // accB and accA are contiguous
    if (tid == 0) { s_last = -__builtin_inff(); s_n = 0; s_bad = (M > CAP) ? 1 : 0; s_pad = 0; s_nan = 0; }
    __syncthreads();

    // 1. unique ids ascending (torch.unique), -1 skipped: repeated "smallest value above the last one"
    for (int it = 0; it < CAP + 2; ++it) {
        const float last = s_last;
        float mn = __builtin_inff();
        for (int s = tid; s < S; s += MM_THREADS) {
            const float x = ids[s];
            if (x > last && x < mn) mn = x;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
        if (lane == 0) red[wave] = mn;
        __syncthreads();
        if (tid == 0) {
            float g = red[0];
            for (int w = 1; w < MM_THREADS / 64; ++w) g = fminf(g, red[w]);
            s_last = g;
            if (g != __builtin_inff() && g != -1.0f) {
                if (s_n < CAP) uniq[s_n] = g; else s_bad = 1;
                s_n = s_n + 1;
            }
        }
        __syncthreads();
        if (s_last == __builtin_inff()) break;
    }
    const int Kb = min(s_n, CAP);
    const bool bad = s_bad != 0;
    if (bad) {
        if (tid == 0) { status[b] = MP_MATCH_TOO_MANY_IDS; n_targets[b] = s_n; }
        for (int m = tid; m < M; m += MM_THREADS) match_col[(size_t)b * M + m] = -1;
        return;
    }
    // 2. rank of every segment's id among the unique ids (-1 for the padding id, which the reference asserts is never a
    // target: loss_handler.py:852)
    for (int s = tid; s < S; s += MM_THREADS) {
        const float x = ids[s];
        int r = -1;
        for (int k = 0; k < Kb; ++k) r = (uniq[k] == x) ? k : r;
        rank[s] = r;
        if (r < 0) s_pad = 1;      // benign race: every writer stores 1
    }
    __syncthreads();
    // 3. cost sums.  Binary targets: sum_s BCE(x, t) = sum_s f0(x) - sum_{s in mask k} x.  Smooth targets (target value
    // c_s inside the mask, 0 outside): sum_s (x - t)^2 = sum_s x^2 - sum_{s in mask k} (2*x*c_s - c_s^2).
    for (int m = 0; m < M; ++m) {
        double a = 0.0;
        for (int s = tid; s < S; s += MM_THREADS) {
            const float x = pm[(size_t)m * S + s];
            const int r = rank[s];
            if (tv) {
                a += (double)x * (double)x;
                const double c = (double)tv[s];
                if (r >= 0) atomicAdd(&accB[m * CAP + r], 2.0 * (double)x * c - c * c);
            } else {
                const float f0 = fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
                a += (double)f0;
                if (r >= 0) atomicAdd(&accB[m * CAP + r], (double)x);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) atomicAdd(&accA[m], a);
    }
    __syncthreads();
    for (int e = tid; e < M * CAP; e += MM_THREADS) {
        const int m = e / CAP, k = e - m * CAP;
        const float c = k < Kb ? (float)(accA[m] - accB[m * CAP + k]) : 0.0f;
        cost[e] = c;
        if (c != c) s_nan = 1;
        if (cost_out) cost_out[((size_t)b * M + m) * CAP + k] = c;
    }
    for (int k = tid; k < CAP; k += MM_THREADS) uniq_ids[(size_t)b * CAP + k] = k < Kb ? uniq[k] : 0.0f;
    __syncthreads();
    // 4. LAP on one wave
    if (wave == 0) {
        bool feasible = s_nan == 0;          // scipy: NaN entries are rejected before the solve
        if (Kb > 0 && feasible) feasible = lsap_wave(cost, M, Kb, match);
        else if (lane < M) match[lane] = -1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < M) match_col[(size_t)b * M + lane] = (int64_t)match[lane];
        if (lane == 0) {
            n_targets[b] = Kb;
            status[b] = (feasible ? 0 : MP_MATCH_INFEASIBLE) | (s_pad ? MP_MATCH_PADDING_ID : 0);
        }
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
    const size_t smem = sizeof(double) * (CAP * CAP + CAP) + sizeof(float) * (CAP * CAP + CAP + MM_THREADS / 64) +
                        sizeof(int) * (CAP + (size_t)S);
    static mp::DynLds lds;   // see common.h
    if (!lds.ensure(reinterpret_cast<const void*>(mask_match_kernel), smem)) return MP_ELAUNCH;
    hipLaunchKernelGGL(mask_match_kernel, dim3((unsigned)B), dim3(MM_THREADS), smem, mp_stream(stream_), pred_masks,
                       target_ids, target_value, (int)M, (int)S, match_col, uniq_ids, n_targets, cost, status);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
