// Farthest point sampling for gfx950.
//
// Reference: models/pointnet2_utils.py:65-86 (farthest_point_sample).  S dependent steps, each a
// "distance to the newest centroid, running min, arg-max" over the N points of one cloud.  This is a
// LATENCY-bound loop: the only lever is the length of one step.
//
// Design: one workgroup per cloud, the whole cloud resident on chip.
//   - coordinates and running min-distances live in VGPRs: thread t owns the PPT CONTIGUOUS points
//     [t*PPT, (t+1)*PPT), so "lowest lane among the maxima" == "lowest point index" and the arg-max needs
//     only a value reduction + ballot (no (value,index) pairs through the cross-lane network);
//   - distances are >= +0, so their bit patterns order like u32: reductions run on integer max;
//   - in-wave reduction = 4 DPP row steps + 4 v_readlane + 3 s_max (no LDS); across waves one 8-byte
//     LDS slot per wave, double buffered => ONE s_barrier per step;
//   - the winner's coordinates come from an LDS copy of the cloud (SoA, broadcast read);
//   - the S selected indices are buffered in LDS and written once at the end (a per-step global store
//     would put a vmcnt(0) drain in front of every barrier).
// Arithmetic is the reference's: d = (dx*dx + dy*dy) + dz*dz, separately rounded (no FMA: this file is
// compiled with -ffp-contract=off), update `if (d < dist) dist = d`, first maximum wins.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace {

__device__ __forceinline__ unsigned max3_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float min_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// FLOOR = true: the measurement variant behind mp_fps_floor_f32 -- the same dependent chain per step (centroid read from
// LDS, wave arg-max, LDS atomic, barrier, broadcast read) with the per-point distance work removed: S * t_iter of it is
// the latency floor this design cannot go below, which bench.py reports the real kernel against.
template <int T, int PPT, bool FLOOR = false, int MODE = 0>      // MODE: how the winner's slot is found (timing builds: see the step loop)
__global__ __launch_bounds__(T) void fps_kernel(const float* __restrict__ xyz, int N, int S,
                                                const int64_t* __restrict__ start_idx,
                                                int64_t* __restrict__ out_idx, float* __restrict__ out_xyz)
{
    constexpr int NW = T / MP_WAVE;
    constexpr int NPAD = T * PPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sx = reinterpret_cast<float*>(smem_raw);
    float* sy = sx + NPAD;
    float* sz = sy + NPAD;
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(sz + NPAD);  // [4] rotating block arg-max keys
    int* sel = reinterpret_cast<int*>(slots + 4);                                   // [S]

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* p = xyz + (size_t)b * N * 3;

    // stage the cloud: coalesced AoS read, SoA LDS image
    for (int i = tid; i < 3 * N; i += T) {
        const float v = p[i];
        const int pt = i / 3;
        const int c = i - 3 * pt;
        (c == 0 ? sx : (c == 1 ? sy : sz))[pt] = v;
    }
    for (int i = N + tid; i < NPAD; i += T) { sx[i] = 0.f; sy[i] = 0.f; sz[i] = 0.f; }
    if (tid < 4) slots[tid] = 0ull;
    __syncthreads();

    // Points are held two to a register pair so that the distance runs on packed fp32 (v_pk_add_f32 / v_pk_mul_f32: two
    // IEEE operations per lane and issue slot -- the step is VALU-bound, ~10 instructions per point before packing);
    // every operation is still separately rounded, so the result is bit-identical to the scalar form.
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int PP2 = (PPT + 1) / 2;
    f2 px[PP2], py[PP2], pz[PP2], dist[PP2];
    const int base = tid * PPT;
#pragma unroll
    for (int j = 0; j < PP2; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 2 * j + h;
            const bool own = q < PPT;            // odd PPT: the spare half-slot behaves like a padded point
            px[j][h] = own ? sx[base + q] : 0.0f;
            py[j][h] = own ? sy[base + q] : 0.0f;
            pz[j][h] = own ? sz[base + q] : 0.0f;
            // padded slots hold 0 forever: min(0, d>=0) == 0 and a real point of lower index wins every tie
            dist[j][h] = (own && base + q < N) ? 1e10f : 0.0f;
        }
    }

    int far = (int)start_idx[b];
    far = far < 0 ? 0 : (far >= N ? N - 1 : far);  // the reference indexes with it unchecked; stay in bounds
    for (int s = 0; s < S; ++s) {
        if (tid == 0) sel[s] = far;
        if (s == S - 1) break;
        const float cxs = sx[far], cys = sy[far], czs = sz[far];
        const f2 cx = {cxs, cxs}, cy = {cys, cys}, cz = {czs, czs};
        // [r6] How a lane finds the slot of its maximum (MODE; all bit-identical, measured at B = 32, N = 5120, 256 x 20 / 512 x 10):
        //   2  (r1 .. r5) value and slot together, compare + two selects per point inside the distance loop: 358 / 350 us;
        //   1  values through a v_max3 tree, then compare + select per point against the lane's maximum: 372 / 344 us;
        //   0  values through the tree, the slot recovered for the ONE winner lane on the scalar unit (a compare of every distance register
        //      against the wave's maximum writes a lane mask; s_bitcmp1 + s_cselect per slot): 381 / 364 us.
        // The step's ~200 VALU instructions became ~160 (1) and ~130 + 45 scalar (0) and the time did not follow: with one wave per SIMD the
        // step is a chain of DEPENDENT issues (the tree and the recovery are behind the last distance; mode 2's selects ride in the distance
        // loop's shadow) in front of a ~590-cycle synchronisation chain (mp_fps_floor_f32).  Two waves per SIMD fill those slots for each
        // other: 512 x 10 with mode 1 is the default for the 5120-point level.  (__builtin_fminf also cost a canonicalising v_max_f32 x, x
        // per point: min_f32 is the bare instruction.)
        unsigned bits[2 * PP2];
        if constexpr (FLOOR) {
#pragma unroll
            for (int q = 0; q < 2 * PP2; ++q) bits[q] = 0u;
            bits[0] = __float_as_uint(__builtin_fabsf((cxs + cys) + czs) + (float)(tid ^ s));   // depends on the LDS read, varies per step
        }
#pragma unroll
        for (int j = 0; j < (FLOOR ? 0 : PP2); ++j) {
            const f2 dx = px[j] - cx;
            const f2 dy = py[j] - cy;
            const f2 dz = pz[j] - cz;
            const f2 d = (dx * dx + dy * dy) + dz * dz;
            // == `if (d < dist) dist = d`.  (v_min_f32 spelled out: __builtin_fminf adds a canonicalising v_max_f32 x, x per value -- 20 of the
            // step's VALU instructions for NaNs that finite coordinates cannot produce)
            const f2 cur = {min_f32(dist[j].x, d.x), min_f32(dist[j].y, d.y)};
            dist[j] = cur;
            bits[2 * j] = __float_as_uint(cur.x);          // distances are >= +0: their bit patterns order like u32
            bits[2 * j + 1] = __float_as_uint(cur.y);      // (odd PPT: the spare half-slot holds 0 and loses every tie to a real slot below it)
        }
        unsigned key = bits[0];
        int bj = 0;
        if constexpr (MODE == 2 && !FLOOR) {
            // (r1 .. r5) value and slot together: compare + two selects per point
#pragma unroll
            for (int q = 1; q < PPT; ++q)
                if (bits[q] > key) { key = bits[q]; bj = q; }
        } else {
#pragma unroll
            for (int q = 1; q + 1 < 2 * PP2; q += 2) key = max3_u32(key, bits[q], bits[q + 1]);
            key = key > bits[2 * PP2 - 1] ? key : bits[2 * PP2 - 1];
        }
        if constexpr (MODE == 1 && !FLOOR) {
            // every lane finds the slot of ITS maximum: compare + select per point (descending: the lowest slot wins)
#pragma unroll
            for (int q = PPT - 1; q >= 0; --q) bj = bits[q] == key ? q : bj;
        }
        const unsigned wmax = mp::wave_max_u32(key);
        const unsigned long long m = __ballot(key == wmax);
        const int src = (int)__builtin_ctzll(m);
        if constexpr (MODE == 0 && !FLOOR) {
            // the slot of the ONE lane that holds the wave's maximum, on the scalar unit: all compares first (their lane masks land in PPT
            // SGPR pairs back to back), then two scalar instructions per slot (descending: the lowest slot wins)
            unsigned long long mq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; ++q) mq[q] = __ballot(bits[q] == wmax);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = PPT - 1; q >= 0; --q)
                asm("s_bitcmp1_b64 %1, %2\n\ts_cselect_b32 %0, %3, %0" : "+s"(bj) : "s"(mq[q]), "s"(src), "n"(q) : "scc");
        } else if constexpr (!FLOOR) {
            bj = __builtin_amdgcn_readlane(bj, src);
        }
        const int widx = src * PPT + wave * (MP_WAVE * PPT) + bj;     // == base + bj of lane `src`
        if constexpr (NW == 1) {
            far = widx;
        } else {
            // key = (distance bits, ~index): a 64-bit max is "largest distance, lowest index".  One LDS atomic per wave,
            // one barrier, one broadcast read.  Three slots rotate so that the slot being cleared was last read two
            // barriers ago.
            unsigned long long* slot = slots + (s % 3);
            if (lane == 0) {
                atomicMax(slot, ((unsigned long long)wmax << 32) | (unsigned long long)(~(unsigned)widx));
                if (wave == 0) slots[(s + 1) % 3] = 0ull;
            }
            __syncthreads();
            far = (int)(~(unsigned)(*slot));
        }
    }
    __syncthreads();
    for (int s = tid; s < S; s += T) {
        const int i = sel[s];
        out_idx[(size_t)b * S + s] = (int64_t)i;
        if (out_xyz) {
            float* o = out_xyz + ((size_t)b * S + s) * 3;
            o[0] = sx[i];
            o[1] = sy[i];
            o[2] = sz[i];
        }
    }
}

template <int T, int PPT, bool FLOOR = false, int MODE = 0>
int launch_fps_mode(const float* xyz, int B, int N, int S, const int64_t* start, int64_t* out_idx, float* out_xyz,
               hipStream_t stream)
{

    const size_t smem = (size_t)3 * T * PPT * sizeof(float) + 4 * sizeof(unsigned long long) + (size_t)S * sizeof(int);
    if (smem > 160 * 1024) return MP_EUNSUPPORTED;
    auto kern = fps_kernel<T, PPT, FLOOR, MODE>;
    // opt-in to > 64 KB of dynamic LDS once per size (not a stream operation: it must not run inside a graph capture)
    static mp::DynLds lds;      // per kernel instantiation, per device
    if (!lds.ensure(reinterpret_cast<const void*>(kern), smem)) return MP_ELAUNCH;
    char tag[48];
    snprintf(tag, sizeof tag, FLOOR ? "fps_floor_kernel<%d, %d>" : "fps_kernel<%d, %d>", T, PPT);
    MP_LAUNCH(tag, 8.0 * B * (double)N * S, (double)B * (N * 12.0 + S * 8.0 + (out_xyz ? S * 12.0 : 0.0)), kern, dim3(B),
              dim3(T), smem, stream, xyz, N, S, start, out_idx, out_xyz);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

template <int T, int PPT, bool FLOOR = false>
int launch_fps(const float* xyz, int B, int N, int S, const int64_t* start, int64_t* out_idx, float* out_xyz, hipStream_t stream)
{
    // MP_FPS_MODE (timing aid; every mode is bit-identical): 1 (default) value tree + per-lane slot recovery, 0 value tree + scalar recovery
    // for the winner lane, 2 the r1-r5 (value, slot) select chain
    static const int mode = [] { const char* e = getenv("MP_FPS_MODE"); return e ? atoi(e) : 1; }();
    if constexpr (!FLOOR) {
        if (mode == 0) return launch_fps_mode<T, PPT, false, 0>(xyz, B, N, S, start, out_idx, out_xyz, stream);
        if (mode == 2) return launch_fps_mode<T, PPT, false, 2>(xyz, B, N, S, start, out_idx, out_xyz, stream);
    }
    return launch_fps_mode<T, PPT, FLOOR, 1>(xyz, B, N, S, start, out_idx, out_xyz, stream);
}

}  // namespace

extern "C" int mp_fps_f32(const float* xyz, int64_t B, int64_t N, int64_t S, const int64_t* start_idx,
                          int64_t* out_idx, float* out_xyz, mp_stream_t stream_)
{
    if (B < 0 || N <= 0 || S < 0) return MP_EINVAL;
    if (B == 0 || S == 0) return MP_OK;
    if (!xyz || !start_idx || !out_idx) return MP_EINVAL;
    if (N > 13312 || S > 8192) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n = (int)N, s = (int)S;
#define MP_FPS_CASE(T, P) \
    if (n <= (T) * (P)) return launch_fps<T, P>(xyz, b, n, s, start_idx, out_idx, out_xyz, stream)
    // one wave: no barrier at all
    MP_FPS_CASE(64, 1);
    MP_FPS_CASE(64, 2);
    MP_FPS_CASE(64, 4);
    MP_FPS_CASE(64, 8);
    MP_FPS_CASE(256, 3);
    MP_FPS_CASE(256, 4);
    MP_FPS_CASE(256, 6);
    MP_FPS_CASE(256, 8);
    // [r1] N = 5120: 256 threads x 20 points 339 us, 512 x 10 350 us, 1024 x 5 366 us (one LDS atomic + barrier per step)
    MP_FPS_CASE(256, 12);
    MP_FPS_CASE(256, 16);
    {   // [r6] the 5120-point level on 512 threads x 10 points (two waves per SIMD fill each other's dependent-issue slots): 344 vs 358 us.
        // MP_FPS_SHAPE=256 / 1024: the other workgroup sizes for this level (timing aid; 1024 x 5: 400 us)
        static const int shape = [] { const char* e = getenv("MP_FPS_SHAPE"); return e ? atoi(e) : 512; }();
        if (shape == 512 && n > 256 * 16) { MP_FPS_CASE(512, 10); }
        if (shape == 1024 && n > 256 * 16) { MP_FPS_CASE(1024, 5); }
    }
    MP_FPS_CASE(256, 20);
    MP_FPS_CASE(512, 12);
    MP_FPS_CASE(512, 16);
    MP_FPS_CASE(512, 20);
    MP_FPS_CASE(512, 24);
    MP_FPS_CASE(1024, 8);
    MP_FPS_CASE(1024, 10);
    MP_FPS_CASE(1024, 13);
#undef MP_FPS_CASE
    return MP_EUNSUPPORTED;
}

// Measurement aid (bench.py): the latency floor of the kernel mp_fps_f32 would pick for this size -- identical launch shape and
// per-step synchronisation, no distance arithmetic.  Output contents are meaningless (in-range indices).
extern "C" int mp_fps_floor_f32(const float* xyz, int64_t B, int64_t N, int64_t S, const int64_t* start_idx,
                                int64_t* out_idx, float* out_xyz, mp_stream_t stream_)
{
    if (B < 0 || N <= 0 || S < 0) return MP_EINVAL;
    if (B == 0 || S == 0) return MP_OK;
    if (!xyz || !start_idx || !out_idx) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    const int b = (int)B, n = (int)N, s = (int)S;
    if (S > 8192) return MP_EUNSUPPORTED;
    if (n <= 512) return launch_fps<64, 8, true>(xyz, b, n, s, start_idx, out_idx, out_xyz, stream);
    if (n <= 5120) return launch_fps<512, 10, true>(xyz, b, n, s, start_idx, out_idx, out_xyz, stream);      // ([r6] the 5120-point level's shape)
    if (n <= 10240) return launch_fps<512, 20, true>(xyz, b, n, s, start_idx, out_idx, out_xyz, stream);
    return MP_EUNSUPPORTED;
}
