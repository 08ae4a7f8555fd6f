// Shared device helpers for the gfx950 kernels (wave64 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/maskplanner_hip.h"

#define MP_WAVE 64

#define MP_CHECK_LAUNCH()                                   \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return MP_ELAUNCH; \
    } while (0)

static inline hipStream_t mp_stream(mp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Optional per-kernel device timing (bench.py's roofline leg): when enabled through mp_profiler_enable(), every
// MP_LAUNCH is bracketed by two HIP events recorded on the launch stream and tagged with the kernel's name and its
// algorithmic work.  Disabled (the default) it costs one relaxed load per launch.
namespace mp {
bool prof_on();
void prof_begin(const char* tag, double flops, double bytes, hipStream_t stream);
void prof_end(hipStream_t stream);
}  // namespace mp

#define MP_LAUNCH(tag, flops, bytes, kernel, grid, block, smem, stream, ...)          \
    do {                                                                              \
        const bool mp_prof_ = mp::prof_on();                                          \
        if (mp_prof_) mp::prof_begin(tag, flops, bytes, stream);                      \
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);           \
        if (mp_prof_) mp::prof_end(stream);                                           \
    } while (0)

namespace mp {

// DPP controls (CDNA ISA): quad_perm packs four 2-bit selectors; row_* act inside 16-lane rows.
constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// max over each 16-lane row, result in every lane of the row (4 DPP steps, no LDS).
__device__ __forceinline__ unsigned row16_max_u32(unsigned v)
{
    unsigned t;
    t = dpp_u32<DPP_QUAD_XOR1>(v); v = v > t ? v : t;
    t = dpp_u32<DPP_QUAD_XOR2>(v); v = v > t ? v : t;
    t = dpp_u32<DPP_ROW_HALF_MIRROR>(v); v = v > t ? v : t;
    t = dpp_u32<DPP_ROW_MIRROR>(v); v = v > t ? v : t;
    return v;
}

__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
    unsigned t;
    t = dpp_u32<DPP_QUAD_XOR1>(v); v = v < t ? v : t;
    t = dpp_u32<DPP_QUAD_XOR2>(v); v = v < t ? v : t;
    t = dpp_u32<DPP_ROW_HALF_MIRROR>(v); v = v < t ? v : t;
    t = dpp_u32<DPP_ROW_MIRROR>(v); v = v < t ? v : t;
    return v;
}

// wave-wide max of a u32 key as a wave-uniform (scalar) value.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = row16_max_u32(v);
    unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
    unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
    unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    a = a > b ? a : b;
    c = c > d ? c : d;
    return a > c ? a : c;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = row16_min_u32(v);
    unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
    unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
    unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    a = a < b ? a : b;
    c = c < d ? c : d;
    return a < c ? a : c;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace mp

// Opt-in to more than 64 KB of dynamic LDS for one kernel.  hipFuncSetAttribute applies to the CURRENT device's function
// object, so the largest size configured so far is tracked per device (a process that drives several GPUs -- or a test that
// maps ranks onto `local % device_count` -- must set it on each); atomics because callers may come from several threads.
// Not a stream operation: a size is configured the first time it is seen, which for a recorded step is one of the eager
// warm-up steps before the capture.
namespace mp {
struct DynLds {
    static constexpr int MAX_DEV = 64;
    std::atomic<size_t> configured[MAX_DEV];
    DynLds() { for (auto& c : configured) c.store(64 * 1024, std::memory_order_relaxed); }
    bool ensure(const void* kernel, size_t smem)
    {
        if (smem <= 64 * 1024) return true;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool tracked = dev >= 0 && dev < MAX_DEV;
        if (tracked && smem <= configured[dev].load(std::memory_order_acquire)) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return false;
        if (tracked) {
            size_t cur = configured[dev].load(std::memory_order_relaxed);
            while (cur < smem && !configured[dev].compare_exchange_weak(cur, smem, std::memory_order_release)) {}
        }
        return true;
    }
};
}  // namespace mp

// Zero fill as an ordinary kernel.  hipMemsetAsync must not be used on the compute path: recorded into a hipGraph it
// becomes a memset node, and on ROCm 7.2 replays were observed (tools/graph_nan_check.py) to run the following kernel
// against the buffer's stale contents now and then -- a dW accumulated by atomics on top of garbage.  A kernel node
// keeps the plain stream order.
namespace mp {
static __global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n4, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.0f;
}

// Zero arena (mp_zero_arena_arm): a caller-owned range that ONE launch has just cleared on `stream`; an output of the library that lies
// inside it is handed out once per arming and needs no clear of its own.  Defined in api.hip.
bool zero_arena_covers(const void* p, size_t bytes, hipStream_t stream);

static inline bool zero_async(float* p, size_t n, hipStream_t stream)
{
    if (n == 0) return true;
    if (zero_arena_covers(p, n * sizeof(float), stream)) return true;
    const size_t n4 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : n / 4;   // float4 stores need 16-byte alignment
    size_t blocks = ((n4 ? n4 : n) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n4, n);
    return hipGetLastError() == hipSuccess;
}
}  // namespace mp
