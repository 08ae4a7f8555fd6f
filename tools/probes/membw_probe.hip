// [r5] What read rate do the streaming kernels' load paths reach on an otherwise idle chip?  (NOTEBOOK: "the 256-output backward layer on the LDS
// ring": 402 MB of reads took 100 us = 4 TB/s inside the step whatever the ring looked like.)  Standalone: hipcc -O3 --offload-arch=gfx950
// tools/probes/membw_probe.hip -o /tmp/membw && /tmp/membw.  Reads only; every variant walks the same bytes once per launch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* src, void* lds_dst)
{
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// plain 16-byte loads into registers, U in flight per lane; a workgroup walks `per_wg` bytes (contiguous, or chunk-interleaved over the grid)
template <int NT, int U, bool INTERLEAVE>
__global__ __launch_bounds__(NT) void k_plain(const uint8_t* __restrict__ a, size_t bytes, float* out)
{
    constexpr size_t CH = (size_t)NT * 16 * U;
    const size_t nch = bytes / CH;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t per = (nch + gridDim.x - 1) / gridDim.x;
    for (size_t i = 0; i < per; ++i) {
        const size_t c = INTERLEAVE ? i * gridDim.x + blockIdx.x : blockIdx.x * per + i;
        if (c >= nch) break;
        const uint8_t* p = a + c * CH + threadIdx.x * 16;
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)u * NT * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = 1.0f;
}

// LDS ring filled by global_load_lds: NW waves, each LPW loads of 1 KB per chunk (chunk = NW * LPW KB), R slots, R - 1 chunks in flight;
// BAR: the waves meet at a barrier per chunk (the product kernels' structure), otherwise each wave runs its own ring
template <int NW, int LPW, int R, bool BAR, bool INTERLEAVE>
__global__ __launch_bounds__(NW * 64) void k_ring(const uint8_t* __restrict__ a, size_t bytes, float* out)
{
    constexpr int CH = NW * LPW * 1024;
    __shared__ __attribute__((aligned(16))) uint8_t ring[R][CH];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t nch = bytes / CH;
    const size_t per = (nch + gridDim.x - 1) / gridDim.x;
    size_t mine = 0;
    for (size_t i = 0; i < per; ++i) { const size_t c = INTERLEAVE ? i * gridDim.x + blockIdx.x : blockIdx.x * per + i; if (c < nch) mine = i + 1; }
    if (!mine) return;
    auto issue = [&](size_t i) {
        if (i >= mine) i = mine - 1;
        const size_t c = INTERLEAVE ? i * gridDim.x + blockIdx.x : blockIdx.x * per + i;
        const int slot = (int)(i % R);
#pragma unroll
        for (int j = 0; j < LPW; ++j) glds16(a + c * CH + (size_t)(wave * LPW + j) * 1024 + lane * 16, ring[slot] + (wave * LPW + j) * 1024);
    };
#pragma unroll
    for (int c = 0; c < R - 1; ++c) issue(c);
    float acc = 0.f;
    for (size_t kc = 0; kc < mine; ++kc) {
        wait_vm<(R - 2) * LPW>();
        if (BAR) lds_barrier();
        acc += *reinterpret_cast<const float*>(&ring[kc % R][(wave * LPW) * 1024 + lane * 4]);
        if (BAR) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(kc + R - 1);
    }
    wait_vm<0>();
    if (acc == 1234.5f) out[0] = acc;
}

// copy: the same loads, each stored to a second buffer (half the bytes each way)
template <int NT, int U, bool INTERLEAVE, bool NTS>
__global__ __launch_bounds__(NT) void k_copy(const uint8_t* __restrict__ a, size_t bytes, float* out)
{
    constexpr size_t CH = (size_t)NT * 16 * U;
    const size_t half = bytes / 2;
    const size_t nch = half / CH;
    uint8_t* dst = const_cast<uint8_t*>(a) + half;
    const size_t per = (nch + gridDim.x - 1) / gridDim.x;
    for (size_t i = 0; i < per; ++i) {
        const size_t c = INTERLEAVE ? i * gridDim.x + blockIdx.x : blockIdx.x * per + i;
        if (c >= nch) break;
        const uint8_t* p = a + c * CH + threadIdx.x * 16;
        uint8_t* q = dst + c * CH + threadIdx.x * 16;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f4*>(p + (size_t)u * NT * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], reinterpret_cast<f4*>(q + (size_t)u * NT * 16));
            else *reinterpret_cast<f4*>(q + (size_t)u * NT * 16) = v[u];
        }
    }
}

template <typename F>
static void timeit(const char* name, size_t bytes, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    const int reps = 12;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %8.1f us  %6.2f TB/s\n", name, ms * 1000.0 / reps, (double)bytes * reps / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)402653184;   // 384 MiB: Z_l [262144 x 256] + Z_{l-1} [262144 x 128] fp32
    const int NSET = 3;                        // three buffers in rotation: no launch finds its bytes in the 256 MiB Infinity Cache
    std::vector<uint8_t*> bufs(NSET);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    float* out; CK(hipMalloc(&out, 4));
#define RUN(name, kern, grid, nt) timeit(name, bytes, [&](int i) { hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 0, 0, bufs[i % NSET], bytes, out); })
    RUN("plain 256 thr x4, 256 WGs contiguous", (k_plain<256, 4, false>), 256, 256);
    RUN("plain 256 thr x4, 1024 WGs contiguous", (k_plain<256, 4, false>), 1024, 256);
    RUN("plain 256 thr x4, 2048 WGs interleaved", (k_plain<256, 4, true>), 2048, 256);
    RUN("plain 512 thr x4, 256 WGs contiguous", (k_plain<512, 4, false>), 256, 512);
    RUN("plain 512 thr x8, 256 WGs interleaved", (k_plain<512, 8, true>), 256, 512);
    RUN("plain 512 thr x4, 1024 WGs interleaved", (k_plain<512, 4, true>), 1024, 512);
    RUN("ring 8 waves x3 KB, 3 slots, barriers, 256 WGs contig", (k_ring<8, 3, 3, true, false>), 256, 512);
    RUN("ring 8 waves x3 KB, 5 slots, barriers, 256 WGs contig", (k_ring<8, 3, 5, true, false>), 256, 512);
    RUN("ring 8 waves x3 KB, 5 slots, barriers, 256 WGs interl", (k_ring<8, 3, 5, true, true>), 256, 512);
    RUN("ring 8 waves x3 KB, 5 slots, no barrier, 256 WGs contig", (k_ring<8, 3, 5, false, false>), 256, 512);
    RUN("ring 8 waves x3 KB, 5 slots, no barrier, 256 WGs interl", (k_ring<8, 3, 5, false, true>), 256, 512);
    RUN("ring 4 waves x4 KB, 4 slots, barriers, 512 WGs interl", (k_ring<4, 4, 4, true, true>), 512, 256);
    RUN("ring 4 waves x4 KB, 4 slots, no barrier, 512 WGs interl", (k_ring<4, 4, 4, false, true>), 512, 256);
    RUN("ring 1 wave x8 KB, 8 slots, 256 WGs contig", (k_ring<1, 8, 8, false, false>), 256, 64);
    RUN("ring 1 wave x8 KB, 8 slots, 256 WGs interl", (k_ring<1, 8, 8, false, true>), 256, 64);
    RUN("ring 2 waves x8 KB, 8 slots, no barrier, 256 WGs interl", (k_ring<2, 8, 8, false, true>), 256, 128);
    RUN("copy 256 thr x4, 2048 WGs interleaved (r + w bytes)", (k_copy<256, 4, true, false>), 2048, 256);
    RUN("copy 256 thr x4, 2048 WGs interleaved, nt stores", (k_copy<256, 4, true, true>), 2048, 256);
    RUN("copy 512 thr x4, 256 WGs contiguous", (k_copy<512, 4, false, false>), 256, 512);
    RUN("copy 512 thr x4, 256 WGs contiguous, nt stores", (k_copy<512, 4, false, true>), 256, 512);
    RUN("copy 512 thr x4, 256 WGs interleaved, nt stores", (k_copy<512, 4, true, true>), 256, 512);
    return 0;
}
