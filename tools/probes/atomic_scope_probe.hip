// [r5] The dW epilogues of the backward kernels are bound by the rate of device-scope float atomics (one 256-byte wave-instruction per ~50 ns
// and CU).  Are atomics of a narrower scope -- performed in the XCD's own L2 -- faster, with one dW copy per XCD (adders of a copy all on that
// XCD: s_getreg XCC_ID) and a tiny sum of the eight copies afterwards?  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics ... && run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// every workgroup adds a TILE of `n` floats (n = 32768: the roles kernel's 256 x 128 dW) held as 64 values per thread
template <int SCOPE, bool PER_XCD>
__global__ __launch_bounds__(512) void k_add(float* __restrict__ dst, int n, unsigned* __restrict__ seen)
{
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicOr(&seen[x], 1u << (blockIdx.x & 31));
    float* d = dst + (PER_XCD ? (size_t)x * n : 0);
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        float* p = d + (size_t)i * 512 + threadIdx.x;
        if (SCOPE == 0) atomicAdd(p, acc[i]);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 2) __hip_atomic_fetch_add(p, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

__global__ void k_sum8(const float* __restrict__ src, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float s = 0.f; for (int x = 0; x < 8; ++x) s += src[(size_t)x * n + i]; out[i] = s; }
}

int main()
{
    const int n = 32768, grid = 256;
    float *dst, *out; unsigned* seen;
    CK(hipMalloc(&dst, (size_t)8 * n * 4)); CK(hipMalloc(&out, (size_t)n * 4)); CK(hipMalloc(&seen, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h(n), ref(n);
    for (int i = 0; i < n; ++i) ref[i] = grid * (float)((i % 512) + (i / 512)) * 1e-3f;
#define RUN(name, kern, check8) do {                                                                                          \
        float best = 1e9f;                                                                                                    \
        for (int rep = 0; rep < 5; ++rep) {                                                                                   \
            CK(hipMemset(dst, 0, (size_t)8 * n * 4)); CK(hipMemset(seen, 0, 64)); CK(hipDeviceSynchronize());                  \
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, dst, n, seen); CK(hipEventRecord(e1)); \
            CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;          \
        }                                                                                                                     \
        if (check8) { hipLaunchKernelGGL(k_sum8, dim3((n + 255) / 256), dim3(256), 0, 0, dst, n, out); CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost)); } \
        else CK(hipMemcpy(h.data(), dst, n * 4, hipMemcpyDeviceToHost));                                                      \
        double worst = 0; for (int i = 0; i < n; ++i) { double e = fabs((double)h[i] - ref[i]) / (fabs((double)ref[i]) + 1e-6); worst = e > worst ? e : worst; } \
        unsigned hs[16]; CK(hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost)); int nx = 0; for (int i = 0; i < 16; ++i) nx += hs[i] != 0; \
        printf("%-44s %7.1f us   max rel err %.1e   XCDs seen %d\n", name, best * 1e3, worst, nx);                              \
    } while (0)
    RUN("atomicAdd, one copy", (k_add<0, false>), false);
    RUN("agent scope, one copy", (k_add<1, false>), false);
    RUN("agent scope, copy per XCD", (k_add<1, true>), true);
    RUN("workgroup scope, copy per XCD", (k_add<2, true>), true);
    RUN("wavefront scope, copy per XCD", (k_add<3, true>), true);
    RUN("workgroup scope, ONE copy (incoherent: expect err)", (k_add<2, false>), false);
    return 0;
}
