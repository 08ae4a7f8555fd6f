// Internal interface of sa_lean.hip (the pooled layer without its stored activation) to sa_mlp.hip's level drivers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mp {
bool lean_enabled();
bool lean_supported(int64_t P, int64_t K, int64_t CO, int64_t CI);
size_t lean_workspace_bytes(int64_t P, int64_t K, int64_t CO, int64_t CI);
int lean_bwd(const float* z1, const float* s1, const float* t1, const float* gamma1, const float* beta1, const float* W, const float* a,
             const float* e, const float* f, const float* gp, const int* argk, int64_t P, int64_t K, int CO, int CI, float* G1,
             float* partials, int* nblk_out, float* dW, void* workspace, hipStream_t stream);
}  // namespace mp
