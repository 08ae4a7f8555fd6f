// Batch collation on the device (gfx950): ragged per-sample sequences -> one padded [B, R, D] tensor.
//
// Reference: utils/dataset/paintnet_ODv1.py:726-748 (Paintnet_ODv1_CollateBatch.__call__) pads every sample's `traj`,
// `traj_as_pc` (fake rows of -100, add_fake_vectors_v2 :887-904) and `stroke_ids` (-1, add_fake_values_v2 :907-925) on the host
// with one numpy concatenate + torch.as_tensor + torch.stack per sample and key, then the training loop copies the stacked
// tensors to the GPU.  Here the samples of a key travel as ONE flat buffer (a single host-to-device copy) and this kernel lays
// them out: out[b, r, :] = r < len_b ? flat[off_b + r, :] : fill.  Pure data movement, bound by the write of the padded tensor.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pad_ragged_kernel(const float* __restrict__ flat, const int64_t* __restrict__ offsets, int R,
                                                         int D, float fill, int64_t total, float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int64_t row = e / D;                  // b * R + r
    const int d = (int)(e - row * D);
    const int64_t b = row / R;
    const int64_t r = row - b * R;
    const int64_t o0 = offsets[b], len = offsets[b + 1] - o0;
    out[e] = r < len ? flat[(o0 + r) * D + d] : fill;
}

}  // namespace

extern "C" int mp_pad_ragged_f32(const float* flat, const int64_t* offsets, int64_t B, int64_t R, int64_t D, float fill, float* out,
                                 mp_stream_t stream_)
{
    if (B < 0 || R < 0 || D < 0) return MP_EINVAL;
    const int64_t total = B * R * D;
    if (total == 0) return MP_OK;
    if (!offsets || !out) return MP_EINVAL;          // flat may be NULL when every sample is empty
    if (R > ((int64_t)1 << 30) || D > ((int64_t)1 << 30) || total > ((int64_t)1 << 40)) return MP_EUNSUPPORTED;
    MP_LAUNCH("pad_ragged_kernel", 0.0, 8.0 * (double)total, pad_ragged_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
              mp_stream(stream_), flat, offsets, (int)R, (int)D, fill, total, out);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
