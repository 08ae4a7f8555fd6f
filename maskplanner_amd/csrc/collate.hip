// Batch collation on the device (gfx950): ragged per-sample sequences -> one padded [B, R, D] tensor.
//
// Reference: utils/dataset/paintnet_ODv1.py:726-748 (Paintnet_ODv1_CollateBatch.__call__) pads every sample's `traj`,
// `traj_as_pc` (fake rows of -100, add_fake_vectors_v2 :887-904) and `stroke_ids` (-1, add_fake_values_v2 :907-925) on the host
// with one numpy concatenate + torch.as_tensor + torch.stack per sample and key, then the training loop copies the stacked
// tensors to the GPU.  Here the samples of a key travel as ONE flat buffer (a single host-to-device copy) and this kernel lays
// them out: out[b, r, :] = r < len_b ? flat[off_b + r, :] : fill.  Pure data movement, bound by the write of the padded tensor.
#include <cstring>

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

// Segments ("mini-sequences") of lambda consecutive poses per stroke, for a whole batch, straight into the padded tensors the
// loss consumes.  Reference: utils/pointcloud.py:294-413 get_sequences_of_lambda_points (+ add_padding :98-105), run on the host
// per sample by the dataset (utils/dataset/paintnet_ODv1.py:294), then padded again by the collate function.
//   per stroke of L poses (ids ascending 0, 0, .., 1, 1, ..):  L <  lambda: dropped (and the following strokes renumbered)
//       overlapping > 0:  (L - lambda) / (lambda - overlapping) + 1 windows, window j starts at pose j * (lambda - overlapping)
//       overlapping = 0:  L / lambda windows, the stroke centred: window j starts at pose (L % lambda) / 2 + j * lambda
//   rows behind a sample's last window: -100 (stroke id -1).
// One workgroup per sample: stroke starts from the id changes (LDS), window counts and their prefix sum by one wave, then every
// thread copies output rows.
constexpr int SEG_MAX_STROKES = 1024;
__global__ __launch_bounds__(256) void lambda_segments_kernel(const float* __restrict__ poses, const float* __restrict__ ids,
                                                              const int64_t* __restrict__ offsets, int D, int lambda, int overlap,
                                                              int R, float* __restrict__ out_traj, float* __restrict__ out_ids,
                                                              int32_t* __restrict__ status)
{
    __shared__ int s_start[SEG_MAX_STROKES + 1];   // first pose of stroke s (relative to the sample)
    __shared__ int s_first[SEG_MAX_STROKES + 1];   // first output row of stroke s (prefix sum of the window counts)
    __shared__ int s_newid[SEG_MAX_STROKES];       // id after renumbering the kept strokes
    __shared__ int s_nstrokes, s_bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t o0 = offsets[b];
    const int n = (int)(offsets[b + 1] - o0);
    const float* p = poses + o0 * D;
    const float* id = ids + o0;
    if (tid == 0) { s_nstrokes = n > 0 ? (int)id[n - 1] + 1 : 0; s_bad = 0; }
    __syncthreads();
    const int ns = s_nstrokes;
    if (ns > SEG_MAX_STROKES || ns < 0) {   // unsupported sample: its rows are all padding (-100 / -1), never uninitialised memory
        if (tid == 0) status[b] = MP_EUNSUPPORTED;
        const int W = lambda * D;
        for (int64_t e = tid; e < (int64_t)R * W; e += 256) out_traj[(int64_t)b * R * W + e] = -100.0f;
        for (int r = tid; r < R; r += 256) out_ids[(int64_t)b * R + r] = -1.0f;
        return;
    }
    for (int s = tid; s <= ns; s += 256) s_start[s] = s == ns ? n : -1;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int s = (int)id[i];
        if (i == 0 || id[i - 1] != id[i]) {
            if (s < 0 || s >= ns || (i > 0 && id[i - 1] > id[i])) s_bad = 1; else s_start[s] = i;
        }
    }
    __syncthreads();
    if (tid == 0) {     // <= 1024 strokes: a serial prefix sum is a few microseconds at most
        int rows = 0, kept = 0;
        for (int s = 0; s < ns; ++s) {
            if (s_start[s] < 0) { s_bad = 1; break; }      // an id of the range never occurs: the reference's argmax would misfire
            const int L = s_start[s + 1] - s_start[s];
            s_first[s] = rows;
            s_newid[s] = kept;
            if (L >= lambda) {
                rows += overlap > 0 ? (L - lambda) / (lambda - overlap) + 1 : L / lambda;
                ++kept;
            }
        }
        s_first[ns] = rows;
        if (rows > R) s_bad = 1;
        status[b] = s_bad ? MP_EINVAL : MP_OK;
    }
    __syncthreads();
    const int W = lambda * D;
    float* ot = out_traj + (size_t)b * R * W;
    float* oi = out_ids + (size_t)b * R;
    const int rows = s_bad ? 0 : s_first[ns];
    for (int e = tid; e < R * W; e += 256) {
        const int r = e / W, w = e - r * W;
        float v = -100.0f;
        if (r < rows) {
            int lo = 0, hi = ns - 1;                      // last stroke whose first row is <= r and which has rows
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_first[mid] <= r) lo = mid; else hi = mid - 1; }
            while (s_first[lo + 1] == s_first[lo]) --lo;    // skip dropped strokes that share the row index (never below 0: r < rows)
            const int L = s_start[lo + 1] - s_start[lo];
            const int j = r - s_first[lo];
            const int pose0 = s_start[lo] + (overlap > 0 ? j * (lambda - overlap) : (L % lambda) / 2 + j * lambda);
            v = p[(size_t)pose0 * D + w];                 // the window's lambda poses are contiguous in memory
            if (w == 0) oi[r] = (float)s_newid[lo];
        } else if (w == 0) {
            oi[r] = -1.0f;
        }
        ot[e] = v;
    }
}

}  // namespace

extern "C" int mp_lambda_segments_f32(const float* poses, const float* stroke_ids, const int64_t* offsets, int64_t B, int64_t D,
                                      int64_t lambda, int64_t overlapping, int64_t R, float* out_traj, float* out_ids,
                                      int32_t* status, mp_stream_t stream_)
{
    if (B < 0 || D <= 0 || lambda <= 0 || overlapping < 0 || overlapping >= lambda || R < 0) return MP_EINVAL;
    if (B == 0) return MP_OK;
    if (!offsets || !status || (R > 0 && (!out_traj || !out_ids))) return MP_EINVAL;
    if (R * lambda * D > ((int64_t)1 << 30)) return MP_EUNSUPPORTED;
    MP_LAUNCH("lambda_segments_kernel", 0.0, 8.0 * (double)(B * R * lambda * D), lambda_segments_kernel, dim3((unsigned)B), dim3(256), 0,
              mp_stream(stream_), poses, stroke_ids, offsets, (int)D, (int)lambda, (int)overlapping, (int)R, out_traj, out_ids, status);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

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
