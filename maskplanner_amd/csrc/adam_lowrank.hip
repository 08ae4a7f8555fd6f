// Fused "gradient from factors + Adam" update for the weight-heavy regression heads.
//
// Context ("next" row of the scope table): in MaskPlanner 97 % of the parameters sit in three
// Linear layers fed by a [B,1024] feature (fc3, fc_normals, sm_fc3: models/pointnet2_cls_ssg.py:270-290).  Their
// weight gradient is an outer-product sum of rank B:  dW[o,i] = sum_b g[b,o] * x[b,i]  with g = dLoss/dy [B,O] and
// x the layer input [B,I].  The reference materialises dW (143 MB), torch.optim.Adam (train_maskplanner.py:159) reads
// it back, and a data-parallel run would all-reduce it.  Here the factors (x, g) -- 1.5 MB -- are what gets exchanged
// between GPUs, and this kernel forms each gradient element on the fly inside the Adam update, so dW never exists:
// per step and parameter it moves 24 B (p, m, v read + write) instead of 28 B + a GEMM that writes another 4 B.
//
// Arithmetic = torch's Adam (no amsgrad / weight decay / maximize):
//   m = lerp(m, grad, 1-beta1);  v = beta2*v + (1-beta2)*grad^2;
//   p -= (lr / (1-beta1^t)) * m / (sqrt(v) / sqrt(1-beta2^t) + eps)
// with grad accumulated in fp32 over b in ascending order (fma chain).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int AL_TO = 64, AL_TI = 64, AL_BC = 32;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// MFMA = false: each thread rebuilds a 4 x 4 patch of the 64 x 64 gradient tile with scalar FMAs (16 per factor row): fine for
// one GPU's 32 factor rows, where the kernel is bound by the Adam stream.  MFMA = true (data-parallel runs: the all-gathered
// factors have 32 * world rows, 256 on a full node): the tile is four 32 x 32 v_mfma_f32_32x32x2_f32 accumulators, one per wave,
// K = factor rows -- the same k-ordered fp32 fma chain, so both forms give bit-identical gradients, but the rebuild stays under
// the kernel's HBM time instead of growing with the world size.
template <bool MFMA>
__global__ __launch_bounds__(256) void adam_lowrank_kernel(float* __restrict__ p, float* __restrict__ m,
                                                           float* __restrict__ v, const float* __restrict__ x,
                                                           const float* __restrict__ g, int Bg, int O, int I,
                                                           float gscale, float lr_c1, float beta1, float beta2,
                                                           float inv_sqrt_c2, float eps, const float* __restrict__ step_dev,
                                                           float lr)
{
    if (step_dev) {   // graph-capturable form: the step count lives on the device, bias corrections are formed here (fp64)
        const double st = (double)step_dev[0];
        lr_c1 = (float)((double)lr / (1.0 - pow((double)beta1, st)));
        inv_sqrt_c2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, st)));
    }
    // tile of the weight matrix per workgroup: 64 x 64 for the matrix-core rebuild (four 32 x 32 accumulators); [r4] 16 rows x 256 columns
    // for the scalar rebuild -- a wave then streams 1 KB contiguous pieces of four rows of p, m and v instead of 256-byte pieces of
    // sixteen (DRAM page locality: the kernel is a 24-byte-per-parameter stream and nothing else)
    constexpr int TO = MFMA ? AL_TO : 16, TI = MFMA ? AL_TI : 256;
    __shared__ __attribute__((aligned(16))) float sg[AL_BC][TO];
    __shared__ __attribute__((aligned(16))) float sx[AL_BC][TI];
    const int tid = threadIdx.x;
    const int to = MFMA ? (tid >> 4) : (tid >> 6), ti = MFMA ? (tid & 15) : (tid & 63);
    const int o0 = blockIdx.y * TO, i0 = blockIdx.x * TI;
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wo = (wave >> 1) * 32, wi = (wave & 1) * 32;        // this wave's 32 x 32 sub-tile (MFMA form)
    float acc[4][4];
    f32x16 macc;
    // [r4] the scalar-rebuild form streams 24 bytes per parameter and is bound by HBM latency x bytes in flight, not by the rebuild:
    // the 12 float4 of (p, m, v) a thread updates are requested BEFORE the factor slabs are staged and multiplied (non-temporal: every
    // byte is touched once per step), so that their latency runs under the rebuild instead of behind it
    typedef float vf4 __attribute__((ext_vector_type(4)));          // (the non-temporal builtins take native vectors)
    const bool fast = !MFMA && (I & 3) == 0 && i0 + ti * 4 + 3 < I;
    vf4 p4[4], m4[4], v4[4];
    if (fast) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + to * 4 + r;
            const size_t off = (size_t)(o < O ? o : 0) * I + i0 + ti * 4;
            p4[r] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p + off));
            m4[r] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(m + off));
            v4[r] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(v + off));
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) macc[r] = 0.0f;

    for (int b0 = 0; b0 < Bg; b0 += AL_BC) {
        __syncthreads();
        // stage the [AL_BC x TO] slab of g and the [AL_BC x TI] slab of x (row-major in memory: coalesced along the columns)
        auto ld_row4 = [](const float* base, int64_t row, int64_t ld, int col, int ncol) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* q = base + (size_t)row * ld + col;
            if (col + 3 < ncol && (ld & 3) == 0) v = *reinterpret_cast<const float4*>(q);   // rows 16-B aligned
            else if (col + 3 < ncol) { v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3]; }
            else { if (col < ncol) v.x = q[0]; if (col + 1 < ncol) v.y = q[1]; if (col + 2 < ncol) v.z = q[2]; }
            return v;
        };
        for (int e = tid; e < AL_BC * (TO / 4); e += 256) {
            const int b = e / (TO / 4), q = (e % (TO / 4)) * 4;
            *reinterpret_cast<float4*>(&sg[b][q]) = (b0 + b < Bg) ? ld_row4(g, b0 + b, O, o0 + q, O) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = tid; e < AL_BC * (TI / 4); e += 256) {
            const int b = e / (TI / 4), q = (e % (TI / 4)) * 4;
            *reinterpret_cast<float4*>(&sx[b][q]) = (b0 + b < Bg) ? ld_row4(x, b0 + b, I, i0 + q, I) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if constexpr (MFMA) {
            // A[m = o][k = b] = g[b][o], B[k = b][n = i] = x[b][i]: lane (l31, hi) feeds k = 2 * s + hi (rows past Bg are zero)
#pragma unroll
            for (int kk = 0; kk < AL_BC; kk += 2)
                macc = __builtin_amdgcn_mfma_f32_32x32x2f32(sg[kk + hi][wo + l31], sx[kk + hi][wi + l31], macc, 0, 0, 0);
        } else {
#pragma unroll 8
            for (int b = 0; b < AL_BC; ++b) {
                const float4 gv = *reinterpret_cast<const float4*>(&sg[b][to * 4]);
                const float4 xv = *reinterpret_cast<const float4*>(&sx[b][ti * 4]);
                const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
                const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fmaf(ga[r], xa[c], acc[r][c]);
            }
        }
    }
    auto upd = [&](float grad, float& pp, float& mm, float& vv) {
        mm = mm + (1.0f - beta1) * (grad - mm);                 // lerp form, as torch's fused Adam
        vv = beta2 * vv + (1.0f - beta2) * grad * grad;
        pp -= lr_c1 * (mm / (sqrtf(vv) * inv_sqrt_c2 + eps));
    };
    if constexpr (MFMA) {
        // accumulator register r of this lane: row (r & 3) + 8 * (r >> 2) + 4 * hi, column l31 of the wave's sub-tile: a half-wave
        // covers 32 consecutive floats of one weight row (128-byte segments of p, m, v)
        const int i = i0 + wi + l31;
        if (i < I) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (o < O) {
                    const size_t off = (size_t)o * I + i;
                    upd(macc[r] * gscale, p[off], m[off], v[off]);
                }
            }
        }
        return;
    }
    const int i = i0 + ti * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + to * 4 + r;
        if (o >= O) continue;
        const size_t off = (size_t)o * I + i;
        if (fast) {   // whole float4 inside the row, rows 16-byte aligned: prefetched above
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float pp = p4[r][c], mm = m4[r][c], vv = v4[r][c];
                upd(acc[r][c] * gscale, pp, mm, vv);
                p4[r][c] = pp; m4[r][c] = mm; v4[r][c] = vv;
            }
            __builtin_nontemporal_store(p4[r], reinterpret_cast<vf4*>(p + off));
            __builtin_nontemporal_store(m4[r], reinterpret_cast<vf4*>(m + off));
            __builtin_nontemporal_store(v4[r], reinterpret_cast<vf4*>(v + off));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (i + c < I) upd(acc[r][c] * gscale, p[off + c], m[off + c], v[off + c]);
        }
    }
}

// Skinny input gradient of a Linear layer: out[b, i] = sum_o g[b, o] * W[o, i]  (B <= 32 rows, O up to tens of
// thousands).  rocBLAS picks a 0.6 TB/s kernel for this shape; the product is bound by ONE read of W, so each block
// streams a slab of SK_ROWS weight rows (coalesced float4 across the 256 threads), keeps the [B, 4] partial sums of
// its 4 columns in registers and writes them as float4 rows of a per-block partial [nblk][B][I]; a second tiny kernel
// sums the partials in block order (deterministic; 16-byte-strided float atomics would run at a fraction of the
// atomic rate).
constexpr int SK_ROWS = 64, SK_B = 32;
__global__ __launch_bounds__(256) void linear_dx_skinny_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                               int B, int O, int I, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float sg[SK_ROWS][SK_B];   // g^T slab: [o][b]
    const int tid = threadIdx.x;
    const int o0 = blockIdx.y * SK_ROWS;
    const int i = (blockIdx.x * 256 + tid) * 4;
    for (int e = tid; e < SK_ROWS * SK_B; e += 256) {
        const int b = e / SK_ROWS, o = e - b * SK_ROWS;      // coalesced along o in global memory
        sg[o][b] = (b < B && o0 + o < O) ? g[(size_t)b * O + o0 + o] : 0.0f;
    }
    __syncthreads();
    if (i >= I) return;
    // two-wide accumulators: the inner product is VALU-bound with scalar FMAs (128 per weight row per thread), and
    // v_pk_fma_f32 retires two per lane per pass
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[SK_B][2];
#pragma unroll
    for (int b = 0; b < SK_B; ++b) acc[b][0] = acc[b][1] = f2{0.0f, 0.0f};
    const int rows = min(SK_ROWS, O - o0);
    // Software pipeline: one wave per SIMD runs here (188 workgroups of 64 rows for the 11988-row heads), so nothing
    // else hides the HBM latency -- the next 8 weight rows are requested before the current 8 are consumed.
    constexpr int U = 8;
    const float* Wc = W + (size_t)o0 * I + i;
    auto fetch = [&](float4 (&w)[U], int ob) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int o = min(ob + u, rows - 1);   // clamped: the duplicate rows are skipped in consume()
            w[u] = *reinterpret_cast<const float4*>(Wc + (size_t)o * I);
        }
    };
    auto consume = [&](const float4 (&w)[U], int ob) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ob + u < rows) {
                const f2 wlo = {w[u].x, w[u].y}, whi = {w[u].z, w[u].w};
#pragma unroll
                for (int b4 = 0; b4 < SK_B; b4 += 4) {
                    const float4 gv = *reinterpret_cast<const float4*>(&sg[ob + u][b4]);   // broadcast read of 4 batch rows
                    const float gb[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f2 gg = {gb[q], gb[q]};
                        acc[b4 + q][0] = __builtin_elementwise_fma(gg, wlo, acc[b4 + q][0]);
                        acc[b4 + q][1] = __builtin_elementwise_fma(gg, whi, acc[b4 + q][1]);
                    }
                }
            }
        }
    };
    float4 wa[U], wb[U];
    fetch(wa, 0);
    for (int ob = 0; ob < rows; ob += 2 * U) {
        fetch(wb, ob + U);
        consume(wa, ob);
        fetch(wa, ob + 2 * U);
        consume(wb, ob + U);
    }
    float* pb = partial + (size_t)blockIdx.y * B * I;
#pragma unroll   // compile-time register indices: a run-time loop bound would push acc[][] to scratch
    for (int b = 0; b < SK_B; ++b) {
        if (b < B) *reinterpret_cast<float4*>(pb + (size_t)b * I + i) = make_float4(acc[b][0].x, acc[b][0].y, acc[b][1].x, acc[b][1].y);
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nblk, int n4,
                                                           float* __restrict__ out)
{
    // 64 float4 elements of the [B, I] result per workgroup; the nblk partials of an element are split into four
    // contiguous slices (one per wave, 8 loads in flight each) and the slice sums are added in slice order, so the
    // summation tree is fixed: the result does not depend on scheduling.
    __shared__ float4 part[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int per = (nblk + 3) / 4;
    const int k0 = min(slice * per, nblk), k1 = min(k0 + per, nblk);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
        const float4* src = reinterpret_cast<const float4*>(partial) + e;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; k < k1; ++k) {
            const float4 v = src[(size_t)k * n4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    part[slice][lane] = a;
    __syncthreads();
    if (slice == 0 && e < n4) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 v = part[q][lane];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(out)[e] = a;
    }
}

}  // namespace

extern "C" int mp_adam_lowrank_f32(float* param, float* exp_avg, float* exp_avg_sq, const float* x, const float* g,
                                   int64_t Bg, int64_t O, int64_t I, double grad_scale, double lr, double beta1,
                                   double beta2, double eps, int64_t step, const float* step_dev, mp_stream_t stream_)
{
    if (Bg < 0 || O < 0 || I < 0 || (step <= 0 && !step_dev)) return MP_EINVAL;
    if (O == 0 || I == 0) return MP_OK;
    if (!param || !exp_avg || !exp_avg_sq || (Bg > 0 && (!x || !g))) return MP_EINVAL;
    const double sh = step > 0 ? (double)step : 1.0;   // placeholders when the device-side count is used
    const double c1 = 1.0 - pow(beta1, sh), c2 = 1.0 - pow(beta2, sh);
    const bool mfma_form = Bg > 32;
    const int64_t TI = mfma_form ? AL_TI : 256, TO = mfma_form ? AL_TO : 16;       // (the kernel's tile: see adam_lowrank_kernel)
    const dim3 grid((unsigned)((I + TI - 1) / TI), (unsigned)((O + TO - 1) / TO));
    // more factor rows than one GPU's batch (all-gathered factors of a data-parallel run): rebuild on the matrix cores
    const bool mfma = mfma_form;
    if (mfma)
        MP_LAUNCH("adam_lowrank_kernel<mfma>", 2.0 * (double)Bg * O * I, 24.0 * (double)O * I + 4.0 * Bg * (double)(O + I),
                  adam_lowrank_kernel<true>, grid, dim3(256), 0, mp_stream(stream_), param, exp_avg, exp_avg_sq, x, g, (int)Bg, (int)O,
                  (int)I, (float)grad_scale, (float)(lr / c1), (float)beta1, (float)beta2, (float)(1.0 / sqrt(c2)), (float)eps, step_dev,
                  (float)lr);
    else
        MP_LAUNCH("adam_lowrank_kernel", 2.0 * (double)Bg * O * I, 24.0 * (double)O * I + 4.0 * Bg * (double)(O + I),
                  adam_lowrank_kernel<false>, grid, dim3(256), 0, mp_stream(stream_), param, exp_avg, exp_avg_sq, x, g, (int)Bg, (int)O,
                  (int)I, (float)grad_scale, (float)(lr / c1), (float)beta1, (float)beta2, (float)(1.0 / sqrt(c2)), (float)eps, step_dev,
                  (float)lr);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" size_t mp_linear_dx_skinny_workspace_bytes(int64_t B, int64_t O, int64_t I)
{
    return (size_t)((O + SK_ROWS - 1) / SK_ROWS) * (size_t)B * (size_t)I * sizeof(float);
}

extern "C" int mp_linear_dx_skinny_f32(const float* g, const float* weight, int64_t B, int64_t O, int64_t I, float* grad_x,
                                       void* workspace, size_t workspace_bytes, mp_stream_t stream_)
{
    if (B < 0 || O < 0 || I < 0) return MP_EINVAL;
    if (B == 0 || I == 0) return MP_OK;
    if (!grad_x || (O > 0 && (!g || !weight || !workspace))) return MP_EINVAL;
    if (B > SK_B || (I & 3)) return MP_EUNSUPPORTED;
    if (workspace_bytes < mp_linear_dx_skinny_workspace_bytes(B, O, I)) return MP_EWORKSPACE;
    hipStream_t stream = mp_stream(stream_);
    if (O == 0) return mp::zero_async(grad_x, (size_t)(B * I), stream) ? MP_OK : MP_ELAUNCH;
    const int nblk = (int)((O + SK_ROWS - 1) / SK_ROWS);
    const dim3 grid((unsigned)((I / 4 + 255) / 256), (unsigned)nblk);
    MP_LAUNCH("linear_dx_skinny_kernel", 2.0 * (double)B * O * I, 4.0 * ((double)O * I + (double)B * (O + I)), linear_dx_skinny_kernel,
              grid, dim3(256), 0, stream, g, weight, (int)B, (int)O, (int)I, reinterpret_cast<float*>(workspace));
    MP_CHECK_LAUNCH();
    const int n4 = (int)(B * I / 4);
    MP_LAUNCH("sum_partials_kernel", 0.0, 4.0 * (double)nblk * B * I, sum_partials_kernel, dim3((n4 + 63) / 64), dim3(256), 0, stream,
              reinterpret_cast<const float*>(workspace), nblk, n4, grad_x);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
