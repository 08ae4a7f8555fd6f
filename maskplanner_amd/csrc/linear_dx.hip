// Input gradient of the weight-heavy head Linears on the matrix cores:  grad_x [B <= 64, I] = g [B, O] * W [O, I].
//
// Reference: autograd of nn.Linear for fc3 / fc_normals / sm_fc3 (models/pointnet2_cls_ssg.py:311, 327, 336): a [32, O] x [O, 1024]
// GEMM with O = 6 000 .. 12 000 -- 25 .. 49 MB of weights read for 0.4 .. 0.8 GF: an HBM stream.  rocBLAS picks a 0.6 TB/s kernel
// for the shape; linear_dx_skinny_kernel (adam_lowrank.hip: VALU inner products, one wave per SIMD, ordered partial sums) reaches
// ~1.5 TB/s and stays the deterministic form.  Here the product runs as v_mfma_f32_32x32x16_bf16 on (h, m, l) operand planes (six
// plane products per fp32 product, fp32 accumulation: fp32-accurate, see sa_mlp.hip split3): the batch is the 32-row side of the
// tile, a wave owns 32 columns of W and a slice of its rows, and the VALU only splits what streams past.
//   grid (I / 128, K slices), 256 threads: the four waves of a workgroup share the slice's g rows -- staged once into LDS as
//   ready-made A fragments -- and take 32 columns each; the weight rows are requested DEPTH k-steps (of 16 rows) ahead, eight
//   128-byte row segments per k-step and wave; the K slices add their tiles into grad_x with fp32 atomics (two 128-byte segments
//   per instruction), so grad_x must be zero on entry and the summation order is not fixed.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int LD_NS_MAX = 12;     // k-steps (of 16 weight rows) per workgroup: ALL of a wave's weight rows are requested up front (96 registers), before
                                  // the g slice is staged -- a workgroup lives for one memory latency, not one per prefetch round

struct Planes8 { bf16x8 h, m, l; };
__device__ __forceinline__ Planes8 split8(const float (&x)[8])
{
    Planes8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        r.h[i] = h;
        r.m[i] = m;
        r.l[i] = (__bf16)(r1 - (float)m);
    }
    return r;
}

// (g2, W2, O2): a second Linear fed by the same activation (fc3 / fc_normals): K slices >= slices1 belong to it, both add into gx.
// [r5] RT: 32-row tiles of the batch -- 1 (B <= 32) or 2 (B <= 64: the weight fragments are split once and meet both tiles).
template <int RT>
__global__ __launch_bounds__(256, RT == 1 ? 2 : 1) void linear_dx_mfma_kernel(const float* __restrict__ g1, const float* __restrict__ W1, int B, int O1, int I,
                                                                int ns, float* __restrict__ gx, const float* __restrict__ g2,
                                                                const float* __restrict__ W2, int O2, int slices1)
{
    __shared__ __attribute__((aligned(16))) __bf16 sG[RT][3][LD_NS_MAX][64][8];     // A fragments: [row tile][plane][k-step][lane][8 k values]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const bool second = (int)blockIdx.y >= slices1;
    const float* __restrict__ g = second ? g2 : g1;
    const float* __restrict__ W = second ? W2 : W1;
    const int O = second ? O2 : O1;
    const int k_base = ((int)blockIdx.y - (second ? slices1 : 0)) * ns * 16;
    // ---- B operand: lane (column c, half h) of k-step s holds W[k_base + 16 s + 8 h + j][col], j = 0..7 (rows past O: clamped --
    // their g values are zeros).  Requested first: everything below runs under their latency.
    const int col = blockIdx.x * 128 + wave * 32 + l31;
    const float* wc = W + col;
    float wq[LD_NS_MAX][8];
#pragma unroll
    for (int s = 0; s < LD_NS_MAX; ++s) {
        if (s < ns) {
            const int k0 = k_base + 16 * s + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) wq[s][j] = wc[(size_t)min(k0 + j, O - 1) * I];
        }
    }
    // ---- the slice's g rows as A fragments: lane (batch row r, half h) of k-step s holds g[r][k_base + 16 s + 8 h .. + 7]
    // (consecutive threads take consecutive 8-float pieces of one batch row: coalesced reads of an L2-resident table)
    for (int e = tid; e < RT * ns * 64; e += 256) {
        const int r = e / (2 * ns), sh = e - r * 2 * ns;      // r: batch row 0 .. 32 RT - 1
        const int s = sh >> 1, ln = (r & 31) + 32 * (sh & 1);
        const int k0 = k_base + 8 * sh;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (r < B && k0 + j < O) ? g[(size_t)r * O + k0 + j] : 0.0f;
        const Planes8 p = split8(x);
        *reinterpret_cast<bf16x8*>(&sG[r >> 5][0][s][ln][0]) = p.h;
        *reinterpret_cast<bf16x8*>(&sG[r >> 5][1][s][ln][0]) = p.m;
        *reinterpret_cast<bf16x8*>(&sG[r >> 5][2][s][ln][0]) = p.l;
    }
    __syncthreads();
    f32x16 acc[RT], cor[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; cor[t][r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < LD_NS_MAX; ++s) {
        if (s < ns) {
            const Planes8 b = split8(wq[s]);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sG[t][0][s][lane][0]);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(&sG[t][1][s][lane][0]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sG[t][2][s][lane][0]);
                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.h, cor[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.h, acc[t], 0, 0, 0);
                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.l, cor[t], 0, 0, 0);
                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b.m, cor[t], 0, 0, 0);
                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b.h, cor[t], 0, 0, 0);
                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.m, cor[t], 0, 0, 0);
            }
        }
    }
    // ---- the slice's tile into grad_x: register r of lane (c, h) is batch row (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < B) atomicAdd(gx + (size_t)row * I + col, acc[t][r] + cor[t][r]);
        }
}

// Dense weight gradient of the same layers for callers that need it materialised (an unchanged training loop with torch.optim.Adam
// over every parameter): dW [O, I] = g^T x, a rank-B outer-product sum (B <= 32) -- 32 FMAs per element against 4 bytes written:
// a write stream.  rocBLAS takes 85 us for the 11988 x 1024 heads (0.58 TB/s).  A workgroup owns 64 rows x 256 columns: g^T and the
// x columns in LDS, a thread 16 rows x 4 columns (b ascending: an fma chain per element), float4 stores.
__global__ __launch_bounds__(256) void linear_dw_outer_kernel(const float* __restrict__ g, const float* __restrict__ x, int B, int O, int I,
                                                              float* __restrict__ dW)
{
    __shared__ __attribute__((aligned(16))) float sg[32][64 + 1];      // [b][o]
    __shared__ __attribute__((aligned(16))) float sx[32][256];         // [b][i]
    const int tid = threadIdx.x;
    const int o0 = blockIdx.y * 64, i0 = blockIdx.x * 256;
    const int cq = tid & 63, rg = tid >> 6;            // columns i0 + 4 cq .. + 3, rows o0 + 16 rg .. + 15
    float4 acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = 0; b0 < B; b0 += 32) {               // ([r5] batches beyond 32 rows: slabs of 32, b ascending throughout)
        if (b0) __syncthreads();
        for (int e = tid; e < 32 * 64; e += 256) {
            const int b = b0 + (e >> 6), o = e & 63;
            sg[e >> 6][o] = (b < B && o0 + o < O) ? g[(size_t)b * O + o0 + o] : 0.0f;
        }
        for (int e = tid; e < 32 * 64; e += 256) {
            const int b = b0 + (e >> 6), q = e & 63;
            const float4 v = (b < B && i0 + 4 * q < I) ? *reinterpret_cast<const float4*>(x + (size_t)b * I + i0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&sx[e >> 6][4 * q]) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int b = 0; b < 32; ++b) {
            const float4 xv = *reinterpret_cast<const float4*>(&sx[b][4 * cq]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float gv = sg[b][16 * rg + r];       // (broadcast: one address per wave)
                acc[r].x = __builtin_fmaf(gv, xv.x, acc[r].x); acc[r].y = __builtin_fmaf(gv, xv.y, acc[r].y);
                acc[r].z = __builtin_fmaf(gv, xv.z, acc[r].z); acc[r].w = __builtin_fmaf(gv, xv.w, acc[r].w);
            }
        }
    }
    if (i0 + 4 * cq < I) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + 16 * rg + r;
            if (o < O) *reinterpret_cast<float4*>(dW + (size_t)o * I + i0 + 4 * cq) = acc[r];
        }
    }
}

}  // namespace

extern "C" int mp_linear_dw_outer_f32(const float* g, const float* x, int64_t B, int64_t O, int64_t I, float* dW, mp_stream_t stream_)
{
    if (B < 0 || O < 0 || I < 0) return MP_EINVAL;
    if (O == 0 || I == 0) return MP_OK;
    if (!dW || (B > 0 && (!g || !x))) return MP_EINVAL;
    if (B > 4096 || (I & 3) || O >= ((int64_t)1 << 30)) return MP_EUNSUPPORTED;
    MP_LAUNCH("linear_dw_outer_kernel", 2.0 * (double)B * O * I, 4.0 * ((double)O * I + (double)B * (O + I)), linear_dw_outer_kernel,
              dim3((unsigned)((I + 255) / 256), (unsigned)((O + 63) / 64)), dim3(256), 0, mp_stream(stream_), g, x, (int)B, (int)O, (int)I, dW);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

static int linear_dx_mfma2(const float* g1, const float* w1, int64_t O1, const float* g2, const float* w2, int64_t O2, int64_t B, int64_t I,
                           float* grad_x, mp_stream_t stream_)
{
    if (B < 0 || O1 < 0 || O2 < 0 || I < 0) return MP_EINVAL;
    if (B == 0 || I == 0) return MP_OK;
    if (!grad_x || (O1 > 0 && (!g1 || !w1)) || (O2 > 0 && (!g2 || !w2))) return MP_EINVAL;
    if (B > 64 || (I % 128) != 0 || O1 >= ((int64_t)1 << 30) || O2 >= ((int64_t)1 << 30)) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    if (!mp::zero_async(grad_x, (size_t)(B * I), stream)) return MP_ELAUNCH;
    if (O1 == 0 && O2 == 0) return MP_OK;
    if (O1 == 0) { O1 = O2; g1 = g2; w1 = w2; O2 = 0; }
    const int64_t Omax = O1 > O2 ? O1 : O2, ksteps = (Omax + 15) / 16, ncol = I / 128;
    int64_t slices = (768 + ncol - 1) / ncol;                 // ~three workgroups per CU
    if (slices > ksteps) slices = ksteps;
    int64_t ns = (ksteps + slices - 1) / slices;
    if (ns > LD_NS_MAX) { ns = LD_NS_MAX; }
    const int64_t s1 = ((O1 + 15) / 16 + ns - 1) / ns, s2 = O2 > 0 ? ((O2 + 15) / 16 + ns - 1) / ns : 0;
    const double Ot = (double)(O1 + O2);
    if (B <= 32)
        MP_LAUNCH("linear_dx_mfma_kernel", 2.0 * (double)B * Ot * I, 4.0 * (Ot * I + (double)B * (Ot + I)), linear_dx_mfma_kernel<1>,
                  dim3((unsigned)ncol, (unsigned)(s1 + s2)), dim3(256), 0, stream, g1, w1, (int)B, (int)O1, (int)I, (int)ns, grad_x, g2, w2, (int)O2, (int)s1);
    else
        MP_LAUNCH("linear_dx_mfma_kernel", 2.0 * (double)B * Ot * I, 4.0 * (Ot * I + (double)B * (Ot + I)), linear_dx_mfma_kernel<2>,
                  dim3((unsigned)ncol, (unsigned)(s1 + s2)), dim3(256), 0, stream, g1, w1, (int)B, (int)O1, (int)I, (int)ns, grad_x, g2, w2, (int)O2, (int)s1);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_linear_dx_mfma_f32(const float* g, const float* weight, int64_t B, int64_t O, int64_t I, float* grad_x, mp_stream_t stream_)
{
    return linear_dx_mfma2(g, weight, O, nullptr, nullptr, 0, B, I, grad_x, stream_);
}

extern "C" int mp_linear_dx_mfma2_f32(const float* g1, const float* w1, int64_t O1, const float* g2, const float* w2, int64_t O2, int64_t B,
                                      int64_t I, float* grad_x, mp_stream_t stream_)
{
    return linear_dx_mfma2(g1, w1, O1, g2, w2, O2, B, I, grad_x, stream_);
}
