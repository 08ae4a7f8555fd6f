// Shared MLP + BatchNorm + ReLU + max-pool of a set-abstraction level, forward and backward, for gfx950.
//
// Reference: models/pointnet2_utils.py:208-214 -- `relu(bn(conv1x1(x)))` x L over [B,C,K,S], then max over K --
// executed there as 3L+1 library kernels that each materialise a [B,C,K,S] tensor, plus autograd's mirror image.
//
// Here a level is a chain of fp32 GEMMs over positions-major activations X[P, C] (P = B*S*K rows):
//     Z_l = act_{l-1}(Z_{l-1}) * W_l^T ,   act_l(z) = relu(z * scale_l + shift_l)
// where (scale, shift) fold BatchNorm (batch statistics in training, running statistics in eval) and the conv
// bias.  Only the RAW pre-BN activations Z_l are ever written to HBM:
//   * BN + ReLU of layer l-1 are applied while the A tile of GEMM l is staged global -> LDS;
//   * the per-channel sums that BatchNorm needs (sum z, sum z^2) are produced by the GEMM epilogue as per-block
//     partials and reduced in fp64 by a tiny finalize kernel (deterministic, no atomics);
//   * the max over K (with arg-max, first index wins) applies BN + ReLU of the last layer on the fly.
// Backward mirrors it: dZ_l = a*relu'(.)*G_l + e*Z_l + f (BatchNorm backward folded into three per-channel
// constants) is formed while staging tiles, never stored:
//   * G_{l-1} = dZ_l * W_l            (GEMM, epilogue accumulates the BN-backward sums of layer l-1)
//   * dW_l    = dZ_l^T * act(Z_{l-1}) (split-K GEMM over P, fp32 atomics into dW)
//   * the gradient of the max-pool is never densified: the loader reads (argmax, pooled grad) per group.
//
// GEMM core: v_mfma_f32_32x32x2_f32 (exact fp32, == an fma chain in k order), 256 threads = 4 waves, block tile
// 128 x 128 (2x2 waves of 64x64) or 128 x 64 (4x1 waves of 32x64), K chunks of 32 staged through LDS with register
// prefetch of the next chunk (global loads in flight under the MFMAs).  LDS tiles are either [row][32+1] (operand
// whose K is contiguous in memory; odd stride => conflict-free ds_read_b32 fragments) or [k][row] (operand whose
// rows are contiguous: a straight float4 copy).
#include "sa_common.h"

namespace {


// out = relu(z* * scale + shift) with z* = group max (scale >= 0) or group min (scale < 0) of the raw activations
__global__ __launch_bounds__(256) void pool_select_kernel(PoolOut po, const float* s, const float* t, int64_t G, int C,      // (s, t: written by bn_prologue)
                                                          float* __restrict__ out, int* __restrict__ argk,
                                                          float* __restrict__ zmax, BnSite bn)
{   // four channels per thread (C % 4 == 0, every buffer 16-byte aligned: the host checks); the min side is only read where a
    // scale is negative.  Grid-stride: with its own BatchNorm prologue (bn.slots) the host launches fewer, longer workgroups.
    __shared__ __attribute__((aligned(16))) float bn_lds[2 * BN_POOL_CMAX];
    const int64_t tot = G * C;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    // the first element's loads are issued BEFORE the prologue (its slot reads then share their latency)
    float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 k0 = make_int4(0, 0, 0, 0);
    if (e0 < tot) { z0 = *reinterpret_cast<const float4*>(po.vmax + e0); k0 = *reinterpret_cast<const int4*>(po.imax + e0); }
    bn_prologue(bn, bn_lds, BN_POOL_CMAX, 0, C, blockIdx.x == 0);
    const float* sp = bn.slots ? bn_lds : s;
    const float* tp = bn.slots ? bn_lds + BN_POOL_CMAX : t;
    for (int64_t e = e0; e < tot; e += (int64_t)gridDim.x * 1024) {
        const int c = (int)(e % C);
        const float4 sc = *reinterpret_cast<const float4*>(sp + c), sh = *reinterpret_cast<const float4*>(tp + c);
        float4 z = e == e0 ? z0 : *reinterpret_cast<const float4*>(po.vmax + e);
        int4 k = e == e0 ? k0 : *reinterpret_cast<const int4*>(po.imax + e);
        if (sc.x < 0.0f || sc.y < 0.0f || sc.z < 0.0f || sc.w < 0.0f) {
            const float4 zn = *reinterpret_cast<const float4*>(po.vmin + e);
            const int4 kn = *reinterpret_cast<const int4*>(po.imin + e);
            if (sc.x < 0.0f) { z.x = zn.x; k.x = kn.x; }
            if (sc.y < 0.0f) { z.y = zn.y; k.y = kn.y; }
            if (sc.z < 0.0f) { z.z = zn.z; k.z = kn.z; }
            if (sc.w < 0.0f) { z.w = zn.w; k.w = kn.w; }
        }
        float4 y;
        y.x = z.x * sc.x + sh.x; y.y = z.y * sc.y + sh.y; y.z = z.z * sc.z + sh.z; y.w = z.w * sc.w + sh.w;
        y.x = y.x > 0.0f ? y.x : 0.0f; y.y = y.y > 0.0f ? y.y : 0.0f; y.z = y.z > 0.0f ? y.z : 0.0f; y.w = y.w > 0.0f ? y.w : 0.0f;
        *reinterpret_cast<float4*>(out + e) = y;
        *reinterpret_cast<int4*>(argk + e) = k;
        *reinterpret_cast<float4*>(zmax + e) = z;
    }
}

__global__ __launch_bounds__(256) void pool_select_scalar_kernel(PoolOut po, const float* s, const float* t, int64_t G, int C,
                                                                 float* __restrict__ out, int* __restrict__ argk,
                                                                 float* __restrict__ zmax, BnSite bn)
{
    __shared__ float bn_lds[2 * BN_POOL_CMAX];
    bn_prologue(bn, bn_lds, BN_POOL_CMAX, 0, C, blockIdx.x == 0);
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= G * C) return;
    const int c = (int)(e % C);
    const float sc = bn.slots ? bn_lds[c] : s[c];
    const bool up = sc >= 0.0f;
    const float z = up ? po.vmax[e] : po.vmin[e];
    const float y = z * sc + (bn.slots ? bn_lds[BN_POOL_CMAX + c] : t[c]);
    out[e] = y > 0.0f ? y : 0.0f;
    argk[e] = up ? po.imax[e] : po.imin[e];
    zmax[e] = z;
}

// =================================================================================================================
// Kernel 1b: forward layer as a position stream, for the layers behind the first one (CI in {64, 128} inputs, CO in
// {64, 128, 256} outputs).  A workgroup walks p_per_block positions in chunks of 32: the activated input chunk is staged
// in LDS, every wave owns CO / waves output columns whose weight fragments stay in registers for the whole kernel
// (v_mfma_f32_16x16x4_f32: 2 row tiles x HT column tiles per wave and chunk), and the epilogue of a chunk -- raw Z store,
// per-column BatchNorm sums, running max / min of the groups for the fused pool -- is 8 rows per lane, interleaved with the
// other waves' MFMAs instead of being a separate phase of a 128 x 128 tile.  One partial-sum row per workgroup
// (P / p_per_block rows instead of P / 128) also makes the BatchNorm finalize kernel 8x shorter.
// =================================================================================================================
#ifndef MP_FPD2_ONE
#define MP_FPD2_ONE 1       // [r4] ... of the one-plane (bf16 variant) forward kernels: on (config 5: 6.06 -> 6.02 ms, the forward kernels 1 080 -> 1 063 us)
#endif
#ifndef MP_FPD2
#define MP_FPD2 0           // [r3] position-stream forward: two chunks of loads in flight
#endif
// ONE (with SPLIT): the bf16 variant of BASELINE configs[4] -- operands rounded once to bf16 (the h plane alone), one product instead of six
template <int CI, int CO, bool POOL, int MODE_A = SRC_ACT, int TAIL = 0, bool SPLIT = false, bool STORE = true, bool ONE = false>   // TAIL = 4: input rows are [CI | 4] wide (features | xyz + pad)
__global__ __launch_bounds__(CO > 128 ? 512 : 256) void fwd_chunk_kernel(PosOperand A, int P, int p_per_block,
                                                                         const float* __restrict__ W, float* __restrict__ Z,
                                                                         BnOut partials, PoolOut po,
                                                                         const float* __restrict__ gamma)
{   // gamma (POOL): this layer's BatchNorm weight -- its sign is the sign of the affine scale, i.e. whether the pool of a column
    // takes the group's largest or smallest raw value, so every lane tracks ONE extremum
    bn_zero(partials);
    __shared__ __attribute__((aligned(16))) float bn_lds[2 * CI];           // (scale, shift) of the input activation when this kernel derives them
    constexpr int NT = CO > 128 ? 512 : 256, NW = NT / 64;
    constexpr int CW = CO / NW;                       // columns per wave: 32 (one 32x32 MFMA tile) or 16 (two 16x16 tiles)
    constexpr bool BIG = CW == 32;                    // v_mfma_f32_32x32x2_f32: a lane's 32 consecutive columns = whole 128-B lines
    constexpr int DBK = 32, LDA = CI + 4;             // 16-byte aligned rows; stride = 4 mod 32 banks
    constexpr int PA = DBK * CI / 4 / NT;             // float4 loads per thread and chunk
    constexpr int NFR = BIG ? CI / 2 : CI / 4;        // weight fragment registers per lane
    constexpr int NV = BIG ? 16 : 8;                  // rows of the chunk held by one lane
    static_assert((CW == 32 || CW == 16) && PA >= 1, "shape");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // SPLIT: the chunk as three bf16 planes [plane][row][k] (rows 16-byte aligned, stride = 4 dwords mod 64 banks)
    constexpr int LDH = CI + 8;
    __shared__ __attribute__((aligned(16))) float sA[2][SPLIT ? 4 : DBK * LDA];
    constexpr int NPLN = ONE ? 1 : 3;        // ([r4] ONE: the single plane only -- see bwd_fused_kernel)
    __shared__ __attribute__((aligned(16))) __bf16 sH[2][NPLN][SPLIT ? DBK * LDH : 8];
    __shared__ float4 sT[2][TAIL ? DBK : 1];           // TAIL: the 4 extra input columns of the chunk's rows
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lc = BIG ? (lane & 31) : (lane & 15);   // column of this lane inside the wave's tile
    const int kq = BIG ? (lane >> 5) : (lane >> 4);   // k sub-index / row group of this lane
    const int col = wave * CW + lc;
    // [r5] negative p_per_block: interleaved workgroups (see bwd_fused_kernel) -- workgroup w takes units w, w + grid, ... where a unit is a
    // chunk, or with the fused pool a whole group of K positions (its chunks stay in order inside one workgroup)
    const bool il = p_per_block < 0;
    if (il) p_per_block = -p_per_block;
    const int cpu_ = POOL ? po.K / DBK : 1;           // chunks per unit
    const int U = cpu_ * DBK;
    const int p0 = il ? 0 : blockIdx.x * p_per_block;
    const int p1 = il ? P : min(P, p0 + p_per_block);
    const int nchunks = il ? (((P + U - 1) / U - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * cpu_ : (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;
    auto cpos = [&](int c) {                          // first position of this workgroup's chunk c
        if (!il) return p0 + c * DBK;
        const int u = POOL ? c / cpu_ : c, r = POOL ? c - u * cpu_ : 0;
        return (u * (int)gridDim.x + (int)blockIdx.x) * U + r * DBK;
    };

    // B[k][n] = W[n][k]: 32x32x2 lane (n, kq) holds W[col][2*st + kq]; 16x16x4 lane (n, kq) holds W[col][4*st + kq]
    float wfrag[SPLIT ? 1 : NFR];
    constexpr int KST = BIG ? 16 : 32;               // K per split MFMA step: 32x32x16 (two k-groups of 8) or 16x16x32 (four)
    bf16x8 wsp[SPLIT ? 3 : 1][SPLIT ? CI / KST : 1];  // SPLIT: lane (n, kq) holds W[col][KST*st + 8*kq .. + 7] as (h, m, l) planes
    if constexpr (SPLIT) {
#pragma unroll
        for (int st = 0; st < CI / KST; ++st) {
            const float* wp = W + (size_t)col * (CI + TAIL) + KST * st + 8 * kq;
            const Split4 lo = split3(ld4(wp)), hi = split3(ld4(wp + 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wsp[0][st][i] = lo.h[i]; wsp[0][st][4 + i] = hi.h[i];
                wsp[1][st][i] = lo.m[i]; wsp[1][st][4 + i] = hi.m[i];
                wsp[2][st][i] = lo.l[i]; wsp[2][st][4 + i] = hi.l[i];
            }
        }
    } else {
#pragma unroll
        for (int st = 0; st < NFR; ++st) wfrag[st] = W[(size_t)col * (CI + TAIL) + (BIG ? 2 : 4) * st + kq];
    }
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);     // TAIL: this lane's column of the 4 extra weight columns (VALU, not MFMA:
    if constexpr (TAIL != 0) wt = ld4(W + (size_t)col * (CI + TAIL) + CI);   // a 32-wide k tile would be 1/8 full)
    if constexpr (TAIL != 0 && ONE) { wt.x = (float)(__bf16)wt.x; wt.y = (float)(__bf16)wt.y; wt.z = (float)(__bf16)wt.z; wt.w = (float)(__bf16)wt.w; }

    const int ca = (tid % (CI / 4)) * 4, ka0 = tid / (CI / 4);
    constexpr int KA_STEP = NT / (CI / 4);
    ChanConst kc;
    struct RSet { Raw4<MODE_A> a[PA]; Raw4<MODE_A> t; };
    RSet rs0, rs1;                             // (rs1: FPD2 only)
    auto gload = [&](int pk, RSet& rs) {
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) raw_load<MODE_A, ONE>(A, p1, pk + ka0 + ps * KA_STEP, ca, rs.a[ps]);
        if constexpr (TAIL != 0) { if (tid < DBK) raw_load<MODE_A, ONE>(A, p1, pk + tid, CI, rs.t); }
    };
    auto sstore = [&](int buf, RSet& rs) {
        auto& ra = rs.a;
        auto& rt = rs.t;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            if constexpr (SPLIT) {
                const Split4 sp = split3(finish<MODE_A>(ra[ps], kc));
                const int o = (ka0 + ps * KA_STEP) * LDH + ca;
                *reinterpret_cast<bf16x4*>(&sH[buf][0][o]) = sp.h;
                if constexpr (!ONE) {
                    *reinterpret_cast<bf16x4*>(&sH[buf][1][o]) = sp.m;
                    *reinterpret_cast<bf16x4*>(&sH[buf][2][o]) = sp.l;
                }
            } else {
                *reinterpret_cast<float4*>(&sA[buf][(ka0 + ps * KA_STEP) * LDA + ca]) = finish<MODE_A>(ra[ps], kc);
            }
        }
        if constexpr (TAIL != 0) {
            if (tid < DBK) {
                float4 xt = finish<MODE_A>(rt, kc);
                if constexpr (ONE) { xt.x = (float)(__bf16)xt.x; xt.y = (float)(__bf16)xt.y; xt.z = (float)(__bf16)xt.z; xt.w = (float)(__bf16)xt.w; }
                sT[buf][tid] = xt;
            }
        }
    };

    double s1 = 0.0, s2 = 0.0;     // the chunk's values are summed in fp32, the 16-32 chunks of a workgroup in fp64
    float gbest = 0.0f;
    int gibest = 0;
    const int cpg = POOL ? po.K / DBK : 1;            // chunks per group (1, 2 or 4)
    // fp32 MFMAs run on the SIMD's own FMA lanes: every other vector instruction of the wave pair takes time away from them
    // (profiles/r02: MFMA busy + VALU busy + LDS ~ 100 %), so the epilogue is written for instruction count:
    //  * raw Z goes out through buffer stores: the row part that is uniform over the wave sits in the scalar offset, the lane
    //    part in one VGPR advanced once per chunk, and rows past the workgroup's last position are dropped by the hardware's
    //    range check -- no 64-bit address arithmetic, no EXEC masking per row;
    //  * the BatchNorm sums run on register pairs (v_pk_add_f32 / v_pk_mul_f32); rows past P are exact zeros and need no mask;
    //  * the pool tracks one extremum per column (sign of gamma), as a maximum of the sign-flipped value.
    typedef float f2 __attribute__((ext_vector_type(2)));
    // ONE: Z is STORED as bf16 (2 bytes per element): adjacent lanes hold adjacent columns, so a lane pair exchanges one value per row
    // pair and each lane stores one dword -- the even lane (col, col + 1) of the even row, the odd lane the same columns of the odd row
    constexpr int ZB = ONE ? 2 : 4;
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(Z) + (size_t)p0 * CO * ZB, 0, (p1 - p0) * CO * ZB, 0x00020000);
    const int zoff0 = ONE ? ((4 * kq) * CO + (col & ~1)) * 2 + ((lane & 1) ? CO * 2 : 0)
                   : ((BIG ? 4 * kq : 4 * kq) * CO + col) * 4;     // byte offset of this lane's first row inside the workgroup's slab
    bool neg = false;
    if constexpr (POOL) neg = gamma[col] < 0.0f;
    const unsigned smask = neg ? 0x80000000u : 0u;

    // ([r3] tried: the staggered barrier of bwd_fused_kernel's DESYNC for the 512-thread form -- 110.7 vs 111.0 us, not kept)
    // FPD2: two chunks of loads in flight (two register sets, the loop unrolled by two), as bwd_fused_kernel's PD2
    constexpr bool FPD2 = (ONE ? MP_FPD2_ONE : MP_FPD2) && SPLIT;
    gload(cpos(0), rs0);
    bn_prologue(A.bn, bn_lds, CI, 0, CI, blockIdx.x == 0);      // (behind the first chunk's loads: its slot reads share their latency)
    load_consts<MODE_A>(A, ca, kc, bn_lds, CI);
    sstore(0, rs0);
    if constexpr (FPD2) {
        if (nchunks > 1) gload(cpos(1), rs0);
        if (nchunks > 2) gload(cpos(2), rs1);
    }
    __syncthreads();
    auto body = [&](const int kcn, RSet& rs) {
        const int cur = kcn & 1;
        if (!FPD2 && kcn + 1 < nchunks) gload(cpos(kcn + 1), rs);
        const int pkc = cpos(kcn);
        const int zoff = zoff0 + (pkc - p0) * CO * ZB;
        float v[NV];
        if constexpr (SPLIT && !BIG) {
            // 16 columns per wave: two 16x16 row tiles, v_mfma_f32_16x16x32_bf16 (lane (row, kq) holds A[row][32*st + 8*kq .. + 7])
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            const int ao = lc * LDH + 8 * kq;
#pragma unroll
            for (int st = 0; st < CI / 32; ++st) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int o = ao + rt * 16 * LDH + 32 * st;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sH[cur][0][o]);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(&sH[cur][ONE ? 0 : 1][o]);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sH[cur][ONE ? 0 : 2][o]);
                    f32x4& a = rt ? a1 : a0;
                    f32x4& c = rt ? c1 : c0;
                    if constexpr (!ONE) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wsp[0][st], c, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[0][st], a, 0, 0, 0);
                    if constexpr (!ONE) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[2][st], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[1][st], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[0][st], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[1][st], c, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a0[i] + c0[i]; v[4 + i] = a1[i] + c1[i]; }   // rows 4 * kq + i and 16 + 4 * kq + i
        } else if constexpr (SPLIT) {
            // two accumulators: the leading products and the five corrections (summed among themselves first, and two
            // independent MFMA chains instead of one)
            f32x16 acc, cor;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; cor[r] = 0.0f; }
            const int ao = (lane & 31) * LDH + 8 * kq;                // A[row = lane & 31][k = 16*st + 8*kq .. + 7]
#pragma unroll
            for (int st = 0; st < CI / 16; ++st) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sH[cur][0][ao + 16 * st]);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(&sH[cur][ONE ? 0 : 1][ao + 16 * st]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sH[cur][ONE ? 0 : 2][ao + 16 * st]);
                if constexpr (!ONE) cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wsp[0][st], cor, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wsp[0][st], acc, 0, 0, 0);
                if constexpr (!ONE) {
                    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wsp[2][st], cor, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wsp[1][st], cor, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wsp[0][st], cor, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wsp[1][st], cor, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r] + cor[r];
        } else if constexpr (BIG) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const float* ap = sA[cur] + (lane & 31) * LDA + kq;       // A[row = lane & 31][k = 2*st + kq]
#pragma unroll
            for (int st = 0; st < NFR; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], wfrag[st], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r];               // rows (r & 3) + 8 * (r >> 2) + 4 * kq: ascending in r
        } else {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            const float* a0p = sA[cur] + lc * LDA + kq;               // A[row = 16*rt + lc][k = 4*st + kq]
            const float* a1p = a0p + 16 * LDA;
#pragma unroll
            for (int st = 0; st < NFR; ++st) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[4 * st], wfrag[st], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[4 * st], wfrag[st], a1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a0[i]; v[4 + i] = a1[i]; }   // rows 4 * kq + i and 16 + 4 * kq + i
        }
        // row of accumulator r relative to the lane part (4 * kq): uniform over the wave
        auto rowc = [](int r) { return BIG ? (r & 3) + 8 * (r >> 2) : (r < 4 ? r : 16 + (r - 4)); };
        if constexpr (TAIL != 0) {   // k = CI .. CI+3 appended to each element's FMA chain, in k order
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const float4 xt = sT[cur][4 * kq + rowc(r)];
                v[r] = __builtin_fmaf(xt.w, wt.w, __builtin_fmaf(xt.z, wt.z, __builtin_fmaf(xt.y, wt.y, __builtin_fmaf(xt.x, wt.x, v[r]))));
            }
        }
        f2 c1 = {0.0f, 0.0f}, c2 = {0.0f, 0.0f};
        float lbest = -__builtin_inff();
        int libest = 0;
#pragma unroll
        for (int r = 0; r < NV; r += 2) {
            if constexpr (STORE && ONE) {
                const bool odd = lane & 1;
                const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[r] : v[r + 1]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                __builtin_amdgcn_raw_buffer_store_b32(odd ? pack_bf16(got, v[r + 1]) : pack_bf16(v[r], got), zrsrc, zoff, rowc(r) * CO * 2, MP_STORE_AUX);
            } else if constexpr (STORE) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), zrsrc, zoff, rowc(r) * CO * 4, MP_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r + 1]), zrsrc, zoff, rowc(r + 1) * CO * 4, MP_STORE_AUX);
            }
            const f2 x = {v[r], v[r + 1]};
            c1 += x;
            c2 += x * x;
            if constexpr (POOL) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float t = __uint_as_float(__float_as_uint(v[r + u]) ^ smask);   // -x where the column pools its minimum
                    if (t > lbest) { lbest = t; libest = rowc(r + u); }                    // strict: the first extremum stays
                }
            }
        }
        s1 += (double)(c1.x + c1.y);
        s2 += (double)(c2.x + c2.y);
        if constexpr (POOL) {
            libest += 4 * kq;
            // the lane groups of a column hold interleaved rows: lexicographic (value, row) combine
#pragma unroll
            for (int d = BIG ? 32 : 16; d <= 32; d <<= 1) {
                const float ox = __shfl_xor(lbest, d, 64);
                const int oix = __shfl_xor(libest, d, 64);
                if (ox > lbest || (ox == lbest && oix < libest)) { lbest = ox; libest = oix; }
            }
            const int cig = kcn % cpg;                // chunk inside its group (p0 is a multiple of K)
            if (cig == 0 || lbest > gbest) { gbest = lbest; gibest = cig * DBK + libest; }   // earlier chunk wins ties
            if (cig == cpg - 1 && kq == 0) {
                const int pk = pkc;
                const size_t o = (size_t)((unsigned)(pk / po.K) * (unsigned)CO + (unsigned)col);
                const float val = __uint_as_float(__float_as_uint(gbest) ^ smask);
                if (neg) { po.vmin[o] = val; po.imin[o] = gibest; }
                else { po.vmax[o] = val; po.imax[o] = gibest; }
            }
        }
        if (kcn + 1 < nchunks) sstore(cur ^ 1, rs);
        if (FPD2 && kcn + 3 < nchunks) gload(cpos(kcn + 3), rs);
        __syncthreads();
    };
    if constexpr (FPD2) {
        for (int kcn = 0; kcn < nchunks; kcn += 2) {
            body(kcn, rs0);
            if (kcn + 1 < nchunks) body(kcn + 1, rs1);
        }
    } else {
        for (int kcn = 0; kcn < nchunks; ++kcn) body(kcn, rs0);
    }
#pragma unroll
    for (int d = BIG ? 32 : 16; d <= 32; d <<= 1) {
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
    }
    if (kq == 0) {
        bn_emit(partials, CO, blockIdx.x, col, s1, s2);
    }
}

// (Kernel 4 / 4b -- the fused backward of the single-tile layers and its role-split form -- live in sa_bwd_fused.hip.)


// BatchNorm statistics of a RECOMPUTED first layer (SRC_*_RC): per workgroup the sums of z and z^2 of its positions, in the
// partials layout of the GEMM epilogues ([block][2][C]) -- the layer's forward pass is this kernel and nothing else.
// A wave loads 64 input rows with one coalesced access (lane i: position base + i) and every lane (= channel) walks them
// through v_readlane; sums run in fp32 over the 64 positions and in fp64 across them.
__global__ __launch_bounds__(256) void rc_stats_kernel(const float* __restrict__ x0, const float* __restrict__ W0, int P, int p_per_block,
                                                       BnOut partials)
{
    bn_zero(partials);
    constexpr int C = 64;
    __shared__ double red[4][2][C];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float4 w = ld4(W0 + (size_t)lane * 4);
    const int p0 = blockIdx.x * p_per_block, p1 = min(P, p0 + p_per_block);
    double S1 = 0.0, S2 = 0.0;
    for (int base = p0 + wave * 64; base < p1; base += 256) {
        const int pos = base + lane;
        const float4 xr = pos < p1 ? ld4(x0 + (size_t)pos * 4) : make_float4(0.f, 0.f, 0.f, 0.f);   // rows past the end give z = 0
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            float4 x;
            x.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr.x), j));
            x.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr.y), j));
            x.z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr.z), j));
            x.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xr.w), j));
            const float z = dot4_rc(x, w);
            s1 += z;
            s2 = __builtin_fmaf(z, z, s2);
        }
        S1 += (double)s1;
        S2 += (double)s2;
    }
    red[wave][0][lane] = S1;
    red[wave][1][lane] = S2;
    __syncthreads();
    if (tid < 2 * C) {
        const int st = tid / C, c = tid - st * C;
        const double v = ((red[0][st][c] + red[1][st][c]) + red[2][st][c]) + red[3][st][c];
        if (partials.slots) atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + st) * C + c, v);
        else partials.rows[((size_t)blockIdx.x * 2 + st) * C + c] = (float)v;
    }
}

// dW for a 4-channel input (the first layer of the first level: xyz + pad): a [CO x 4] result is no MFMA shape (a 128 x 32
// tile would be 1/16 full), and the kernel is a pure stream over dZ = (Z_1, G_1).  Thread (position lane, channel quad):
// 64 position lanes x (CO/4) quads; every thread walks positions lane, lane+PL, ... of its workgroup's slice with float4
// loads of dZ and of the input row, 16 FMAs each; lanes are combined through LDS and added to dW with atomics.
// Kernel 4b: the same single pass for the FIRST layer of a level whose grouped input is [128 features | xyz | pad] (132
// columns): dW [128 x 132] (128 columns by MFMA, the 4 coordinate columns by plain FMAs on the staged tiles) and the
// feature part of grad_x0 (128 columns; coordinates carry no gradient), no BatchNorm sums (there is no layer below).
template <int MODE_DZ, bool SPLIT = false, int MODE_IN = SRC_ID>
__global__ __launch_bounds__(512) void bwd_first_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block,
                                                           const float* __restrict__ W, float* __restrict__ dW,
                                                           float* __restrict__ G)
{
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * 128];
    bn_prologue(DZ.bn, bn_lds, 128, 0, 128, blockIdx.x == 0);
    constexpr int CO = 128, CIW = 132, CIX = 128, DBK = 16, LDA = CO + 4, LDB = CIW, KPL = CO / 4;   // LDA, KPL: see bwd_fused_kernel
    constexpr int NT = 512;                             // eight waves: half the accumulators / weight fragments per wave
    constexpr int PA = DBK * CO / 4 / NT;               // 1
    constexpr int NB4 = DBK * CIW / 4;                  // 528 float4 of the input chunk
    constexpr int PB = (NB4 + NT - 1) / NT;             // 2 (the last pass has 16 live threads)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // SPLIT: dZ and the 128 feature columns of the input as (h, m, l) bf16 planes in the K-packed layout of bwd_fused_kernel; the
    // fp32 dZ chunk and the 4 coordinate columns stay beside them for the four coordinate columns of dW (plain FMAs)
    constexpr int GS = DBK * 8 + 32;
    __shared__ __attribute__((aligned(16))) float sA[2][DBK * LDA];
    __shared__ __attribute__((aligned(16))) float sB[2][SPLIT ? 4 : DBK * LDB];
    __shared__ __attribute__((aligned(16))) __bf16 hA[2][3][SPLIT ? (CO / 8) * GS : 8];
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][3][SPLIT ? (CIX / 8) * GS : 8];
    __shared__ float4 sT[2][SPLIT ? DBK : 1];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, l31 = lane & 31;
    const int wrow0 = (wave >> 1) * 32, wcol0 = (wave & 1) * 64;   // dW tiles: waves 4 x 2, 1 x 2 tiles of 32 x 32 each
    const int p0 = blockIdx.x * p_per_block;
    const int p1 = min(P, p0 + p_per_block);
    const int nchunks = (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;

    f32x16 accW[1][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[0][ni][r] = 0.0f;
    float tacc = 0.0f;                                  // dW[tid & 127][128 + (tid >> 7)]

    const int xcol0 = wave * 16;                        // this wave's 16 grad_x0 columns (one 16x16 tile)
    float wfrag[SPLIT ? 1 : CO / 4];
    bf16x8 wsp[SPLIT ? CO / 32 : 1][3];                 // SPLIT (16x16x32): lane (col, kq) holds W[32*st + 8*kq .. + 7][col] as planes
    if constexpr (SPLIT) {
#pragma unroll
        for (int st = 0; st < CO / 32; ++st) {
            const float* wp = W + (size_t)(32 * st + 8 * kq) * CIW + xcol0 + l15;
            const Split4 lo = split3(make_float4(wp[0], wp[CIW], wp[2 * CIW], wp[3 * CIW]));
            const Split4 hi = split3(make_float4(wp[4 * CIW], wp[5 * CIW], wp[6 * CIW], wp[7 * CIW]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wsp[st][0][i] = lo.h[i]; wsp[st][0][4 + i] = hi.h[i];
                wsp[st][1][i] = lo.m[i]; wsp[st][1][4 + i] = hi.m[i];
                wsp[st][2][i] = lo.l[i]; wsp[st][2][4 + i] = hi.l[i];
            }
        }
    } else {
#pragma unroll
        for (int st = 0; st < CO / 4; ++st) wfrag[st] = W[(size_t)(kq * KPL + st) * CIW + xcol0 + l15];   // permuted k: k = kq * KPL + st
    }
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(G + (size_t)p0 * CIW, 0, (p1 - p0) * CIW * 4, 0x00020000);
    int goff = (4 * kq * CIW + xcol0 + l15) * 4;

    // SPLIT: a wave stages 4 positions x 64 channels (see bwd_fused_kernel): all 16 positions of dZ and of the features in one pass
    // (lane -> quad / row as in bwd_fused_kernel: conflict-free ds_write_b64 into the K-packed planes)
    const int ca = SPLIT ? (wave & 1) * 64 + 4 * (MP_MAPWIDE ? (lane & 15) : ((lane & 7) + 8 * (lane >> 5))) : (tid % (CO / 4)) * 4;
    const int ka0 = SPLIT ? (wave >> 1) * 4 + (MP_MAPWIDE ? (lane >> 4) : ((lane >> 3) & 3)) : tid / (CO / 4);
    constexpr int KA_STEP = NT / (CO / 4);
    ChanConst ka, kb;
    load_consts<MODE_DZ>(DZ, ca, ka, bn_lds, 128);
    Raw4<MODE_DZ> ra[PA];
    Raw4<MODE_IN> rb[PB];
    int brow[PB], bcol[PB];                              // this thread's (row, column) of the input chunk, per pass
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) {
        const int e = ps * NT + tid;
        brow[ps] = e / (CIW / 4);
        bcol[ps] = (e - brow[ps] * (CIW / 4)) * 4;
    }
    if constexpr (SPLIT) {      // pass 0: the features, in the plane mapping; pass 1: threads 0..15 take the coordinate quad of row tid
        brow[0] = ka0; bcol[0] = ca;
        brow[1] = tid & 15; bcol[1] = CIX;
    }
    auto gload = [&](int pk) {
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) raw_load<MODE_DZ>(DZ, p1, pk + ka0 + ps * KA_STEP, ca, ra[ps]);
#pragma unroll
        for (int ps = 0; ps < PB; ++ps)
            if (SPLIT ? (ps == 0 || tid < DBK) : (ps * NT + tid < NB4)) raw_load<MODE_IN>(IN, p1, pk + brow[ps], bcol[ps], rb[ps]);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            const float4 dz = finish<MODE_DZ>(ra[ps], ka);
            *reinterpret_cast<float4*>(&sA[buf][(ka0 + ps * KA_STEP) * LDA + ca]) = dz;
            if constexpr (SPLIT) {
                const Split4 sp = split3(dz);
                const int o = (ca >> 3) * GS + ka0 * 8 + (ca & 7);
                *reinterpret_cast<bf16x4*>(&hA[buf][0][o]) = sp.h;
                *reinterpret_cast<bf16x4*>(&hA[buf][1][o]) = sp.m;
                *reinterpret_cast<bf16x4*>(&hA[buf][2][o]) = sp.l;
            }
        }
        if constexpr (SPLIT) {
            const Split4 sp = split3(finish<MODE_IN>(rb[0], kb));
            const int o = (ca >> 3) * GS + ka0 * 8 + (ca & 7);
            *reinterpret_cast<bf16x4*>(&hB[buf][0][o]) = sp.h;
            *reinterpret_cast<bf16x4*>(&hB[buf][1][o]) = sp.m;
            *reinterpret_cast<bf16x4*>(&hB[buf][2][o]) = sp.l;
            if (tid < DBK) sT[buf][tid] = finish<MODE_IN>(rb[1], kb);
        } else {
#pragma unroll
            for (int ps = 0; ps < PB; ++ps) {
                const int e = ps * NT + tid;
                if (e < NB4) *reinterpret_cast<float4*>(&sB[buf][e * 4]) = finish<MODE_IN>(rb[ps], kb);
            }
        }
    };

    gload(p0);
    sstore(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nchunks) gload(p0 + (kc + 1) * DBK);
        if constexpr (SPLIT) {   // dW[:, 0:128] += dZ^T * X: six plane products per tile, one dZ plane live at a time
            bf16x8 fb[3][2], fa;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[0][ni] = tr_frag_packed<GS>(hB[cur][0], 0, wcol0 + ni * 32);
            fa = tr_frag_packed<GS>(hA[cur][2], 0, wrow0);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) accW[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[0][ni], accW[0][ni], 0, 0, 0);
            fa = tr_frag_packed<GS>(hA[cur][0], 0, wrow0);
#pragma unroll
            for (int pl = 2; pl >= 1; --pl)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) fb[pl][ni] = tr_frag_packed<GS>(hB[cur][pl], 0, wcol0 + ni * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) accW[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[pl][ni], accW[0][ni], 0, 0, 0);
            fa = tr_frag_packed<GS>(hA[cur][1], 0, wrow0);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) accW[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[pl][ni], accW[0][ni], 0, 0, 0);
        } else {
            mma_chunk_pipelined<true, true, LDA, LDB, 1, 2, DBK>(sA[cur], sB[cur], wrow0, wcol0, accW);   // dW[:, 0:128] += dZ^T * X
        }
        {   // the 4 coordinate columns of dW
            const float* a = sA[cur] + (tid & 127);
            if constexpr (SPLIT) {
                const float* t = reinterpret_cast<const float*>(sT[cur]) + (tid >> 7);
#pragma unroll
                for (int k = 0; k < DBK; ++k) tacc = __builtin_fmaf(a[k * LDA], t[k * 4], tacc);
            } else {
                const float* t = sB[cur] + CIX + (tid >> 7);
#pragma unroll
                for (int k = 0; k < DBK; ++k) tacc = __builtin_fmaf(a[k * LDA], t[k * LDB], tacc);
            }
        }
        {   // grad_x0 chunk [16 x 128] = dZ [16 x 128] * W[:, 0:128]
            f32x4 ax = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                f32x4 cx = {0.f, 0.f, 0.f, 0.f};
                const int ao = kq * GS + l15 * 8;
                bf16x8 af[2][3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao]);
#pragma unroll
                for (int st = 0; st < CO / 32; ++st) {
                    if (st + 1 < CO / 32) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) af[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao + 4 * (st + 1) * GS]);
                    }
                    const bf16x8 ah = af[st & 1][0], am = af[st & 1][1], al = af[st & 1][2];
                    cx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wsp[st][0], cx, 0, 0, 0);
                    ax = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[st][0], ax, 0, 0, 0);
                    cx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[st][2], cx, 0, 0, 0);
                    cx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[st][1], cx, 0, 0, 0);
                    cx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[st][0], cx, 0, 0, 0);
                    cx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[st][1], cx, 0, 0, 0);
                }
                ax += cx;
            }
            const float4* arow = reinterpret_cast<const float4*>(sA[cur] + l15 * LDA + kq * KPL);
            constexpr int AB = 2, NB = SPLIT ? 0 : KPL / (4 * AB);
            float4 abuf[2][AB];
#pragma unroll
            for (int j = 0; j < (SPLIT ? 0 : AB); ++j) abuf[0][j] = arow[j];
#pragma unroll
            for (int bt = 0; bt < NB; ++bt) {
                if (bt + 1 < NB) {
#pragma unroll
                    for (int j = 0; j < AB; ++j) abuf[(bt + 1) & 1][j] = arow[(bt + 1) * AB + j];
                }
#pragma unroll
                for (int j = 0; j < AB; ++j) {
                    const float4 a4 = abuf[bt & 1][j];
                    const int st = (bt * AB + j) * 4;
                    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wfrag[st], ax, 0, 0, 0);
                    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wfrag[st + 1], ax, 0, 0, 0);
                    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wfrag[st + 2], ax, 0, 0, 0);
                    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wfrag[st + 3], ax, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)      // rows past the workgroup's last position: dropped by the buffer's range check
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ax[i]), grsrc, goff, i * CIW * 4, MP_STORE_AUX);
            // the 4 coordinate / padding columns of grad_x0 carry no gradient: written as zeros so that the buffer is fully defined
            if (tid < DBK * 4) __builtin_amdgcn_raw_buffer_store_b32(0u, grsrc, ((kc * DBK + (tid >> 2)) * CIW + CIX + (tid & 3)) * 4, 0, 0);
            goff += DBK * CIW * 4;
        }
        if (kc + 1 < nchunks) sstore(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = wcol0 + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + acc_row_in_tile(r);
            atomicAdd(dW + (size_t)(row * CIW + col), accW[0][ni][r]);
        }
    }
    atomicAdd(dW + (size_t)((tid & 127) * CIW + CIX + (tid >> 7)), tacc);
}

// =================================================================================================================
// small kernels
// =================================================================================================================
// Reduce per-block partial sums [nblk][2][C] in fp64: a block owns FIN_CH channels, FIN_SL threads per channel walk
// the partial rows strided (coalesced across channels), then combine through LDS.  Result valid in threads < FIN_CH.
constexpr int FIN_CH = 16, FIN_SL = 64;
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partials, int nblk, int C, double& s1, double& s2)
{
    __shared__ double red[2][FIN_SL][FIN_CH];
    const int cl = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int i = sl;
        for (; i + 3 * FIN_SL < nblk; i += 4 * FIN_SL) {   // four independent loads in flight per sum
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a1[u] += (double)partials[((size_t)(i + u * FIN_SL) * 2 + 0) * C + c];
                a2[u] += (double)partials[((size_t)(i + u * FIN_SL) * 2 + 1) * C + c];
            }
        }
        for (; i < nblk; i += FIN_SL) {
            a1[0] += (double)partials[((size_t)i * 2 + 0) * C + c];
            a2[0] += (double)partials[((size_t)i * 2 + 1) * C + c];
        }
    }
    red[0][sl][cl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    red[1][sl][cl] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    __syncthreads();
    // tree over the slices: 64 -> 1
    for (int h = FIN_SL / 2; h > 0; h >>= 1) {
        if (sl < h) {
            red[0][sl][cl] += red[0][sl + h][cl];
            red[1][sl][cl] += red[1][sl + h][cl];
        }
        __syncthreads();
    }
    if (threadIdx.x < FIN_CH) {
        s1 = red[0][0][cl];
        s2 = red[1][0][cl];
    }
}

// SyncBN (data-parallel runs with global-batch statistics): the per-channel sums of THIS rank in fp64, written
// to an exchange buffer [2][C] (and a second copy that stays local); the caller all-reduces the buffer and the finalize
// kernels below take the global sums from it instead of reducing the partials themselves.
__global__ void bn_sums_kernel(const float* __restrict__ partials, int nblk, int C, double* __restrict__ out_a,
                               double* __restrict__ out_b)
{
    double s1 = 0.0, s2 = 0.0;
    reduce_partials(partials, nblk, C, s1, s2);
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    out_a[c] = s1;
    out_a[C + c] = s2;
    if (out_b) { out_b[c] = s1; out_b[C + c] = s2; }
}

// BatchNorm forward statistics -> affine (scale, shift); running-stat update; saves mean / rstd.
// gsums != NULL: the (all-reduced) sums [2][C] over the global batch; invP / unbias then refer to the global position count.
__global__ void bn_fwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C, double invP, double unbias,
                                       int training, double momentum, double eps, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, const float* __restrict__ bias,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                       float* __restrict__ scale, float* __restrict__ shift, const double* __restrict__ gsums)
{
    double s1 = 0.0, s2 = 0.0;
    if (training && !gsums) reduce_partials(partials, nblk, C, s1, s2);
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    if (training && gsums) { s1 = gsums[c]; s2 = gsums[C + c]; }
    const float b = bias ? bias[c] : 0.0f;
    if (training) {
        const double mean = s1 * invP;
        double var = s2 * invP - mean * mean;  // biased; fp64 keeps the cancellation harmless
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + eps);
        const float sc = (float)((double)gamma[c] * rstd);
        scale[c] = sc;
        shift[c] = (float)((double)beta[c] - mean * (double)sc);
        mean_out[c] = (float)mean;   // of the bias-free z; the conv bias cancels inside train-mode BN
        rstd_out[c] = (float)rstd;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * (mean + (double)b));
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * var * unbias);
        }
    } else {
        const double rs = 1.0 / sqrt((double)running_var[c] + eps);
        const float sc = (float)((double)gamma[c] * rs);
        scale[c] = sc;
        shift[c] = (float)((double)beta[c] + ((double)b - (double)running_mean[c]) * (double)sc);
        mean_out[c] = running_mean[c] - b;  // so that zhat = (z - mean_out) * rstd_out in both modes
        rstd_out[c] = (float)rs;
    }
}

// BatchNorm backward sums (sum dy, sum dy*z) -> dgamma, dbeta, dbias and the dZ constants (a, e, f).
// SyncBN (gsums / lsums != NULL, both [2][C]): dgamma / dbeta are this rank's contribution (local sums; the gradient exchange
// averages them like every other parameter gradient), the dZ constants come from the global sums and the global count.
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C, double invP, int training,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ dbias, float* __restrict__ ca,
                                       float* __restrict__ ce, float* __restrict__ cf, const double* __restrict__ gsums,
                                       const double* __restrict__ lsums)
{
    double s1 = 0.0, s2 = 0.0;
    if (!gsums) reduce_partials(partials, nblk, C, s1, s2);
    const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH);
    if (c >= C || threadIdx.x >= FIN_CH) return;
    const double mu = mean[c], rs = rstd[c], g = gamma[c];
    if (gsums) { s1 = lsums[c]; s2 = lsums[C + c]; }
    double dbe = s1;
    double dga = rs * (s2 - mu * s1);
    dbeta[c] = (float)dbe;
    dgamma[c] = (float)dga;
    if (gsums) { dbe = gsums[c]; dga = rs * (gsums[C + c] - mu * gsums[c]); }
    const double a = g * rs;
    if (training) {
        const double c1 = dbe * invP, c2 = dga * invP;
        ca[c] = (float)a;
        ce[c] = (float)(-a * c2 * rs);
        cf[c] = (float)(-a * c1 + a * c2 * rs * mu);
        if (dbias) dbias[c] = 0.0f;  // d/dbias of train-mode BN output is identically zero
    } else {
        ca[c] = (float)a;
        ce[c] = 0.0f;
        cf[c] = 0.0f;
        if (dbias) dbias[c] = (float)(a * dbe);
    }
}

__global__ __launch_bounds__(256) void zero_cols_kernel(float* __restrict__ x, int64_t P, int C, int c0)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int w = C - c0;
    if (e >= P * w) return;
    x[(e / w) * C + c0 + (int)(e % w)] = 0.0f;
}

// out[g,c] = max_k relu(z[g*K+k, c]*s+t); first maximum wins; keeps arg-max and the raw z there.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ z, const float* s, const float* t, int64_t G, int K, int C,
                                                       float* __restrict__ out, int* __restrict__ argk,
                                                       float* __restrict__ zmax)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= G * C) return;
    const int64_t g = e / C;
    const int c = (int)(e - g * C);
    const float sc = s[c], sh = t[c];
    const float* zp = z + (g * K) * C + c;
    float best = -1.0f, bz = 0.0f;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        const float zz = zp[(int64_t)k * C];
        float y = zz * sc + sh;
        y = y > 0.0f ? y : 0.0f;
        if (y > best) { best = y; bk = k; bz = zz; }
    }
    out[e] = best;
    argk[e] = bk;
    zmax[e] = bz;
}

// gp = g * [out > 0]; per-block partial sums (sum gp, sum gp*zmax) per channel.
// grid (ceil(G / rows), ceil(C / 256)): a block handles `rows` <= POOL_ROWS groups x 256 channels, lanes along channels; the host
// picks fewer rows per block when G is small (the group_all level has 32 groups: 8 workgroups of 16 rows took 19 us, most of it
// the dW clear below on 2048 threads).
constexpr int POOL_ROWS = 16;
__global__ __launch_bounds__(256) void pool_bwd_prep_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                            const float* __restrict__ zmax, int64_t G, int C, int rows,
                                                            float* __restrict__ gp, BnOut partials,
                                                            float* __restrict__ clear, size_t clear_n)
{
    // clear: the level's dW block (accumulated with atomics by the kernels that follow) is zeroed here, in the first launch of the
    // level's backward, instead of by a launch of its own (16-byte stores when the block is aligned)
    bn_zero(partials);
    {
        const size_t nthreads = (size_t)gridDim.x * gridDim.y * 256;
        const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        const size_t n4 = (reinterpret_cast<uintptr_t>(clear) & 15) ? 0 : clear_n / 4;
        for (size_t i = me; i < n4; i += nthreads) reinterpret_cast<float4*>(clear)[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
        for (size_t i = 4 * n4 + me; i < clear_n; i += nthreads) clear[i] = 0.0f;
    }
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t g0 = (int64_t)blockIdx.x * rows;
    const int n = (int)(min(G, g0 + rows) - g0);
    // all of the block's rows in flight at once (3 x 16 independent loads per thread), then the sums in row order
    float o[POOL_ROWS], gv[POOL_ROWS], zv[POOL_ROWS];
#pragma unroll
    for (int r = 0; r < POOL_ROWS; ++r) {
        const bool in = r < n;
        const size_t e = (size_t)(g0 + (in ? r : 0)) * C + c;
        o[r] = in ? out[e] : 0.0f;
        gv[r] = in ? gout[e] : 0.0f;
        zv[r] = in ? zmax[e] : 0.0f;
    }
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int r = 0; r < POOL_ROWS; ++r) {
        if (r < n) {
            const float v = o[r] > 0.0f ? gv[r] : 0.0f;
            gp[(size_t)(g0 + r) * C + c] = v;
            s1 += v;
            s2 += v * zv[r];
        }
    }
    bn_emit(partials, C, blockIdx.x, c, (double)s1, (double)s2);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int log2_or_neg(int64_t k)
{
    if (k <= 0 || (k & (k - 1))) return -1;
    int s = 0;
    while ((int64_t(1) << s) < k) ++s;
    return s;
}

// MP_CHUNK_FWD=0 keeps the tiled GEMM kernel for every forward layer (A/B timing)
// The fp32 contractions run as six bf16 MFMAs on (h, m, l) operand planes (split3) -- fp32-accurate results at 2.7x the
// matrix rate of v_mfma_f32_*_f32, and beside the VALU instead of on it.  MP_SA_SPLIT=0 selects the fp32-MFMA kernels (A/B timing,
// bit-for-bit comparisons with the k-ordered FMA chain).
inline bool split_enabled()
{
    static const bool on = !(getenv("MP_SA_SPLIT") && atoi(getenv("MP_SA_SPLIT")) == 0);
    return on;
}

// workgroups the position-stream forward aims for: positions per workgroup halve from 1024 until there are that many
// [r2] same-box sweep: 512 for most shapes; the 256-output kernel (512 threads, one workgroup per CU) is best with one round of 256, the
// HBM-bound 64 -> 64 layer with 2048 small workgroups
// [r5] fwd_chunk_kernel: chunk-interleaved workgroups (negative p_per_block) while byte offsets into Z fit 31 bits.  Same box, four alternations:
// <64,128,pool> 95 -> 87 us, <128,256,pool> 105 -> 105, <128,128> 65 -> 66, <64,64> 54 -> 55: on for the first shape only (MP_FWD_IL=2: every
// shape, 0: none)
static int fwd_il(int ppb, int64_t P, int Co, bool gains)
{
    static const int mode = [] { const char* e = getenv("MP_FWD_IL"); return e ? atoi(e) : 1; }();
    return ((mode == 2 || (mode == 1 && gains)) && (uint64_t)P * (uint64_t)Co * 4u < (1ull << 31)) ? -ppb : ppb;
}
inline int fwd_wgs_wanted(int dflt = 512) { return dflt; }

// [r5] operand planes of the fp32 BACKWARD contractions on the position-stream kernels: 2 (default: h, m -- three plane products, split2) or 3
// (MP_BWD_PLANES=3: h, m, l -- six products, as the forward pass).  Read on every call.
inline int bwd_planes()
{
    const char* e = getenv("MP_BWD_PLANES");
    return (e && atoi(e) == 3) ? 3 : 2;
}

inline bool chunk_fwd_enabled()
{
    static const bool on = !(getenv("MP_CHUNK_FWD") && atoi(getenv("MP_CHUNK_FWD")) == 0);
    return on;
}

// MP_FUSED_BWD=0 keeps the separate dX / dW kernels for the single-tile layers (A/B timing)
inline bool fused_bwd_enabled()
{
    static const bool on = !(getenv("MP_FUSED_BWD") && atoi(getenv("MP_FUSED_BWD")) == 0);
    return on;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------

static size_t ws_core_bytes(int64_t P, int64_t K, int n_layers, const int64_t* channels, int backward);
extern "C" size_t mp_sa_mlp_workspace_bytes(int64_t P, int64_t K, int n_layers, const int64_t* channels, int backward)
{
    if (P <= 0 || n_layers <= 0 || !channels) return 0;
    return ws_core_bytes(P, K, n_layers, channels, backward);
}

static size_t ws_core_bytes(int64_t P, int64_t K, int n_layers, const int64_t* channels, int backward)
{
    int64_t cmax = 0;
    for (int l = 0; l <= n_layers; ++l) cmax = channels[l] > cmax ? channels[l] : cmax;
    const size_t nblk = (size_t)((P + 63) / 64);                           // row tiles are 128 or (small grids) 64 high
    size_t bytes = align_up(nblk * 2 * (size_t)cmax * sizeof(float), 256);  // epilogue partials
    bytes += 3 * align_up((size_t)cmax * sizeof(float), 256);              // dZ constants a, e, f
    if (!backward) bytes += 4 * align_up((size_t)(P / (K > 0 ? K : 1)) * (size_t)channels[n_layers] * sizeof(float), 256);  // fused pool
    if (backward) {
        bytes += 2 * align_up((size_t)P * (size_t)cmax * sizeof(float), 256);            // G ping-pong
        bytes += align_up((size_t)(P / (K > 0 ? K : 1)) * (size_t)channels[n_layers] * sizeof(float), 256);  // gp
    }
    return bytes;
}

// Can the first layer of this chain be recomputed instead of stored (layers[0].z = NULL in forward AND backward)?  Yes for a
// 4-channel input into 64 channels followed by a 64 -> 64 / 128 layer that is not the pooled one, with the position-stream
// kernels enabled; the caller must not need grad_x0.  MP_RECOMPUTE_FIRST=0 switches it off (read on every call).
extern "C" int mp_sa_mlp_recompute_first(int n_layers, const int64_t* channels, int64_t K)
{
    if (!channels || n_layers < 3) return 0;
    const char* e = getenv("MP_RECOMPUTE_FIRST");
    if (e && atoi(e) == 0) return 0;
    if (!chunk_fwd_enabled() || !fused_bwd_enabled()) return 0;
    return channels[0] == 4 && channels[1] == 64 && (channels[2] == 64 || channels[2] == 128) && K > 0;
}

// bf16 variant (mp_sa_mlp_*_bf16 / _gather_bf16): does this chain keep its activations IN MEMORY as bf16?  Yes when every layer runs on
// the position-stream kernels -- first layer recomputed (first_layer = 1) or factorised (2), every later layer 64 / 128 inputs into
// 64 / 128 / 256 outputs with a fused pool -- which then read and write Z_l / G_l as bf16 (half the bytes of kernels that are HBM-bound on
// them).  The caller allocates layers[l].z as bf16 [P, c_out] exactly when this returns 1.  Any other bf16 chain runs on the tiled
// kernels with fp32 storage (a recomputed first layer is then not available: pass layers[0].z).
static bool chain_store16(int n_layers, const int64_t* ch, int64_t K, int first)
{
    if ((first != 1 && first != 2) || n_layers < 2) return false;
    if (!chunk_fwd_enabled() || !fused_bwd_enabled() || !(K == 32 || K == 64 || K == 128)) return false;
    if (first == 1 && !mp_sa_mlp_recompute_first(n_layers, ch, K)) return false;
    if (first == 2 && !(ch[0] == 4 && (ch[1] == 64 || ch[1] == 128 || ch[1] == 256))) return false;
    for (int l = 1; l < n_layers; ++l) {
        const int64_t Ci = ch[l], Co = ch[l + 1];
        if (!(Ci == 64 || Ci == 128) || !(Co == 64 || Co == 128 || Co == 256)) return false;
        if ((Co == 64 && Ci == 128) || (Co == 256 && Ci != 128)) return false;
    }
    return true;
}
// dW [Co, Ci] = dz^T x over P rows, both operands as stored (the per-source-point half of a factorised first layer: dW_f = dA^T F over the
// B*N source points, sa_mlp._PerPointFirst) -- the position-sliced weight-gradient GEMM of the levels' interior layers with identity
// operands: slices of the rows add their tiles with atomics (summation order not fixed); dW is cleared first unless it lies in the armed
// zero arena.
extern "C" int mp_dw_gemm_f32(const float* dz, const float* x, int64_t P, int64_t Co, int64_t Ci, float* dW, mp_stream_t stream_)
{
    if (P < 0 || Co <= 0 || Ci <= 0) return MP_EINVAL;
    if (!dW || (P > 0 && (!dz || !x))) return MP_EINVAL;
    if (P >= ((int64_t)1 << 31) || (Co & 3) || (Ci & 3) || Co > 1024 || Ci > 1024) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    if (!mp::zero_async(dW, (size_t)(Co * Ci), stream)) return MP_ELAUNCH;
    if (P == 0) return MP_OK;
    PosOperand DZ{}, IN{};
    DZ.x = dz; DZ.C = (int)Co; DZ.K = 1; DZ.kshift = 0;
    IN.x = x; IN.C = (int)Ci; IN.K = 1; IN.kshift = 0;
    return mp_dw_gemm_launch(SRC_ID, SRC_ID, !split_enabled() ? 0 : (bwd_planes() == 2 ? 2 : 3), &DZ, &IN, P, dW, stream);     // (a gradient: two planes, [r6])
}

extern "C" int mp_sa_mlp_bf16_storage(int n_layers, const int64_t* channels, int64_t K, int first_layer)
{
    return (channels && K > 0 && chain_store16(n_layers, channels, K, first_layer)) ? 1 : 0;
}

// The gathered-input form (mp_sa_mlp_{fwd,bwd}_gather_f32) covers what BASELINE's second set-abstraction level is: a first layer of
// [128 features | xyz | pad] -> 128 behind at least one more layer, fp32 results, fused kernels enabled.
// =================================================================================================================
// Factorised first layer of a level with input features (models/pointnet2_utils.py:138 + :208-213).  The layer is linear in
// [f_i ; x_i - c], so  Z_0[p] = A[b, idx[p]] + W_x (x[b, idx[p]] - c[group(p)])  with  A = F W_f^T  computed ONCE per source
// point by the caller (a [B*N, CF] x [CF, Co] GEMM, 16x fewer rows than the grouped tensor at K = 64 / N = 512 / S = 128): a
// gather-add instead of a GEMM over the grouped rows, which need not exist.  The gather descriptor then carries A as its
// `feats` ([B, N, Co], CF == Co) and layers[0] is the coordinate part alone: weight [Co, 4] (W_x | 0), c_in == 4.
//   forward : this kernel writes the raw Z_0 and the BatchNorm partial sums (one row per workgroup), everything after it is the
//             ordinary chain (layer 1 reads act(Z_0) like any other layer);
//   backward: one pass over (Z_0, G_0) writes dZ_0 [P, Co + 4] -- it IS the gradient of the gathered A rows; the caller reduces it
//             over the gathering rows (mp_group_bwd_f32) and finishes dW_f, dF with two small GEMMs -- and accumulates dW_x from
//             the gathered coordinates (first_factored_bwd_kernel).
// Same mathematics as the grouped GEMM, a different fp32 summation order (W_f f is rounded before W_x d is added).
// =================================================================================================================
#ifndef MP_FACT_U
#define MP_FACT_U 4        // rows in flight per lane group (timing builds: tools/fact_variants.sh)
#endif
#ifndef MP_FACT_PPB
#define MP_FACT_PPB 256    // positions per workgroup of the forward kernel = per BatchNorm partial row (the backward kernel takes twice as many)
#endif
// NV = 2 ([r4], bf16 storage of Z_0): eight channels per lane -- one 16-byte store of the rounded row piece instead of an 8-byte one
// (the bf16 variant's launches wrote half the bytes with the same number of store instructions: 1.8 TB/s)
template <int Q, int NV>   // Q = Co / 4; Q / NV lanes per row
__global__ __launch_bounds__(256) void first_factored_fwd_kernel(const float* __restrict__ A, const float* __restrict__ xyz,
                                                                 const float* __restrict__ new_xyz, const int64_t* __restrict__ idx,
                                                                 const float* __restrict__ Wx, int P, int K, int kshift, int N, int per,
                                                                 int gshift, int ppb, float* __restrict__ Z0, BnOut partials, int r16, int h16)
{   // r16: the bf16 variant -- W_x and the centred coordinates rounded to bf16 (A comes from rounded operands already); h16: Z_0 stored as bf16
    bn_zero(partials);
    constexpr int CO = 4 * Q, CPL = 4 * NV, QL = CO / CPL, RW = 64 / QL, RB = 4 * RW, U = MP_FACT_U;      // rows per wave / per workgroup pass, passes in flight
    __shared__ float red[2][RB][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane % QL, slot = wave * RW + lane / QL;
    float4 w[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) w[j] = rb16(ld4(Wx + (size_t)(CPL * ql + j) * 4), r16);
    const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
    float4 s1[NV], s2[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s1[v] = s2[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = p0 + slot; p < p1; p += RB * U) {
        float4 a[U][NV];
        float dx[U], dy[U], dz[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = min(p + u * RB, p1 - 1);
            const unsigned b = gshift >= 0 ? (unsigned)pp >> gshift : (unsigned)pp / (unsigned)per;
            const unsigned grp = kshift >= 0 ? (unsigned)pp >> kshift : (unsigned)pp / (unsigned)K;
            const size_t src = (size_t)b * (unsigned)N + (size_t)idx[pp];
#pragma unroll
            for (int v = 0; v < NV; ++v) a[u][v] = ld4(A + src * CO + CPL * ql + 4 * v);
            const float* x = xyz + src * 3;
            const float* c = new_xyz + (size_t)grp * 3;
            dx[u] = rb16(x[0] - c[0], r16); dy[u] = rb16(x[1] - c[1], r16); dz[u] = rb16(x[2] - c[2], r16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + u * RB;
            if (pp < p1) {
                float4 z[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    z[v].x = a[u][v].x + __builtin_fmaf(w[4 * v + 0].z, dz[u], __builtin_fmaf(w[4 * v + 0].y, dy[u], w[4 * v + 0].x * dx[u]));
                    z[v].y = a[u][v].y + __builtin_fmaf(w[4 * v + 1].z, dz[u], __builtin_fmaf(w[4 * v + 1].y, dy[u], w[4 * v + 1].x * dx[u]));
                    z[v].z = a[u][v].z + __builtin_fmaf(w[4 * v + 2].z, dz[u], __builtin_fmaf(w[4 * v + 2].y, dy[u], w[4 * v + 2].x * dx[u]));
                    z[v].w = a[u][v].w + __builtin_fmaf(w[4 * v + 3].z, dz[u], __builtin_fmaf(w[4 * v + 3].y, dy[u], w[4 * v + 3].x * dx[u]));
                    s1[v].x += z[v].x; s1[v].y += z[v].y; s1[v].z += z[v].z; s1[v].w += z[v].w;
                    s2[v].x += z[v].x * z[v].x; s2[v].y += z[v].y * z[v].y; s2[v].z += z[v].z * z[v].z; s2[v].w += z[v].w * z[v].w;
                }
                if constexpr (NV == 2) {     // (bf16 storage only)
                    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(Z0) + (size_t)pp * CO + CPL * ql) =
                        make_uint4(pack_bf16(z[0].x, z[0].y), pack_bf16(z[0].z, z[0].w), pack_bf16(z[NV - 1].x, z[NV - 1].y), pack_bf16(z[NV - 1].z, z[NV - 1].w));
                } else {
                    if (h16) st4h(Z0, (size_t)pp * CO + 4 * ql, z[0]);
                    else *reinterpret_cast<float4*>(Z0 + (size_t)pp * CO + 4 * ql) = z[0];
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        *reinterpret_cast<float4*>(&red[0][slot][CPL * ql + 4 * v]) = s1[v];
        *reinterpret_cast<float4*>(&red[1][slot][CPL * ql + 4 * v]) = s2[v];
    }
    __syncthreads();
    for (int e = tid; e < 2 * CO; e += 256) {
        const int st = e / CO, c = e - st * CO;
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < RB; ++r) v += red[st][r][c];
        if (partials.slots) atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + st) * CO + c, (double)v);
        else partials.rows[((size_t)blockIdx.x * 2 + st) * CO + c] = v;
    }
}

// Backward of the factorised first layer in ONE pass over (Z_0, G_0): dZ_0 rows written out [P, Co + 4] (the gradient of the gathered
// A rows) and dW_x[c, j] += sum_p dZ_0[p, c] * (x[idx[p]] - c[group(p)])_j, reduced over the workgroup's rows in LDS and added to
// dW [Co, 4] with one atomic per element and workgroup.
template <int Q>
__global__ __launch_bounds__(256) void first_factored_bwd_kernel(PosOperand DZ, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                                 const int64_t* __restrict__ idx, int P, int K, int kshift, int N, int per,
                                                                 int gshift, int ppb, float* __restrict__ dz_out, float* __restrict__ dW, int r16, int h16)
{   // r16: the bf16 variant -- dZ_0 (as written out, too: its reduction over the gathering rows then sums rounded values) and the
    // centred coordinates rounded to bf16
    constexpr int CO = 4 * Q, RW = 64 / Q, RB = 4 * RW, U = MP_FACT_U;
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * CO];
    bn_prologue(DZ.bn, bn_lds, CO, 0, CO, blockIdx.x == 0);
    __shared__ float red[3][RB][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane % Q, slot = wave * RW + lane / Q;
    ChanConst k;
    load_consts<SRC_DZ>(DZ, 4 * ql, k, bn_lds, CO);
    const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ay = ax, az = ax;
    for (int p = p0 + slot; p < p1; p += RB * U) {
        float4 z[U], g[U];
        float dx[U], dy[U], dzc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = min(p + u * RB, p1 - 1);
            const unsigned b = gshift >= 0 ? (unsigned)pp >> gshift : (unsigned)pp / (unsigned)per;
            const unsigned grp = kshift >= 0 ? (unsigned)pp >> kshift : (unsigned)pp / (unsigned)K;
            const size_t src = (size_t)b * (unsigned)N + (size_t)idx[pp];
            z[u] = ldz4(DZ.x, (size_t)pp * CO + 4 * ql, h16);
            g[u] = ldz4(DZ.g, (size_t)pp * CO + 4 * ql, h16);
            const float* x = xyz + src * 3;
            const float* c = new_xyz + (size_t)grp * 3;
            dx[u] = rb16(x[0] - c[0], r16); dy[u] = rb16(x[1] - c[1], r16); dzc[u] = rb16(x[2] - c[2], r16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pp = p + u * RB;
            if (pp < p1) {
                float4 d;
                d.x = xf1<SRC_DZ>(z[u].x, g[u].x, k.s.x, k.t.x, k.a.x, k.e.x, k.f.x);
                d.y = xf1<SRC_DZ>(z[u].y, g[u].y, k.s.y, k.t.y, k.a.y, k.e.y, k.f.y);
                d.z = xf1<SRC_DZ>(z[u].z, g[u].z, k.s.z, k.t.z, k.a.z, k.e.z, k.f.z);
                d.w = xf1<SRC_DZ>(z[u].w, g[u].w, k.s.w, k.t.w, k.a.w, k.e.w, k.f.w);
                d = rb16(d, r16);
                *reinterpret_cast<float4*>(dz_out + (size_t)pp * (CO + 4) + 4 * ql) = d;
                ax.x = __builtin_fmaf(d.x, dx[u], ax.x); ax.y = __builtin_fmaf(d.y, dx[u], ax.y); ax.z = __builtin_fmaf(d.z, dx[u], ax.z); ax.w = __builtin_fmaf(d.w, dx[u], ax.w);
                ay.x = __builtin_fmaf(d.x, dy[u], ay.x); ay.y = __builtin_fmaf(d.y, dy[u], ay.y); ay.z = __builtin_fmaf(d.z, dy[u], ay.z); ay.w = __builtin_fmaf(d.w, dy[u], ay.w);
                az.x = __builtin_fmaf(d.x, dzc[u], az.x); az.y = __builtin_fmaf(d.y, dzc[u], az.y); az.z = __builtin_fmaf(d.z, dzc[u], az.z); az.w = __builtin_fmaf(d.w, dzc[u], az.w);
            }
        }
    }
    *reinterpret_cast<float4*>(&red[0][slot][4 * ql]) = ax;
    *reinterpret_cast<float4*>(&red[1][slot][4 * ql]) = ay;
    *reinterpret_cast<float4*>(&red[2][slot][4 * ql]) = az;
    __syncthreads();
    for (int e = tid; e < 3 * CO; e += 256) {
        const int j = e / CO, c = e - j * CO;
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < RB; ++r) v += red[j][r][c];
        atomicAdd(dW + (size_t)c * 4 + j, v);
    }
}

// ---- the same backward without the dZ_0 round trip ---------------------------------------------------------------------------
// csr_rows_kernel sorts the rows of every cloud by the source point they gathered (counting sort in LDS, one workgroup per cloud;
// the order inside a point follows the atomics, so the summation order below is not fixed: the non-deterministic mode only).
// first_factored_reduce_kernel then walks the sorted rows in chunks of `chunk` rows per wave -- perfect balance whatever the hot
// points are --, forms dZ_0 from (Z_0, G_0) on the fly, keeps a running sum while the point stays the same and adds it to dA with one
// atomic per channel when the point changes; dW_x as in first_factored_bwd_kernel.  dA must be zero on entry.
__global__ __launch_bounds__(1024) void csr_rows_kernel(const int64_t* __restrict__ idx, int N, int M, int* __restrict__ order, int* __restrict__ pts)
{
    extern __shared__ int csr_lds[];          // [N] histogram -> cursors, [1024] scan scratch
    int* hist = csr_lds;
    int* part = csr_lds + N;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int64_t* bi = idx + (size_t)b * M;
    for (int i = tid; i < N; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int m = tid; m < M; m += 1024) {
        const int64_t i64 = bi[m];
        atomicAdd(&hist[(int)(i64 < 0 ? 0 : (i64 >= N ? N - 1 : i64))], 1);
    }
    __syncthreads();
    // exclusive scan: every thread owns a run of consecutive points, the run totals are scanned over the workgroup
    const int run = (N + 1023) / 1024, i0 = tid * run, i1 = min(N, i0 + run);
    int tot = 0;
    for (int i = i0; i < i1; ++i) tot += hist[i];
    part[tid] = tot;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int base = part[tid] - tot;
    for (int i = i0; i < i1; ++i) { const int c = hist[i]; hist[i] = base; base += c; }
    __syncthreads();
    for (int m = tid; m < M; m += 1024) {
        const int64_t i64 = bi[m];
        const int i = (int)(i64 < 0 ? 0 : (i64 >= N ? N - 1 : i64));
        const int pos = atomicAdd(&hist[i], 1);
        order[(size_t)b * M + pos] = m;
        pts[(size_t)b * M + pos] = i;
    }
}

static int launch_csr_rows(const int64_t* idx, int64_t Bc, int Np, int M, int* order, int* pts, hipStream_t stream)
{
    static mp::DynLds lds;
    const size_t smem = sizeof(int) * ((size_t)Np + 1024);
    if (!lds.ensure(reinterpret_cast<const void*>(csr_rows_kernel), smem)) return MP_ELAUNCH;
    hipLaunchKernelGGL(csr_rows_kernel, dim3((unsigned)Bc), dim3(1024), smem, stream, idx, Np, M, order, pts);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

// One gathered row's share of a lane: NV = 1: four channels (fp32 storage: 16 bytes; bf16 storage: 8 bytes), NV = 2: eight channels of a
// bf16-stored row (16 bytes, kept packed until they are used).
template <int NV> struct FactRow;
template <> struct FactRow<1> {
    float4 v;
    __device__ __forceinline__ void load(const float* base, size_t elem, int h16) { v = ldz4(base, elem, h16); }
    __device__ __forceinline__ float4 get(int) const { return v; }
};
template <> struct FactRow<2> {
    uint4 r;
    __device__ __forceinline__ void load(const float* base, size_t elem, int) { r = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + elem); }
    __device__ __forceinline__ float4 get(int i) const
    {
        const unsigned a = i ? r.z : r.x, c = i ? r.w : r.y;
        return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u), __uint_as_float(c << 16), __uint_as_float(c & 0xffff0000u));
    }
};

// NV = 2 ([r4], bf16 activation storage only): a lane takes EIGHT channels of a row, i.e. one 16-byte load per tensor instead of an
// 8-byte one -- with four channels per lane the bf16 variant moved half the bytes with the same number of load instructions and rows
// in flight, and ran at 1 TB/s (134 us for the 8 192 x 32-row level that takes 80 us in fp32)
template <int Q, int NV>
__global__ __launch_bounds__(256) void first_factored_reduce_kernel(PosOperand DZ, const int* __restrict__ order, const int* __restrict__ pts,
                                                                    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int M,
                                                                    int K, int kshift, int chunk, float* __restrict__ dA, float* __restrict__ dW, int r16, int h16)
{   // r16: see first_factored_bwd_kernel
#ifndef MP_FACT_RU
#define MP_FACT_RU 8          // rows in flight per slot ([r4] 4 -> 8: 87 -> 79 us)
#endif
    constexpr int CO = 4 * Q, CPL = 4 * NV, QL = CO / CPL, RW = 64 / QL, U = MP_FACT_RU;
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * CO];
    bn_prologue(DZ.bn, bn_lds, CO, 0, CO, blockIdx.x == 0 && blockIdx.y == 0);
    __shared__ float red[3][4 * RW][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane % QL, sub = lane / QL;
    const int b = blockIdx.y;
    // [r4] every row-slot of a wave (`sub`: the QL lanes that hold one row) walks its OWN contiguous piece of the wave's chunk: a point's run of
    // sorted rows then stays in one slot (interleaved, a run of R rows was flushed min(R, RW) times), and a run that neither opens nor closes
    // the piece belongs to this slot alone -- it is written with 16-byte stores instead of atomics (dA is zero on entry and a point's rows
    // are contiguous in the sorted order, so nobody else adds to that row)
    const int len = chunk / RW;
    const int j0 = (blockIdx.x * 4 + wave) * chunk + sub * len, j1 = min(M, j0 + len);
    const int S = M / K;
    ChanConst k[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) load_consts<SRC_DZ>(DZ, CPL * ql + 4 * v, k[v], bn_lds, CO);
    const int* bo = order + (size_t)b * M;
    const int* bp = pts + (size_t)b * M;
    float* dst = dA + (size_t)b * N * CO + CPL * ql;
    int cur = -1;
    bool opening = true;       // the run in `acc` is the first of this piece (it may have begun in the piece before)
    float4 acc[NV], ax[NV], ay[NV], az[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = ax[v] = ay[v] = az[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&](bool shared) {
        if (cur >= 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float* d = dst + (size_t)cur * CO + 4 * v;
                if (shared) { atomicAdd(d + 0, acc[v].x); atomicAdd(d + 1, acc[v].y); atomicAdd(d + 2, acc[v].z); atomicAdd(d + 3, acc[v].w); }
                else *reinterpret_cast<float4*>(d) = acc[v];
            }
        }
    };
    // the sorted (row, point) pairs are fetched one batch AHEAD of the rows they name: the row loads depend on them
    int nm[U], npt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int jj = min(min(j0 + u, max(j1 - 1, j0)), M - 1);
        nm[u] = bo[jj];
        npt[u] = bp[jj];
    }
    for (int j = j0; j < j1; j += U) {
        FactRow<NV> z[U], g[U];
        float dx[U], dy[U], dzc[U];
        int pt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = nm[u];
            pt[u] = npt[u];
            const int jn = min(j + U + u, j1 - 1);
            nm[u] = bo[jn];
            npt[u] = bp[jn];
            const size_t row = (size_t)b * M + m;
            z[u].load(DZ.x, row * CO + CPL * ql, h16);
            g[u].load(DZ.g, row * CO + CPL * ql, h16);
            const unsigned grp = (unsigned)b * (unsigned)S + (kshift >= 0 ? (unsigned)m >> kshift : (unsigned)m / (unsigned)K);
            const float* x = xyz + ((size_t)b * N + pt[u]) * 3;
            const float* c = new_xyz + (size_t)grp * 3;
            dx[u] = rb16(x[0] - c[0], r16); dy[u] = rb16(x[1] - c[1], r16); dzc[u] = rb16(x[2] - c[2], r16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u < j1) {
                if (pt[u] != cur) { flush(opening); opening = cur < 0; cur = pt[u];
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 zz = z[u].get(v), gg = g[u].get(v);
                    float4 d;
                    d.x = xf1<SRC_DZ>(zz.x, gg.x, k[v].s.x, k[v].t.x, k[v].a.x, k[v].e.x, k[v].f.x);
                    d.y = xf1<SRC_DZ>(zz.y, gg.y, k[v].s.y, k[v].t.y, k[v].a.y, k[v].e.y, k[v].f.y);
                    d.z = xf1<SRC_DZ>(zz.z, gg.z, k[v].s.z, k[v].t.z, k[v].a.z, k[v].e.z, k[v].f.z);
                    d.w = xf1<SRC_DZ>(zz.w, gg.w, k[v].s.w, k[v].t.w, k[v].a.w, k[v].e.w, k[v].f.w);
                    d = rb16(d, r16);
                    acc[v].x += d.x; acc[v].y += d.y; acc[v].z += d.z; acc[v].w += d.w;
                    ax[v].x = __builtin_fmaf(d.x, dx[u], ax[v].x); ax[v].y = __builtin_fmaf(d.y, dx[u], ax[v].y); ax[v].z = __builtin_fmaf(d.z, dx[u], ax[v].z); ax[v].w = __builtin_fmaf(d.w, dx[u], ax[v].w);
                    ay[v].x = __builtin_fmaf(d.x, dy[u], ay[v].x); ay[v].y = __builtin_fmaf(d.y, dy[u], ay[v].y); ay[v].z = __builtin_fmaf(d.z, dy[u], ay[v].z); ay[v].w = __builtin_fmaf(d.w, dy[u], ay[v].w);
                    az[v].x = __builtin_fmaf(d.x, dzc[u], az[v].x); az[v].y = __builtin_fmaf(d.y, dzc[u], az[v].y); az[v].z = __builtin_fmaf(d.z, dzc[u], az[v].z); az[v].w = __builtin_fmaf(d.w, dzc[u], az[v].w);
                }
            }
        }
    }
    flush(true);               // the closing run may go on in the next piece
    const int slot = wave * RW + sub;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        *reinterpret_cast<float4*>(&red[0][slot][CPL * ql + 4 * v]) = ax[v];
        *reinterpret_cast<float4*>(&red[1][slot][CPL * ql + 4 * v]) = ay[v];
        *reinterpret_cast<float4*>(&red[2][slot][CPL * ql + 4 * v]) = az[v];
    }
    __syncthreads();
    for (int e = tid; e < 3 * CO; e += 256) {
        const int jc = e / CO, c = e - jc * CO;
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < 4 * RW; ++r) v += red[jc][r][c];
        atomicAdd(dW + (size_t)c * 4 + jc, v);
    }
}

// the factorised first layer's shapes: gather descriptor carrying A [B, N, Co], layers[0] = (W_x | 0) [Co, 4]
static bool factored_ok(const mp_gather_t* g, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers, bool bf16)
{
    if (!g || !g->feats || !g->xyz || !g->new_xyz || !g->idx || g->N <= 0 || g->S <= 0) return false;
    (void)bf16;     // (the bf16 variant rounds W_x, the centred coordinates and dZ_0 inside the factorised kernels; A comes rounded)
    if (n_layers < 2 || layers[0].c_in != 4 || g->CF != layers[0].c_out) return false;
    if (g->CF != 64 && g->CF != 128 && g->CF != 256) return false;
    if (P % (g->S * K) != 0 || !layers[0].z) return false;
    return true;
}

// The tiled GEMMs live in sa_gemm.hip ([r6]); PREC by the call's arithmetic: bf16 operands 1, split planes 3 (forward) / 2 (gradients, bwd_planes()), fp32 MFMA 0.
static int pos_gemm(int mode, bool krow, int epi, int prec, const PosOperand& A, int64_t P, const float* W, int N, int Kd, float* C, BnOut partials,
                    const float* zprev, const float* sprev, const float* tprev, hipStream_t stream, int* nblk_out, PoolOut po = PoolOut{}, int ldw = 0,
                    int ldc = 0)
{
    return mp_pos_gemm_launch(mode, krow ? 1 : 0, epi, prec, &A, P, W, N, Kd, C, &partials, zprev, sprev, tprev, stream, nblk_out, &po, ldw, ldc);
}
static int dw_gemm(int mode_dz, int mode_in, int prec, const PosOperand& DZ, const PosOperand& IN, int64_t P, float* dW, hipStream_t stream)
{
    return mp_dw_gemm_launch(mode_dz, mode_in, prec, &DZ, &IN, P, dW, stream);
}
#define MP_PREC_F (bf16 ? 1 : (split_enabled() ? 3 : 0))
#define MP_PREC_B (bf16 ? 1 : (split_enabled() ? (npl == 2 ? 2 : 3) : 0))
#define MP_POS_GEMM_B(MODE, KROW, EPI, ...) pos_gemm(MODE, KROW, EPI, MP_PREC_B, __VA_ARGS__)
#define MP_DW_GEMM_B(MODE_DZ, MODE_IN, ...) dw_gemm(MODE_DZ, MODE_IN, MP_PREC_B, __VA_ARGS__)
#define MP_POS_GEMM(MODE, KROW, EPI, ...) pos_gemm(MODE, KROW, EPI, MP_PREC_F, __VA_ARGS__)

// bf16 = true: every contraction runs on the bf16 matrix cores with both operands rounded to bf16 while they are staged -- [r3] through the
// same position-stream kernels as the fp32 path with ONE operand plane instead of three (fwd_chunk / bwd_fused <..., ONE>), the
// recomputed and the factorised first layers included (their VALU products on rounded operands: rb16); widths outside those
// kernels take the generic tiled kernels.  Everything else as in fp32.
static int sa_mlp_fwd(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                      int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                      void* workspace, size_t workspace_bytes, mp_stream_t stream_, bool bf16, const mp_syncbn_t* sync,
                      const mp_gather_t* gather = nullptr)
{
    if (sync && (!sync->allreduce || !sync->exchange || sync->world < 1)) return MP_EINVAL;
    if (!training) sync = nullptr;      // running statistics: nothing to exchange
    if (P < 0 || K <= 0 || n_layers <= 0 || !layers) return MP_EINVAL;
    if (P == 0) return MP_OK;
    if ((!x0 && !gather) || !out || !argk || !zmax || !workspace || (P % K) != 0) return MP_EINVAL;
    const bool factored = gather && factored_ok(gather, P, K, n_layers, layers, bf16);
    if (gather && !factored) return MP_EUNSUPPORTED;      // the gather descriptor carries the factorised first layer only
    if (n_layers > 8 || P > ((int64_t)1 << 31)) return MP_EUNSUPPORTED;
    int64_t ch[9];
    ch[0] = layers[0].c_in;
    for (int l = 0; l < n_layers; ++l) {
        const mp_mlp_layer_t& L = layers[l];
        if (!L.weight || !L.gamma || !L.beta || (!L.z && l > 0 && l != n_layers - 1) || !L.mean || !L.rstd || !L.scale || !L.shift) return MP_EINVAL;
        if (!training && (!L.running_mean || !L.running_var)) return MP_EINVAL;
        if (L.c_in != ch[l] || L.c_out <= 0 || L.c_out > 4096 || L.c_in > 4096) return MP_EINVAL;
        if ((L.c_in & 3) || (L.c_out & 3)) return MP_EUNSUPPORTED;      // float4 granularity: pad channels to x4
        if (P * (L.c_in > L.c_out ? L.c_in : L.c_out) >= ((int64_t)1 << 31)) return MP_EUNSUPPORTED;  // 32-bit offsets
        ch[l + 1] = L.c_out;
    }
    if (workspace_bytes < mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 0)) return MP_EWORKSPACE;
    // layers[0].z == NULL: the caller asks for the first layer to be recomputed instead of stored (mp_sa_mlp_recompute_first)
    const bool rc_first = layers[0].z == nullptr;
    if (rc_first && !mp_sa_mlp_recompute_first(n_layers, ch, K)) return MP_EINVAL;      // (bf16: x0 and layers[0].weight come pre-rounded)
    // bf16: activations stored as bf16 iff the whole chain runs on the position-stream kernels (mp_sa_mlp_bf16_storage); otherwise the
    // bf16 contractions of every layer behind the first run on the tiled kernels with fp32 storage
    const bool store16 = bf16 && chain_store16(n_layers, ch, K, rc_first ? 1 : (factored ? 2 : 0));
    const bool stream_ok = !bf16 || store16;
    if (bf16 && rc_first && !store16) return MP_EINVAL;
    if (n_layers > 1 && layers[n_layers - 1].z == nullptr) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    // consumer-side BatchNorm finalize (bn_prologue): train mode, per-replica statistics, every layer with its persistent state
    bool bn_fused = training && !sync && n_layers >= 2;
    for (int l = 0; l < n_layers; ++l) bn_fused = bn_fused && layers[l].bn_state != nullptr;
    BnOut partials{reinterpret_cast<float*>(workspace), nullptr, nullptr, 0, nullptr, 0};
    if (bn_fused) {     // the first kernel of the call zeroes what earlier calls left behind: the last forward site, the first-layer backward site
        const mp_mlp_layer_t &Lz = layers[n_layers - 1], &L0 = layers[0];
        partials.z0 = bn_fwd_slots(Lz.bn_state, (int)Lz.c_out);
        partials.n0 = BN_NS * 2 * (int)Lz.c_out;
        partials.z1 = bn_bwd_slots(L0.bn_state, (int)L0.c_out);
        partials.n1 = BN_NS * 2 * (int)L0.c_out;
    }
    // the forward site of layer l, to be consumed by the kernel launched after layer l's own (which also zeroes the rows of site l-1)
    auto fwd_site = [&](int l) {
        BnSite b{};
        if (!bn_fused) return b;
        const mp_mlp_layer_t& Lb = layers[l];
        const int C = (int)Lb.c_out;
        b.slots = bn_fwd_slots(Lb.bn_state, C);
        if (l > 0) { b.z0 = bn_fwd_slots(layers[l - 1].bn_state, (int)layers[l - 1].c_out); b.n0 = BN_NS * 2 * (int)layers[l - 1].c_out; }
        b.C = C;
        b.kind = 1;
        b.invP = 1.0 / (double)P;
        b.unbias = P > 1 ? (double)P / ((double)P - 1.0) : 1.0;
        b.momentum = momentum;
        b.eps = eps;
        b.gamma = Lb.gamma; b.beta = Lb.beta; b.bias = Lb.bias;
        b.running_mean = Lb.running_mean; b.running_var = Lb.running_var;
        b.mean = Lb.mean; b.rstd = Lb.rstd; b.o0 = Lb.scale; b.o1 = Lb.shift;
        return b;
    };
    // a consumer without a prologue (tiled GEMMs, the unfused pool): the site's algebra as a launch of its own
    auto settle = [&](BnSite& b) -> int {
        if (!b.slots) return MP_OK;
        hipLaunchKernelGGL(bn_site_finalize_kernel, dim3((unsigned)((b.C + 255) / 256)), dim3(256), 0, stream, b);
        b = BnSite{};
        return hipGetLastError() == hipSuccess ? MP_OK : MP_ELAUNCH;
    };
    // fused max-pool: group size a multiple of the 32-row MFMA tile that divides the 128-row block tile
    const bool fused_pool = (K == 32 || K == 64 || K == 128);
    PoolOut po{};
    {
        int64_t cmax = 0;
        for (int l = 0; l <= n_layers; ++l) cmax = ch[l] > cmax ? ch[l] : cmax;
        unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
        w += align_up((size_t)((P + 63) / 64) * 2 * (size_t)cmax * sizeof(float), 256);
        w += 3 * align_up((size_t)cmax * sizeof(float), 256);
        const size_t pb = align_up((size_t)(P / K) * (size_t)ch[n_layers] * sizeof(float), 256);
        po.vmax = reinterpret_cast<float*>(w);
        po.vmin = reinterpret_cast<float*>(w + pb);
        po.imax = reinterpret_cast<int*>(w + 2 * pb);
        po.imin = reinterpret_cast<int*>(w + 3 * pb);
        po.K = (int)K;
    }

    PosOperand A{};
    A.x = x0;
    A.C = (int)ch[0];
    A.K = (int)K;
    A.kshift = log2_or_neg(K);
    for (int l = 0; l < n_layers; ++l) {
        const mp_mlp_layer_t& L = layers[l];
        int nblk = 0, rc = MP_OK;
        partials.slots = bn_fused ? bn_fwd_slots(L.bn_state, (int)L.c_out) : nullptr;
        const bool fuse_pool = (l == n_layers - 1) && fused_pool;
        const int Ci_ = (int)L.c_in, Co_ = (int)L.c_out;
        const bool last_unfused = (l == n_layers - 1) && !fused_pool;
        if (l == 0 && factored) {
            const int ppb = MP_FACT_PPB;
            nblk = (int)((P + ppb - 1) / ppb);
            const double by = (store16 ? 2.0 : 4.0) * (double)P * Co_ + 4.0 * ((double)P * Co_ + 5.0 * (double)P);    // Z_0 written, A rows gathered (L2), indices + coordinates
            const int per = (int)(gather->S * K);
#define MP_FACT(Q_, NV_)                                                                                                         \
    MP_LAUNCH("first_factored_fwd_kernel", 8.0 * (double)P * Co_, by, (first_factored_fwd_kernel<Q_, NV_>), dim3((unsigned)nblk), dim3(256), 0,   \
              stream, gather->feats, gather->xyz, gather->new_xyz, gather->idx, L.weight, (int)P, (int)K, log2_or_neg(K), (int)gather->N, per, \
              log2_or_neg(per), ppb, L.z, partials, (int)bf16, (int)store16)
            if (store16) { if (Co_ == 64) MP_FACT(16, 2); else if (Co_ == 128) MP_FACT(32, 2); else MP_FACT(64, 2); }
            else { if (Co_ == 64) MP_FACT(16, 1); else if (Co_ == 128) MP_FACT(32, 1); else MP_FACT(64, 1); }
#undef MP_FACT
            MP_CHECK_LAUNCH();
        } else if (l == 0 && rc_first) {
            const int ppb = 1024;
            nblk = (int)((P + ppb - 1) / ppb);
            MP_LAUNCH("rc_stats_kernel", 8.0 * (double)P * Co_, 16.0 * (double)P, rc_stats_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, x0,
                      L.weight, (int)P, ppb, partials);
            MP_CHECK_LAUNCH();
        } else if (!bf16 && l == 0 && Ci_ == 132 && Co_ == 128 && !fuse_pool && chunk_fwd_enabled()) {
            // first layer of a level with a [128 features | xyz | pad] input: the position-stream kernel with the 4 extra columns
            // on the VALU (the tiled kernel pays a whole 32-wide k tile for them)
            int ppb = 1024;
            while ((P + ppb - 1) / ppb < fwd_wgs_wanted() && ppb > 128) ppb >>= 1;
            const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            const double eb = store16 ? 2.0 : 4.0;      // bytes per stored activation (bf16 storage of the bf16 variant's stream chains)
            const double fl = 2.0 * (double)P * Co_ * Ci_, by = eb * (double)P * (Ci_ + Co_) + 4.0 * (double)Co_ * Ci_;
            if (split_enabled())
                MP_LAUNCH("fwd_chunk_kernel<128, 128, false, 0, 4, split>", fl, by, (fwd_chunk_kernel<128, 128, false, SRC_ID, 4, true>), dim3(gx), dim3(256), 0, stream, A,
                          (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else
                MP_LAUNCH("fwd_chunk_kernel<128, 128, false, 0, 4>", fl, by, (fwd_chunk_kernel<128, 128, false, SRC_ID, 4>), dim3(gx), dim3(256), 0, stream, A,
                          (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            MP_CHECK_LAUNCH();
            nblk = (int)gx;
        } else if (l == 1 && rc_first) {
            int ppb = 1024;
            while ((P + ppb - 1) / ppb < fwd_wgs_wanted(Co_ == 64 ? 2048 : 512) && ppb > 128) ppb >>= 1;
            const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            const double fl = 2.0 * (double)P * Co_ * Ci_, by = 4.0 * ((double)P * 4 + (double)Co_ * Ci_) + (store16 ? 2.0 : 4.0) * (double)P * Co_;
            char tg[64];
            snprintf(tg, sizeof tg, "fwd_chunk_kernel<%d, %d, false, 4>", Ci_, Co_);
            int rc16 = 0;
            if (bf16 && store16) {      // [r5] sa_stream16.hip
                char tg16[64];
                snprintf(tg16, sizeof tg16, "fwd_stream16_kernel<%d, %d, false, 4>", Ci_, Co_);
                rc16 = mp_s16_fwd_launch(0, 1, Ci_, Co_, &A, P, ppb, L.weight, L.z, &partials, &po, L.gamma, tg16, fl, by, stream);
                if (rc16 < 0) return rc16;
            }
            if (rc16 == 1) {
            } else if (bf16 && Co_ == 64)
                MP_LAUNCH("fwd_chunk_bf16_kernel<64, 64, false, 4>", fl, by, (fwd_chunk_kernel<64, 64, false, SRC_ACT_RC, 0, true, true, true>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else if (bf16)
                MP_LAUNCH("fwd_chunk_bf16_kernel<64, 128, false, 4>", fl, by, (fwd_chunk_kernel<64, 128, false, SRC_ACT_RC, 0, true, true, true>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else if (Co_ == 64 && split_enabled())
                MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<64, 64, false, SRC_ACT_RC, 0, true>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else if (Co_ == 64)
                MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<64, 64, false, SRC_ACT_RC>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else if (split_enabled())
                MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<64, 128, false, SRC_ACT_RC, 0, true>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            else
                MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<64, 128, false, SRC_ACT_RC>), dim3(gx), dim3(256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, false), L.weight, L.z, partials, po, L.gamma);
            MP_CHECK_LAUNCH();
            nblk = (int)gx;
        } else if (l > 0 && stream_ok && (Ci_ == 64 || Ci_ == 128) && (Co_ == 64 || Co_ == 128 || Co_ == 256) && (P % K) == 0 && (1024 % K == 0 || !fuse_pool) &&
            chunk_fwd_enabled() && !(fuse_pool && (K % 32) != 0) && !(bf16 && Co_ == 64 && Ci_ == 128)) {
            (void)last_unfused;
            int ppb = 1024;
            while ((P + ppb - 1) / ppb < fwd_wgs_wanted(Co_ == 256 ? 256 : 512) && ppb > 128 && (!fuse_pool || (ppb / 2) % K == 0)) ppb >>= 1;   // >= 512 workgroups when P allows
            const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            const double eb = store16 ? 2.0 : 4.0;      // bytes per stored activation (bf16 storage of the bf16 variant's stream chains)
            const double fl = 2.0 * (double)P * Co_ * Ci_, by = eb * (double)P * (Ci_ + Co_) + 4.0 * (double)Co_ * Ci_;
            char tg[64];
            snprintf(tg, sizeof tg, bf16 ? "fwd_chunk_bf16_kernel<%d, %d, %s>" : "fwd_chunk_kernel<%d, %d, %s>", Ci_, Co_, fuse_pool ? "true" : "false");
            int rc16 = 0;
            if (bf16 && store16) {      // [r5] sa_stream16.hip
                char tg16[64];
                snprintf(tg16, sizeof tg16, "fwd_stream16_kernel<%d, %d, %s>", Ci_, Co_, fuse_pool ? "true" : "false");
                rc16 = mp_s16_fwd_launch(fuse_pool ? 1 : 0, 0, Ci_, Co_, &A, P, ppb, L.weight, L.z, &partials, &po, L.gamma, tg16, fl, by, stream);
                if (rc16 < 0) return rc16;
            }
#define MP_FWD(CI, CO, PL)                                                                                                     \
    MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<CI, CO, PL>), dim3(gx), dim3(CO > 128 ? 512 : 256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, CI == 64 && PL), L.weight, L.z, \
              partials, po, L.gamma)
#define MP_FWD_SPLIT(CI, CO, PL)                                                                                               \
    if (bf16)                                                                                                                  \
        MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<CI, CO, PL, SRC_ACT, 0, true, true, true>), dim3(gx), dim3(CO > 128 ? 512 : 256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, CI == 64 && PL), \
                  L.weight, L.z, partials, po, L.gamma);                                                                       \
    else                                                                                                                       \
        MP_LAUNCH(tg, fl, by, (fwd_chunk_kernel<CI, CO, PL, SRC_ACT, 0, true>), dim3(gx), dim3(CO > 128 ? 512 : 256), 0, stream, A, (int)P, fwd_il(ppb, P, Co_, CI == 64 && PL), \
                  L.weight, L.z, partials, po, L.gamma)
#define MP_FWD_CO(CI, PL)                                  \
    if (Co_ == 64) { if (bf16) MP_FWD_SPLIT(CI, 64, PL); else MP_FWD(CI, 64, PL); }                     \
    else if (Co_ == 128) { if (split_enabled() || bf16) MP_FWD_SPLIT(CI, 128, PL); else MP_FWD(CI, 128, PL); }              \
    else { if (split_enabled() || bf16) MP_FWD_SPLIT(CI, 256, PL); else MP_FWD(CI, 256, PL); }
            if (rc16 == 1) {
            } else if (fuse_pool) { if (Ci_ == 64) { MP_FWD_CO(64, true); } else { MP_FWD_CO(128, true); } }
            else { if (Ci_ == 64) { MP_FWD_CO(64, false); } else { MP_FWD_CO(128, false); } }
#undef MP_FWD_CO
#undef MP_FWD_SPLIT
#undef MP_FWD
            MP_CHECK_LAUNCH();
            nblk = (int)gx;
        } else if (fuse_pool) {
            if (int r2 = settle(A.bn)) return r2;
            if (l == 0)
                rc = MP_POS_GEMM(SRC_ID, false, EPI_SQ_POOL, A, P, L.weight, (int)L.c_out, (int)L.c_in, L.z, partials,
                                 nullptr, nullptr, nullptr, stream, &nblk, po);
            else
                rc = MP_POS_GEMM(SRC_ACT, false, EPI_SQ_POOL, A, P, L.weight, (int)L.c_out, (int)L.c_in, L.z, partials,
                                 nullptr, nullptr, nullptr, stream, &nblk, po);
        } else if (l == 0)
            rc = MP_POS_GEMM(SRC_ID, false, EPI_SQ, A, P, L.weight, (int)L.c_out, (int)L.c_in, L.z, partials, nullptr,
                             nullptr, nullptr, stream, &nblk);
        else {
            if (int r2 = settle(A.bn)) return r2;
            rc = MP_POS_GEMM(SRC_ACT, false, EPI_SQ, A, P, L.weight, (int)L.c_out, (int)L.c_in, L.z, partials, nullptr,
                             nullptr, nullptr, stream, &nblk);
        }
        if (rc != MP_OK) return rc;
        partials.z0 = partials.z1 = nullptr;      // (only the first kernel of the call has the zeroing duty)
        const int C = (int)L.c_out;
        if (!bn_fused) {
            double Pg = (double)P;
            const double* gsums = nullptr;
            if (sync) {     // global-batch statistics: local fp64 sums -> all-reduce (the caller's collective) -> finalize
                hipLaunchKernelGGL(bn_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, stream, partials.rows, nblk, C,
                                   sync->exchange, (double*)nullptr);
                MP_CHECK_LAUNCH();
                if (sync->allreduce(sync->user, sync->exchange, 2 * (int64_t)C, stream_) != 0) return MP_ELAUNCH;
                Pg = (double)P * (double)sync->world;
                gsums = sync->exchange;
            }
            hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, stream, partials.rows, nblk, C,
                               1.0 / Pg, Pg > 1.0 ? Pg / (Pg - 1.0) : 1.0, training, momentum, eps, L.gamma,
                               L.beta, L.bias, L.running_mean, L.running_var, L.mean, L.rstd, L.scale, L.shift, gsums);
            MP_CHECK_LAUNCH();
        }
        A = PosOperand{};
        A.x = L.z;
        A.s = L.scale;
        A.t = L.shift;
        A.C = C;
        A.K = (int)K;
        A.kshift = log2_or_neg(K);
        A.bn = fwd_site(l);            // the next kernel that reads act(Z_l) derives (scale, shift) of layer l in its prologue
        if (l == 0 && rc_first) { A.rx = x0; A.rw = L.weight; }
    }
    const mp_mlp_layer_t& LL = layers[n_layers - 1];
    const int64_t G = P / K;
    const int64_t tot = G * LL.c_out;
    BnSite last_site = fwd_site(n_layers - 1);
    if (!fused_pool || LL.c_out > BN_POOL_CMAX) { if (int r2 = settle(last_site)) return r2; }
    if (fused_pool) {
        const bool wide = !(LL.c_out & 3) && !((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(argk) | reinterpret_cast<uintptr_t>(zmax) |
                                                reinterpret_cast<uintptr_t>(LL.scale) | reinterpret_cast<uintptr_t>(LL.shift) | reinterpret_cast<uintptr_t>(po.vmax) |
                                                reinterpret_cast<uintptr_t>(po.vmin) | reinterpret_cast<uintptr_t>(po.imax) | reinterpret_cast<uintptr_t>(po.imin)) & 15);
        if (wide) {
            unsigned gx = (unsigned)((tot / 4 + 255) / 256);
            if (last_site.slots && gx > 512) gx = 512;       // every workgroup pays the BatchNorm prologue: fewer, grid-striding ones
            MP_LAUNCH("pool_select_kernel", 0.0, 28.0 * (double)tot, pool_select_kernel, dim3(gx),
                      dim3(256), 0, stream, po, LL.scale, LL.shift, G, (int)LL.c_out, out, argk, zmax, last_site);
        }
        else
            MP_LAUNCH("pool_select_kernel", 0.0, 28.0 * (double)tot, pool_select_scalar_kernel, dim3((unsigned)((tot + 255) / 256)),
                      dim3(256), 0, stream, po, LL.scale, LL.shift, G, (int)LL.c_out, out, argk, zmax, last_site);
        MP_CHECK_LAUNCH();
        return MP_OK;
    }
    MP_LAUNCH("pool_fwd_kernel", 0.0, 4.0 * (double)P * LL.c_out + 12.0 * (double)tot, pool_fwd_kernel,
              dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, LL.z, LL.scale, LL.shift, G, (int)K, (int)LL.c_out, out,
              argk, zmax);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_sa_mlp_fwd_f32(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                 int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                 void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    return sa_mlp_fwd(x0, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, false, nullptr);
}

extern "C" int mp_sa_mlp_fwd_gather_f32(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                        int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                        void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    if (!gather) return MP_EINVAL;
    return sa_mlp_fwd(nullptr, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, false, nullptr,
                      gather);
}

extern "C" int mp_sa_mlp_fwd_gather_bf16(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                         int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                         void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    if (!gather || n_layers < 1 || !layers || layers[0].c_in != 4) return MP_EINVAL;      // the factorised form only
    return sa_mlp_fwd(nullptr, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, true, nullptr,
                      gather);
}

extern "C" int mp_sa_mlp_fwd_gather_ex(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                       int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                       void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream)
{   // the gather forms with the SyncBN hook of mp_sa_mlp_fwd_ex (bf16: the factorised form only, as mp_sa_mlp_fwd_gather_bf16)
    if (!gather || n_layers < 1 || !layers || (bf16 && layers[0].c_in != 4)) return MP_EINVAL;
    return sa_mlp_fwd(nullptr, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, bf16 != 0, sync,
                      gather);
}

extern "C" int mp_sa_mlp_fwd_bf16(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                  int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                  void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    return sa_mlp_fwd(x0, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, true, nullptr);
}

extern "C" int mp_sa_mlp_fwd_ex(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                int training, double momentum, double eps, float* out, int32_t* argk, float* zmax,
                                void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream)
{
    return sa_mlp_fwd(x0, P, K, n_layers, layers, training, momentum, eps, out, argk, zmax, workspace, workspace_bytes, stream, bf16 != 0, sync);
}

static int sa_mlp_bwd(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                      int training, const float* grad_out, const float* out, const int32_t* argk,
                      const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                      void* workspace, size_t workspace_bytes, mp_stream_t stream_, bool bf16, const mp_syncbn_t* sync,
                      const mp_gather_t* gather = nullptr)
{
    if (sync && (!sync->allreduce || !sync->exchange || sync->world < 1)) return MP_EINVAL;
    if (!training) sync = nullptr;
    if (P < 0 || K <= 0 || n_layers <= 0 || !layers || !grads) return MP_EINVAL;
    if (P == 0) return MP_OK;
    if ((!x0 && !gather) || !grad_out || !out || !argk || !zmax || !workspace || (P % K) != 0) return MP_EINVAL;
    const bool factored = gather && factored_ok(gather, P, K, n_layers, layers, bf16);
    // factored: grad_x0_cols == Co: dZ_0 rows [P, Co + 4] for the caller to reduce; grad_x0_cols == 0: grad_x0 IS dA [B, N, Co], reduced here
    if (factored && (!grad_x0 || (grad_x0_cols != layers[0].c_out && grad_x0_cols != 0))) return MP_EINVAL;
    if (factored && grad_x0_cols == 0 && (gather->N > 15000 || gather->S * K >= ((int64_t)1 << 24))) return MP_EUNSUPPORTED;
    if (gather && !factored) return MP_EUNSUPPORTED;
    if (n_layers > 8) return MP_EUNSUPPORTED;
    int64_t ch[9];
    ch[0] = layers[0].c_in;
    int64_t cmax = ch[0];
    for (int l = 0; l < n_layers; ++l) {
        ch[l + 1] = layers[l].c_out;
        cmax = ch[l + 1] > cmax ? ch[l + 1] : cmax;
        if (!grads[l].d_weight || !grads[l].d_gamma || !grads[l].d_beta) return MP_EINVAL;
        if ((layers[l].c_in & 3) || (layers[l].c_out & 3)) return MP_EUNSUPPORTED;
    }
    if (P * cmax >= ((int64_t)1 << 31)) return MP_EUNSUPPORTED;
    if (workspace_bytes < mp_sa_mlp_workspace_bytes(P, K, n_layers, ch, 1)) return MP_EWORKSPACE;
    const bool rc_first = layers[0].z == nullptr;   // the forward pass did not store Z_0 (mp_sa_mlp_recompute_first)
    if (rc_first && (grad_x0 || !mp_sa_mlp_recompute_first(n_layers, ch, K))) return MP_EINVAL;
    const bool store16 = bf16 && chain_store16(n_layers, ch, K, rc_first ? 1 : (factored ? 2 : 0));     // (as in the forward pass)
    const bool stream_ok = !bf16 || store16;
    if (bf16 && rc_first && !store16) return MP_EINVAL;
    for (int l = 1; l < n_layers; ++l)
        if (!layers[l].z) return MP_EINVAL;
    const int npl = bwd_planes();
    hipStream_t stream = mp_stream(stream_);
    // carve the workspace (same order as mp_sa_mlp_workspace_bytes)
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    const size_t nblk_max = (size_t)((P + 63) / 64);
    bool bn_fused = training && !sync && n_layers >= 2;      // consumer-side BatchNorm finalize (bn_prologue), as in the forward pass
    for (int l = 0; l < n_layers; ++l) bn_fused = bn_fused && layers[l].bn_state != nullptr;
    BnOut partials{reinterpret_cast<float*>(w), nullptr, nullptr, 0, nullptr, 0};
    if (bn_fused) {     // pool_bwd_prep_kernel, the first kernel of the call: the rows the forward pass and an earlier backward pass left behind
        const mp_mlp_layer_t &Lz = layers[n_layers - 1], &L0 = layers[0];
        partials.z0 = bn_fwd_slots(Lz.bn_state, (int)Lz.c_out);
        partials.n0 = BN_NS * 2 * (int)Lz.c_out;
        partials.z1 = bn_bwd_slots(L0.bn_state, (int)L0.c_out);
        partials.n1 = BN_NS * 2 * (int)L0.c_out;
    }
    w += align_up(nblk_max * 2 * (size_t)cmax * sizeof(float), 256);
    float* cbuf[3];
    for (int i = 0; i < 3; ++i) {
        cbuf[i] = reinterpret_cast<float*>(w);
        w += align_up((size_t)cmax * sizeof(float), 256);
    }
    float* gbuf[2];
    for (int i = 0; i < 2; ++i) {
        gbuf[i] = reinterpret_cast<float*>(w);
        w += align_up((size_t)P * (size_t)cmax * sizeof(float), 256);
    }
    float* gp = reinterpret_cast<float*>(w);
    w += align_up((size_t)(P / K) * (size_t)ch[n_layers] * sizeof(float), 256);

    const int64_t G = P / K;
    const int L = n_layers;
    const mp_mlp_layer_t& last = layers[L - 1];
    // the backward sums of layer `li`: where its producer adds them, and the site its first consumer derives (a, e, f) from
    auto bwd_slots = [&](int li) -> double* { return bn_fused ? bn_bwd_slots(layers[li].bn_state, (int)layers[li].c_out) : nullptr; };
    auto bwd_site = [&](int li) {
        BnSite b{};
        if (!bn_fused) return b;
        const mp_mlp_layer_t& Lb = layers[li];
        const int C = (int)Lb.c_out;
        b.slots = bn_bwd_slots(Lb.bn_state, C);
        if (li + 1 < n_layers) { b.z0 = bn_bwd_slots(layers[li + 1].bn_state, (int)layers[li + 1].c_out); b.n0 = BN_NS * 2 * (int)layers[li + 1].c_out; }
        b.C = C;
        b.kind = 2;
        b.invP = 1.0 / (double)P;
        b.gamma = Lb.gamma;
        b.mean = Lb.mean; b.rstd = Lb.rstd;
        b.o0 = cbuf[0]; b.o1 = cbuf[1]; b.o2 = cbuf[2];
        b.dgamma = grads[li].d_gamma; b.dbeta = grads[li].d_beta; b.dbias = grads[li].d_bias;
        return b;
    };
    // BatchNorm-backward finalize of layer `li` from `nb` partial rows (SyncBN: local sums kept, global sums exchanged)
    auto finalize_bwd = [&](int li, int nb, int C) -> int {
        if (bn_fused) return MP_OK;          // the first kernel that forms dZ_li does it in its prologue
        const mp_mlp_layer_t& Lf = layers[li];
        double Pg = (double)P;
        const double *gs = nullptr, *ls = nullptr;
        if (sync) {
            hipLaunchKernelGGL(bn_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, stream, partials.rows, nb, C,
                               sync->exchange, sync->exchange + 2 * (size_t)C);
            if (hipGetLastError() != hipSuccess) return MP_ELAUNCH;
            if (sync->allreduce(sync->user, sync->exchange, 2 * (int64_t)C, stream_) != 0) return MP_ELAUNCH;
            Pg = (double)P * (double)sync->world;
            gs = sync->exchange;
            ls = sync->exchange + 2 * (size_t)C;
        }
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_SL), 0, stream, partials.rows, nb, C,
                           1.0 / Pg, training, Lf.gamma, Lf.mean, Lf.rstd, grads[li].d_gamma, grads[li].d_beta, grads[li].d_bias,
                           cbuf[0], cbuf[1], cbuf[2], gs, ls);
        return hipGetLastError() == hipSuccess ? MP_OK : MP_ELAUNCH;
    };
    // dW is accumulated with atomics and must start from zero: one clear for the whole level when the caller laid the
    // layers' buffers out back to back (the Python layer does) -- folded into the launch below --, one per layer otherwise
    bool dw_joint = L > 1;
    size_t dw_total = 0;
    for (int l = 0; l < L; ++l) {
        if (l + 1 < L && grads[l + 1].d_weight != grads[l].d_weight + (size_t)layers[l].c_out * layers[l].c_in) dw_joint = false;
        dw_total += (size_t)layers[l].c_out * layers[l].c_in;
    }
    // pooled gradient through the last ReLU + its BatchNorm-backward sums
    {
        const int C = (int)last.c_out;
        // rows of pooled groups per workgroup: POOL_ROWS, fewer while that leaves fewer than ~512 workgroups and the partial rows fit
        int rows = POOL_ROWS;
        while (rows > 1 && ((G + rows - 1) / rows) * ((C + 255) / 256) < 512 && (size_t)((G + rows / 2 - 1) / (rows / 2)) <= nblk_max) rows >>= 1;
        const int nb = (int)((G + rows - 1) / rows);
        if ((size_t)nb > nblk_max) return MP_EUNSUPPORTED;  // partials hold P/64 rows: needs K >= 4
        partials.slots = bwd_slots(L - 1);
        hipLaunchKernelGGL(pool_bwd_prep_kernel, dim3(nb, (C + 255) / 256), dim3(256), 0, stream, grad_out, out, zmax, G, C, rows, gp,
                           partials, dw_joint ? grads[0].d_weight : nullptr, dw_joint ? dw_total : (size_t)0);
        MP_CHECK_LAUNCH();
        partials.z0 = partials.z1 = nullptr;
        if (int rc = finalize_bwd(L - 1, nb, C)) return rc;
    }
    const float* G_cur = nullptr;  // dense gradient w.r.t. the activation output of layer l (l < L-1)
    for (int l = L - 1; l >= 0; --l) {
        const mp_mlp_layer_t& Ly = layers[l];
        const int Co = (int)Ly.c_out, Ci = (int)Ly.c_in;
        const bool pooled = (l == L - 1);
        PosOperand DZ{};
        DZ.x = Ly.z;
        DZ.s = Ly.scale;
        DZ.t = Ly.shift;
        DZ.a = cbuf[0];
        DZ.e = cbuf[1];
        DZ.f = cbuf[2];
        DZ.C = Co;
        DZ.K = (int)K;
        DZ.kshift = log2_or_neg(K);
        if (pooled) { DZ.g = gp; DZ.argk = argk; } else { DZ.g = G_cur; }
        DZ.bn = bwd_site(l);                            // the first kernel below that forms dZ_l derives its constants
        partials.slots = l > 0 ? bwd_slots(l - 1) : nullptr;      // ... and the kernel that computes G_{l-1} adds the sums of layer l-1
        PosOperand IN{};
        IN.C = Ci;
        IN.K = (int)K;
        IN.kshift = log2_or_neg(K);
        if (l == 0) { IN.x = x0; }
        else { IN.x = layers[l - 1].z; IN.s = layers[l - 1].scale; IN.t = layers[l - 1].shift; }
        if (rc_first && l == 1) { IN.rx = x0; IN.rw = layers[0].weight; }   // act(Z_0) and raw Z_0 from the input rows
        if (rc_first && l == 0) { DZ.rx = x0; DZ.rw = Ly.weight; }          // dZ_0 = f(Z_0, G_0) likewise

        // dW_l = dZ_l^T * act(Z_{l-1})
        if (!dw_joint && !mp::zero_async(grads[l].d_weight, (size_t)Co * Ci, stream)) return MP_ELAUNCH;
        if (l > 0 && stream_ok && (Ci == 64 || Ci == 128) && (Co == 64 || Co == 128 || (Co == 256 && Ci == 128)) && fused_bwd_enabled() && !(bf16 && Co == 64 && Ci == 128)) {
            // single-tile layer: dX, dW and the BatchNorm-backward sums of layer l-1 in one pass over dZ_l (bwd_fused_kernel)
            const mp_mlp_layer_t& Pv = layers[l - 1];
            float* Gn = gbuf[l & 1];
            const int ppb = 1024;      // positions per workgroup
            const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            if ((size_t)gx > nblk_max) return MP_EUNSUPPORTED;
            const double fl = 4.0 * (double)P * Co * Ci;
            const double eb = store16 ? 2.0 : 4.0;      // bytes per stored Z / G element
            const double by = eb * ((pooled ? 1.0 : 2.0) * (double)P * Co + 2.0 * (double)P * Ci);
            if (bf16 && store16) {
                // [r5] the kernels written for one plane and bf16 storage (sa_stream16.hip: rows straight into an LDS ring)
                char tg16[64];
                const bool rc_in = rc_first && l == 1;
                snprintf(tg16, sizeof tg16, rc_in ? "bwd_stream16_kernel<%d, %d, %d, 4>" : "bwd_stream16_kernel<%d, %d, %d>", pooled ? 3 : 2, Co, Ci);
                const int rc16 = mp_s16_bwd_launch(pooled ? 1 : 0, rc_in ? 1 : 0, Co, Ci, &DZ, &IN, P, ppb, Ly.weight, grads[l].d_weight, Gn, &partials, tg16,
                                                   fl, rc_in ? by - eb * (double)P * Ci + 16.0 * (double)P : by, stream);
                if (rc16 < 0) return rc16;
                if (rc16 == 1) {
                    if (int rc = finalize_bwd(l - 1, (int)gx, Ci)) return rc;
                    G_cur = Gn;
                    continue;
                }
            }
            {
                const bool rc_in = rc_first && l == 1;
                if (int rc = mp_bwd_fused_launch(pooled ? 1 : 0, rc_in ? 1 : 0, bf16 ? 1 : 0, split_enabled() ? 1 : 0, npl, Co, Ci, &DZ, &IN, P, ppb, Ly.weight,
                                                 grads[l].d_weight, Gn, &partials, fl, by, by - eb * (double)P * Ci + 16.0 * (double)P, stream))
                    return rc;
            }
            MP_CHECK_LAUNCH();
            (void)Pv;
            if (int rc = finalize_bwd(l - 1, (int)gx, Ci)) return rc;
            G_cur = Gn;
            continue;
        }
        if (!bf16 && l == 0 && grad_x0 && Co == 128 && Ci == 132 && grad_x0_cols == 128 && fused_bwd_enabled()) {
            // first layer of a level with a [128 features | xyz | pad] input: dW and the feature columns of grad_x0 in one pass
            const int ppb = 1024;      // positions per workgroup
            const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            const double fl = 2.0 * (double)P * Co * (Ci + 128), by = 4.0 * ((pooled ? 1.0 : 2.0) * (double)P * Co + (double)P * (Ci + 128));
            if (pooled && split_enabled())
                MP_LAUNCH("bwd_first_kernel<3>", fl, by, (bwd_first_kernel<SRC_DZ_POOLED, true>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb,
                          Ly.weight, grads[l].d_weight, grad_x0);
            else if (pooled)
                MP_LAUNCH("bwd_first_kernel<3>", fl, by, (bwd_first_kernel<SRC_DZ_POOLED>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb,
                          Ly.weight, grads[l].d_weight, grad_x0);
            else if (split_enabled())
                MP_LAUNCH("bwd_first_kernel<2>", fl, by, (bwd_first_kernel<SRC_DZ, true>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, Ly.weight,
                          grads[l].d_weight, grad_x0);
            else
                MP_LAUNCH("bwd_first_kernel<2>", fl, by, (bwd_first_kernel<SRC_DZ>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, Ly.weight,
                          grads[l].d_weight, grad_x0);
            MP_CHECK_LAUNCH();
            continue;
        }
        if (factored && l == 0) {
            // dW_x from the gathered coordinates, dZ_0 itself (= the gradient of the gathered A rows) out to the caller
            const double fl = 2.0 * (double)P * Co * 4, by = (store16 ? 2.0 : 4.0) * 2.0 * (double)P * Co + 20.0 * (double)P;   // Z_0 and G_0 rows in their storage type
            if (grad_x0_cols == 0) {
                // sorted-row reduce: no dZ_0 round trip.  Scratch: the G buffer that is free at this layer (2 ints per position).
                const int M = (int)(gather->S * K), Np = (int)gather->N;
                const int64_t Bc = P / M;
                // [r4] gather->rows: the sorted row lists prepared by the caller (mp_csr_rows_i64: they depend on idx alone -- a training
                // harness sorts them with the sampling plan, off the step's stream); NULL: sorted here
                const int* order = gather->rows;
                const int* pts = order ? order + P : nullptr;
                if (!order) {
                    int* o_ = reinterpret_cast<int*>(gbuf[0]);
                    if (int rc = launch_csr_rows(gather->idx, Bc, Np, M, o_, o_ + P, stream)) return rc;
                    order = o_;
                    pts = o_ + P;
                }
                if (!mp::zero_async(grad_x0, (size_t)Bc * Np * Co, stream)) return MP_ELAUNCH;
#ifndef MP_FACT_RCHUNK
#define MP_FACT_RCHUNK 256
#endif
                // rows per wave: long pieces (fewer shared runs, more rows in flight per slot) as long as the grid still covers the chip.
                // Measured at the bench shape (8 192 rows x 32 clouds, 512-byte rows): 64: 122 us, 128: 90, 256 (one workgroup per CU): 80,
                // 512: 122; the multi-scale level's 256-byte rows (4 096 x 32) prefer two workgroups per CU
                int chunk = MP_FACT_RCHUNK;
                const int64_t min_wg = Co >= 128 ? 256 : 512;
                const bool wide = store16 && Co >= 64;          // bf16 storage: eight channels (16 bytes) per lane
                const int slots = 64 / (int)(Co / (wide ? 8 : 4));       // rows per wave instruction
                while (chunk > 16 * slots && (int64_t)((M + 4 * chunk - 1) / (4 * chunk)) * Bc < min_wg) chunk >>= 1;
                const unsigned gxr = (unsigned)((M + 4 * chunk - 1) / (4 * chunk));
#define MP_FACT_R(Q_, NV_)                                                                                                       \
    MP_LAUNCH("first_factored_reduce_kernel", fl, by, (first_factored_reduce_kernel<Q_, NV_>), dim3(gxr, (unsigned)Bc), dim3(256), 0, stream, DZ, order, pts, \
              gather->xyz, gather->new_xyz, Np, M, (int)K, log2_or_neg(K), chunk, grad_x0, grads[l].d_weight, (int)bf16, (int)store16)
                if (wide) { if (Co == 64) MP_FACT_R(16, 2); else if (Co == 128) MP_FACT_R(32, 2); else MP_FACT_R(64, 2); }
                else { if (Co == 64) MP_FACT_R(16, 1); else if (Co == 128) MP_FACT_R(32, 1); else MP_FACT_R(64, 1); }
#undef MP_FACT_R
                MP_CHECK_LAUNCH();
                continue;
            }
            const int ppb = 2 * MP_FACT_PPB, per = (int)(gather->S * K);      // measured: 512 here (fewer dW_x atomics), 256 forward
            const unsigned gxf = (unsigned)((P + ppb - 1) / ppb);
#define MP_FACT_B(Q_)                                                                                                            \
    MP_LAUNCH("first_factored_bwd_kernel", fl, by + 4.0 * (double)P * Co, (first_factored_bwd_kernel<Q_>), dim3(gxf), dim3(256), 0, stream, DZ, gather->xyz, \
              gather->new_xyz, gather->idx, (int)P, (int)K, log2_or_neg(K), (int)gather->N, per, log2_or_neg(per), ppb, grad_x0, grads[l].d_weight, (int)bf16, (int)store16)
            if (Co == 64) MP_FACT_B(16); else if (Co == 128) MP_FACT_B(32); else MP_FACT_B(64);
#undef MP_FACT_B
            MP_CHECK_LAUNCH();
            continue;
        }
        if (rc_first && l == 0) {
            const double fl = 2.0 * (double)P * Co * Ci, by = 4.0 * ((double)P * Co + 2.0 * (double)P * Ci);
            if (int rc = mp_dw_ci4_rc_launch(&DZ, &IN, P, grads[l].d_weight, (int)bf16, (int)store16, fl, by, stream)) return rc;
            continue;
        }
        // [r6] both products of the layer as ONE launch where the joint kernel applies (sa_gemm.hip: bwd_pair_kernel -- two planes, the
        // group_all level's shapes): the constants of dZ_l from a finalize launch in front, then dX and dW tiles side by side
        if (!bf16 && split_enabled() && npl == 2 && bn_fused && (l > 0 || grad_x0)) {
            const int mdz = pooled ? SRC_DZ_POOLED : SRC_DZ;
            const float *zp = nullptr, *sp = nullptr, *tp = nullptr;
            float* Gout = nullptr;
            int ncols = Ci, ldw_ = 0, ldc_ = 0, epi = EPI_DY, min_ = SRC_ACT;
            bool ok = true, zero_tail = false;
            BnOut part = partials;
            if (l > 0) {
                Gout = gbuf[l & 1];
                zp = layers[l - 1].z; sp = layers[l - 1].scale; tp = layers[l - 1].shift;
            } else {
                const bool compact = grad_x0_cols < 0;
                const int64_t gcols = compact ? -grad_x0_cols : grad_x0_cols;
                if (compact && ((gcols & 3) || gcols > Ci)) return MP_EINVAL;
                ncols = (gcols > 0 && gcols < Ci) ? (int)((gcols + 3) / 4 * 4) : Ci;
                ldw_ = Ci;
                ldc_ = compact ? ncols : Ci;
                zero_tail = !compact && ncols < Ci;
                Gout = grad_x0;
                epi = EPI_NONE;
                min_ = SRC_ID;
                part = BnOut{nullptr, nullptr};
            }
            ok = mp_bwd_pair_launch(mdz, min_, epi, &DZ, &IN, P, Ly.weight, ncols, Co, Gout, &part, zp, sp, tp, ldw_, ldc_, grads[l].d_weight, stream, nullptr, 1) == 1;
            if (ok) {
                if (DZ.bn.slots) {
                    hipLaunchKernelGGL(bn_site_finalize_kernel, dim3((unsigned)((DZ.bn.C + 255) / 256)), dim3(256), 0, stream, DZ.bn);
                    MP_CHECK_LAUNCH();
                    DZ.bn = BnSite{};
                }
                int nblk = 0;
                const int rc = mp_bwd_pair_launch(mdz, min_, epi, &DZ, &IN, P, Ly.weight, ncols, Co, Gout, &part, zp, sp, tp, ldw_, ldc_, grads[l].d_weight, stream, &nblk, 0);
                if (rc < 0) return rc;
                if (rc == 1) {
                    if (l > 0) {
                        if (int rc2 = finalize_bwd(l - 1, nblk, Ci)) return rc2;
                        G_cur = Gout;
                    } else if (zero_tail) {
                        hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((P * (Ci - ncols) + 255) / 256)), dim3(256), 0, stream, grad_x0, P, Ci, ncols);
                        MP_CHECK_LAUNCH();
                    }
                    continue;
                }
            }
        }
        {
            int rc;
            if (pooled) rc = (l == 0) ? MP_DW_GEMM_B(SRC_DZ_POOLED, SRC_ID, DZ, IN, P, grads[l].d_weight, stream)
                                      : MP_DW_GEMM_B(SRC_DZ_POOLED, SRC_ACT, DZ, IN, P, grads[l].d_weight, stream);
            else rc = (l == 0) ? MP_DW_GEMM_B(SRC_DZ, SRC_ID, DZ, IN, P, grads[l].d_weight, stream)
                               : MP_DW_GEMM_B(SRC_DZ, SRC_ACT, DZ, IN, P, grads[l].d_weight, stream);
            if (rc != MP_OK) return rc;
            DZ.bn = BnSite{};        // (consumed by the weight-gradient kernel: the kernels below read the constants it wrote)
        }
        // G_{l-1} = dZ_l * W_l  (+ BN-backward sums of layer l-1)
        if (l > 0) {
            const mp_mlp_layer_t& Pv = layers[l - 1];
            float* Gn = gbuf[l & 1];
            int nblk = 0, rc;
            if (pooled)
                rc = MP_POS_GEMM_B(SRC_DZ_POOLED, true, EPI_DY, DZ, P, Ly.weight, Ci, Co, Gn, partials, Pv.z, Pv.scale, Pv.shift, stream, &nblk);
            else
                rc = MP_POS_GEMM_B(SRC_DZ, true, EPI_DY, DZ, P, Ly.weight, Ci, Co, Gn, partials, Pv.z, Pv.scale, Pv.shift, stream, &nblk);
            if (rc != MP_OK) return rc;
            // the constants of layer l are still being read by the kernels above: they are stream-ordered, so
            // overwriting cbuf for layer l-1 here is safe.
            if (int rc2 = finalize_bwd(l - 1, nblk, Ci)) return rc2;
            G_cur = Gn;
        } else if (grad_x0) {
            // only the first grad_x0_cols input channels need a gradient (the features; the centred coordinates that
            // follow them are not differentiated): skip the N tiles beyond them.  Rows of grad_x0 stay Ci apart.
            // [r4] grad_x0_cols < 0: COMPACT -- grad_x0 is [P, -grad_x0_cols] (a multiple of 4), the gradient of the feature columns alone
            // with nothing behind them (the caller differentiates the features, not the concatenated rows: no slice, no copy, no clear)
            const bool compact = grad_x0_cols < 0;
            const int64_t gcols = compact ? -grad_x0_cols : grad_x0_cols;
            if (compact && ((gcols & 3) || gcols > Ci)) return MP_EINVAL;
            int ncols = (gcols > 0 && gcols < Ci) ? (int)((gcols + 3) / 4 * 4) : Ci;
            const int ldc = compact ? ncols : Ci;
            int rc;
            if (pooled)
                rc = MP_POS_GEMM_B(SRC_DZ_POOLED, true, EPI_NONE, DZ, P, Ly.weight, ncols, Co, grad_x0, BnOut{nullptr, nullptr}, nullptr, nullptr, nullptr, stream, nullptr, PoolOut{}, Ci, ldc);
            else
                rc = MP_POS_GEMM_B(SRC_DZ, true, EPI_NONE, DZ, P, Ly.weight, ncols, Co, grad_x0, BnOut{nullptr, nullptr}, nullptr, nullptr, nullptr, stream, nullptr, PoolOut{}, Ci, ldc);
            if (rc != MP_OK) return rc;
            if (!compact && ncols < Ci) {   // the columns that carry no gradient: defined (zero), so that no consumer can read garbage
                hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((P * (Ci - ncols) + 255) / 256)), dim3(256), 0, stream, grad_x0, P, Ci, ncols);
                MP_CHECK_LAUNCH();
            }
        }
    }
    return MP_OK;
}

extern "C" int mp_sa_mlp_bwd_f32(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                 int training, const float* grad_out, const float* out, const int32_t* argk,
                                 const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                 void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    return sa_mlp_bwd(x0, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, false, nullptr);
}

extern "C" int mp_sa_mlp_bwd_gather_f32(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                        int training, const float* grad_out, const float* out, const int32_t* argk,
                                        const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                        void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    if (!gather) return MP_EINVAL;
    return sa_mlp_bwd(nullptr, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, false, nullptr, gather);
}

extern "C" int mp_sa_mlp_bwd_gather_bf16(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                         int training, const float* grad_out, const float* out, const int32_t* argk,
                                         const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                         void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    if (!gather || n_layers < 1 || !layers || layers[0].c_in != 4) return MP_EINVAL;
    return sa_mlp_bwd(nullptr, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, true, nullptr, gather);
}

extern "C" int mp_sa_mlp_bwd_gather_ex(const mp_gather_t* gather, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                       int training, const float* grad_out, const float* out, const int32_t* argk,
                                       const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                       void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream)
{
    if (!gather || n_layers < 1 || !layers || (bf16 && layers[0].c_in != 4)) return MP_EINVAL;
    return sa_mlp_bwd(nullptr, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, bf16 != 0, sync, gather);
}

extern "C" int mp_sa_mlp_bwd_bf16(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                  int training, const float* grad_out, const float* out, const int32_t* argk,
                                  const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                  void* workspace, size_t workspace_bytes, mp_stream_t stream)
{
    return sa_mlp_bwd(x0, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, true, nullptr);
}

extern "C" int mp_sa_mlp_bwd_ex(const float* x0, int64_t P, int64_t K, int n_layers, const mp_mlp_layer_t* layers,
                                int training, const float* grad_out, const float* out, const int32_t* argk,
                                const float* zmax, const mp_mlp_grads_t* grads, float* grad_x0, int64_t grad_x0_cols,
                                void* workspace, size_t workspace_bytes, int bf16, const mp_syncbn_t* sync, mp_stream_t stream)
{
    return sa_mlp_bwd(x0, P, K, n_layers, layers, training, grad_out, out, argk, zmax, grads, grad_x0, grad_x0_cols, workspace,
                      workspace_bytes, stream, bf16 != 0, sync);
}

// rows [2][B][M] int32 (M = S * K): per cloud the rows 0 .. M-1 sorted by the source point idx[b, row] (rows[0]) and that point (rows[1]) --
// what mp_sa_mlp_bwd_gather_* sorts for itself when mp_gather_t::rows is NULL.  Depends on idx alone.
extern "C" int mp_csr_rows_i64(const int64_t* idx, int64_t B, int64_t N, int64_t M, int32_t* rows, mp_stream_t stream_)
{
    if (B < 0 || N <= 0 || M < 0) return MP_EINVAL;
    if (B == 0 || M == 0) return MP_OK;
    if (!idx || !rows) return MP_EINVAL;
    if (N > 15000 || M >= ((int64_t)1 << 24) || B * M >= ((int64_t)1 << 31)) return MP_EUNSUPPORTED;
    return launch_csr_rows(idx, B, (int)N, (int)M, rows, rows + B * M, mp_stream(stream_));
}
