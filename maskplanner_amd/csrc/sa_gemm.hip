// Tiled GEMM kernels of the set-abstraction MLPs, for the layer widths outside the position-stream kernels (the group_all level:
// 4096 positions x 260 -> 256 -> 512 -> 1024; the multi-scale levels' 32-wide scale) and the factorised first layer's per-point weight
// gradient.  [r6] Split out of sa_mlp.hip (which keeps the level's control flow, the position-stream forward kernels and the small
// kernels); the dispatchers at the end are what sa_mlp.hip calls.
//
// Reference: models/pointnet2_utils.py:208-214 (`relu(bn(conv1x1(x)))` x L, max over K) and its autograd.
//   forward   Z_l = act(Z_{l-1}) W_l^T                      pos_gemm_kernel, NT form, BatchNorm sums / fused max-pool tracking in the epilogue
//   backward  G_{l-1} = dZ_l W_l                             pos_gemm_kernel, NN form, BatchNorm-backward sums of layer l-1 in the epilogue
//             dW_l = dZ_l^T act(Z_{l-1})                     dw_gemm_kernel (split over positions, fp32 atomics), dw_ci4_kernel (4 input channels)
// Operands are formed while they are staged (activation, dZ = a dy + e z + f, the pooled gradient from (argmax, pooled grad)); PREC picks the
// arithmetic: 0 fp32 MFMA, 1 bf16 operands, 3 three bf16 planes / six products (fp32-accurate, forward), 2 two planes / three products (gradients).
#include <cstdio>

#include "sa_common.h"

#ifndef MP_SPLIT2_NBUF
#define MP_SPLIT2_NBUF 1        // LDS buffers of the two-plane tiled GEMMs ([r6] 2 measured: the pooled dX product 41 -> 64 us, dW 50 -> 60 -- the second buffer costs the co-resident workgroup)
#endif

namespace {

// =================================================================================================================
// Kernel 1/2: C[M=P, N] = posop(A)[P, Kd] * Wmat   with per-column epilogue sums.
//   NT (W_KROW=false): Wmat = W^T, W row-major [N, Kd]           -> forward:   Z_l = act(Z_{l-1}) * W_l^T
//   NN (W_KROW=true) : Wmat = W,   W row-major [Kd, N]           -> backward:  G_{l-1} = dZ_l * W_l
// Epilogue sums (per block partials [gridDim.x][2][N]):
//   EPI_SQ : (sum c, sum c^2)                                  (BatchNorm forward statistics)
//   EPI_DY : with dy = relu'(zp*s+t) ? c : 0 : (sum dy, sum dy*zp)   zp = previous layer's raw Z (same shape as C)
// =================================================================================================================


// LDS layout of the two kernel bodies (one raw buffer per workgroup, carved here: bwd_pair_kernel gives both bodies the SAME buffer -- as
// static arrays inside the bodies the joint kernel would reserve their sum for every workgroup and halve the workgroups per CU)
constexpr size_t lds_al(size_t v) { return (v + 15) & ~(size_t)15; }
template <bool W_KROW, int EPI, int WAVES_M, int WAVES_N, int TM, int TN, int PREC>
struct PosLds {
    static constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    static constexpr bool SPL = PREC == 3 || PREC == 2, BF16 = PREC != 0;
    static constexpr int BK = (SPL && BM * BN <= 64 * 64) ? 64 : MP_BK;
    static constexpr int NPL = SPL ? PREC : 1;
    static constexpr int ES = BF16 ? 2 : 4;                         // bytes per LDS tile element
    static constexpr int LDA = BF16 ? BK + 8 : BK + 1;
    static constexpr int LDB = BF16 ? (W_KROW ? tr_ld(BN) : BK + 8) : (W_KROW ? BN : BK + 1);
    static constexpr int PSA = BM * LDA, PSB = W_KROW ? BK * LDB : BN * LDB;
    static constexpr int NBUF = (SPL && !(PREC == 2 && MP_SPLIT2_NBUF == 2)) ? 1 : 2;
    static constexpr bool POOL = EPI == EPI_SQ_POOL;
    static constexpr size_t OFF_A = 0;
    static constexpr size_t OFF_B = OFF_A + lds_al((size_t)NBUF * NPL * PSA * ES);
    static constexpr size_t OFF_RED = OFF_B + lds_al((size_t)NBUF * NPL * PSB * ES);
    static constexpr size_t OFF_PV = OFF_RED + lds_al((size_t)WAVES_M * 2 * BN * 4);
    static constexpr size_t OFF_PI = OFF_PV + lds_al(POOL ? (size_t)2 * (BM / 32) * BN * 4 : 16);
    static constexpr size_t BYTES = OFF_PI + lds_al(POOL ? (size_t)2 * (BM / 32) * BN * 4 : 16);
};
template <int WAVES_M, int WAVES_N, int TM, int TN, int PREC>
struct DwLds {
    static constexpr int DBK = 32, BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    static constexpr bool SPL = PREC == 3 || PREC == 2, BF16 = PREC != 0;
    static constexpr int NPL = SPL ? PREC : 1;
    static constexpr int ES = BF16 ? 2 : 4;
    static constexpr int LDA = BF16 ? tr_ld(BM) : BM, LDB = BF16 ? tr_ld(BN) : BN;
    static constexpr int PSA = DBK * LDA, PSB = DBK * LDB;
    static constexpr int NBUF = (SPL && !(PREC == 2 && MP_SPLIT2_NBUF == 2)) ? 1 : 2;
    static constexpr size_t OFF_A = 0;
    static constexpr size_t OFF_B = OFF_A + lds_al((size_t)NBUF * NPL * PSA * ES);
    static constexpr size_t OFF_T = OFF_B + lds_al((size_t)NBUF * NPL * PSB * ES);
    static constexpr size_t OFF_BN = OFF_T + lds_al((size_t)NBUF * DBK * 4 * 4);
    static constexpr size_t BYTES = OFF_BN + lds_al((size_t)3 * BM * 4);
};

template <int MODE, bool W_KROW, int EPI, int WAVES_M, int WAVES_N, int TM, int TN, int PREC = 0>   // PREC: 0 fp32 MFMA, 1 bf16, 3 split (h, m, l) planes, 2 [r6] split (h, m) planes: gradients
__device__ __forceinline__ void pos_gemm_body(const uint3 bid, PosOperand A, int P, const float* __restrict__ W, int N,
                                              int Kd, float* __restrict__ C, BnOut partials,
                                              const float* __restrict__ zprev,
                                              const float* __restrict__ sprev,
                                              const float* __restrict__ tprev, PoolOut po, int ldw,
                                              int ldc, unsigned char* lds)
{   // lds: PosLds<...>::BYTES of the workgroup's LDS.  ldw: row stride of W in the NN form (>= N: only the first N columns are produced); ldc: row stride of C
    bn_zero(partials);
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    // (these shadow the file-level constants) split planes on the 64 x 64 tile: K chunks of 64 -- the six-product chunk of 32 is over
    // before the next chunk's loads have landed, and a barrier pair per 12 MFMAs is too many
    constexpr bool SPL = PREC == 3 || PREC == 2;
    constexpr int BK = (SPL && BM * BN <= 64 * 64) ? 64 : MP_BK;
    constexpr int TPR = BK / 4, RPP = 256 / TPR, LDK = BK + 1;
    constexpr bool BF16 = PREC != 0;
    constexpr int NPL = SPL ? PREC : 1;                           // operand planes in LDS
    using TL = std::conditional_t<BF16, __bf16, float>;           // element type of the LDS tiles
    constexpr int LDA = BF16 ? BK + 8 : LDK;                      // bf16: 80-byte rows (16-byte aligned, conflict-free b128 reads)
    constexpr int LDB = BF16 ? (W_KROW ? tr_ld(BN) : BK + 8) : (W_KROW ? BN : LDK);
    constexpr int PSA = BM * LDA, PSB = W_KROW ? BK * LDB : BN * LDB;   // plane strides
    constexpr bool SUMS = (EPI == EPI_SQ || EPI == EPI_DY || EPI == EPI_SQ_POOL);
    using L = PosLds<W_KROW, EPI, WAVES_M, WAVES_N, TM, TN, PREC>;
    static_assert(L::LDA == LDA && L::LDB == LDB && L::PSA == PSA && L::PSB == PSB && L::BK == BK && L::NPL == NPL && L::ES == (int)sizeof(TL), "PosLds mirrors these constants");
    float (*pool_v)[BM / 32][EPI == EPI_SQ_POOL ? BN : 1] = reinterpret_cast<float (*)[BM / 32][EPI == EPI_SQ_POOL ? BN : 1]>(lds + L::OFF_PV);
    int (*pool_i)[BM / 32][EPI == EPI_SQ_POOL ? BN : 1] = reinterpret_cast<int (*)[BM / 32][EPI == EPI_SQ_POOL ? BN : 1]>(lds + L::OFF_PI);
    constexpr int A_PASSES = BM / RPP;                // TPR threads x float4 per row, RPP rows per pass
    constexpr int B_PASSES = W_KROW ? (BK * BN / 4 / THREADS) : (BN / RPP);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    // split planes: ONE buffer (three planes of each operand are 3x the fp32 tile's bytes; two or three workgroups per CU cover
    // each other's staging instead of a second buffer)
    constexpr int NBUF = (SPL && !(PREC == 2 && MP_SPLIT2_NBUF == 2)) ? 1 : 2;      // [r6] two planes, two buffers: measured slower, see MP_SPLIT2_NBUF
    static_assert(L::NBUF == NBUF, "PosLds mirrors NBUF");
    TL (*sA)[NPL * PSA] = reinterpret_cast<TL (*)[NPL * PSA]>(lds + L::OFF_A);
    TL (*sB)[NPL * PSB] = reinterpret_cast<TL (*)[NPL * PSB]>(lds + L::OFF_B);
    float (*red)[2][BN] = reinterpret_cast<float (*)[2][BN]>(lds + L::OFF_RED);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int wrow0 = wm * TM * 32, wcol0 = wn * TN * 32;
    const int m0 = bid.x * BM;
    const int n0 = bid.y * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    Raw4<MODE> ra[A_PASSES];
    float4 rb[B_PASSES];
    ChanConst kc;
    const int arow = tid / TPR, acol = (tid % TPR) * 4;
    auto gload = [&](int k0) {
        load_consts<MODE>(A, k0 + acol, kc);
#ifdef MP_ABLATE_LOAD
        if (k0 > 0) return;   // only the first chunk is really loaded
#endif
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) raw_load<MODE>(A, P, m0 + ps * RPP + arow, k0 + acol, ra[ps]);
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            if constexpr (W_KROW) {  // slab [BK][BN] of W[Kd, N]
                const int e = (ps * THREADS + tid) * 4;
                rb[ps] = ld4_plain(W, Kd, N, ldw, k0 + e / BN, n0 + e % BN);
            } else {                 // rows of W[N, Kd], K contiguous
                rb[ps] = ld4_plain(W, N, Kd, Kd, n0 + ps * RPP + arow, k0 + acol);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const float4 v = finish<MODE>(ra[ps], kc);
            if constexpr (SPL) {
                const Split4 sp = splitn<NPL>(v);
                const int o = (ps * RPP + arow) * LDA + acol;
                *reinterpret_cast<bf16x4*>(&sA[buf][o]) = sp.h;
                *reinterpret_cast<bf16x4*>(&sA[buf][PSA + o]) = sp.m;
                if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(&sA[buf][2 * PSA + o]) = sp.l;
            } else if constexpr (BF16) {
                *reinterpret_cast<bf16x4*>(&sA[buf][(ps * RPP + arow) * LDA + acol]) = to_bf16x4(v);
            } else {
                float* d = &sA[buf][(ps * RPP + arow) * LDA + acol];
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            if constexpr (SPL) {
                const int e = (ps * THREADS + tid) * 4;
                const int o = W_KROW ? (e / BN) * LDB + e % BN : (ps * RPP + arow) * LDB + acol;
                const Split4 sp = splitn<NPL>(rb[ps]);
                *reinterpret_cast<bf16x4*>(&sB[buf][o]) = sp.h;
                *reinterpret_cast<bf16x4*>(&sB[buf][PSB + o]) = sp.m;
                if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(&sB[buf][2 * PSB + o]) = sp.l;
            } else if constexpr (BF16) {
                if constexpr (W_KROW) {   // slab element e = row k, column n of the [BK][BN] slab
                    const int e = (ps * THREADS + tid) * 4;
                    *reinterpret_cast<bf16x4*>(&sB[buf][(e / BN) * LDB + e % BN]) = to_bf16x4(rb[ps]);
                } else {
                    *reinterpret_cast<bf16x4*>(&sB[buf][(ps * RPP + arow) * LDB + acol]) = to_bf16x4(rb[ps]);
                }
            } else if constexpr (W_KROW) {
                *reinterpret_cast<float4*>(&sB[buf][(ps * THREADS + tid) * 4]) = rb[ps];
            } else {
                float* d = &sB[buf][(ps * RPP + arow) * LDB + acol];
                d[0] = rb[ps].x; d[1] = rb[ps].y; d[2] = rb[ps].z; d[3] = rb[ps].w;
            }
        }
    };

    const int nchunks = (Kd + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kc_ = 0; kc_ < nchunks; ++kc_) {
        const int cur = NBUF == 1 ? 0 : (kc_ & 1);
        if (kc_ + 1 < nchunks) gload((kc_ + 1) * BK);
#ifndef MP_ABLATE_MFMA
        if constexpr (SPL) mma_chunk_split<false, W_KROW, LDA, LDB, TM, TN, BK, PSA, PSB, NPL>(sA[cur], sB[cur], wrow0, wcol0, acc);
        else if constexpr (BF16) mma_chunk_bf16<false, W_KROW, LDA, LDB, TM, TN, BK>(sA[cur], sB[cur], wrow0, wcol0, acc);
        else mma_chunk<false, W_KROW, LDA, LDB, TM, TN, BK>(sA[cur], sB[cur], wrow0, wcol0, acc);
#endif
        if constexpr (NBUF == 1) __syncthreads();      // every wave is done reading the chunk
        if (kc_ + 1 < nchunks) sstore(NBUF == 1 ? 0 : (cur ^ 1));
        __syncthreads();
    }

    // ---- epilogue: store C, per-column sums.  One uniform branch selects the unchecked body for interior tiles. ----
    const int l31 = lane & 31;
    const bool interior = (m0 + BM <= P) && (n0 + BN <= N);
    auto epilogue = [&](auto checked_tag) {
        constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = n0 + wcol0 + ni * 32 + l31;
            const bool cok = !CHECKED || col < N;
            float s1 = 0.0f, s2 = 0.0f, sp = 0.0f, tp = 0.0f;
            if constexpr (EPI == EPI_DY) {
                if (cok) { sp = sprev[col]; tp = tprev[col]; }
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int rbase = m0 + wrow0 + mi * 32;
                float zp[16];
                if constexpr (EPI == EPI_DY) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + acc_row_in_tile(r);
                        const bool ok = cok && (!CHECKED || row < P);
                        zp[r] = ok ? zprev[(size_t)((unsigned)row * (unsigned)N + (unsigned)col)] : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + acc_row_in_tile(r);
                    const bool ok = cok && (!CHECKED || row < P);
                    const float v = acc[mi][ni][r];
                    if (ok && C) C[(size_t)((unsigned)row * (unsigned)ldc + (unsigned)col)] = v;
                    if constexpr (EPI == EPI_SQ || EPI == EPI_SQ_POOL) {
                        if (ok) { s1 += v; s2 += v * v; }
                    } else if constexpr (EPI == EPI_DY) {
                        const float dy = (ok && zp[r] * sp + tp > 0.0f) ? v : 0.0f;
                        s1 += dy;
                        s2 += dy * zp[r];
                    }
                }
            }
            if constexpr (EPI == EPI_SQ_POOL) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    // registers ascend with the row for a fixed half-wave: strict compares keep the first extremum
                    float bmax = acc[mi][ni][0], bmin = acc[mi][ni][0];
                    int imax = acc_row_in_tile(0), imin = imax;
#pragma unroll
                    for (int r = 1; r < 16; ++r) {
                        const float v = acc[mi][ni][r];
                        const int ri = acc_row_in_tile(r);
                        if (v > bmax) { bmax = v; imax = ri; }
                        if (v < bmin) { bmin = v; imin = ri; }
                    }
                    const float omax = __shfl_xor(bmax, 32, 64), omin = __shfl_xor(bmin, 32, 64);
                    const int oimax = __shfl_xor(imax, 32, 64), oimin = __shfl_xor(imin, 32, 64);
                    if (omax > bmax || (omax == bmax && oimax < imax)) { bmax = omax; imax = oimax; }
                    if (omin < bmin || (omin == bmin && oimin < imin)) { bmin = omin; imin = oimin; }
                    if (lane < 32) {
                        const int tr = wrow0 / 32 + mi, cc = wcol0 + ni * 32 + lane;
                        pool_v[0][tr][cc] = bmax; pool_i[0][tr][cc] = imax;
                        pool_v[1][tr][cc] = bmin; pool_i[1][tr][cc] = imin;
                    }
                }
            }
            if constexpr (SUMS) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    red[wm][0][wcol0 + ni * 32 + lane] = s1;
                    red[wm][1][wcol0 + ni * 32 + lane] = s2;
                }
            }
        }
    };
#ifdef MP_ABLATE_EPI
    if (acc[0][0][0] == 12345.678f) epilogue(std::true_type{});   // keeps acc alive, never taken
#else
    if (interior) epilogue(std::false_type{}); else epilogue(std::true_type{});
#endif
    if constexpr (SUMS) {
        __syncthreads();
        for (int e = tid; e < 2 * BN; e += THREADS) {
            const int st = e / BN, c = e - st * BN;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) v += red[w][st][c];
            if (n0 + c < N) {
                if (partials.slots) atomicAdd(partials.slots + ((size_t)(bid.x & (BN_NS - 1)) * 2 + st) * N + n0 + c, (double)v);
                else partials.rows[((size_t)bid.x * 2 + st) * N + n0 + c] = v;
            }
        }
    }
    if constexpr (EPI == EPI_SQ_POOL) {
        // combine the 32-row tiles of each group (ascending rows; strict compares keep the first extremum)
        const int tpg = po.K / 32;            // tiles per group: 1, 2 or 4
        const int groups = BM / po.K;
        for (int e = tid; e < groups * BN; e += THREADS) {
            const int gl = e / BN, c = e - gl * BN;
            const int t0 = gl * tpg;
            float bmax = pool_v[0][t0][c], bmin = pool_v[1][t0][c];
            int imax = pool_i[0][t0][c], imin = pool_i[1][t0][c];
            for (int t = 1; t < tpg; ++t) {
                const float vx = pool_v[0][t0 + t][c], vn = pool_v[1][t0 + t][c];
                if (vx > bmax) { bmax = vx; imax = t * 32 + pool_i[0][t0 + t][c]; }
                if (vn < bmin) { bmin = vn; imin = t * 32 + pool_i[1][t0 + t][c]; }
            }
            const int grow = m0 / po.K + gl;
            if ((grow + 1) * po.K <= P && n0 + c < N) {
                const size_t o = (size_t)((unsigned)grow * (unsigned)N + (unsigned)(n0 + c));
                po.vmax[o] = bmax; po.imax[o] = imax;
                po.vmin[o] = bmin; po.imin[o] = imin;
            }
        }
    }
}

// [r6] bid: the tile's (row block, column block) -- blockIdx of the kernel below, or the position inside bwd_pair_kernel's joint grid
template <int MODE, bool W_KROW, int EPI, int WAVES_M, int WAVES_N, int TM, int TN, int PREC = 0>
__global__ __launch_bounds__(THREADS) void pos_gemm_kernel(PosOperand A, int P, const float* __restrict__ W, int N, int Kd, float* __restrict__ C,
                                                           BnOut partials, const float* __restrict__ zprev, const float* __restrict__ sprev,
                                                           const float* __restrict__ tprev, PoolOut po, int ldw, int ldc)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[PosLds<W_KROW, EPI, WAVES_M, WAVES_N, TM, TN, PREC>::BYTES];
    pos_gemm_body<MODE, W_KROW, EPI, WAVES_M, WAVES_N, TM, TN, PREC>(make_uint3(blockIdx.x, blockIdx.y, 0), A, P, W, N, Kd, C, partials, zprev, sprev, tprev, po, ldw,
                                                                      ldc, lds);
}

// =================================================================================================================
// Kernel 3: dW[Co, Ci] += sum_p dZ[p, Co] * act(Zin)[p, Ci]   (split over P, fp32 atomics)
//   both operands are positions-major slabs [BK positions][channels] -> LDS [k][row] layout, straight copies.
// =================================================================================================================
template <int MODE_DZ, int MODE_IN, int WAVES_M, int WAVES_N, int TM, int TN, int PREC = 0>
__device__ __forceinline__ void dw_gemm_body(const uint3 bid, const unsigned gdz, PosOperand DZ, PosOperand IN, int P, int p_per_block,
                                             float* __restrict__ dW, int ci_base, int tail_ci, unsigned char* lds)
{   // lds: DwLds<...>::BYTES of the workgroup's LDS.  tail_ci >= 0: the 4 input channels [tail_ci, tail_ci + 4) (132 = 128 + 4, 260 = 256 + 4: the centred xyz + pad of a
    // grouped input) are handled by the workgroups of the LAST column tile with plain FMAs on the staged dZ tile, instead of
    // a second launch that would stream dZ from HBM again for a 97 % empty MFMA tile
    constexpr int DBK = 32;                 // positions per K chunk
    constexpr int BM = WAVES_M * TM * 32;   // output channels (rows of dW)
    constexpr int BN = WAVES_N * TN * 32;   // input channels  (cols of dW)
    constexpr int PA = DBK * BM / 4 / THREADS;
    constexpr int PB = DBK * BN / 4 / THREADS;
    static_assert(WAVES_M * WAVES_N == 4 && PA >= 1 && PB >= 1, "tile");
    static_assert(BM == 128, "the tail-column path maps 256 threads onto 128 rows x 2 column pairs");
    constexpr bool BF16 = PREC != 0;
    constexpr bool SPL = PREC == 3 || PREC == 2;       // [r6] 2: the two-plane form (h, m; three products), see mma_chunk_split
    constexpr int NPL = SPL ? PREC : 1;
    using TL = std::conditional_t<BF16, __bf16, float>;
    constexpr int LDA = BF16 ? tr_ld(BM) : BM, LDB = BF16 ? tr_ld(BN) : BN;   // bf16: [k][row] tiles read through ds_read_b64_tr_b16
    constexpr int PSA = DBK * LDA, PSB = DBK * LDB;
    constexpr int NBUF = (SPL && !(PREC == 2 && MP_SPLIT2_NBUF == 2)) ? 1 : 2;       // split planes: one buffer, see pos_gemm_kernel
    using L = DwLds<WAVES_M, WAVES_N, TM, TN, PREC>;
    static_assert(L::LDA == LDA && L::LDB == LDB && L::PSA == PSA && L::PSB == PSB && L::NBUF == NBUF && L::NPL == NPL && L::ES == (int)sizeof(TL), "DwLds mirrors these constants");
    TL (*sA)[NPL * PSA] = reinterpret_cast<TL (*)[NPL * PSA]>(lds + L::OFF_A);
    TL (*sB)[NPL * PSB] = reinterpret_cast<TL (*)[NPL * PSB]>(lds + L::OFF_B);
    float (*sT)[DBK * 4] = reinterpret_cast<float (*)[DBK * 4]>(lds + L::OFF_T);
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wrow0 = (wave / WAVES_N) * TM * 32, wcol0 = (wave % WAVES_N) * TN * 32;
    const int co0 = bid.y * BM, ci0 = ci_base + bid.z * BN;
    float* bn_lds = reinterpret_cast<float*>(lds + L::OFF_BN);          // (a, e, f) of this workgroup's BM output channels
    bn_prologue(DZ.bn, bn_lds, BM, co0, BM, bid.x == 0 && bid.z == 0);
    const bool do_tail = tail_ci >= 0 && bid.z == gdz - 1;
    float tacc0 = 0.0f, tacc1 = 0.0f;
    ChanConst kt;
    Raw4<MODE_IN> rt;
    if (do_tail) load_consts<MODE_IN>(IN, tail_ci, kt);
    const int p0 = bid.x * p_per_block;
    const int p1 = min(P, p0 + p_per_block);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // K runs over positions here, so every thread keeps the SAME channels for the whole kernel
    const int ca = (tid * 4) % BM, cb = (tid * 4) % BN;
    const int ka0 = (tid * 4) / BM, kb0 = (tid * 4) / BN;       // first slab row of this thread
    constexpr int KA_STEP = THREADS * 4 / BM, KB_STEP = THREADS * 4 / BN;
    ChanConst ka, kb;
    load_consts<MODE_DZ>(DZ, co0 + ca, ka, bn_lds, BM, co0);
    load_consts<MODE_IN>(IN, ci0 + cb, kb);

    Raw4<MODE_DZ> ra[PA];
    Raw4<MODE_IN> rb[PB];
    auto gload = [&](int pk) {
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) raw_load<MODE_DZ>(DZ, p1, pk + ka0 + ps * KA_STEP, co0 + ca, ra[ps]);
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) raw_load<MODE_IN>(IN, p1, pk + kb0 + ps * KB_STEP, ci0 + cb, rb[ps]);
        if (do_tail && tid < DBK) raw_load<MODE_IN>(IN, p1, pk + tid, tail_ci, rt);
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            if constexpr (SPL) {
                const Split4 sp = splitn<NPL>(finish<MODE_DZ>(ra[ps], ka));
                const int o = (ka0 + ps * KA_STEP) * LDA + ca;
                *reinterpret_cast<bf16x4*>(&sA[buf][o]) = sp.h;
                *reinterpret_cast<bf16x4*>(&sA[buf][PSA + o]) = sp.m;
                if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(&sA[buf][2 * PSA + o]) = sp.l;
            }
            else if constexpr (BF16) *reinterpret_cast<bf16x4*>(&sA[buf][(ka0 + ps * KA_STEP) * LDA + ca]) = to_bf16x4(finish<MODE_DZ>(ra[ps], ka));
            else *reinterpret_cast<float4*>(&sA[buf][(ps * THREADS + tid) * 4]) = finish<MODE_DZ>(ra[ps], ka);
        }
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) {
            if constexpr (SPL) {
                const Split4 sp = splitn<NPL>(finish<MODE_IN>(rb[ps], kb));
                const int o = (kb0 + ps * KB_STEP) * LDB + cb;
                *reinterpret_cast<bf16x4*>(&sB[buf][o]) = sp.h;
                *reinterpret_cast<bf16x4*>(&sB[buf][PSB + o]) = sp.m;
                if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(&sB[buf][2 * PSB + o]) = sp.l;
            }
            else if constexpr (BF16) *reinterpret_cast<bf16x4*>(&sB[buf][(kb0 + ps * KB_STEP) * LDB + cb]) = to_bf16x4(finish<MODE_IN>(rb[ps], kb));
            else *reinterpret_cast<float4*>(&sB[buf][(ps * THREADS + tid) * 4]) = finish<MODE_IN>(rb[ps], kb);
        }
        if (do_tail && tid < DBK) {
            float4 v = finish<MODE_IN>(rt, kt);
            if constexpr (PREC == 1) { const bf16x4 h = to_bf16x4(v); v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]); }
            *reinterpret_cast<float4*>(&sT[buf][tid * 4]) = v;
        }
    };
    const int nchunks = (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;
    gload(p0);
    sstore(0);
    __syncthreads();
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = NBUF == 1 ? 0 : (kc & 1);
        if (kc + 1 < nchunks) gload(p0 + (kc + 1) * DBK);
        if constexpr (SPL) mma_chunk_split<true, true, LDA, LDB, TM, TN, DBK, PSA, PSB, NPL>(sA[cur], sB[cur], wrow0, wcol0, acc);
        else if constexpr (BF16) mma_chunk_bf16<true, true, LDA, LDB, TM, TN, DBK>(sA[cur], sB[cur], wrow0, wcol0, acc);
        else mma_chunk<true, true, LDA, LDB, TM, TN, DBK>(sA[cur], sB[cur], wrow0, wcol0, acc);
        if (do_tail) {   // thread = (output channel tid & 127, column pair tid >> 7); bf16: the rounded dZ, fp32 coordinates
            const TL* a = sA[cur] + (tid & (BM - 1));
            const float* t = sT[cur] + 2 * (tid >> 7);
#pragma unroll
            for (int k = 0; k < DBK; ++k) {
                float av = (float)a[k * LDA];
                if constexpr (PREC == 3) av = (av + (float)a[PSA + k * LDA]) + (float)a[2 * PSA + k * LDA];   // h + m + l: the fp32 dZ again
                if constexpr (PREC == 2) av = av + (float)a[PSA + k * LDA];                                    // h + m
                tacc0 = __builtin_fmaf(av, t[k * 4], tacc0);
                tacc1 = __builtin_fmaf(av, t[k * 4 + 1], tacc1);
            }
        }
        if constexpr (NBUF == 1) __syncthreads();
        if (kc + 1 < nchunks) sstore(NBUF == 1 ? 0 : (cur ^ 1));
        __syncthreads();
    }
    const int l31 = lane & 31;
    const int Co = DZ.C, Ci = IN.C;
    if (do_tail) {
        const int row = co0 + (tid & (BM - 1)), col = tail_ci + 2 * (tid >> 7);
        if (row < Co) {
            if (col < Ci) atomicAdd(dW + (size_t)((unsigned)row * (unsigned)Ci + (unsigned)col), tacc0);
            if (col + 1 < Ci) atomicAdd(dW + (size_t)((unsigned)row * (unsigned)Ci + (unsigned)(col + 1)), tacc1);
        }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = ci0 + wcol0 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + wrow0 + mi * 32 + acc_row_in_tile(r);
                if (row < Co && col < Ci) atomicAdd(dW + (size_t)((unsigned)row * (unsigned)Ci + (unsigned)col), acc[mi][ni][r]);
            }
        }
}

template <int MODE_DZ, int MODE_IN, int WAVES_M, int WAVES_N, int TM, int TN, int PREC = 0>
__global__ __launch_bounds__(THREADS) void dw_gemm_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block, float* __restrict__ dW, int ci_base, int tail_ci)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[DwLds<WAVES_M, WAVES_N, TM, TN, PREC>::BYTES];
    dw_gemm_body<MODE_DZ, MODE_IN, WAVES_M, WAVES_N, TM, TN, PREC>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), gridDim.z, DZ, IN, P, p_per_block, dW, ci_base,
                                                                   tail_ci, lds);
}

// [r6] G_{l-1} = dZ_l W_l and dW_l = dZ_l^T act(Z_{l-1}) of one group_all layer as ONE launch: the 64 x 64 tiles of the first and the
// 128 x 128 tiles of the second alternate in a joint 1-D grid.  Neither product fills the chip at 4096 positions (0.3-1.5 waves per SIMD,
// r5's counters) and both read the same dZ_l; the dW product feeds nothing but the optimizer.  As two launches on one stream they ran
// one after the other; on two streams the fork / join of the recorded graph cost more than the overlap returned (experiments/r6_fork_dw.patch).
// Two planes (PREC 2) only; the constants of dZ_l come from a finalize launch in front (both halves read the global arrays).
template <int MODE_DZ, int MODE_IN, int EPI>
__global__ __launch_bounds__(THREADS) void bwd_pair_kernel(PosOperand DZ, PosOperand IN, int P, const float* __restrict__ W, int N, int Kd,
                                                           float* __restrict__ G, BnOut partials, const float* __restrict__ zprev,
                                                           const float* __restrict__ sprev, const float* __restrict__ tprev, int ldw, int ldc, int gm, int gn,
                                                           int ppb, float* __restrict__ dW, int tail_ci, int wx, int wy, int wz)
{
    constexpr size_t LB = PosLds<true, EPI, 2, 2, 1, 1, 2>::BYTES > DwLds<2, 2, 2, 2, 2>::BYTES ? PosLds<true, EPI, 2, 2, 1, 1, 2>::BYTES : DwLds<2, 2, 2, 2, 2>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LB];          // ONE buffer for whichever body this workgroup runs
    const unsigned n_dx = (unsigned)(gm * gn), n_dw = (unsigned)(wx * wy * wz);
    const unsigned lo = n_dx < n_dw ? n_dx : n_dw;          // the first 2 * lo blocks alternate, the longer list's rest follows
    unsigned b = blockIdx.x;
    bool dx;
    if (b < 2 * lo) { dx = (b & 1) == 0; b >>= 1; }
    else { dx = n_dx > n_dw; b -= lo; }
    if (dx) {
        pos_gemm_body<MODE_DZ, true, EPI, 2, 2, 1, 1, 2>(make_uint3(b % (unsigned)gm, b / (unsigned)gm, 0), DZ, P, W, N, Kd, G, partials, zprev, sprev, tprev, PoolOut{},
                                                         ldw, ldc, lds);
    } else {
        const unsigned x = b % (unsigned)wx, yz = b / (unsigned)wx;
        dw_gemm_body<MODE_DZ, MODE_IN, 2, 2, 2, 2, 2>(make_uint3(x, yz % (unsigned)wy, yz / (unsigned)wy), (unsigned)wz, DZ, IN, P, ppb, dW, 0, tail_ci, lds);
    }
}

template <int MODE_DZ, int MODE_IN>
__global__ __launch_bounds__(256) void dw_ci4_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block, float* __restrict__ dW, int r16, int h16)
{   // r16: the bf16 variant -- dZ and the input rows rounded to bf16 before the products; h16: the dZ operand's Z / G stored as bf16
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * 1024];          // (Co <= 1024: launch_dw's condition for this kernel)
    bn_prologue(DZ.bn, bn_lds, 1024, 0, DZ.C, blockIdx.x == 0);
    __shared__ float red[256][16 + 1];
    const int tid = threadIdx.x;
    const int Co = DZ.C;
    const int nq = Co / 4;                        // channel quads (Co % 4 == 0): 16 for Co = 64
    const int q = tid % nq, pl = tid / nq;        // this thread's quad and position lane
    const int PL = 256 / nq;                      // position lanes per workgroup
    const int p0 = blockIdx.x * p_per_block, p1 = min(P, p0 + p_per_block);
    ChanConst ka, kb;
    load_consts<MODE_DZ>(DZ, 4 * q, ka, bn_lds, 1024);
    load_consts<MODE_IN>(IN, 0, kb);
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    if (pl < PL) {
        for (int p = p0 + pl; p < p1; p += 4 * PL) {
            Raw4<MODE_DZ> rz[4];
            Raw4<MODE_IN> rx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // four positions in flight
                if (h16) raw_load<MODE_DZ, true>(DZ, p1, p + u * PL, 4 * q, rz[u]); else raw_load<MODE_DZ, false>(DZ, p1, p + u * PL, 4 * q, rz[u]);
                raw_load<MODE_IN>(IN, p1, p + u * PL, 0, rx[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 dz = rb16(finish<MODE_DZ>(rz[u], ka), r16);
                const float4 xv = rb16(finish<MODE_IN>(rx[u], kb), r16);
                const float d[4] = {dz.x, dz.y, dz.z, dz.w}, x[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fmaf(d[a], x[b], acc[a][b]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) red[tid][4 * a + b] = acc[a][b];
    __syncthreads();
    // element e = (quad, a, b): sum over the position lanes, fixed order
    for (int e = tid; e < nq * 16; e += 256) {
        const int qq = e / 16, ab = e - qq * 16;
        float s = 0.0f;
        for (int l = 0; l < PL; ++l) s += red[l * nq + qq][ab];
        const int co = 4 * qq + ab / 4, ci = ab & 3;
        if (ci < IN.C) atomicAdd(dW + (size_t)co * IN.C + ci, s);
    }
}

template <int MODE_DZ, int MODE_IN, int PREC = 0>
int launch_dw(const PosOperand& DZ, const PosOperand& IN, int64_t P64, float* dW, hipStream_t stream)
{
    const int Co = DZ.C, Ci = IN.C;
    const int P = (int)P64;
    // positions per workgroup: 1024, fewer when that leaves the chip short of workgroups (group_all: P = 4096).  Every
    // workgroup ends with one fp32 atomic per dW element of its tile, so slices are not made smaller than needed for
    // ~512 workgroups in all (32 output tiles x 128 slices of 128 positions spent more time in atomics than in MFMA).
    const unsigned gy = (Co + 127) / 128;
    int ppb = 1024;
    {
        const int64_t tiles = (int64_t)gy * ((Ci + 127) / 128);
        static const int wgs = [] { const char* e = getenv("MP_DW_WGS"); return e ? atoi(e) : 512; }();      // (timing aid)
        const int64_t want = (wgs + tiles - 1) / tiles;                     // slices wanted
        while ((P + ppb - 1) / ppb < want && ppb > 128) ppb >>= 1;
    }
    const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
    // input-channel tiling: 128-wide tiles, a narrow remainder (132 = 128 + 4, 260 = 2*128 + 4) gets 32-wide tiles so
    // that it does not pay for a whole 128-column MFMA tile of zeros
    if (Ci == 4 && (Co & 3) == 0 && Co <= 1024 && 256 % (Co / 4) == 0) {      // (PREC == 1: the same kernel with both operands rounded to bf16)
        double fl = 2.0 * (double)P * Co * Ci, by = 4.0 * ((MODE_DZ == SRC_DZ ? 2.0 : 1.0) * (double)P * Co + (double)P * Ci);
        char tg[64];
        snprintf(tg, sizeof tg, "dw_ci4_kernel<%d, %d>", MODE_DZ, MODE_IN);
        MP_LAUNCH(tg, fl, by, (dw_ci4_kernel<MODE_DZ, MODE_IN>), dim3((unsigned)((P + 1023) / 1024)), dim3(256), 0, stream, DZ, IN, P, 1024, dW, PREC == 1 ? 1 : 0, 0);
        MP_CHECK_LAUNCH();
        return MP_OK;
    }
    const int main_ci = (Ci > 128 && Ci % 128 != 0 && Ci % 128 <= 32) ? (Ci / 128) * 128 : Ci;
    const int tail_ci = (main_ci < Ci && Ci - main_ci == 4) ? main_ci : -1;   // 4 leftover columns ride along (see the kernel)
    char tag[96];
    auto work = [&](int cols, double& flops, double& bytes) {
        flops = 2.0 * (double)P * Co * cols;
        bytes = 4.0 * ((MODE_DZ == SRC_DZ ? 2.0 : 1.0) * (double)P * Co + (double)P * cols + (double)Co * cols);
    };
    double flops, bytes;
    work(tail_ci >= 0 ? Ci : main_ci, flops, bytes);
    const char* kn = PREC == 1 ? "dw_gemm_bf16_kernel" : (PREC == 3 ? "dw_gemm_split_kernel" : (PREC == 2 ? "dw_gemm_split2_kernel" : "dw_gemm_kernel"));
    if (main_ci <= 32) {
        snprintf(tag, sizeof tag, "%s<%d, %d, 4, 1, 1, 1>", kn, MODE_DZ, MODE_IN);
        MP_LAUNCH(tag, flops, bytes, (dw_gemm_kernel<MODE_DZ, MODE_IN, 4, 1, 1, 1, PREC>), dim3(gx, gy, (main_ci + 31) / 32), dim3(THREADS), 0, stream, DZ, IN, P, ppb, dW, 0, tail_ci);
    } else if (main_ci <= 64) {
        snprintf(tag, sizeof tag, "%s<%d, %d, 4, 1, 1, 2>", kn, MODE_DZ, MODE_IN);
        MP_LAUNCH(tag, flops, bytes, (dw_gemm_kernel<MODE_DZ, MODE_IN, 4, 1, 1, 2, PREC>), dim3(gx, gy, (main_ci + 63) / 64), dim3(THREADS), 0, stream, DZ, IN, P, ppb, dW, 0, tail_ci);
    } else {
        snprintf(tag, sizeof tag, "%s<%d, %d, 2, 2, 2, 2>", kn, MODE_DZ, MODE_IN);
        MP_LAUNCH(tag, flops, bytes, (dw_gemm_kernel<MODE_DZ, MODE_IN, 2, 2, 2, 2, PREC>), dim3(gx, gy, (main_ci + 127) / 128), dim3(THREADS), 0, stream, DZ, IN, P, ppb, dW, 0, tail_ci);
    }
    MP_CHECK_LAUNCH();
    if (main_ci < Ci && tail_ci < 0) {
        work(Ci - main_ci, flops, bytes);
        snprintf(tag, sizeof tag, "%s<%d, %d, 4, 1, 1, 1>", kn, MODE_DZ, MODE_IN);
        PosOperand DZ2 = DZ;
        DZ2.bn = BnSite{};       // (the launch above derived the constants: its prologue consumed the slot rows)
        MP_LAUNCH(tag, flops, bytes, (dw_gemm_kernel<MODE_DZ, MODE_IN, 4, 1, 1, 1, PREC>), dim3(gx, gy, (Ci - main_ci + 31) / 32), dim3(THREADS), 0, stream, DZ2, IN, P, ppb, dW, main_ci, -1);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}

template <int MODE, bool W_KROW, int EPI, int PREC = 0>
int launch_pos_gemm(const PosOperand& A, int64_t P, const float* W, int N, int Kd, float* C, BnOut partials,
                    const float* zprev, const float* sprev, const float* tprev, hipStream_t stream, int* nblk_out,
                    PoolOut po = PoolOut{}, int ldw = 0, int ldc = 0)
{
    if (ldw == 0) ldw = N;
    if (ldc == 0) ldc = N;
    // algorithmic work of one launch: 2*P*N*Kd flops; bytes = operand(s) read once + result written once + weights
    const double flops = 2.0 * (double)P * N * Kd;
    const double rd = (MODE == SRC_DZ ? 2.0 : 1.0) * (double)P * Kd + (EPI == EPI_DY ? (double)P * N : 0.0);
    const double bytes = 4.0 * (rd + (C ? (double)P * N : 0.0) + (double)N * Kd);
    // Tile shape: 128x128 (2x2 waves of 64x64) by default, 128x64 for N <= 64.  A launch needs well over 256 workgroups
    // to fill the chip: when the default grid is smaller (the group_all level: P = 4096 => 32 row tiles) narrower, then
    // lower tiles are used.  The fused max-pool epilogue needs whole groups inside a row tile, so it keeps 128 rows.
    const int64_t t128 = ((P + 127) / 128) * ((N + 127) / 128);
    const int64_t t128x64 = ((P + 127) / 128) * ((N + 63) / 64);
    int shape = (N <= 64) ? 1 : 0;                       // 0: 128x128, 1: 128x64, 2: 64x64
    if (shape == 0 && t128 < 384) shape = (t128x64 >= 384 || EPI == EPI_SQ_POOL) ? 1 : 2;
    if (shape == 1 && N > 64 && EPI != EPI_SQ_POOL && t128x64 < 384) shape = 2;
    // (split planes, [r2]: 64 x 64 tiles on all CUs run at 66-80 TFLOP/s on the group_all level; 128 x 128 tiles on half of them were
    // slower: 55 -> 93 us, one workgroup's K loop alone does not cover its load latency)
    char tag[96];
    const char* kn = PREC == 1 ? "pos_gemm_bf16_kernel" : (PREC == 3 ? "pos_gemm_split_kernel" : (PREC == 2 ? "pos_gemm_split2_kernel" : "pos_gemm_kernel"));
    if (shape == 1) {
        const unsigned gm = (unsigned)((P + 127) / 128);
        if (nblk_out) *nblk_out = (int)gm;
        snprintf(tag, sizeof tag, "%s<%d, %s, %d, 4, 1, 1, 2>", kn, MODE, W_KROW ? "true" : "false", EPI);
        MP_LAUNCH(tag, flops, bytes, (pos_gemm_kernel<MODE, W_KROW, EPI, 4, 1, 1, 2, PREC>), dim3(gm, (N + 63) / 64),
                  dim3(THREADS), 0, stream, A, (int)P, W, N, Kd, C, partials, zprev, sprev, tprev, po, ldw, ldc);
    } else if (shape == 2) {
        if constexpr (EPI != EPI_SQ_POOL) {
            const unsigned gm = (unsigned)((P + 63) / 64);
            if (nblk_out) *nblk_out = (int)gm;
            snprintf(tag, sizeof tag, "%s<%d, %s, %d, 2, 2, 1, 1>", kn, MODE, W_KROW ? "true" : "false", EPI);
            MP_LAUNCH(tag, flops, bytes, (pos_gemm_kernel<MODE, W_KROW, EPI, 2, 2, 1, 1, PREC>), dim3(gm, (N + 63) / 64),
                      dim3(THREADS), 0, stream, A, (int)P, W, N, Kd, C, partials, zprev, sprev, tprev, po, ldw, ldc);
        }
    } else {
        const unsigned gm = (unsigned)((P + 127) / 128);
        if (nblk_out) *nblk_out = (int)gm;
        snprintf(tag, sizeof tag, "%s<%d, %s, %d, 2, 2, 2, 2>", kn, MODE, W_KROW ? "true" : "false", EPI);
        MP_LAUNCH(tag, flops, bytes, (pos_gemm_kernel<MODE, W_KROW, EPI, 2, 2, 2, 2, PREC>), dim3(gm, (N + 127) / 128),
                  dim3(THREADS), 0, stream, A, (int)P, W, N, Kd, C, partials, zprev, sprev, tprev, po, ldw, ldc);
    }
    MP_CHECK_LAUNCH();
    return MP_OK;
}

}  // namespace

// ---- what sa_mlp.hip calls (operands / partials / pool outputs by address: PosOperand, BnOut, PoolOut of sa_common.h) ----------------
int mp_pos_gemm_launch(int mode, int w_krow, int epi, int prec, const void* a_, int64_t P, const float* W, int N, int Kd, float* C, const void* partials_,
                       const float* zprev, const float* sprev, const float* tprev, hipStream_t stream, int* nblk_out, const void* po_, int ldw, int ldc)
{
    const PosOperand& A = *static_cast<const PosOperand*>(a_);
    const BnOut partials = partials_ ? *static_cast<const BnOut*>(partials_) : BnOut{nullptr, nullptr, nullptr, 0, nullptr, 0};
    const PoolOut po = po_ ? *static_cast<const PoolOut*>(po_) : PoolOut{};
#define MP_PG(MODE_, KROW_, EPI_, P2_)                                                                                                  \
    if (mode == MODE_ && (w_krow != 0) == KROW_ && epi == EPI_) {                                                                        \
        switch (prec) {                                                                                                                  \
            case 0: return launch_pos_gemm<MODE_, KROW_, EPI_, 0>(A, P, W, N, Kd, C, partials, zprev, sprev, tprev, stream, nblk_out, po, ldw, ldc); \
            case 1: return launch_pos_gemm<MODE_, KROW_, EPI_, 1>(A, P, W, N, Kd, C, partials, zprev, sprev, tprev, stream, nblk_out, po, ldw, ldc); \
            case 2: return launch_pos_gemm<MODE_, KROW_, EPI_, P2_>(A, P, W, N, Kd, C, partials, zprev, sprev, tprev, stream, nblk_out, po, ldw, ldc); \
            case 3: return launch_pos_gemm<MODE_, KROW_, EPI_, 3>(A, P, W, N, Kd, C, partials, zprev, sprev, tprev, stream, nblk_out, po, ldw, ldc); \
            default: return MP_EINVAL;                                                                                                   \
        }                                                                                                                                \
    }
    // forward (two planes are a gradient form: a forward call that asks for them gets three)
    MP_PG(SRC_ID, false, EPI_SQ_POOL, 3)
    MP_PG(SRC_ACT, false, EPI_SQ_POOL, 3)
    MP_PG(SRC_ID, false, EPI_SQ, 3)
    MP_PG(SRC_ACT, false, EPI_SQ, 3)
    // backward
    MP_PG(SRC_DZ_POOLED, true, EPI_DY, 2)
    MP_PG(SRC_DZ, true, EPI_DY, 2)
    MP_PG(SRC_DZ_POOLED, true, EPI_NONE, 2)
    MP_PG(SRC_DZ, true, EPI_NONE, 2)
#undef MP_PG
    return MP_EUNSUPPORTED;
}

int mp_dw_gemm_launch(int mode_dz, int mode_in, int prec, const void* dz_, const void* in_, int64_t P, float* dW, hipStream_t stream)
{
    const PosOperand& DZ = *static_cast<const PosOperand*>(dz_);
    const PosOperand& IN = *static_cast<const PosOperand*>(in_);
#define MP_DG(MDZ_, MIN_)                                                              \
    if (mode_dz == MDZ_ && mode_in == MIN_) {                                          \
        switch (prec) {                                                                \
            case 0: return launch_dw<MDZ_, MIN_, 0>(DZ, IN, P, dW, stream);            \
            case 1: return launch_dw<MDZ_, MIN_, 1>(DZ, IN, P, dW, stream);            \
            case 2: return launch_dw<MDZ_, MIN_, 2>(DZ, IN, P, dW, stream);            \
            case 3: return launch_dw<MDZ_, MIN_, 3>(DZ, IN, P, dW, stream);            \
            default: return MP_EINVAL;                                                 \
        }                                                                              \
    }
    MP_DG(SRC_DZ_POOLED, SRC_ID)
    MP_DG(SRC_DZ_POOLED, SRC_ACT)
    MP_DG(SRC_DZ, SRC_ID)
    MP_DG(SRC_DZ, SRC_ACT)
    MP_DG(SRC_ID, SRC_ID)
#undef MP_DG
    return MP_EUNSUPPORTED;
}

// [r6] dX and dW of one layer as one launch (bwd_pair_kernel): returns 1 when it launched, 0 when the shapes are not the ones the joint kernel is
// built for (the caller then launches the two products separately), < 0 on error.  Built for: two planes, the 64 x 64 tiling of the dX product
// and the 128 x 128 tiling of the dW product without a remainder launch -- the group_all level.  DZ.bn must already be settled.
int mp_bwd_pair_launch(int mode_dz, int mode_in, int epi, const void* dz_, const void* in_, int64_t P64, const float* W, int N, int Kd, float* G,
                       const void* partials_, const float* zprev, const float* sprev, const float* tprev, int ldw, int ldc, float* dW, hipStream_t stream,
                       int* nblk_out, int probe)
{   // probe != 0: only answer whether the joint kernel applies (DZ.bn is ignored: the caller settles it before the real call)
    static const int level = [] { const char* e = getenv("MP_BWD_PAIR"); return e ? atoi(e) : 2; }();      // 0: never, 1: the dense layers only, 2 (default): the pooled layer too
    if (level == 0) return 0;
    const PosOperand& DZ = *static_cast<const PosOperand*>(dz_);
    const PosOperand& IN = *static_cast<const PosOperand*>(in_);
    const BnOut partials = partials_ ? *static_cast<const BnOut*>(partials_) : BnOut{nullptr, nullptr, nullptr, 0, nullptr, 0};
    if ((!probe && DZ.bn.slots != nullptr) || P64 <= 0 || P64 >= ((int64_t)1 << 31)) return 0;
    if (!((mode_dz == SRC_DZ || (mode_dz == SRC_DZ_POOLED && level >= 2)) && ((mode_in == SRC_ACT && epi == EPI_DY) || (mode_in == SRC_ID && epi == EPI_NONE)))) return 0;
    const int P = (int)P64;
    if (ldw == 0) ldw = N;
    if (ldc == 0) ldc = N;
    // the dX product's tiling (launch_pos_gemm): only its 64 x 64 shape
    {
        const int64_t t128 = ((P64 + 127) / 128) * ((N + 127) / 128), t128x64 = ((P64 + 127) / 128) * ((N + 63) / 64);
        int shape = (N <= 64) ? 1 : 0;
        if (shape == 0 && t128 < 384) shape = (t128x64 >= 384) ? 1 : 2;
        if (shape == 1 && N > 64 && t128x64 < 384) shape = 2;
        if (shape != 2) return 0;
    }
    const int gm = (int)((P64 + 63) / 64), gn = (N + 63) / 64;
    // the dW product's tiling (launch_dw): only 128 x 128 tiles, the 4 leftover columns of a 4k + 4 input riding along
    const int Co = DZ.C, Ci = IN.C;
    if (Co != Kd || Ci == 4) return 0;
    const int main_ci = (Ci > 128 && Ci % 128 != 0 && Ci % 128 <= 32) ? (Ci / 128) * 128 : Ci;
    const int tail_ci = (main_ci < Ci && Ci - main_ci == 4) ? main_ci : -1;
    if (main_ci <= 64 || (main_ci < Ci && tail_ci < 0)) return 0;
    const int wy = (Co + 127) / 128, wz = (main_ci + 127) / 128;
    int ppb = 1024;
    {
        const int64_t tiles = (int64_t)wy * ((Ci + 127) / 128);
        const int64_t want = (512 + tiles - 1) / tiles;
        while ((P + ppb - 1) / ppb < want && ppb > 128) ppb >>= 1;
    }
    const int wx = (P + ppb - 1) / ppb;
    if (probe) return 1;
    if (nblk_out) *nblk_out = gm;
    const double dzr = (mode_dz == SRC_DZ ? 2.0 : 1.0) * (double)P * Co;
    const double flops = 2.0 * (double)P * N * Kd + 2.0 * (double)P * Co * Ci;
    const double bytes = 4.0 * (2.0 * dzr + (epi == EPI_DY ? (double)P * N : 0.0) + (G ? (double)P * N : 0.0) + (double)N * Kd + (double)P * Ci + (double)Co * Ci);
    const dim3 grid((unsigned)(gm * gn + wx * wy * wz));
    char tag[96];
#define MP_PAIR(MDZ_, MIN_, EPI_)                                                                                                              \
    if (mode_dz == MDZ_ && mode_in == MIN_ && epi == EPI_) {                                                                                    \
        snprintf(tag, sizeof tag, "bwd_pair_kernel<%d, %d, %d>", MDZ_, MIN_, EPI_);                                                             \
        MP_LAUNCH(tag, flops, bytes, (bwd_pair_kernel<MDZ_, MIN_, EPI_>), grid, dim3(THREADS), 0, stream, DZ, IN, P, W, N, Kd, G, partials, zprev, sprev, \
                  tprev, ldw, ldc, gm, gn, ppb, dW, tail_ci, wx, wy, wz);                                                                       \
        MP_CHECK_LAUNCH();                                                                                                                      \
        return 1;                                                                                                                               \
    }
    // (same box, alternating, the group_all level: dense layers 30-35 us against 28 + 23 and 27-29 against 27 + 15, the pooled layer 80-84 against
    // 49 + 42 -- with ONE LDS buffer for whichever body a workgroup runs; as static arrays of both bodies the joint kernel reserved their sum,
    // halved the workgroups per CU and took 130 us for the pooled layer)
    MP_PAIR(SRC_DZ, SRC_ACT, EPI_DY)
    MP_PAIR(SRC_DZ, SRC_ID, EPI_NONE)
    MP_PAIR(SRC_DZ_POOLED, SRC_ACT, EPI_DY)
#undef MP_PAIR
    return 0;
}

// the first layer of a level whose input is the 4-channel coordinate rows and is recomputed, not stored (dZ_0 from (Z_0 recomputed, G_0))
int mp_dw_ci4_rc_launch(const void* dz_, const void* in_, int64_t P, float* dW, int r16, int h16, double flops, double bytes, hipStream_t stream)
{
    const PosOperand& DZ = *static_cast<const PosOperand*>(dz_);
    const PosOperand& IN = *static_cast<const PosOperand*>(in_);
    MP_LAUNCH("dw_ci4_kernel<5, 0>", flops, bytes, (dw_ci4_kernel<SRC_DZ_RC, SRC_ID>), dim3((unsigned)((P + 1023) / 1024)), dim3(256), 0, stream, DZ, IN,
              (int)P, 1024, dW, r16, h16);
    MP_CHECK_LAUNCH();
    return MP_OK;
}
