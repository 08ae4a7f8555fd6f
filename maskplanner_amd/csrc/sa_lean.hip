// "Lean" pooled layer of a set-abstraction level: the last layer of the shared MLP (the one under the max-pool) forward WITHOUT
// storing its raw activation Z_L and backward WITHOUT reading it -- and without ever forming its dense dZ_L.
//
// Reference: models/pointnet2_utils.py:208-214 (conv1x1 -> BatchNorm2d -> ReLU -> max over K) and autograd's mirror image.
//
// Backward of the pooled layer L (CO outputs, CI inputs, A = act(Z_{L-1}) its activated input, W its weight):
//     dZ_L[p, c] = a_c * gm[p, c] + e_c * z[p, c] + f_c                       (BatchNorm backward folded into a, e, f;  gm = pooled
//                                                                               gradient at the group's arg-max member, 0 elsewhere)
// is one nonzero per (group, channel) plus a term that is LINEAR in z = A W^T.  With any shift m~ (here: the mean of relu of a
// Gaussian with the previous BatchNorm's shift and scale -- it only has to be near the column means), Ac = A - m~:
//     G_{L-1} = dZ_L W       = Ac M + v~ + sparse           M = W^T diag(e) W [CI, CI],  v~ = W^T f~,  f~ = f + e .* (W m~)
//                                                            sparse[p, :] = sum_{c : argmax(g(p), c) = p} a_c gp[g, c] W[c, :]
//     dW_L    = dZ_L^T A     = a .* S + e .* (W Gp) + f~ (x) colsum(A)        S[c, :]  = sum_g gp[g, c] A[argmax(g, c), :]
//                                                            Gp[k', k] = (Ac^T Ac)[k', k] + m~_k colsum(Ac)[k']
// Every dense product left has contraction / output width CI instead of CO (half the matrix-core work for the 128 -> 256 and
// 64 -> 128 layers), Z_L [P, CO] is neither written by the forward pass nor read here (-2 x 268 MB per level at the bench shape),
// and the kernel stages ONE operand (Ac as three bf16 planes) instead of two.  The shift keeps the reformulation as accurate as the
// direct one: z - mean(z) is formed from centred inputs, so e * z + f never cancels two large numbers (tests/test_gpu_split.py,
// tests/test_gpu_arbiter.py hold it against fp64).
//
// bwd_lean_kernel: a workgroup walks ppb positions in chunks of 16, 64-position blocks of 4 chunks:
//   * staging: raw z chunk -> act -> Ac -> (h, m, l) planes in the K-packed LDS image of sa_mlp.hip's fused backward, act and raw z
//     as fp32 beside them;
//   * Gram += Ac^T Ac   (32x32x16 bf16 MFMA, K = positions, six plane products, registers for the whole workgroup);
//   * G chunk = Ac M    (16x16x32 bf16 MFMA, M planes in registers), + v~ + the chunk's sparse rows, BatchNorm-backward sums of
//     layer L-1 in the epilogue;
//   * S: thread (channel c, column part) adds gp * act row of the group's arg-max member when that member is in the chunk;
//   * sparse rows of the NEXT chunk: each wave takes the (group, channel) entries assigned to it whose member lies in that chunk
//     and adds coef * W[c, :] (a coalesced row of an L2-resident table) into an LDS tile with ds_add_f32.
// Gram, S and colsum(Ac) leave as per-workgroup partial tiles (plain stores, no atomics); lean_reduce / lean_dw finish dW_L --
// bit-reproducible, unlike the atomics of the direct kernels.
#include <cstdio>

#include "common.h"
#include "sa_lean.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// x = h + m + l with three bf16 numbers (sa_mlp.hip: split3)
struct Split4 { bf16x4 h, m, l; };
__device__ __forceinline__ Split4 split3(const float4& v)
{
    Split4 r;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        r.h[i] = h;
        r.m[i] = m;
        r.l[i] = (__bf16)r2;
    }
    return r;
}

// K-packed plane image (sa_mlp.hip: tr_frag_packed): element (row, c) of a [rows][C] chunk at (c / 8) * GS + row * 8 + c % 8
template <int GS>
__device__ __forceinline__ bf16x8 tr_frag_packed(const __bf16* tile, int k0, int c0)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = i >> 2, pp = i & 3, nh = (lane >> 4) & 1, h = lane >> 5;
    const __bf16* p = tile + ((c0 >> 3) + 2 * nh + (pp >> 1)) * GS + (k0 + 8 * h + q) * 8 + 4 * (pp & 1);
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 4 * 8));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

#ifndef MP_LEAN_DX_WAVES
#define MP_LEAN_DX_WAVES 4      // minimum waves per SIMD of lean_dx_kernel: two 512-thread (four 256-thread) workgroups per CU, 128 registers
#endif

__device__ __forceinline__ int acc_row_in_tile(int r) { return (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }

struct LeanArgs {
    const float* z1;        // raw Z_{L-1} [P, CI]
    const float* s1;        // its BatchNorm affine: act = relu(z * s1 + t1)
    const float* t1;
    const float* mt;        // m~ [CI]
    const float* Mm;        // M [CI, CI] row-major (M[k'][k])
    const float* vt;        // v~ [CI]
    const float* W;         // [CO, CI]
    const float* a;         // a [CO]
    const float* gp;        // pooled gradient, relu-masked [G, CO]
    const int* argk;        // arg-max member inside the group [G, CO]
    float* G1;              // out: gradient w.r.t. act(Z_{L-1}) [P, CI]
    float* partials;        // BatchNorm-backward partial sums of layer L-1 [nblk][2][CI]
    float* wgS;             // per-workgroup S [nblk][CO * CI]
    float* wgGram;          // per-workgroup Ac^T Ac [nblk][CI * CI]
    float* wgCs;            // per-workgroup colsum(Ac) [nblk][CI]
    int P, K, ppb;          // ppb: positions per workgroup of lean_dx_kernel
    int ppb2;               // ... of lean_dws_kernel (its per-workgroup partial tiles make fewer, larger workgroups cheaper)
    int skip;               // timing experiments (MP_LEAN_SKIP bit mask): 1 sparse rows, 2 S, 4 Gram, 8 G product
};

// GPB: groups per 64-position block (K >= 64: 1, K = 32: 2)
// ---- kernel 1: G_{L-1} = Ac M + v~ + sparse rows, BatchNorm-backward sums of layer L-1 ------------------------------------------------
template <int CO, int CI, int GPB>
__global__ __launch_bounds__(CI == 128 ? 512 : 256, MP_LEAN_DX_WAVES) void lean_dx_kernel(LeanArgs A)
{
    constexpr int NT = CI == 128 ? 512 : 256, NW = NT / 64;
    constexpr int DBK = 16, GS = DBK * 8 + 32;
    constexpr int NST = CI / 32;                 // k-steps of the G product (16x16x32)
    constexpr int NBB = CI / 64;
    constexpr int NE = CO * GPB;                 // (group, channel) entries per block
    constexpr int RPW = DBK / NW, LOG_RPW = RPW == 2 ? 1 : 2;     // sparse rows of a chunk owned by one wave
    static_assert(NE <= NT && NE % 64 == 0 && NW * 16 == CI && (RPW == 2 || RPW == 4), "shape");
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][3][(CI / 8) * GS];
    __shared__ __attribute__((aligned(16))) float sZ[2][DBK * CI];
    __shared__ __attribute__((aligned(16))) float sSp[2][DBK * CI];
    __shared__ int eRib[2][NE];
    __shared__ float eCoef[2][NE];
    __shared__ int hitE[NW][128];             // per wave: the entries of the chunk whose rows it owns (drained above 64)
    __shared__ float red[2][CI];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, kq = lane >> 4;
    const int K = A.K;
    const int p0 = blockIdx.x * A.ppb;
    const int p1 = min(A.P, p0 + A.ppb);
    const int nchunks = (p1 - p0) / DBK;         // P and ppb are multiples of 64
    if (nchunks <= 0) return;
    const int nblocks = nchunks / 4;

    // staging: one float4 per thread and chunk (mapping of the fused backward: conflict-free K-packed plane writes)
    const int cq = (lane & 7) + 8 * (lane >> 5), pr = (lane >> 3) & 3;
    const int cb = (wave % NBB) * 64 + 4 * cq;
    const int kb0 = (wave / NBB) * 4 + pr;
    const float4 ks = ld4(A.s1 + cb), kt = ld4(A.t1 + cb), km = ld4(A.mt + cb);
    // G tile of this wave: columns xcol0 .. +15; M planes of those columns in registers: lane (col, kq) holds M[32 st + 8 kq .. + 7][col]
    const int xcol0 = wave * 16;
    bf16x8 wsp[NST][3];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const float* mp_ = A.Mm + (size_t)(32 * st + 8 * kq) * CI + xcol0 + l15;
        const Split4 lo = split3(make_float4(mp_[0], mp_[CI], mp_[2 * CI], mp_[3 * CI]));
        const Split4 hi = split3(make_float4(mp_[4 * CI], mp_[5 * CI], mp_[6 * CI], mp_[7 * CI]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wsp[st][0][i] = lo.h[i]; wsp[st][0][4 + i] = hi.h[i];
            wsp[st][1][i] = lo.m[i]; wsp[st][1][4 + i] = hi.m[i];
            wsp[st][2][i] = lo.l[i]; wsp[st][2][4 + i] = hi.l[i];
        }
    }
    const int ecol = xcol0 + l15;
    const float spx = A.s1[ecol], tpx = A.t1[ecol], vtx = A.vt[ecol];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 sx1 = {0.0f, 0.0f}, sx2 = {0.0f, 0.0f};
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(A.G1 + (size_t)p0 * CI, 0, (p1 - p0) * CI * 4, 0x00020000);
    int goff = ((4 * kq) * CI + ecol) * 4;

    // entries of a block: thread q < NE owns entry q = gi * CO + c: (row of the arg-max member inside the block or -1, a_c * gp)
    const int ec = tid % CO, egi = tid / CO;
    const float a_c = A.a[ec];
    int raw_k = 0;
    float raw_g = 0.0f;
    auto load_entry = [&](int bb) {               // raw loads only: nothing here waits for them
        if (tid < NE && bb < nblocks) {
            const size_t o = (size_t)((p0 + 64 * bb) / K + egi) * CO + ec;
            raw_k = A.argk[o];
            raw_g = A.gp[o];
        }
    };
    auto write_entry = [&](int bb) {
        if (tid < NE) {
            const int pos = p0 + 64 * bb;
            const int rowbase = pos - (pos / K) * K;       // != 0 only for K > 64 (a block is a part of one group)
            const int r = egi * K + raw_k - rowbase;
            eRib[bb & 1][tid] = (bb < nblocks && r >= 0 && r < 64) ? r : -1;
            eCoef[bb & 1][tid] = a_c * raw_g;
        }
    };
    // sparse rows of chunk nx: this wave owns rows wave * RPW .. + RPW - 1 of the chunk and sums coef * W[c, :] over the entries whose
    // member is one of them, in entry order (registers; a plain store at the end: no atomics, bit-reproducible)
    auto sparse_rows = [&](int nx) {
        if (nx >= nchunks) return;
        const int par = (nx >> 2) & 1, want = (nx & 3) * NW + wave;
        float acc[RPW][NBB];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int h = 0; h < NBB; ++h) acc[r][h] = 0.0f;
        if (!(A.skip & 1)) {
            // this wave's hits of the entry table, compacted into its own list (entry order kept; entry | row slot << 16), and
            // consumed in rounds of RND W rows in flight; the list is drained whenever it holds more than 64 entries
            constexpr int RND = 8 / NBB;
            auto drain = [&](int n) {
                for (int i0 = 0; i0 < n; i0 += RND) {
                    float wv[RND][NBB], cfu[RND];
                    int rsu[RND];
#pragma unroll
                    for (int u = 0; u < RND; ++u) {
                        const bool live = i0 + u < n;
                        const int ent = __builtin_amdgcn_readfirstlane(hitE[wave][live ? i0 + u : 0]);
                        const int q = ent & 0xffff;
                        const float cfl = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, eCoef[par][q])));
                        cfu[u] = live ? cfl : 0.0f;
                        rsu[u] = ent >> 16;
                        const int c = q % CO;
#pragma unroll
                        for (int h = 0; h < NBB; ++h) wv[u][h] = A.W[(size_t)c * CI + lane + 64 * h];
                    }
#pragma unroll
                    for (int u = 0; u < RND; ++u)
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            const float cr = rsu[u] == r ? cfu[u] : 0.0f;       // (wave-uniform select)
#pragma unroll
                            for (int h = 0; h < NBB; ++h) acc[r][h] = __builtin_fmaf(cr, wv[u][h], acc[r][h]);
                        }
                }
            };
            int n = 0;
            for (int e0 = 0; e0 < NE; e0 += 64) {
                const int rib = eRib[par][e0 + lane];
                const bool hit = (rib >> LOG_RPW) == want;
                const unsigned long long mask = __ballot(hit);
                if (hit) hitE[wave][n + mp::prefix_popc(mask)] = (e0 + lane) | ((rib & (RPW - 1)) << 16);
                n += __builtin_popcountll(mask);
                if (n > 64) { drain(n); n = 0; }
            }
            drain(n);
        }
        float* tile = sSp[nx & 1] + (wave * RPW) * CI + lane;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int h = 0; h < NBB; ++h) tile[r * CI + 64 * h] = acc[r][h];
    };

    float4 rz;
    auto gload = [&](int kc) { rz = ld4(A.z1 + (size_t)(p0 + kc * DBK + kb0) * CI + cb); };
    auto sstore = [&](int buf) {
        float4 ac;
        ac.x = fmaxf(rz.x * ks.x + kt.x, 0.0f) - km.x; ac.y = fmaxf(rz.y * ks.y + kt.y, 0.0f) - km.y;
        ac.z = fmaxf(rz.z * ks.z + kt.z, 0.0f) - km.z; ac.w = fmaxf(rz.w * ks.w + kt.w, 0.0f) - km.w;
        const Split4 sp = split3(ac);
        const int oh = (cb >> 3) * GS + kb0 * 8 + (cb & 7);
        *reinterpret_cast<bf16x4*>(&hB[buf][0][oh]) = sp.h;
        *reinterpret_cast<bf16x4*>(&hB[buf][1][oh]) = sp.m;
        *reinterpret_cast<bf16x4*>(&hB[buf][2][oh]) = sp.l;
        *reinterpret_cast<float4*>(&sZ[buf][kb0 * CI + cb]) = rz;
    };

    // ---- prologue: entries of block 0, first chunk staged, its sparse rows
    load_entry(0);
    write_entry(0);
    gload(0);
    sstore(0);
    load_entry(1);
    __syncthreads();
    sparse_rows(0);
    __syncthreads();

    for (int it = 0; it < nchunks; ++it) {
        const int cur_b = it & 1, j = it & 3, b = it >> 2;
        // sparse rows of the NEXT chunk first: the loads it waits for are then older than the prefetch below
        sparse_rows(it + 1);
        if (j == 1) write_entry(b + 1);               // (loaded at j == 0 of this block, or in the prologue)
        if (j == 0 && it > 0) load_entry(b + 1);
        if (it + 1 < nchunks) gload(it + 1);
        // ---- G chunk [16 x 16 of this wave] = Ac M
        f32x4 ax = {0.f, 0.f, 0.f, 0.f}, cx = {0.f, 0.f, 0.f, 0.f};
        if (!(A.skip & 8)) {
            const int ao = kq * GS + l15 * 8;
            bf16x8 af[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(&hB[cur_b][pl][ao]);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) af[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(&hB[cur_b][pl][ao + 4 * (st + 1) * GS]);
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
        // ---- epilogue: + v~ + sparse rows, store, BatchNorm-backward sums of layer L-1
        {
            const float* sp = &sSp[cur_b][(4 * kq) * CI + ecol];
            const float* zr = &sZ[cur_b][(4 * kq) * CI + ecol];
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const float g0 = ax[i] + (vtx + sp[i * CI]), g1 = ax[i + 1] + (vtx + sp[(i + 1) * CI]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g0), grsrc, goff, i * CI * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g1), grsrc, goff, (i + 1) * CI * 4, 0);
                const f2 zp = {zr[i * CI], zr[(i + 1) * CI]};
                const f2 y = zp * f2{spx, spx} + f2{tpx, tpx};
                const f2 dy = {y.x > 0.0f ? g0 : 0.0f, y.y > 0.0f ? g1 : 0.0f};
                sx1 += dy;
                sx2 += dy * zp;
            }
            goff += DBK * CI * 4;
        }
        if (it + 1 < nchunks) sstore(cur_b ^ 1);
        __syncthreads();
    }

    // ---- BatchNorm-backward partial sums of layer L-1
    {
        float s1x = sx1.x + sx1.y, s2x = sx2.x + sx2.y;
        s1x += __shfl_xor(s1x, 16, 64); s1x += __shfl_xor(s1x, 32, 64);
        s2x += __shfl_xor(s2x, 16, 64); s2x += __shfl_xor(s2x, 32, 64);
        if (lane < 16) { red[0][ecol] = s1x; red[1][ecol] = s2x; }
    }
    __syncthreads();
    for (int e = tid; e < 2 * CI; e += NT) A.partials[(size_t)blockIdx.x * 2 * CI + e] = (&red[0][0])[e];
}

// ---- kernel 2: Gram = Ac^T Ac, S (gather of the arg-max members' act rows), colsum(Ac) as per-workgroup partial tiles -------------
template <int CO, int CI, int GPB>
__global__ __launch_bounds__(CI == 128 ? 512 : 256, CI == 128 ? 1 : 2) void lean_dws_kernel(LeanArgs A)
{
    constexpr int NT = CI == 128 ? 512 : 256, NW = NT / 64;
    constexpr int DBK = 16, GS = DBK * 8 + 32;
    constexpr int TNW = CI / 64;                 // Gram tiles per wave (one tile row, TNW tile columns): NW = 2 * (CI / 32) waves
    constexpr int SP = NT / CO, SW = CI / SP;    // S: threads per channel, columns per thread
    constexpr int LDACT = CI + 4;                // fp32 act rows: 16 distinct rows hit 16 distinct 4-bank slots
    constexpr int NBB = CI / 64;
    static_assert(CO <= NT && NT % CO == 0 && SW % 4 == 0 && NW == 2 * (CI / 32), "shape");
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][3][(CI / 8) * GS];
    __shared__ __attribute__((aligned(16))) float sAct[2][DBK * LDACT];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31;
    const int K = A.K;
    const int p0 = blockIdx.x * A.ppb2;
    const int p1 = min(A.P, p0 + A.ppb2);
    const int nchunks = (p1 - p0) / DBK;
    if (nchunks <= 0) return;
    const int nblocks = nchunks / 4;

    const int cq = (lane & 7) + 8 * (lane >> 5), pr = (lane >> 3) & 3;
    const int cb = (wave % NBB) * 64 + 4 * cq;
    const int kb0 = (wave / NBB) * 4 + pr;
    const float4 ks = ld4(A.s1 + cb), kt = ld4(A.t1 + cb), km = ld4(A.mt + cb);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int grow0 = (wave >> 1) * 32, gcol0 = (wave & 1) * TNW * 32;
    f32x16 accG[TNW];
#pragma unroll
    for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) accG[ni][r] = 0.0f;
    // S: channel sc, columns scol0 .. + SW
    const int sc = tid % CO, spart = tid / CO, scol0 = spart * SW;
    float accS[SW];
#pragma unroll
    for (int j = 0; j < SW; ++j) accS[j] = 0.0f;
    // (group, channel) pairs of the current block: rib = row of the arg-max member inside the 64-position block, or -1
    int cur_rib[GPB], raw_k[GPB];
    float cur_g[GPB], raw_g[GPB];
    auto load_pairs = [&](int bb) {               // raw loads only
        if (bb >= nblocks) return;
        const int grp0 = (p0 + 64 * bb) / K;
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi) {
            const size_t o = (size_t)(grp0 + gi) * CO + sc;
            raw_k[gi] = A.argk[o];
            raw_g[gi] = A.gp[o];
        }
    };
    auto take_pairs = [&](int bb) {
        const int pos = p0 + 64 * bb;
        const int rowbase = pos - (pos / K) * K;
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi) {
            const int r = gi * K + raw_k[gi] - rowbase;
            cur_rib[gi] = (bb < nblocks && r >= 0 && r < 64) ? r : -1;
            cur_g[gi] = raw_g[gi];
        }
    };
    auto gload = [&](int kc, float4& rz) { if (kc < nchunks) rz = ld4(A.z1 + (size_t)(p0 + kc * DBK + kb0) * CI + cb); };
    auto sstore = [&](int buf, const float4& rz) {
        float4 act, ac;
        act.x = fmaxf(rz.x * ks.x + kt.x, 0.0f); act.y = fmaxf(rz.y * ks.y + kt.y, 0.0f);
        act.z = fmaxf(rz.z * ks.z + kt.z, 0.0f); act.w = fmaxf(rz.w * ks.w + kt.w, 0.0f);
        ac.x = act.x - km.x; ac.y = act.y - km.y; ac.z = act.z - km.z; ac.w = act.w - km.w;
        csum.x += ac.x; csum.y += ac.y; csum.z += ac.z; csum.w += ac.w;
        const Split4 sp = split3(ac);
        const int oh = (cb >> 3) * GS + kb0 * 8 + (cb & 7);
        *reinterpret_cast<bf16x4*>(&hB[buf][0][oh]) = sp.h;
        *reinterpret_cast<bf16x4*>(&hB[buf][1][oh]) = sp.m;
        *reinterpret_cast<bf16x4*>(&hB[buf][2][oh]) = sp.l;
        *reinterpret_cast<float4*>(&sAct[buf][kb0 * LDACT + cb]) = act;
    };
    // one chunk: LDS buffer it & 1 holds chunk it, rx holds chunk it + 1 (in flight since the iteration before), ry is free: the
    // load of chunk it + 2 goes there -- every z load has two iterations to arrive
    auto step = [&](int it, float4& rx, float4& ry) {
        const int cur_b = it & 1, j = it & 3, b = it >> 2;
        gload(it + 2, ry);
        if (j == 0) load_pairs(b + 1);
        if (!(A.skip & 4)) {
            bf16x8 fb[3][TNW], fa;
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni) fb[0][ni] = tr_frag_packed<GS>(hB[cur_b][0], 0, gcol0 + ni * 32);
            fa = tr_frag_packed<GS>(hB[cur_b][2], 0, grow0);
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni) accG[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[0][ni], accG[ni], 0, 0, 0);
            fa = tr_frag_packed<GS>(hB[cur_b][0], 0, grow0);
#pragma unroll
            for (int pl = 2; pl >= 1; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) fb[pl][ni] = tr_frag_packed<GS>(hB[cur_b][pl], 0, gcol0 + ni * 32);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accG[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[pl][ni], accG[ni], 0, 0, 0);
            fa = tr_frag_packed<GS>(hB[cur_b][1], 0, grow0);
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accG[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[pl][ni], accG[ni], 0, 0, 0);
        }
        // S += gp * act row of the arg-max member, when that member is one of this chunk's 16 rows
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi) {
            const int rib = cur_rib[gi];
            if ((rib >> 4) == j && !(A.skip & 2)) {
                const float g = cur_g[gi];
                const float* ar = &sAct[cur_b][(rib & 15) * LDACT + scol0];
#pragma unroll
                for (int q4 = 0; q4 < SW / 4; ++q4) {
                    const float4 v = *reinterpret_cast<const float4*>(ar + 4 * q4);
                    accS[4 * q4 + 0] = __builtin_fmaf(g, v.x, accS[4 * q4 + 0]);
                    accS[4 * q4 + 1] = __builtin_fmaf(g, v.y, accS[4 * q4 + 1]);
                    accS[4 * q4 + 2] = __builtin_fmaf(g, v.z, accS[4 * q4 + 2]);
                    accS[4 * q4 + 3] = __builtin_fmaf(g, v.w, accS[4 * q4 + 3]);
                }
            }
        }
        if (it + 1 < nchunks) sstore(cur_b ^ 1, rx);
        if (j == 3) take_pairs(b + 1);
        __syncthreads();
    };

    float4 rza, rzb;
    load_pairs(0);
    take_pairs(0);
    gload(0, rza);
    gload(1, rzb);
    sstore(0, rza);
    __syncthreads();
    for (int it = 0; it < nchunks; it += 2) {     // (nchunks is a multiple of 4)
        step(it, rzb, rza);
        step(it + 1, rza, rzb);
    }

    // ---- colsum(Ac): the threads that staged the same channels (every row slot) through LDS
    {
        float* cs = &sAct[0][0];        // (free now) [16 row slots][CI]
        *reinterpret_cast<float4*>(&cs[kb0 * CI + cb]) = csum;
        __syncthreads();
        for (int c = tid; c < CI; c += NT) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < DBK; ++r) s += cs[r * CI + c];
            A.wgCs[(size_t)blockIdx.x * CI + c] = s;
        }
    }
    // ---- per-workgroup partial tiles
    float* gt = A.wgGram + (size_t)blockIdx.x * CI * CI;
#pragma unroll
    for (int ni = 0; ni < TNW; ++ni) {
        const int col = gcol0 + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) gt[(size_t)(grow0 + acc_row_in_tile(r)) * CI + col] = accG[ni][r];
    }
    float* st_ = A.wgS + (size_t)blockIdx.x * CO * CI + (size_t)sc * CI + scol0;
#pragma unroll
    for (int q4 = 0; q4 < SW / 4; ++q4)
        *reinterpret_cast<float4*>(st_ + 4 * q4) = make_float4(accS[4 * q4], accS[4 * q4 + 1], accS[4 * q4 + 2], accS[4 * q4 + 3]);
}

// m~ (mean of relu of a Gaussian with the previous BatchNorm's shift beta and scale |gamma|), mu~ = W m~, f~ = f + e mu~, v~ = W^T f~,
// M = W^T diag(e) W; copies of a, e, f~ for the finishing kernel (the caller's constant buffers are reused by the next layer).
// grid: CI + 1 workgroups of 256 threads: workgroup k' < CI computes row k' of M, workgroup CI the vectors.
__global__ __launch_bounds__(256) void lean_prep_kernel(const float* __restrict__ W, int CO, int CI, const float* __restrict__ a,
                                                        const float* __restrict__ e, const float* __restrict__ f,
                                                        const float* __restrict__ gamma1, const float* __restrict__ beta1,
                                                        float* __restrict__ mt, float* __restrict__ Mm, float* __restrict__ vt,
                                                        float* __restrict__ ka, float* __restrict__ ke, float* __restrict__ kf)
{
    __shared__ float s_m[256], s_x[1024];
    __shared__ float s_part[4][256];
    const int tid = threadIdx.x;
    const int kp = blockIdx.x;
    if (kp < CI) {
        // x_c = W[c, k'] * e_c, then M[k', k] = sum_c x_c W[c, k]: thread (k, slice of the channels), four slices
        for (int c = tid; c < CO; c += 256) s_x[c] = W[(size_t)c * CI + kp] * e[c];
        __syncthreads();
        const int per = 256 / 4;                       // threads per slice: CI <= 64 * ... handled by the k loop below
        const int sl = tid / per, k0 = tid % per;
        for (int kb = 0; kb < CI; kb += per) {
            const int k = kb + k0;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < CI) {
                const int c0 = sl * (CO / 4), c1 = c0 + CO / 4;
                for (int c = c0; c < c1; c += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(s_x[c + u], W[(size_t)(c + u) * CI + k], acc[u]);
                }
            }
            s_part[sl][k0] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
            if (sl == 0 && k < CI) Mm[(size_t)kp * CI + k] = (s_part[0][k0] + s_part[1][k0]) + (s_part[2][k0] + s_part[3][k0]);
            __syncthreads();
        }
        return;
    }
    for (int k = tid; k < CI; k += 256) {
        const float mu = beta1[k], sg = fabsf(gamma1[k]);
        float m = fmaxf(mu, 0.0f);
        if (sg > 1e-12f) {
            const float r = mu / sg;
            m = sg * 0.3989422804f * __expf(-0.5f * r * r) + mu * 0.5f * (1.0f + erff(r * 0.7071067812f));
        }
        s_m[k] = m;
        mt[k] = m;
    }
    __syncthreads();
    // mu~_c = W[c, :] . m~ -> f~_c: a thread per channel (rows of an L2-resident table), four independent chains
    for (int c = tid; c < CO; c += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float4* wr = reinterpret_cast<const float4*>(W + (size_t)c * CI);
        for (int k = 0; k < CI; k += 4) {
            const float4 w4 = wr[k >> 2];
            acc[0] = __builtin_fmaf(w4.x, s_m[k], acc[0]); acc[1] = __builtin_fmaf(w4.y, s_m[k + 1], acc[1]);
            acc[2] = __builtin_fmaf(w4.z, s_m[k + 2], acc[2]); acc[3] = __builtin_fmaf(w4.w, s_m[k + 3], acc[3]);
        }
        const float ec = e[c];
        const float ft = f[c] + ec * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        s_x[c] = ft;
        ka[c] = a[c]; ke[c] = ec; kf[c] = ft;
    }
    __syncthreads();
    {   // v~[k] = sum_c f~_c W[c, k]
        const int per = 256 / 4;
        const int sl = tid / per, k0 = tid % per;
        for (int kb = 0; kb < CI; kb += per) {
            const int k = kb + k0;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < CI) {
                const int c0 = sl * (CO / 4), c1 = c0 + CO / 4;
                for (int c = c0; c < c1; c += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(s_x[c + u], W[(size_t)(c + u) * CI + k], acc[u]);
                }
            }
            s_part[sl][k0] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
            if (sl == 0 && k < CI) vt[k] = (s_part[0][k0] + s_part[1][k0]) + (s_part[2][k0] + s_part[3][k0]);
            __syncthreads();
        }
    }
}

// Sum of per-workgroup partial tiles: out[e] = sum_w part[w][e], a workgroup of 256 threads = 16 outputs x 16 slices of the w range
// (every load of a thread in flight at once: a sequential walk over 256 partial tiles costs 256 memory latencies), fixed summation
// order => bit-reproducible.  Shared by the two finishing kernels below.
__device__ __forceinline__ float sum_partials16(const float* __restrict__ part, int nblk, size_t stride, size_t e0, int n_out, float (&s_red)[16][17])
{
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int per = (nblk + 15) / 16;
    const int w0 = sl * per, w1 = min(nblk, w0 + per);
    float acc = 0.0f;
    if ((int)o < n_out) {
        float v[16];
        for (int wb = w0; wb < w1; wb += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (wb + u < w1) ? part[(size_t)(wb + u) * stride + e0 + o] : 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    }
    s_red[sl][o] = acc;
    __syncthreads();
    float tot = 0.0f;
    if (sl == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += s_red[q][o];
    }
    return tot;       // valid in threads < 16
}

// Gp[k', k] = sum_wg Gram_wg[k', k] + m~_k * cs[k'],  cs[k'] = sum_wg colsum_wg[k']
// grid: CI * CI / 16 workgroups (16 consecutive outputs of one row k' each) + CI / 16 workgroups for cs... cs first:
// cs is needed by every Gram output, so each workgroup sums the colsum of ITS row k' itself (256 values, one per thread).
__global__ __launch_bounds__(256) void lean_reduce_kernel(const float* __restrict__ wgGram, const float* __restrict__ wgCs, int nblk, int CI,
                                                          const float* __restrict__ mt, float* __restrict__ Gp, float* __restrict__ cs)
{
    __shared__ float s_red[16][17];
    __shared__ float s_c[256];
    const size_t e0 = (size_t)blockIdx.x * 16;
    const int kp = (int)(e0 / CI), k0 = (int)(e0 - (size_t)kp * CI);
    float c = 0.0f;
    for (int w = threadIdx.x; w < nblk; w += 256) c += wgCs[(size_t)w * CI + kp];
    s_c[threadIdx.x] = c;
    const float g = sum_partials16(wgGram, nblk, (size_t)CI * CI, e0, 16, s_red);      // (its barrier also publishes s_c)
    if (threadIdx.x < 16) {
        float ctot = 0.0f;
        for (int q = 0; q < 256; ++q) ctot += s_c[q];
        Gp[e0 + threadIdx.x] = g + mt[k0 + threadIdx.x] * ctot;
        if (k0 == 0 && threadIdx.x == 0) cs[kp] = ctot;
    }
}

// dW[c, k] = a_c * sum_wg S_wg[c, k] + e_c * sum_k' W[c, k'] Gp[k', k] + f~_c * (cs[k] + P m~_k)
// grid: CO * CI / 16 workgroups of 256 threads (16 consecutive k of one weight row c)
__global__ __launch_bounds__(256) void lean_finish_kernel(const float* __restrict__ wgS, int nblk, int CO, int CI, const float* __restrict__ W,
                                                          const float* __restrict__ Gp, const float* __restrict__ cs, const float* __restrict__ mt,
                                                          const float* __restrict__ ka, const float* __restrict__ ke, const float* __restrict__ kf,
                                                          float P, float* __restrict__ dW)
{
    __shared__ float s_red[16][17];
    __shared__ float s_wg[16][17];
    const size_t e0 = (size_t)blockIdx.x * 16;
    const int c = (int)(e0 / CI), k0 = (int)(e0 - (size_t)c * CI);
    // (W Gp)[c, k0 + o]: 16 outputs x 16 slices of k'
    {
        const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int per = CI / 16;
        float acc = 0.0f;
        for (int q = sl * per; q < (sl + 1) * per; ++q) acc = __builtin_fmaf(W[(size_t)c * CI + q], Gp[(size_t)q * CI + k0 + o], acc);
        s_wg[sl][o] = acc;
    }
    const float S = sum_partials16(wgS, nblk, (size_t)CO * CI, e0, 16, s_red);
    if (threadIdx.x < 16) {
        const int o = threadIdx.x;
        float wg = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) wg += s_wg[q][o];
        dW[e0 + o] = ka[c] * S + (ke[c] * wg + kf[c] * (cs[k0 + o] + P * mt[k0 + o]));
    }
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace mp {

bool lean_enabled()
{
    static const bool on = getenv("MP_LEAN_LAST") && atoi(getenv("MP_LEAN_LAST")) != 0;     // (opt-in while the kernels are being tuned)
    return on;
}

// positions per workgroup: lean_dx_kernel wants >= 2 workgroups per CU (they hide each other's latencies), lean_dws_kernel one per CU
// (every workgroup leaves CO * CI + CI * CI partial sums behind)
int lean_ppb(int64_t P, int which)
{
    static const int env1 = getenv("MP_LEAN_PPB") ? atoi(getenv("MP_LEAN_PPB")) : 512;
    static const int env2 = getenv("MP_LEAN_PPB2") ? atoi(getenv("MP_LEAN_PPB2")) : 1024;
    int ppb = which == 1 ? env1 : env2;
    if (ppb < 64 || (ppb % 64)) ppb = which == 1 ? 512 : 1024;
    const int64_t want = which == 1 ? 512 : 256;
    while (ppb > 64 && (P + ppb - 1) / ppb < want) ppb >>= 1;
    return ppb;
}

bool lean_supported(int64_t P, int64_t K, int64_t CO, int64_t CI)
{
    if (!lean_enabled()) return false;
    if (P <= 0 || (P % 64) != 0 || (P % K) != 0) return false;
    if (!(K == 32 || K == 64 || K == 128)) return false;       // (the forward's fused pool)
    const int gpb = K >= 64 ? 1 : (int)(64 / K);
    if (CI == 128 && CO == 256 && gpb == 1) return true;       // second level of the SSG encoder (K = 64); K = 128 scales of the MSG encoder
    if (CI == 64 && CO == 128 && (gpb == 1 || gpb == 2)) return true;     // first level (K = 32)
    if (CI == 128 && CO == 128 && (gpb == 1 || gpb == 2)) return true;
    return false;
}

size_t lean_workspace_bytes(int64_t P, int64_t K, int64_t CO, int64_t CI)
{
    (void)K;
    const size_t nblk = (size_t)((P + lean_ppb(P, 2) - 1) / lean_ppb(P, 2));
    size_t b = 0;
    b += al256(nblk * (size_t)CO * CI * 4);      // wgS
    b += al256(nblk * (size_t)CI * CI * 4);      // wgGram
    b += al256(nblk * (size_t)CI * 4);           // wgCs
    b += al256((size_t)CI * CI * 4) * 2;         // M, Gp
    b += al256((size_t)CI * 4) * 3;              // m~, v~, cs
    b += al256((size_t)CO * 4) * 3;              // a, e, f~
    return b;
}

// Backward of the pooled layer (see the head of this file).  z1 / s1 / t1 / gamma1 / beta1: raw activation, BatchNorm affine and
// BatchNorm parameters of layer L-1; a / e / f: the dZ constants of layer L; gp / argk: relu-masked pooled gradient and arg-max.
// Writes G1 [P, CI], the BatchNorm-backward partials of layer L-1 ([*nblk_out][2][CI]) and dW [CO, CI].
int lean_bwd(const float* z1, const float* s1, const float* t1, const float* gamma1, const float* beta1, const float* W, const float* a,
             const float* e, const float* f, const float* gp, const int* argk, int64_t P, int64_t K, int CO, int CI, float* G1,
             float* partials, int* nblk_out, float* dW, void* workspace, hipStream_t stream)
{
    const int ppb = lean_ppb(P, 1), ppb2 = lean_ppb(P, 2);
    const int nblk = (int)((P + ppb - 1) / ppb), nblk2 = (int)((P + ppb2 - 1) / ppb2);
    unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
    auto take = [&](size_t bytes) { float* p = reinterpret_cast<float*>(w); w += al256(bytes); return p; };
    float* wgS = take((size_t)nblk2 * CO * CI * 4);
    float* wgGram = take((size_t)nblk2 * CI * CI * 4);
    float* wgCs = take((size_t)nblk2 * CI * 4);
    float* Mm = take((size_t)CI * CI * 4);
    float* Gp = take((size_t)CI * CI * 4);
    float* mt = take((size_t)CI * 4);
    float* vt = take((size_t)CI * 4);
    float* cs = take((size_t)CI * 4);
    float* ka = take((size_t)CO * 4);
    float* ke = take((size_t)CO * 4);
    float* kf = take((size_t)CO * 4);
    MP_LAUNCH("lean_prep_kernel", 2.0 * (double)CO * CI * CI, 4.0 * (double)CO * CI, lean_prep_kernel, dim3((unsigned)CI + 1), dim3(256), 0, stream, W, CO,
              CI, a, e, f, gamma1, beta1, mt, Mm, vt, ka, ke, kf);
    MP_CHECK_LAUNCH();
    LeanArgs A{};
    A.z1 = z1; A.s1 = s1; A.t1 = t1; A.mt = mt; A.Mm = Mm; A.vt = vt; A.W = W; A.a = a; A.gp = gp; A.argk = argk;
    A.G1 = G1; A.partials = partials; A.wgS = wgS; A.wgGram = wgGram; A.wgCs = wgCs;
    A.P = (int)P; A.K = (int)K; A.ppb = ppb; A.ppb2 = ppb2;
    static const int skip_env = getenv("MP_LEAN_SKIP") ? atoi(getenv("MP_LEAN_SKIP")) : 0;
    A.skip = skip_env;
    const int gpb = K >= 64 ? 1 : (int)(64 / K);
    // algorithmic work: G = Ac M (2 P CI^2) + the sparse rows (2 G CO CI) | Gram (2 P CI^2) + S (2 G CO CI); bytes: Z_{L-1} in, G out | Z_{L-1} in
    const double fl = 2.0 * (double)P * CI * CI + 2.0 * (double)(P / K) * CO * CI;
    const double by1 = 4.0 * (2.0 * (double)P * CI + 2.0 * (double)(P / K) * CO), by2 = 4.0 * ((double)P * CI + 2.0 * (double)(P / K) * CO);
    char tg1[64], tg2[64];
    snprintf(tg1, sizeof tg1, "lean_dx_kernel<%d, %d, %d>", CO, CI, gpb);
    snprintf(tg2, sizeof tg2, "lean_dws_kernel<%d, %d, %d>", CO, CI, gpb);
#define MP_LEAN(CO_, CI_, GPB_)                                                                                                               \
    do {                                                                                                                                      \
        MP_LAUNCH(tg1, fl, by1, (lean_dx_kernel<CO_, CI_, GPB_>), dim3((unsigned)nblk), dim3(CI_ == 128 ? 512 : 256), 0, stream, A);        \
        MP_LAUNCH(tg2, fl, by2, (lean_dws_kernel<CO_, CI_, GPB_>), dim3((unsigned)nblk2), dim3(CI_ == 128 ? 512 : 256), 0, stream, A);      \
    } while (0)
    if (CO == 256 && CI == 128 && gpb == 1) MP_LEAN(256, 128, 1);
    else if (CO == 128 && CI == 64 && gpb == 2) MP_LEAN(128, 64, 2);
    else if (CO == 128 && CI == 64 && gpb == 1) MP_LEAN(128, 64, 1);
    else if (CO == 128 && CI == 128 && gpb == 1) MP_LEAN(128, 128, 1);
    else if (CO == 128 && CI == 128 && gpb == 2) MP_LEAN(128, 128, 2);
    else return MP_EUNSUPPORTED;
#undef MP_LEAN
    MP_CHECK_LAUNCH();
    MP_LAUNCH("lean_reduce_kernel", 0.0, 4.0 * (double)nblk2 * CI * CI, lean_reduce_kernel, dim3((unsigned)(CI * CI / 16)), dim3(256), 0, stream, wgGram, wgCs, nblk2, CI, mt, Gp, cs);
    MP_CHECK_LAUNCH();
    MP_LAUNCH("lean_finish_kernel", 2.0 * (double)CO * CI * CI, 4.0 * (double)nblk2 * CO * CI, lean_finish_kernel, dim3((unsigned)(CO * CI / 16)),
              dim3(256), 0, stream, wgS, nblk2, CO, CI, W, Gp, cs, mt, ka, ke, kf, (float)P, dW);
    MP_CHECK_LAUNCH();
    if (nblk_out) *nblk_out = nblk;
    return MP_OK;
}

}  // namespace mp
