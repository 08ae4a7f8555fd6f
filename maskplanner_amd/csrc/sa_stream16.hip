// [r5] Position-stream kernels written for ONE bf16 plane and bf16 activation storage (BASELINE configs[4]: containers, N = 10240, multi-scale
// grouping, "bf16 MFMA grouped-MLP").
//
// Reference: models/pointnet2_utils.py:219-276 (PointNetSetAbstractionMsg: per scale `relu(bn(conv1x1(x)))` x 3 over [B, C, K, S], max over K)
// and autograd's mirror image.  Until r4 the bf16 variant ran sa_mlp.hip's three-plane kernels with two planes switched off: 232-248
// registers (one eight-wave workgroup per CU), the chunk's rows prefetched through REGISTERS one or two chunks ahead (8-byte loads), 16-32 KB
// of reads in flight per CU -- 2.6 TB/s with every unit idle (profiles/r04_sq_counters: MFMA 7.7 %, 65 % of the wave cycles waiting for
// memory).  These kernels are built around the load path instead:
//   * the chunk's raw bf16 rows go STRAIGHT INTO LDS (global_load_lds_dwordx4: 64 lanes x 16 bytes = 1 KB per instruction, no register
//     destination) into a ring of R slots, R - 1 chunks ahead; nothing of a chunk waits in registers;
//   * the instruction is issued through inline assembly with an explicit `s_waitcnt vmcnt(N)` in front of the chunk's barrier -- the
//     compiler neither counts nor drains it (vmcnt retires in order, loads and stores alike: N = the operations this wave issued behind
//     the loads it waits for);
//   * the LDS image of a raw row is linear (the hardware writes lane i at base + 16 i); bank conflicts of the staging reads (a wave reads
//     four rows x 64 channels per pass: rows one or two bank cycles apart) are removed on the SOURCE side: lane i fetches the 16-byte unit
//     u ^ swz(row) of its row, the reader applies the same XOR;
//   * staging LDS -> registers -> LDS: BN / ReLU / dZ algebra in fp32, ONE bf16 plane in the K-packed, row-swizzled image of sa_common.h
//     (tr_frag_packed), which serves dW (32x32x16, transposed reads) and dX (16x16x32, 16-byte rows) alike; the weights as one plane in
//     registers (a third of the three-plane kernels' fragments);
//   * 256-thread workgroups (512 for 256 outputs), 70 KB of LDS: two workgroups per CU, 2 x 2 chunks x 16-24 KB of reads in flight.
// Arithmetic = the one-plane form of sa_mlp.hip (both operands of every contraction rounded to bf16 as they are staged, fp32 accumulation,
// BatchNorm sums from the fp32 accumulators, Z_l / G_l stored as bf16): same oracle (oracle/torch_ref.py: _StoreRound / _GradRound).
#include <cstdlib>

#include "sa_common.h"

namespace {

// 16 bytes per lane from global memory straight into LDS: lane i's data lands at lds_dst + 16 i (lds_dst wave-uniform: it travels in M0).
__device__ __forceinline__ void glds16(const void* src, void* lds_dst)
{
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a six-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// every LDS operation of this wave has completed, then the workgroup barrier (no vmcnt(0): the ring's loads stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// XOR on the 16-byte unit index of a raw row, so that the four rows of a staging pass fall into four different 64-byte bank windows:
// rows of 256 / 512 bytes start on the same bank -> 4 * (row & 3); rows of 128 bytes alternate between two windows -> 4 * ((row >> 1) & 1)
template <int ROWBYTES>
__device__ __forceinline__ int raw_swz(int row) { return ROWBYTES == 128 ? 4 * ((row >> 1) & 1) : 4 * (row & 3); }

__device__ __forceinline__ float4 bf4_to_f4(const uint2 u)
{
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

#ifndef MP_S16_FRING
#define MP_S16_FRING 3         // raw slots of the forward kernels' ring
#endif
#ifndef MP_S16_RING
#define MP_S16_RING 3          // raw slots of the pooled layers' ring (the dense layers stage Z and G: two slots)
#endif

constexpr int s16_threads(int CO) { return CO == 256 ? 512 : 256; }

// dX + dW + the BatchNorm-backward sums of layer l - 1 in one pass over dZ_l (sa_mlp.hip: bwd_fused_kernel), bf16 storage, one plane.
//   DZ.x: Z_l [P, CO] bf16;  pooled: DZ.g [P / K, CO] fp32 (relu-masked pooled gradient), DZ.argk [P / K, CO] int32;  dense: DZ.g = G_l [P, CO] bf16
//   IN.x: Z_{l-1} [P, CI] bf16 (+ its scale / shift);  W [CO, CI] fp32;  G: G_{l-1} [P, CI] bf16 out;  dW [CO, CI] fp32 += (atomics)
//   MODE_IN = SRC_ACT_RC: the input layer is the level's RECOMPUTED first layer (sa_mlp.hip: *_RC): IN.rx = the level's input rows [P, 4] fp32,
//   IN.rw = W_0 [CI, 4]; z_{l-1}[p][c] = dot4_rc(rx[p], rw[c]) is formed while staging (and again in the epilogue's sums), never read.
template <int MODE_DZ, int CO, int CI, int MODE_IN = SRC_ACT>
__global__ __launch_bounds__(s16_threads(CO), CO == 256 ? 1 : 2) void bwd_stream16_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block,
                                                                                          const float* __restrict__ W, float* __restrict__ dW,
                                                                                          float* __restrict__ G, BnOut partials)
{
    constexpr bool POOLED = MODE_DZ == SRC_DZ_POOLED, RC = MODE_IN == SRC_ACT_RC;
    static_assert((MODE_DZ == SRC_DZ || POOLED) && (MODE_IN == SRC_ACT || RC), "operands");
    constexpr int NT = s16_threads(CO), NW = NT / 64, DBK = 32, GS = DBK * 8;
    constexpr int R = POOLED ? MP_S16_RING : 2;
    constexpr int RBA = CO * 2, RBB = RC ? 32 : CI * 2;             // bytes of a raw row (RC: 16-byte input rows, the slot padded to one 1 KB load)
    constexpr int NBA = CO / 64, NBB = CI / 64;                     // 64-channel blocks of a row: one staging pass of a wave = 4 rows x one block
    constexpr int KA_STEP = 4 * (NW / NBA), KB_STEP = 4 * (NW / NBB);
    constexpr int PA = DBK / KA_STEP, PB = DBK / KB_STEP;           // staging passes per wave and chunk
    constexpr int TMW = CO / (32 * (NW / 2)), TNW = CI / 64;        // 32 x 32 dW tiles per wave (waves (NW / 2) x 2)
    constexpr int HT = CI / (16 * NW);                              // 16-column dX tiles per wave (both 16-row tiles of the chunk)
    constexpr int NST = CO / 32;                                    // k-steps of the dX product
    constexpr int LZ = DBK * RBA / 1024 / NW, LI = RC ? 1 : DBK * RBB / 1024 / NW;   // 1 KB load instructions per wave and chunk (RC: every wave, same bytes)
    constexpr int LPI = (CO / 2 + 63) / 64;                         // pooled: (gradient | arg-max) of the chunk's group, every wave (same bytes)
    constexpr int LD = LZ * (POOLED ? 1 : 2) + LI + (POOLED ? LPI : 0);
    constexpr int ST = 4 * HT;                                      // G stores per wave and chunk
    static_assert(LZ >= 1 && LI >= 1 && HT >= 1 && PA >= 1 && PB >= 1 && TMW >= 1 && TNW >= 1, "shape");
    static_assert(R == 2 || R == 3, "the wait counts below are written for two or three slots");
    static_assert((R - 2) * LD + (R - 1) * ST < 64, "vmcnt");

    __shared__ __attribute__((aligned(16))) unsigned char rawZ[R][DBK * RBA];
    __shared__ __attribute__((aligned(16))) unsigned char rawG[POOLED ? 1 : R][POOLED ? 16 : DBK * RBA];
    __shared__ __attribute__((aligned(16))) unsigned char rawI[R][DBK * RBB];
    __shared__ __attribute__((aligned(16))) float rawP[POOLED ? R : 1][POOLED ? LPI * 256 : 4];      // [g: CO floats | argk: CO ints] (CO = 64: half used)
    __shared__ __attribute__((aligned(16))) __bf16 hA[(CO / 8) * GS];
    __shared__ __attribute__((aligned(16))) __bf16 hB[(CI / 8) * GS];
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * CO];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // negative p_per_block: chunk-interleaved workgroups (sa_mlp.hip: bwd_fused_kernel) -- workgroup w takes chunks w, w + grid, ...
    const bool il = p_per_block < 0;
    if (il) p_per_block = -p_per_block;
    const int p0 = il ? 0 : blockIdx.x * p_per_block;
    const int p1 = il ? P : min(P, p0 + p_per_block);
    const int cstep = il ? (int)gridDim.x * DBK : DBK;
    const int cp0 = il ? (int)blockIdx.x * DBK : p0;
    const int nchunks = il ? ((P + DBK - 1) / DBK - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;
    const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(DZ.x);
    const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(DZ.g);
    const unsigned char* isrc = reinterpret_cast<const unsigned char*>(IN.x);

    // ---- the ring: chunk c -> slot c % R --------------------------------------------------------------------------------------------
    auto issue = [&](int c) {
        const int slot = c % R;
        const int pk = cp0 + c * cstep;
        constexpr int UA = RBA / 16, UB = RBB / 16;                 // 16-byte units per row
#pragma unroll
        for (int j = 0; j < LZ; ++j) {
            const int s = (wave * LZ + j) * 64 + lane;              // linear 16-byte slot of the chunk image
            const int row = s / UA, u = (s % UA) ^ raw_swz<RBA>(row);
            const int pr = pk + row < p1 ? pk + row : p0;           // rows past the end re-read row p0 (staged as zeros)
            glds16(zsrc + (size_t)pr * RBA + u * 16, rawZ[slot] + (wave * LZ + j) * 1024);
            if constexpr (!POOLED) glds16(gsrc + (size_t)pr * RBA + u * 16, rawG[slot] + (wave * LZ + j) * 1024);
        }
        if constexpr (RC) {
            const int row = lane & 31;                              // 32 rows x 16 bytes (lanes 32..63 repeat them into the slot's second half)
            const int pr = pk + row < p1 ? pk + row : p0;
            glds16(reinterpret_cast<const unsigned char*>(IN.rx) + (size_t)pr * 16, rawI[slot]);
        } else {
#pragma unroll
            for (int j = 0; j < LI; ++j) {
                const int s = (wave * LI + j) * 64 + lane;
                const int row = s / UB, u = (s % UB) ^ raw_swz<RBB>(row);
                const int pr = pk + row < p1 ? pk + row : p0;
                glds16(isrc + (size_t)pr * RBB + u * 16, rawI[slot] + (wave * LI + j) * 1024);
            }
        }
        if constexpr (POOLED) {
            const size_t grow = (size_t)((unsigned)pk >> DZ.kshift) * CO;      // the chunk lies inside one group (K = 2^kshift >= 32)
#pragma unroll
            for (int j = 0; j < LPI; ++j) {
                const int s = j * 64 + lane;                        // units 0 .. CO/4 - 1: gradient, CO/4 .. CO/2 - 1: arg-max
                const int uu = s < CO / 2 ? s : CO / 2 - 1;
                const unsigned char* src = uu < CO / 4 ? reinterpret_cast<const unsigned char*>(DZ.g + grow) + uu * 16
                                                       : reinterpret_cast<const unsigned char*>(DZ.argk + grow) + (uu - CO / 4) * 16;
                glds16(src, reinterpret_cast<unsigned char*>(rawP[slot]) + j * 1024);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < R - 1; ++c) issue(c < nchunks ? c : nchunks - 1);

    // ---- per-thread constants --------------------------------------------------------------------------------------------------------
    // staging: lane -> (channel quad cq of the wave's 64-channel block, row pr of a pass's four); see sa_mlp.hip (conflict-free plane writes)
    const int cq = (lane & 7) + 8 * (lane >> 5), prw = (lane >> 3) & 3;
    const int ca = (wave % NBA) * 64 + 4 * cq, ka0 = (wave / NBA) * 4 + prw;
    const int cb = (wave % NBB) * 64 + 4 * cq, kb0 = (wave / NBB) * 4 + prw;
    ChanConst ka, kb;
    load_consts<MODE_IN>(IN, cb, kb);
    // dW tiles of this wave
    const int l31 = lane & 31;
    const int wrow0 = (wave >> 1) * TMW * 32, wcol0 = (wave & 1) * TNW * 32;
    f32x16 accW[TMW][TNW];
#pragma unroll
    for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[mi][ni][r] = 0.0f;
    // dX: this wave's HT column tiles, both row tiles; W_l as ONE bf16 plane in registers: lane (col l15, kq) holds W[32 st + 8 kq .. + 7][col]
    const int l15 = lane & 15, kq = lane >> 4;
    const int xcol0 = wave * HT * 16;
    bf16x8 wsp[HT][NST];
#pragma unroll
    for (int h = 0; h < HT; ++h)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const float* wp = W + (size_t)(32 * st + 8 * kq) * CI + xcol0 + 16 * h + l15;
#pragma unroll
            for (int i = 0; i < 8; ++i) wsp[h][st][i] = (__bf16)wp[(size_t)i * CI];
        }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    float spx[HT], tpx[HT];
    float4 w0c[RC ? HT : 1];                                       // RC: W_0 row of this lane's G column
    f2 sx1[HT], sx2[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        const int col = xcol0 + 16 * h + l15;
        spx[h] = IN.s[col];
        tpx[h] = IN.t[col];
        if constexpr (RC) w0c[h] = ld4(IN.rw + (size_t)col * 4);
        sx1[h] = f2{0.0f, 0.0f};
        sx2[h] = f2{0.0f, 0.0f};
    }
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(G) + (size_t)p0 * CI * 2, 0, (p1 - p0) * CI * 2, 0x00020000);
    int goff = ((4 * kq) * CI + xcol0 + (l15 & ~1)) * 2 + ((lane & 1) ? CI * 2 : 0) + (cp0 - p0) * CI * 2;

    bn_prologue(DZ.bn, bn_lds, CO, 0, CO, blockIdx.x == 0);        // (a, e, f) of dZ_l when this kernel is their first consumer (contains a barrier)
    load_consts<MODE_DZ>(DZ, ca, ka, bn_lds, CO);

    for (int kc = 0; kc < nchunks; ++kc) {
        const int slot = kc % R;
        const int pk = cp0 + kc * cstep;
        // this wave's loads of chunk kc have landed: behind them it issued the loads of R - 2 more chunks and the stores of min(kc, R - 1) epilogues
        if (kc >= R - 1) wait_vm<(R - 2) * LD + (R - 1) * ST>();
        else if (R > 2 && kc == 1) wait_vm<(R - 2) * LD + 1 * ST>();
        else wait_vm<(R - 2) * LD>();
        lds_barrier();                                             // B1: every wave's part of the raw chunk is in LDS; the planes are free
        issue(kc + R - 1 < nchunks ? kc + R - 1 : nchunks - 1);    // -> the slot chunk kc - 1 left (past the end: a slot nobody reads again)

        // ---- staging: raw rows -> dz / activated input -> one bf16 plane each ----------------------------------------------------------
        {
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 ak4 = make_int4(0, 0, 0, 0);
            if constexpr (POOLED) {
                g4 = *reinterpret_cast<const float4*>(&rawP[slot][ca]);
                ak4 = *reinterpret_cast<const int4*>(&rawP[slot][CO + ca]);
            }
            const int kk0 = POOLED ? (pk & (DZ.K - 1)) : 0;
#pragma unroll
            for (int ps = 0; ps < PA; ++ps) {
                const int row = ka0 + ps * KA_STEP;
                const int ro = row * RBA + (((ca >> 3) ^ raw_swz<RBA>(row)) * 16) + (cq & 1) * 8;
                const float4 z = bf4_to_f4(*reinterpret_cast<const uint2*>(&rawZ[slot][ro]));
                float4 dz;
                if constexpr (POOLED) {
                    const int kk = kk0 + row;
                    dz.x = xf1<MODE_DZ>(z.x, ak4.x == kk ? g4.x : 0.0f, ka.s.x, ka.t.x, ka.a.x, ka.e.x, ka.f.x);
                    dz.y = xf1<MODE_DZ>(z.y, ak4.y == kk ? g4.y : 0.0f, ka.s.y, ka.t.y, ka.a.y, ka.e.y, ka.f.y);
                    dz.z = xf1<MODE_DZ>(z.z, ak4.z == kk ? g4.z : 0.0f, ka.s.z, ka.t.z, ka.a.z, ka.e.z, ka.f.z);
                    dz.w = xf1<MODE_DZ>(z.w, ak4.w == kk ? g4.w : 0.0f, ka.s.w, ka.t.w, ka.a.w, ka.e.w, ka.f.w);
                } else {
                    const float4 g = bf4_to_f4(*reinterpret_cast<const uint2*>(&rawG[slot][ro]));
                    dz.x = xf1<MODE_DZ>(z.x, g.x, ka.s.x, ka.t.x, ka.a.x, ka.e.x, ka.f.x);
                    dz.y = xf1<MODE_DZ>(z.y, g.y, ka.s.y, ka.t.y, ka.a.y, ka.e.y, ka.f.y);
                    dz.z = xf1<MODE_DZ>(z.z, g.z, ka.s.z, ka.t.z, ka.a.z, ka.e.z, ka.f.z);
                    dz.w = xf1<MODE_DZ>(z.w, g.w, ka.s.w, ka.t.w, ka.a.w, ka.e.w, ka.f.w);
                }
                if (pk + row >= p1) dz = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<bf16x4*>(&hA[(ca >> 3) * GS + (row ^ kswz(ca >> 3)) * 8 + (ca & 7)]) = to_bf16x4(dz);
            }
#pragma unroll
            for (int ps = 0; ps < PB; ++ps) {
                const int row = kb0 + ps * KB_STEP;
                float4 z;
                if constexpr (RC) {
                    const float4 xr = *reinterpret_cast<const float4*>(&rawI[slot][row * 16]);
                    z = make_float4(dot4_rc(xr, kb.w[0]), dot4_rc(xr, kb.w[1]), dot4_rc(xr, kb.w[2]), dot4_rc(xr, kb.w[3]));
                } else {
                    const int ro = row * RBB + (((cb >> 3) ^ raw_swz<RBB>(row)) * 16) + (cq & 1) * 8;
                    z = bf4_to_f4(*reinterpret_cast<const uint2*>(&rawI[slot][ro]));
                }
                float4 x;
                x.x = xf1<SRC_ACT>(z.x, 0.f, kb.s.x, kb.t.x, 0.f, 0.f, 0.f);
                x.y = xf1<SRC_ACT>(z.y, 0.f, kb.s.y, kb.t.y, 0.f, 0.f, 0.f);
                x.z = xf1<SRC_ACT>(z.z, 0.f, kb.s.z, kb.t.z, 0.f, 0.f, 0.f);
                x.w = xf1<SRC_ACT>(z.w, 0.f, kb.s.w, kb.t.w, 0.f, 0.f, 0.f);
                if (pk + row >= p1) x = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<bf16x4*>(&hB[(cb >> 3) * GS + (row ^ kswz(cb >> 3)) * 8 + (cb & 7)]) = to_bf16x4(x);
            }
        }
        lds_barrier();                                             // B2: the planes are complete

        // ---- dW += dZ^T act(Z_{l-1}): two k-steps of 16 positions ---------------------------------------------------------------------
#pragma unroll
        for (int k0 = 0; k0 < DBK; k0 += 16) {
            bf16x8 fb[TNW], fa[TMW];
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni) fb[ni] = tr_frag_packed<GS, true>(hB, k0, wcol0 + ni * 32);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA, k0, wrow0 + mi * 32);
            tr_fence();
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], accW[mi][ni], 0, 0, 0);
        }
        // ---- G_{l-1} chunk [32 x CI] = dZ [32 x CO] W_l [CO x CI]; epilogue: bf16 store + the BatchNorm-backward sums of layer l - 1 ------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 ax[HT];
#pragma unroll
            for (int h = 0; h < HT; ++h) ax[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ao = kq * GS + ((16 * rt + l15) ^ kswz(kq)) * 8;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&hA[ao + 4 * st * GS]);
#pragma unroll
                for (int h = 0; h < HT; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wsp[h][st], ax[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                const int col = xcol0 + 16 * h + l15;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const bool odd = lane & 1;
                    const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? ax[h][i] : ax[h][i + 1]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                    __builtin_amdgcn_raw_buffer_store_b32(odd ? pack_bf16(got, ax[h][i + 1]) : pack_bf16(ax[h][i], got), grsrc,
                                                          goff + (16 * rt * CI + 16 * h) * 2, i * CI * 2, MP_STORE_AUX);
                    const int r0 = 16 * rt + 4 * kq + i, r1 = r0 + 1;
                    f2 zp;
                    if constexpr (RC) {
                        zp = f2{dot4_rc(*reinterpret_cast<const float4*>(&rawI[slot][r0 * 16]), w0c[h]),
                                dot4_rc(*reinterpret_cast<const float4*>(&rawI[slot][r1 * 16]), w0c[h])};
                    } else {
                        const unsigned short z0 = *reinterpret_cast<const unsigned short*>(&rawI[slot][r0 * RBB + (((col >> 3) ^ raw_swz<RBB>(r0)) * 16) + (col & 7) * 2]);
                        const unsigned short z1 = *reinterpret_cast<const unsigned short*>(&rawI[slot][r1 * RBB + (((col >> 3) ^ raw_swz<RBB>(r1)) * 16) + (col & 7) * 2]);
                        zp = f2{__uint_as_float((unsigned)z0 << 16), __uint_as_float((unsigned)z1 << 16)};
                    }
                    const f2 y = zp * f2{spx[h], spx[h]} + f2{tpx[h], tpx[h]};
                    const f2 dy = {y.x > 0.0f ? ax[h][i] : 0.0f, y.y > 0.0f ? ax[h][i + 1] : 0.0f};
                    sx1[h] += dy;
                    sx2[h] += dy * zp;
                }
            }
        }
        goff += cstep * CI * 2;
    }
    wait_vm<0>();
    // ---- BatchNorm-backward sums of layer l - 1: the wave owns its columns; the four row groups of a lane column meet by shuffles -------
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        float s1x = sx1[h].x + sx1[h].y, s2x = sx2[h].x + sx2[h].y;
        s1x += __shfl_xor(s1x, 16, 64); s1x += __shfl_xor(s1x, 32, 64);
        s2x += __shfl_xor(s2x, 16, 64); s2x += __shfl_xor(s2x, 32, 64);
        if (lane < 16) {
            const int col = xcol0 + 16 * h + lane;
            if (partials.slots) {
                atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + 0) * CI + col, (double)s1x);
                atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + 1) * CI + col, (double)s2x);
            } else {
                partials.rows[((size_t)blockIdx.x * 2 + 0) * CI + col] = s1x;
                partials.rows[((size_t)blockIdx.x * 2 + 1) * CI + col] = s2x;
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNW; ++ni) {
            const int col = wcol0 + ni * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wrow0 + mi * 32 + acc_row_in_tile(r);
                atomicAdd(dW + (size_t)(row * CI + col), accW[mi][ni][r]);
            }
        }
}


// Forward layer Z_l = act(Z_{l-1}) W_l^T of the position stream (sa_mlp.hip: fwd_chunk_kernel), bf16 storage, one plane: the input rows through
// the ring, BN + ReLU while staging into a row-major bf16 plane, W_l as one plane in registers; epilogue = bf16 store of Z_l, the layer's
// BatchNorm sums (fp32 per chunk, fp64 across), and for the pooled layer the group's extremum per column (max-pool commutes with the
// monotone BN + ReLU: one tracked extremum, chosen by the sign of gamma).
//   A.x: Z_{l-1} [P, CI] bf16 (MODE_A = SRC_ACT) | A.rx: input rows [P, 4] fp32, A.rw: W_0 [CI, 4] (SRC_ACT_RC);  W [CO, CI] fp32;  Z [P, CO] bf16
template <int CI, int CO, bool POOL, int MODE_A = SRC_ACT>
__global__ __launch_bounds__(s16_threads(CO), CO == 256 ? 1 : 2) void fwd_stream16_kernel(PosOperand A, int P, int p_per_block, const float* __restrict__ W,
                                                                                          float* __restrict__ Z, BnOut partials, PoolOut po,
                                                                                          const float* __restrict__ gamma)
{
    bn_zero(partials);
    constexpr bool RC = MODE_A == SRC_ACT_RC;
    static_assert(MODE_A == SRC_ACT || RC, "input operand");
    constexpr int NT = s16_threads(CO), NW = NT / 64, DBK = 32;
    constexpr int CW = CO / NW;                       // columns per wave: 32 (one 32x32 tile) or 16 (two 16x16 row tiles)
    constexpr bool BIG = CW == 32;
    static_assert(CW == 32 || CW == 16, "shape");
    constexpr int R = MP_S16_FRING;
    constexpr int RB = RC ? 32 : CI * 2;              // bytes of a raw row (RC: the slot padded to one 1 KB load)
    constexpr int LI = RC ? 1 : DBK * RB / 1024 / NW; // 1 KB load instructions per wave and chunk
    constexpr int NV = BIG ? 16 : 8;                  // rows of the chunk held by one lane
    constexpr int SV = NV / 2;                        // Z stores per wave and chunk (the pool's stores at a group's end come on top: the wait
                                                      // counts assume the fewer -- waiting for one or two operations more is always safe)
    constexpr int LDH = CI + 8;                       // plane row stride in halves: 16-byte aligned rows, conflict-free 16-byte fragment reads
    constexpr int KST = BIG ? 16 : 32;
    constexpr int PA = DBK * CI / 4 / NT;             // channel quads staged per thread and chunk
    static_assert(LI >= 1 && PA >= 1 && (R == 2 || R == 3 || R == 4), "shape");
    static_assert((R - 2) * LI + (R - 1) * SV < 64, "vmcnt");

    __shared__ __attribute__((aligned(16))) unsigned char raw[R][RC ? 1024 : DBK * RB];
    __shared__ __attribute__((aligned(16))) __bf16 sH[DBK * LDH];
    __shared__ __attribute__((aligned(16))) float bn_lds[2 * CI];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = BIG ? (lane & 31) : (lane & 15);
    const int kq = BIG ? (lane >> 5) : (lane >> 4);
    const int col = wave * CW + lc;
    // negative p_per_block: interleaved workgroups, a unit = one chunk, or with the fused pool one group of K positions (sa_mlp.hip: fwd_chunk_kernel)
    const bool il = p_per_block < 0;
    if (il) p_per_block = -p_per_block;
    const int cpu_ = POOL ? po.K / DBK : 1;
    const int U = cpu_ * DBK;
    const int p0 = il ? 0 : blockIdx.x * p_per_block;
    const int p1 = il ? P : min(P, p0 + p_per_block);
    const int nchunks = il ? (((P + U - 1) / U - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * cpu_ : (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;
    auto cpos = [&](int c) {
        if (!il) return p0 + c * DBK;
        const int u = POOL ? c / cpu_ : c, r = POOL ? c - u * cpu_ : 0;
        return (u * (int)gridDim.x + (int)blockIdx.x) * U + r * DBK;
    };
    const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(RC ? A.rx : A.x);

    auto issue = [&](int c) {
        const int slot = c % R;
        const int pk = cpos(c);
        if constexpr (RC) {
            const int row = lane & 31;
            const int pr = pk + row < p1 ? pk + row : p0;
            glds16(xsrc + (size_t)pr * 16, raw[slot]);
        } else {
            constexpr int U = RB / 16;
#pragma unroll
            for (int j = 0; j < LI; ++j) {
                const int s = (wave * LI + j) * 64 + lane;
                const int row = s / U, u = s % U;
                const int pr = pk + row < p1 ? pk + row : p0;
                glds16(xsrc + (size_t)pr * RB + u * 16, raw[slot] + (wave * LI + j) * 1024);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < R - 1; ++c) issue(c < nchunks ? c : nchunks - 1);

    // W_l^T fragments, one plane: lane (n = col, kq) holds W[col][KST st + 8 kq .. + 7]
    bf16x8 wsp[CI / KST];
#pragma unroll
    for (int st = 0; st < CI / KST; ++st) {
        const float* wp = W + (size_t)col * CI + KST * st + 8 * kq;
        const float4 lo = ld4(wp), hi = ld4(wp + 4);
        wsp[st][0] = (__bf16)lo.x; wsp[st][1] = (__bf16)lo.y; wsp[st][2] = (__bf16)lo.z; wsp[st][3] = (__bf16)lo.w;
        wsp[st][4] = (__bf16)hi.x; wsp[st][5] = (__bf16)hi.y; wsp[st][6] = (__bf16)hi.z; wsp[st][7] = (__bf16)hi.w;
    }
    const int ca = (tid % (CI / 4)) * 4, ka0 = tid / (CI / 4);
    constexpr int KA_STEP = NT / (CI / 4);
    ChanConst kc;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    double s1 = 0.0, s2 = 0.0;
    float gbest = 0.0f;
    int gibest = 0;
    const int cpg = POOL ? po.K / DBK : 1;
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(Z) + (size_t)p0 * CO * 2, 0, (p1 - p0) * CO * 2, 0x00020000);
    const int zoff0 = ((4 * kq) * CO + (col & ~1)) * 2 + ((lane & 1) ? CO * 2 : 0);
    bool neg = false;
    if constexpr (POOL) neg = gamma[col] < 0.0f;
    const unsigned smask = neg ? 0x80000000u : 0u;

    bn_prologue(A.bn, bn_lds, CI, 0, CI, blockIdx.x == 0);
    load_consts<MODE_A>(A, ca, kc, bn_lds, CI);

    for (int kcn = 0; kcn < nchunks; ++kcn) {
        const int slot = kcn % R;
        const int pk = cpos(kcn);
        const int zoff = zoff0 + (pk - p0) * CO * 2;
        if (kcn >= R - 1) wait_vm<(R - 2) * LI + (R - 1) * SV>();
        else if (R > 2 && kcn == 1) wait_vm<(R - 2) * LI + 1 * SV>();
        else if (R > 3 && kcn == 2) wait_vm<(R - 2) * LI + 2 * SV>();
        else wait_vm<(R - 2) * LI>();
        lds_barrier();                                             // B1: the raw chunk is complete; the plane is free
        issue(kcn + R - 1 < nchunks ? kcn + R - 1 : nchunks - 1);
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            const int row = ka0 + ps * KA_STEP;
            float4 z;
            if constexpr (RC) {
                const float4 xr = *reinterpret_cast<const float4*>(&raw[slot][row * 16]);
                z = make_float4(dot4_rc(xr, kc.w[0]), dot4_rc(xr, kc.w[1]), dot4_rc(xr, kc.w[2]), dot4_rc(xr, kc.w[3]));
            } else {
                z = bf4_to_f4(*reinterpret_cast<const uint2*>(&raw[slot][row * RB + ca * 2]));
            }
            float4 x;
            x.x = xf1<SRC_ACT>(z.x, 0.f, kc.s.x, kc.t.x, 0.f, 0.f, 0.f);
            x.y = xf1<SRC_ACT>(z.y, 0.f, kc.s.y, kc.t.y, 0.f, 0.f, 0.f);
            x.z = xf1<SRC_ACT>(z.z, 0.f, kc.s.z, kc.t.z, 0.f, 0.f, 0.f);
            x.w = xf1<SRC_ACT>(z.w, 0.f, kc.s.w, kc.t.w, 0.f, 0.f, 0.f);
            if (pk + row >= p1) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<bf16x4*>(&sH[row * LDH + ca]) = to_bf16x4(x);
        }
        lds_barrier();                                             // B2: the plane is complete

        float v[NV];
        if constexpr (BIG) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            const int ao = (lane & 31) * LDH + 8 * kq;                // A[row = lane & 31][k = 16 st + 8 kq .. + 7]
#pragma unroll
            for (int st = 0; st < CI / 16; ++st)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&sH[ao + 16 * st]), wsp[st], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r];               // rows (r & 3) + 8 (r >> 2) + 4 kq
        } else {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            const int ao = lc * LDH + 8 * kq;                         // A[row = 16 rt + lc][k = 32 st + 8 kq .. + 7]
#pragma unroll
            for (int st = 0; st < CI / 32; ++st) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&sH[ao + 32 * st]), wsp[st], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&sH[ao + 16 * LDH + 32 * st]), wsp[st], a1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a0[i]; v[4 + i] = a1[i]; }   // rows 4 kq + i and 16 + 4 kq + i
        }
        auto rowc = [](int r) { return BIG ? (r & 3) + 8 * (r >> 2) : (r < 4 ? r : 16 + (r - 4)); };
        f2 c1 = {0.0f, 0.0f}, c2 = {0.0f, 0.0f};
        float lbest = -__builtin_inff();
        int libest = 0;
#pragma unroll
        for (int r = 0; r < NV; r += 2) {
            const bool odd = lane & 1;
            const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v[r] : v[r + 1]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
            __builtin_amdgcn_raw_buffer_store_b32(odd ? pack_bf16(got, v[r + 1]) : pack_bf16(v[r], got), zrsrc, zoff, rowc(r) * CO * 2, MP_STORE_AUX);
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
#pragma unroll
            for (int d = BIG ? 32 : 16; d <= 32; d <<= 1) {
                const float ox = __shfl_xor(lbest, d, 64);
                const int oix = __shfl_xor(libest, d, 64);
                if (ox > lbest || (ox == lbest && oix < libest)) { lbest = ox; libest = oix; }
            }
            const int cig = kcn % cpg;                // chunk inside its group (p0 is a multiple of K)
            if (cig == 0 || lbest > gbest) { gbest = lbest; gibest = cig * DBK + libest; }   // earlier chunk wins ties
            if (cig == cpg - 1 && kq == 0) {
                const size_t o = (size_t)((unsigned)(pk / po.K) * (unsigned)CO + (unsigned)col);
                const float val = __uint_as_float(__float_as_uint(gbest) ^ smask);
                if (neg) { po.vmin[o] = val; po.imin[o] = gibest; }
                else { po.vmax[o] = val; po.imax[o] = gibest; }
            }
        }
    }
    wait_vm<0>();
#pragma unroll
    for (int d = BIG ? 32 : 16; d <= 32; d <<= 1) {
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
    }
    if (kq == 0) bn_emit(partials, CO, blockIdx.x, col, s1, s2);
}

}  // namespace

// ---- launchers (called from sa_mlp.hip's level drivers; operands by pointer: the structs' layout is sa_common.h's) --------------------
// Returns 1 when the shape has a kernel here (and it was launched), 0 when not, < 0 on a launch error.
// MP_S16_IL=0: contiguous position ranges per workgroup (A/B timing); otherwise interleaved chunks while byte offsets fit 31 bits
static bool s16_interleave(int64_t P, int C)
{
    static const bool on = [] { const char* e = getenv("MP_S16_IL"); return !e || atoi(e) != 0; }();
    return on && (uint64_t)P * (uint64_t)C * 2u < (1ull << 31);
}

int mp_s16_bwd_launch(int pooled, int rc_in, int Co, int Ci, const void* dz_, const void* in_, int64_t P, int ppb, const float* W, float* dW, float* G,
                      const void* partials_, const char* tag, double flops, double bytes, hipStream_t stream)
{
    const PosOperand& DZ = *static_cast<const PosOperand*>(dz_);
    const PosOperand& IN = *static_cast<const PosOperand*>(in_);
    const BnOut& partials = *static_cast<const BnOut*>(partials_);
    if (pooled && (DZ.kshift < 5 || (ppb & (DZ.K - 1)))) return 0;          // a 32-position chunk lies inside one group
    {   // MP_S16=0: sa_mlp.hip's one-plane kernels (A/B timing)
        const char* e = getenv("MP_S16");
        if (e && atoi(e) == 0) return 0;
    }
    // positions per workgroup: a workgroup's set-up (weight plane, constants, the ring's first loads) is paid once per slab -- longer slabs
    // while the grid still covers the chip four times over (BatchNorm sums in slot rows only: the partial-row form is indexed by workgroup)
    if (partials.slots)
        while (ppb < 4096 && P / (2 * ppb) >= 1024) ppb *= 2;
    const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
    const int ppb_k = s16_interleave(P, Co > Ci ? Co : Ci) ? -ppb : ppb;
#define MP_S16(MODE, CO_, CI_, ...)                                                                                                \
    do {                                                                                                                           \
        MP_LAUNCH(tag, flops, bytes, (bwd_stream16_kernel<MODE, CO_, CI_, ##__VA_ARGS__>), dim3(gx), dim3(s16_threads(CO_)), 0, stream, DZ, IN, (int)P, \
                  ppb_k, W, dW, G, partials);                                                                                        \
        return hipGetLastError() == hipSuccess ? 1 : MP_ELAUNCH;                                                                   \
    } while (0)
    if (rc_in) {        // the layer behind a recomputed first layer (never the pooled one)
        if (pooled || !IN.rx || !IN.rw) return 0;
        if (Co == 128 && Ci == 64) MP_S16(SRC_DZ, 128, 64, SRC_ACT_RC);
        if (Co == 64 && Ci == 64) MP_S16(SRC_DZ, 64, 64, SRC_ACT_RC);
        return 0;
    }
    if (pooled) {
        if (Co == 128 && Ci == 128) MP_S16(SRC_DZ_POOLED, 128, 128);
        if (Co == 256 && Ci == 128) MP_S16(SRC_DZ_POOLED, 256, 128);
        if (Co == 128 && Ci == 64) MP_S16(SRC_DZ_POOLED, 128, 64);
        if (Co == 64 && Ci == 64) MP_S16(SRC_DZ_POOLED, 64, 64);
    } else {
        if (Co == 128 && Ci == 128) MP_S16(SRC_DZ, 128, 128);
        if (Co == 128 && Ci == 64) MP_S16(SRC_DZ, 128, 64);
        if (Co == 64 && Ci == 64) MP_S16(SRC_DZ, 64, 64);
    }
#undef MP_S16
    return 0;
}

int mp_s16_fwd_launch(int pool, int rc_in, int Ci, int Co, const void* a_, int64_t P, int ppb, const float* W, float* Z, const void* partials_,
                      const void* po_, const float* gamma, const char* tag, double flops, double bytes, hipStream_t stream)
{
    const PosOperand& A = *static_cast<const PosOperand*>(a_);
    const BnOut& partials = *static_cast<const BnOut*>(partials_);
    const PoolOut& po = *static_cast<const PoolOut*>(po_);
    {
        const char* e = getenv("MP_S16");
        if (e && (atoi(e) == 0 || atoi(e) == 2)) return 0;         // MP_S16=2: the backward kernels only
    }
    if (pool && (po.K < 32 || (po.K & 31) || (ppb % po.K))) return 0;
    if (rc_in && (!A.rx || !A.rw)) return 0;
    const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
    static const bool fwd_il_on = [] { const char* e = getenv("MP_S16_FIL"); return !e || atoi(e) != 0; }();
    const int ppb_k = (fwd_il_on && s16_interleave(P, Co > Ci ? Co : Ci)) ? -ppb : ppb;
#define MP_S16F(CI_, CO_, POOL_, ...)                                                                                              \
    do {                                                                                                                           \
        MP_LAUNCH(tag, flops, bytes, (fwd_stream16_kernel<CI_, CO_, POOL_, ##__VA_ARGS__>), dim3(gx), dim3(s16_threads(CO_)), 0, stream, A, (int)P, ppb_k, \
                  W, Z, partials, po, gamma);                                                                                      \
        return hipGetLastError() == hipSuccess ? 1 : MP_ELAUNCH;                                                                   \
    } while (0)
    if (rc_in) {
        if (pool) return 0;
        if (Ci == 64 && Co == 128) MP_S16F(64, 128, false, SRC_ACT_RC);
        if (Ci == 64 && Co == 64) MP_S16F(64, 64, false, SRC_ACT_RC);
        return 0;
    }
    if (pool) {
        if (Ci == 128 && Co == 128) MP_S16F(128, 128, true);
        if (Ci == 128 && Co == 256) MP_S16F(128, 256, true);
        if (Ci == 64 && Co == 128) MP_S16F(64, 128, true);
        if (Ci == 64 && Co == 64) MP_S16F(64, 64, true);
    } else {
        if (Ci == 128 && Co == 128) MP_S16F(128, 128, false);
        if (Ci == 64 && Co == 128) MP_S16F(64, 128, false);
        if (Ci == 64 && Co == 64) MP_S16F(64, 64, false);
    }
#undef MP_S16F
    return 0;
}
