// The fused backward of a set-abstraction level's single-tile layers for gfx950: dX, dW and the BatchNorm-backward sums of the layer below in
// ONE pass over dZ_l (bwd_fused_kernel), and its role-split form for the 256-output layer (bwd_roles_kernel).  Reference: autograd's mirror
// image of models/pointnet2_utils.py:208-214.  [r5] Split out of sa_mlp.hip (which keeps the tiled GEMMs, the forward stream kernels, the
// first-layer kernels and the host side of mp_sa_mlp_*); shared device helpers: sa_common.h; the bf16-storage ring kernels: sa_stream16.hip.
#include <cstdlib>

#include "sa_common.h"

#ifndef MP_BF_ABL
#define MP_BF_ABL 0         // timing builds of bwd_fused_kernel with parts compiled out (tools/bwd_ablate.sh): 1 dW products, 2 dX products, 4 dX epilogue,
#endif                      // 8 staging arithmetic, 16 global loads, 32 staging altogether -- results are wrong by construction
#ifndef MP_PD2
#define MP_PD2 3            // [r3] fused backward: two chunks of loads in flight (two register sets, loop unrolled by two); bit mask, see PD2
#endif
#ifndef MP_PD2_ONE
#define MP_PD2_ONE 2        // [r4] the one-plane (bf16 variant) kernels: two chunks of loads in flight for the 128 x 128 shapes only (the 256-thread
                            // kernels are faster with one: config 5 6.27 -> 6.20 ms; 128 x 128 with two: 668 -> 615 us, 262 -> 257 us)
#endif
#ifndef MP_ONE_DBK32
#define MP_ONE_DBK32 5      // [r4] one-plane fused backward, 32 positions per chunk: bit 0: 128 x 128 (700 -> 660 us, 312 -> 272), bit 1: 256 x 128 (spills: off),
                            // bit 2: the 64-input shapes (195 -> 165, 85 -> 70, 74 -> 60 us; two workgroups per CU instead of three)
#endif
#ifndef MP_DESYNC
#define MP_DESYNC 1         // [r3] fused backward, 256-output layer (eight waves): waves 4..7 take the chunk's barrier between the products and the staging
#endif

namespace {

// =================================================================================================================
// Kernel 4: dX and dW of one layer in ONE pass over dZ_l, for layers whose channels fit a single tile (CO in {64, 128}
// outputs, 64 inputs: the HBM-bound layers of the first set-abstraction level).  Separately, dX streams (Z_l, G_l), reads
// Z_{l-1} and writes G_{l-1}; dW streams (Z_l, G_l, Z_{l-1}) again: 7 tensor passes.  Here a workgroup walks 1024 positions
// in chunks of 32: the dZ chunk [32][CO] and the activated input chunk [32][64] are staged once and feed both
//   dW[CO,64]  += dZ^T * act(Z_{l-1})          (K = positions; accumulated in registers, atomics at the end) and
//   G_{l-1}[32,64] = dZ * W_l                   (K = CO; W_l resident in LDS), whose epilogue also forms the
// BatchNorm-backward sums of layer l-1 from the raw Z_{l-1} chunk kept beside the activated one: 4 tensor passes.
// The dX tile of a chunk is computed as eight 16x16 MFMA tiles, two per wave.
// =================================================================================================================
#ifndef MP_BF_INTERLEAVE
#define MP_BF_INTERLEAVE 1  // [r5] fused backward: chunk-interleaved workgroups (the host asks for it with a negative p_per_block; MP_BF_IL=0 at run time: contiguous ranges)
#endif
#ifndef MP_BWD_KSPLIT
#define MP_BWD_KSPLIT 0     // [r2] measured on one box: 234 us without, 287 us with (the extra barrier and 8 spilled registers cost more than the halved LDS reads return)
#endif
#ifndef MP_MAP256
#define MP_MAP256 0         // 1: the conflict-free 8-lane mapping also for the 1 KB dZ rows of the 256-output layer (234 -> 256 us: 128-byte global segments)
#endif
#ifndef MP_SPLIT_WGS
#define MP_SPLIT_WGS 2     // (3: a third workgroup of the 64-input layers per CU -- tried: 168-register cap, spills in the loop, 150 -> 370 us)
#endif
#ifndef MP_BF_NT256
#define MP_BF_NT256 0       // experiment: the 128-input layers with FOUR waves (one per SIMD, up to 512 registers each) instead of eight; bit 0: 256 outputs, bit 1: 128
#endif
// threads of a fused-backward workgroup: eight waves for the 128-input layers (registers per wave), four otherwise
constexpr int bwd_fused_threads(int CO, int CI, bool SPLIT, bool ONE)
{
    if (CO >= 128 && CI == 128) return (SPLIT && !ONE && ((MP_BF_NT256 >> (CO == 256 ? 0 : 1)) & 1)) ? 256 : 512;
    return 256;
}
template <int MODE_DZ, int CO, int CI, int MODE_IN = SRC_ACT, bool SPLIT = false, bool ONE = false, int PL = 3>      // ONE: see fwd_chunk_kernel; PL: operand planes of the split form (split2 / split3)
__global__ __launch_bounds__(bwd_fused_threads(CO, CI, SPLIT, ONE), (CO >= 128 && CI == 128 ? 1 : ((SPLIT && CI == 64) ? MP_SPLIT_WGS : 2))) void bwd_fused_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block,
                                                            const float* __restrict__ W, float* __restrict__ dW,
                                                            float* __restrict__ G, BnOut partials)
{
    __shared__ __attribute__((aligned(16))) float bn_lds[3 * CO];           // (a, e, f) of dZ_l when this kernel derives them
    constexpr int NT = bwd_fused_threads(CO, CI, SPLIT, ONE), NW = NT / 64;
    // [r4] ONE (the bf16 variant's single plane): 32 positions per chunk -- a chunk's products are a sixth of the three-plane kernel's, so
    // at 16 positions the two barriers and the LDS round trips of a chunk were most of its 3 300 cycles (MFMA 8 %, VALU 29 %, 65 % waiting)
    // MP_ONE_DBK32: bit 0: 128 x 128, bit 1: 256 x 128, bit 2: the 64-input shapes
    constexpr bool ONE32 = SPLIT && ONE && (((MP_ONE_DBK32 & 1) && CO == 128 && CI == 128) || ((MP_ONE_DBK32 & 2) && CO == 256 && CI == 128) || ((MP_ONE_DBK32 & 4) && CI == 64));
    constexpr int DBK = ONE32 ? 32 : ((CI == 128 || SPLIT) ? 16 : 32);
      // positions per chunk (LDS and registers: at least two workgroups per CU)
    constexpr int XW = DBK == 32 ? CI / (NW / 2) : CI / NW; // dX columns per wave (the chunk's dX tile is [DBK x CI])
    constexpr int HT = XW / 16;                             // 16x16 dX tiles per wave and chunk
    constexpr int LDA = CO + 4;                 // 16-byte aligned rows: one ds_write_b128 per staged float4, ds_read_b128 dX fragments
    constexpr int KPL = CO / 4;                 // dX: k values per lane group kq, CONTIGUOUS (k = kq * KPL + s), see below
    constexpr int TMW = CO / (32 * (NW / 2)), TNW = CI / 64; // 32x32 dW tiles per wave (waves (NW/2) x 2)

    constexpr int PA = DBK * CO / 4 / NT, PB = DBK * CI / 4 / NT;
    static_assert((CO == 64 || CO == 128 || CO == 256) && (CI == 64 || CI == 128) && PA >= 1 && PB >= 1 && HT >= 1, "tile");
    static_assert(!SPLIT || DBK == 16 || ONE, "split: one 32x32x16 k-step of positions per chunk (the one-plane form loops over two)");
    // KSPLIT (256 outputs): in the dX product every wave reads the WHOLE dZ chunk from LDS for its 16 columns -- 192 of the 332 KB of LDS
    // traffic per chunk.  Here a wave takes 32 columns (two tiles) and HALF of K, its partner (wave ^ 4) the other half; each
    // finalises one of the two tiles after adding the partner's partial (8 KB through LDS, one extra barrier per chunk).
    constexpr bool KSPLIT = SPLIT && !ONE && PL == 3 && CO == 256 && CI == 128 && MP_BWD_KSPLIT;
    constexpr int HTW = KSPLIT ? 2 : HT, NSTW = KSPLIT ? CO / 64 : CO / 32;     // weight-plane tiles / k-steps per wave
    // SPLIT: K-packed planes (tr_frag_packed), group stride in halves: 16 dwords mod 64 banks.  The 64-input layers are HBM-bound:
    // a smaller pad (8 dwords: some 2-way conflicts in the transposed reads) lets a third workgroup onto the CU -- more loads in flight
    constexpr bool KSWZ = SPLIT && MP_KSWZ;        // row-swizzled image without pad (tr_frag_packed)
    constexpr int GS = DBK * 8 + (KSWZ ? 0 : ((SPLIT && CI == 64 && MP_SPLIT_WGS == 3) ? 16 : 32));
    __shared__ __attribute__((aligned(16))) float sA[2][SPLIT ? 4 : DBK * LDA];
    __shared__ __attribute__((aligned(16))) float sB[2][SPLIT ? 4 : DBK * CI];
    // ([r4] ONE: a single plane -- the two unused ones were a third of the kernel's LDS.  It is the REGISTERS that keep the bf16 variant's
    // 128 x 128 kernels at one eight-wave workgroup per CU (162 VGPRs; forcing 128 spills 34 dwords: 706 -> 1 100 us; four-wave workgroups
    // take 281): two chunks of loads in flight per CU = the 2.3 TB/s they run at, with every unit idle)
    constexpr int NPLN = ONE ? 1 : PL;
    static_assert(PL == 3 || (PL == 2 && SPLIT && !ONE), "two planes: the split form only");
    __shared__ __attribute__((aligned(16))) __bf16 hA[2][NPLN][SPLIT ? (CO / 8) * GS : 8];   // dZ chunk as (h, m, l) planes
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][NPLN][SPLIT ? (CI / 8) * GS : 8];   // activated input chunk
    __shared__ __attribute__((aligned(16))) float sZ[2][DBK * CI];
    __shared__ float red[2][2][CI];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wrow0 = (wave >> 1) * TMW * 32, wcol0 = (wave & 1) * TNW * 32;
    // [r5] MP_BF_INTERLEAVE: workgroup w walks chunks w, w + grid, w + 2 grid, ... instead of a contiguous range of p_per_block positions: at any
    // moment the chip reads (and writes) one window of each operand, the order a copy kernel sweeps memory in (tools/probes/membw_probe.hip:
    // copies at 256 workgroups 4.8 TB/s over contiguous ranges, 5.8 interleaved).  dW and the BatchNorm sums do not care which positions a
    // workgroup sees; byte offsets into G stay 32-bit (host: mp_bf_interleave_ok).
    const bool il = MP_BF_INTERLEAVE && p_per_block < 0;
    if (p_per_block < 0) p_per_block = -p_per_block;
    const int p0 = il ? 0 : blockIdx.x * p_per_block;
    const int p1 = il ? P : min(P, p0 + p_per_block);
    const int cstep = il ? (int)gridDim.x * DBK : DBK;             // positions from one chunk of this workgroup to its next
    const int cp0 = il ? (int)blockIdx.x * DBK : p0;               // its first chunk
    const int nchunks = il ? ((P + DBK - 1) / DBK - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;

    f32x16 accW[TMW][TNW];
#pragma unroll
    for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
        for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[mi][ni][r] = 0.0f;

    // every thread keeps the same channels for the whole kernel.  SPLIT: a wave stages 4 positions x 64 channels per pass (16 lanes x
    // 16 bytes of one row: 256-byte global segments), so that its ds_write_b64 into the K-packed planes touch every bank twice
    constexpr int NBA = CO / 64, NBB = CI / 64;
    // lane -> (channel quad cq of the wave's 64-channel block, row pr of its 4): lanes 0..31 take quads 0..7 of all four rows, lanes
    // 32..63 quads 8..15 -- a 32-lane pass of the ds_write_b64 then covers (4 groups) x (4 rows) x (2 halves) = 32 distinct bank pairs
    // (the 1 KB dZ rows of the 256-output layer keep 16 lanes = 256 contiguous bytes per row: MP_MAP256)
    constexpr bool WIDE_A = (CO == 256 && !MP_MAP256) || MP_MAPWIDE;
    constexpr bool WIDE_B = WIDE_A;                  // ([r2] same box: 236 -> 225 us with both operands of that layer on the wide mapping)
    const int cq = (lane & 7) + 8 * (lane >> 5), pr = (lane >> 3) & 3;
    const int cqa = WIDE_A ? (lane & 15) : cq, pra = WIDE_A ? (lane >> 4) : pr;
    const int ca = SPLIT ? ((tid >> 6) % NBA) * 64 + 4 * cqa : (tid % (CO / 4)) * 4;
    const int cqb = WIDE_B ? (lane & 15) : cq, prb = WIDE_B ? (lane >> 4) : pr;
    const int cb = SPLIT ? ((tid >> 6) % NBB) * 64 + 4 * cqb : (tid % (CI / 4)) * 4;
    const int ka0 = SPLIT ? ((tid >> 6) / NBA) * 4 + pra : tid / (CO / 4);
    const int kb0 = SPLIT ? ((tid >> 6) / NBB) * 4 + prb : tid / (CI / 4);
    constexpr int KA_STEP = SPLIT ? 4 * (NW / NBA) : NT / (CO / 4), KB_STEP = SPLIT ? 4 * (NW / NBB) : NT / (CI / 4);
    static_assert(!SPLIT || (PA * KA_STEP == DBK && PB * KB_STEP == DBK), "split staging covers the chunk");
    ChanConst ka, kb;
    load_consts<MODE_IN>(IN, cb, kb);
    // SPLIT, 256-output layer: the registers hold 96 weight-plane and 64 dW-accumulator values per lane; the per-channel constants of
    // the staging arithmetic wait in LDS between chunks instead (five ds_read_b128 per chunk, no spill code in the loop)
    constexpr bool LDS_CONSTS = SPLIT && (CO == 256 || (CI == 64 && MP_SPLIT_WGS == 3)) && !is_rc(MODE_IN) && !is_rc(MODE_DZ);
    __shared__ float4 sKA[LDS_CONSTS ? 5 : 1][LDS_CONSTS ? CO / 4 : 1];
    __shared__ float4 sKB[LDS_CONSTS ? 2 : 1][LDS_CONSTS ? CI / 4 : 1];
    struct RSet { Raw4<MODE_DZ> a[PA]; Raw4<MODE_IN> b[PB]; };
    RSet rs0, rs1;            // (rs1: PD2 only)
    auto gload = [&](int pk, RSet& rs) {
        if constexpr ((MP_BF_ABL >> 4) & 1) { if (pk != p0) return; }
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) raw_load<MODE_DZ, ONE>(DZ, p1, pk + ka0 + ps * KA_STEP, ca, rs.a[ps]);
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) raw_load<MODE_IN, ONE>(IN, p1, pk + kb0 + ps * KB_STEP, cb, rs.b[ps]);
    };
    auto sstore = [&](int buf, RSet& rs) {
        auto& ra = rs.a;
        auto& rb = rs.b;
        if constexpr ((MP_BF_ABL >> 5) & 1) { if (buf >= 0) return; }
        if constexpr (LDS_CONSTS) {
            ka.s = sKA[0][ca >> 2]; ka.t = sKA[1][ca >> 2]; ka.a = sKA[2][ca >> 2]; ka.e = sKA[3][ca >> 2]; ka.f = sKA[4][ca >> 2];
            kb.s = sKB[0][cb >> 2]; kb.t = sKB[1][cb >> 2];
        }
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            if constexpr (SPLIT) {
                Split4 sp;
                if constexpr ((MP_BF_ABL >> 3) & 1) { const bf16x4 c = to_bf16x4(ra[ps].z); sp.h = c; sp.m = c; sp.l = c; }
                else sp = splitn<PL>(finish<MODE_DZ>(ra[ps], ka));
                const int o = (ca >> 3) * GS + ((ka0 + ps * KA_STEP) ^ (KSWZ ? kswz(ca >> 3) : 0)) * 8 + (ca & 7);
                *reinterpret_cast<bf16x4*>(&hA[buf][0][o]) = sp.h;
                if constexpr (NPLN >= 2) *reinterpret_cast<bf16x4*>(&hA[buf][1][o]) = sp.m;
                if constexpr (NPLN >= 3) *reinterpret_cast<bf16x4*>(&hA[buf][2][o]) = sp.l;
            } else {
                *reinterpret_cast<float4*>(&sA[buf][(ka0 + ps * KA_STEP) * LDA + ca]) = finish<MODE_DZ>(ra[ps], ka);
            }
        }
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) {
            const int o = (kb0 + ps * KB_STEP) * CI + cb;
            if constexpr (SPLIT) {
                Split4 sp;
                if constexpr ((MP_BF_ABL >> 3) & 1) { const bf16x4 c = to_bf16x4(rb[ps].z); sp.h = c; sp.m = c; sp.l = c; }
                else sp = splitn<PL>(finish<MODE_IN>(rb[ps], kb));
                const int oh = (cb >> 3) * GS + ((kb0 + ps * KB_STEP) ^ (KSWZ ? kswz(cb >> 3) : 0)) * 8 + (cb & 7);
                *reinterpret_cast<bf16x4*>(&hB[buf][0][oh]) = sp.h;
                if constexpr (NPLN >= 2) *reinterpret_cast<bf16x4*>(&hB[buf][1][oh]) = sp.m;
                if constexpr (NPLN >= 3) *reinterpret_cast<bf16x4*>(&hB[buf][2][oh]) = sp.l;
            } else {
                *reinterpret_cast<float4*>(&sB[buf][o]) = finish<MODE_IN>(rb[ps], kb);
            }
            *reinterpret_cast<float4*>(&sZ[buf][o]) = rb[ps].ok ? raw_z<MODE_IN>(rb[ps], kb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    const int l31 = lane & 31;
    // dX tiles of this wave: 32-position chunks: two row tiles x NW/2 column groups; 16-position chunks: rows 0..15, columns
    // (CI/NW)*wave..  The W_l fragments of those columns never change: they live in registers for the whole kernel
    // (v_mfma_f32_16x16x4_f32 B operand: lane (l15, kq) holds W[k][col]), so W_l costs no LDS.  The contraction index is
    // PERMUTED: step s of lane group kq covers k = kq * KPL + s instead of 4 * s + kq, so that a lane's A values of consecutive
    // steps are consecutive floats of its dZ row -- four steps per ds_read_b128, fetched a batch ahead of the MFMAs that use
    // them.  (The sum over k is the same set of products in another order: G differs from the k-ordered chain by rounding.)
    const int xrow0 = DBK == 32 ? (wave / (NW / 2)) * 16 : 0;
    const int xcol0 = DBK == 32 ? (wave % (NW / 2)) * XW : wave * XW;
    float wfrag[SPLIT ? 1 : HT][SPLIT ? 1 : CO / 4];
    bf16x8 wsp[SPLIT ? HTW : 1][SPLIT ? NSTW : 1][PL];  // SPLIT (16x16x32): lane (col, kq) holds W[32*st + 8*kq .. + 7][col] as planes
    const int gcol0 = KSPLIT ? (wave & 3) * 32 : xcol0;      // first dX column of this wave's tiles
    const int gst0 = KSPLIT ? (wave >> 2) * NSTW : 0;        // first k-step of its share of K
    if constexpr (SPLIT) {
#pragma unroll
        for (int h = 0; h < HTW; ++h)
#pragma unroll
            for (int st = 0; st < NSTW; ++st) {
                const float* wp = W + (size_t)(32 * (gst0 + st) + 8 * (lane >> 4)) * CI + gcol0 + 16 * h + (lane & 15);
                const Split4 lo = splitn<PL>(make_float4(wp[0], wp[CI], wp[2 * CI], wp[3 * CI]));
                const Split4 hi = splitn<PL>(make_float4(wp[4 * CI], wp[5 * CI], wp[6 * CI], wp[7 * CI]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wsp[h][st][0][i] = lo.h[i]; wsp[h][st][0][4 + i] = hi.h[i];
                    wsp[h][st][1][i] = lo.m[i]; wsp[h][st][1][4 + i] = hi.m[i];
                    if constexpr (PL == 3) { wsp[h][st][2][i] = lo.l[i]; wsp[h][st][2][4 + i] = hi.l[i]; }
                }
            }
    } else {
#pragma unroll
        for (int h = 0; h < HT; ++h)
#pragma unroll
            for (int st = 0; st < CO / 4; ++st) wfrag[h][st] = W[(size_t)((lane >> 4) * KPL + st) * CI + xcol0 + 16 * h + (lane & 15)];
    }
    typedef float f2_ __attribute__((ext_vector_type(2)));
    float spx[HT], tpx[HT];                     // this lane's G columns
    f2_ sx1[HT], sx2[HT];                       // BatchNorm-backward sums of those columns, two row slots each
    const int ecol0 = KSPLIT ? gcol0 + 16 * (wave >> 2) : xcol0;   // first column of the tile(s) this wave FINALISES
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        const int col = ecol0 + 16 * h + (lane & 15);
        spx[h] = IN.s[col];
        tpx[h] = IN.t[col];
        sx1[h] = f2_{0.0f, 0.0f};
        sx2[h] = f2_{0.0f, 0.0f};
    }
    // G_{l-1} rows of this workgroup through a buffer resource: lane part of the offset in one VGPR, row part as immediates
    // ONE: G_{l-1} is STORED as bf16 (see fwd_chunk_kernel: a lane pair exchanges one value per row pair, one dword store per lane)
    constexpr int GB = ONE ? 2 : 4;
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(G) + (size_t)p0 * CI * GB, 0, (p1 - p0) * CI * GB, 0x00020000);
    int goff = ONE ? ((xrow0 + 4 * (lane >> 4)) * CI + ecol0 + ((lane & 15) & ~1)) * 2 + ((lane & 1) ? CI * 2 : 0)
                   : ((xrow0 + 4 * (lane >> 4)) * CI + ecol0 + (lane & 15)) * 4;
    goff += (cp0 - p0) * CI * GB;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 xbuf[KSPLIT ? NW : 1][64];      // KSPLIT: the partial of the tile the partner wave finalises
    f32x4 ax[HT];                                    // the finished dX tile(s) of this wave, between g_mfma and g_epi

    constexpr int PD2M = ONE ? MP_PD2_ONE : MP_PD2;
    constexpr bool PD2 = PD2M && SPLIT && !KSPLIT && ((PD2M >> (NT == 512 ? (CO == 256 ? 2 : 1) : 0)) & 1);
   // bit 0: 256-thread kernels, 1: <.,128,128>, 2: <.,256,128>
    constexpr bool DESYNC = MP_DESYNC && NT == 512 && CO == 256 && SPLIT && !KSPLIT && !PD2 && DBK == 16;   // (<.,128,128>: 128 -> 134 us with it, 124 -> 116 with PD2)
    const int half = DESYNC ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;
    gload(cp0, rs0);
    bn_prologue(DZ.bn, bn_lds, CO, 0, CO, blockIdx.x == 0);     // (behind the first chunk's loads: its slot reads share their latency)
    load_consts<MODE_DZ>(DZ, ca, ka, bn_lds, CO);
    if constexpr (LDS_CONSTS) {
        sKA[0][ca >> 2] = ka.s; sKA[1][ca >> 2] = ka.t; sKA[2][ca >> 2] = ka.a; sKA[3][ca >> 2] = ka.e; sKA[4][ca >> 2] = ka.f;
        sKB[0][cb >> 2] = kb.s; sKB[1][cb >> 2] = kb.t;
    }
    sstore(0, rs0);
    if (DESYNC && half && nchunks > 1) { gload(cp0 + cstep, rs0); sstore(1, rs0); }
    if constexpr (PD2) {     // chunks 1 and 2 on their way before the first product
        if (nchunks > 1) gload(cp0 + cstep, rs0);
        if (nchunks > 2) gload(cp0 + 2 * cstep, rs1);
    }
    __syncthreads();
    // one chunk: products of chunk kc (plane buffer kc & 1), then chunk kc + 1 (held by `rs`) is staged into the other buffer
    auto body = [&](const int kc, RSet& rs) {
        const int cur = kc & 1;
        if (!PD2 && kc + 1 + half < nchunks) gload(cp0 + (kc + 1 + half) * cstep, rs);
        auto do_dw = [&]() {
        if constexpr (SPLIT && ONE) {   // one plane: dW += bf16(dZ)^T * bf16(act(Z_{l-1})), one k-step per 16 positions of the chunk
#pragma unroll
            for (int k0 = 0; k0 < DBK; k0 += 16) {
                bf16x8 fb[TNW], fa[TMW];
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) fb[ni] = tr_frag_packed<GS, KSWZ>(hB[cur][0], k0, wcol0 + ni * 32);
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][0], k0, wrow0 + mi * 32);
                if constexpr (MP_TR_FENCE & 1) tr_fence();
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], accW[mi][ni], 0, 0, 0);
            }
        } else if constexpr (SPLIT && PL == 2) {   // [r5] two planes: h*m, h*h, m*h
            bf16x8 fb[2][TNW], fa[TMW];
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) fb[pl][ni] = tr_frag_packed<GS, KSWZ>(hB[cur][pl], 0, wcol0 + ni * 32);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][0], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][1], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[0][ni], accW[mi][ni], 0, 0, 0);
        } else if constexpr (SPLIT) {   // dW += dZ^T * act(Z_{l-1}): one k-step of 16 positions, six plane products per tile
            // fragments in the order they are consumed (one dZ plane live at a time): l*h, h*l, h*m, h*h, m*m, m*h
            bf16x8 fb[3][TNW], fa[TMW];
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni) fb[0][ni] = tr_frag_packed<GS, KSWZ>(hB[cur][0], 0, wcol0 + ni * 32);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][2], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[0][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][0], 0, wrow0 + mi * 32);
#pragma unroll
            for (int pl = 2; pl >= 1; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) fb[pl][ni] = tr_frag_packed<GS, KSWZ>(hB[cur][pl], 0, wcol0 + ni * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, KSWZ>(hA[cur][1], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 1) & 1) tr_fence();
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
        } else {
            mma_chunk_pipelined<true, true, LDA, CI, TMW, TNW, DBK>(sA[cur], sB[cur], wrow0, wcol0, accW);   // dW += dZ^T * act(Z_{l-1})
        }
        };
        auto g_mfma = [&]() {
        {   // G_{l-1} chunk [DBK x 64] = dZ [DBK x CO] * W_l [CO x 64] as 16x16 tiles, HT per wave (v_mfma_f32_16x16x4_f32:
            // with 32x32 tiles only one or two waves would have work)
#pragma unroll
            for (int h = 0; h < HT; ++h) ax[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int l15 = lane & 15, kq = lane >> 4;
            if constexpr (KSPLIT) {   // two tiles x half of K; the tile this wave does not finalise goes to the partner through LDS
                f32x4 a2[2], c2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) { a2[h] = f32x4{0.f, 0.f, 0.f, 0.f}; c2[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                const int ao = (4 * gst0 + kq) * GS + ((xrow0 + l15) ^ (KSWZ ? kswz(kq) : 0)) * 8;
                bf16x8 af[2][3];
#pragma unroll
                for (int pl = 0; pl < NPLN; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao]);
#pragma unroll
                for (int st = 0; st < NSTW; ++st) {
                    if (st + 1 < NSTW) {
#pragma unroll
                        for (int pl = 0; pl < NPLN; ++pl) af[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao + 4 * (st + 1) * GS]);
                    }
                    const bf16x8 ah = af[st & 1][0], am = af[st & 1][ONE ? 0 : 1], al = af[st & 1][ONE ? 0 : 2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        c2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wsp[h][st][0], c2[h], 0, 0, 0);
                        a2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][0], a2[h], 0, 0, 0);
                        c2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][2], c2[h], 0, 0, 0);
                        c2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][1], c2[h], 0, 0, 0);
                        c2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][0], c2[h], 0, 0, 0);
                        c2[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][1], c2[h], 0, 0, 0);
                    }
                }
                const int mine = wave >> 2;
                ax[0] = mine ? a2[1] + c2[1] : a2[0] + c2[0];
                xbuf[wave][lane] = mine ? a2[0] + c2[0] : a2[1] + c2[1];
            } else if constexpr (SPLIT) {   // v_mfma_f32_16x16x32_bf16: lane (row, kq) holds dZ[row][32*st + 8*kq .. + 7] -- one packed group
                f32x4 cx[HT];
#pragma unroll
                for (int h = 0; h < HT; ++h) cx[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int ao = kq * GS + ((xrow0 + l15) ^ (KSWZ ? kswz(kq) : 0)) * 8;
                bf16x8 af[2][3];       // the fragments of k-step st + 1 are requested before the MFMAs of step st are issued
#pragma unroll
                for (int pl = 0; pl < NPLN; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao]);
#pragma unroll
                for (int st = 0; st < CO / 32; ++st) {
                    if (st + 1 < CO / 32) {
#pragma unroll
                        for (int pl = 0; pl < NPLN; ++pl) af[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao + 4 * (st + 1) * GS]);
                    }
                    const bf16x8 ah = af[st & 1][0], am = af[st & 1][NPLN >= 2 ? 1 : 0], al = af[st & 1][NPLN >= 3 ? 2 : 0];
#pragma unroll
                    for (int h = 0; h < HT; ++h) {
                        if constexpr (NPLN == 3) cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wsp[h][st][0], cx[h], 0, 0, 0);
                        ax[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][0], ax[h], 0, 0, 0);
                        if constexpr (NPLN == 3) {
                            cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][2], cx[h], 0, 0, 0);
                            cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][1], cx[h], 0, 0, 0);
                        }
                        if constexpr (NPLN >= 2) {
                            cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][0], cx[h], 0, 0, 0);
                            cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][1], cx[h], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < HT; ++h) ax[h] += cx[h];
            }
            const float4* arow = reinterpret_cast<const float4*>(sA[cur] + (SPLIT ? 0 : (xrow0 + l15) * LDA + kq * KPL));   // A[row][k = kq*KPL + s]
            constexpr int AB = 2, NB = SPLIT ? 0 : KPL / (4 * AB);      // batches of AB float4 = 8 steps, fetched one batch ahead
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
#pragma unroll
                    for (int h = 0; h < HT; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wfrag[h][st], ax[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < HT; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wfrag[h][st + 1], ax[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < HT; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wfrag[h][st + 2], ax[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < HT; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wfrag[h][st + 3], ax[h], 0, 0, 0);
                }
            }
        }
        };
        auto g_epi = [&]() {
        {
            const int l15 = lane & 15, kq = lane >> 4;
            if constexpr (KSPLIT) ax[0] += xbuf[wave ^ 4][lane];       // (behind the barrier that follows the partner's write)
            // epilogue written for instruction count (see fwd_chunk_kernel): buffer stores (rows past the workgroup's last
            // position are dropped by the range check; their dZ rows were staged as zeros, so they add nothing to the sums),
            // the four rows of a lane as two register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                const float* zr = sZ[cur] + (xrow0 + 4 * kq) * CI + ecol0 + 16 * h + l15;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    if constexpr (ONE) {
                        const bool odd = lane & 1;
                        const float got = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? ax[h][i] : ax[h][i + 1]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                        __builtin_amdgcn_raw_buffer_store_b32(odd ? pack_bf16(got, ax[h][i + 1]) : pack_bf16(ax[h][i], got), grsrc, goff + 16 * h * 2, i * CI * 2, MP_STORE_AUX);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ax[h][i]), grsrc, goff + 16 * h * 4, i * CI * 4, MP_STORE_AUX);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ax[h][i + 1]), grsrc, goff + 16 * h * 4, (i + 1) * CI * 4, MP_STORE_AUX);
                    }
                    const f2 zp = {zr[i * CI], zr[(i + 1) * CI]};
                    const f2 y = zp * f2{spx[h], spx[h]} + f2{tpx[h], tpx[h]};
                    const f2 dy = {y.x > 0.0f ? ax[h][i] : 0.0f, y.y > 0.0f ? ax[h][i + 1] : 0.0f};
                    sx1[h] += dy;
                    sx2[h] += dy * zp;
                }
            }
            goff += cstep * CI * GB;
        }
        };
        // (tried: the two halves of the workgroup walking the two products in opposite order, so that only four waves at a time
        // read the dZ planes for G -- 254 -> 315 us on the 256-output layer: twice the loop code, spills again)
        if constexpr (KSPLIT) {
            g_mfma();
            do_dw();
            __syncthreads();      // every partial is in xbuf
            g_epi();
            if (kc + 1 < nchunks) sstore(cur ^ 1, rs);
        } else if constexpr (DESYNC) {
            // [r3] eight waves, two per SIMD (wave w and w + 4), one barrier per chunk: left alone both waves of a SIMD run the matrix
            // phase together and then the staging arithmetic together -- the matrix pipe idles through the second, the VALU through the
            // first (profiles/r02_sq_counters.md: MFMA 37 % + VALU 27 % of the cycles, one after the other).  Here waves 4..7 take
            // their barrier BETWEEN the products and the staging instead of behind both, and stage one chunk further ahead (chunk
            // kc + 2 into the buffer the products of chunk kc just left): past the first chunk one half's MFMAs run under the
            // other half's VALU work on every SIMD.  Same arithmetic; every wave still passes one barrier per chunk.
            do_dw();
            g_mfma();
            g_epi();
            if (half) __syncthreads();
            if (kc + 1 + half < nchunks) sstore((kc + 1 + half) & 1, rs);
            if (!half) __syncthreads();
        } else {
            if constexpr (!(MP_BF_ABL & 1)) do_dw();
            if constexpr (!((MP_BF_ABL >> 1) & 1)) g_mfma();

            else {
#pragma unroll
                for (int h = 0; h < HT; ++h) ax[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (!((MP_BF_ABL >> 2) & 1)) g_epi();
            if (kc + 1 < nchunks) sstore(cur ^ 1, rs);
            // PD2: the set just emptied is refilled at once with the chunk it stages two iterations from now -- two chunks of
            // loads in flight per workgroup instead of one (an iteration of these kernels lasts about one loaded-HBM round trip)
            if (PD2 && kc + 3 < nchunks) gload(cp0 + (kc + 3) * cstep, rs);
        }
        if constexpr (!DESYNC) __syncthreads();
    };
    if constexpr (PD2) {
        for (int kc = 0; kc < nchunks; kc += 2) {
            body(kc, rs0);
            if (kc + 1 < nchunks) body(kc + 1, rs1);
        }
    } else {
        for (int kc = 0; kc < nchunks; ++kc) body(kc, rs0);
    }
    // BatchNorm-backward partial sums of layer l-1: the four 16-lane row groups of a wave, then (32-position chunks) the two
    // waves that share a column half
    for (int e = tid; e < 2 * 2 * CI; e += NT) (&red[0][0][0])[e] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        float s1x = sx1[h].x + sx1[h].y, s2x = sx2[h].x + sx2[h].y;
        s1x += __shfl_xor(s1x, 16, 64); s1x += __shfl_xor(s1x, 32, 64);
        s2x += __shfl_xor(s2x, 16, 64); s2x += __shfl_xor(s2x, 32, 64);
        if (lane < 16) {
            const int col = ecol0 + 16 * h + lane;
            red[DBK == 32 ? (wave / (NW / 2)) : 0][0][col] = s1x;
            red[DBK == 32 ? (wave / (NW / 2)) : 0][1][col] = s2x;
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * CI; e += NT) {
        const int st = e / CI, c = e - st * CI;
        const float v = red[0][st][c] + red[1][st][c];
        if (partials.slots) atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + st) * CI + c, (double)v);
        else partials.rows[((size_t)blockIdx.x * 2 + st) * CI + c] = v;
    }
    // dW
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

// =================================================================================================================
// [r3] Kernel 4b: the fused backward of the 128-input layers with the two products on DIFFERENT waves (MP_BF_ROLES; default for 256 outputs).
// bwd_fused_kernel gives each of its eight waves a slice of both products; the 96 (256 outputs) weight-fragment registers of the dX
// slice and the 64 accumulators of the dW slice leave no room to fetch operand fragments more than one step ahead, and the two waves
// of a SIMD walk identical phases in lock step (tools/bwd_ablate.sh: each product costs ~1.8x its matrix-pipe time).  Here waves 0..3
// own the dX product (32 columns each: the A fragments of the dZ chunk are read 4x per chunk instead of 8x, 12 MFMAs per fragment
// triple instead of 6) and waves 4..7 the dW product (8 or 4 tiles each, every fragment plane fetched once): one wave of each kind
// per SIMD.  Staging: all eight waves (128 outputs) or, where the dX waves have no registers left (256 outputs: 192 of them hold
// weight planes), the four dW waves.  Same chunk images (row-swizzled K-packed planes), same arithmetic per product.
// =================================================================================================================
#ifndef MP_ROLES_SPLITSTAGE
#define MP_ROLES_SPLITSTAGE 0       // (1: the dX waves stage the input operand -- 13 spilled registers, 203 -> 225 us)
#endif
#ifndef MP_ROLES_BALL
#define MP_ROLES_BALL 1             // [r5] two planes: every wave stages a piece of the input operand (see BALL)
#endif
#ifndef MP_ROLES_MFMA_ORDER
#define MP_ROLES_MFMA_ORDER 0
#endif
#ifndef MP_ROLES_BEARLY
#define MP_ROLES_BEARLY 1
#endif
#ifndef MP_ROLES_PD2
#define MP_ROLES_PD2 0              // 256 outputs: two chunks of loads in flight in the staging (dW) waves (13 spilled registers: 205 -> 270 us)
#endif
#ifndef MP_ROLES_PRIO
#define MP_ROLES_PRIO 1             // s_setprio for one kind of wave: 1 the dW (staging) waves (they are the longer chain: 204.8 -> 200.3 us), 2 the dX waves (no change)
#endif
// -DMP_ROLES_TIMING: per-phase s_memtime sums of bwd_roles_kernel (tools/roles_timing.sh): [kind: 0 dX wave, 1 dW wave][phase] in shader cycles,
// summed over all waves of a launch; mp_debug_roles_times() copies and clears them.  phases dX: 0 fragment loop + MFMAs, 1 epilogue, 2 barrier wait,
// 3 whole loop; dW: 0 fragment reads + MFMAs, 1 staging, 2 barrier wait, 3 whole loop, 4 the gload issue
#ifdef MP_ROLES_TIMING
__device__ unsigned long long g_roles_t[2][8];
#define RT_DECL unsigned long long rt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long rt0_ = __builtin_readcyclecounter(), rtl_ = rt0_
#define RT_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); rt_[i] += n_ - rtl_; rtl_ = n_; }
#define RT_FLUSH(kind) { rt_[3] = __builtin_readcyclecounter() - rt0_; if (lane == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_roles_t[kind][i_], rt_[i_]); } }
#else
#define RT_DECL
#define RT_MARK(i)
#define RT_FLUSH(kind)
#endif
template <int MODE_DZ, int CO, int PL = 3>      // PL: operand planes (3: h, m, l, six products; 2: h, m, three products -- split2)
__global__ __launch_bounds__(512, 1) void bwd_roles_kernel(PosOperand DZ, PosOperand IN, int P, int p_per_block,
                                                           const float* __restrict__ W, float* __restrict__ dW,
                                                           float* __restrict__ G, BnOut partials)
{
    constexpr int CI = 128, DBK = 16, GS = DBK * 8, MODE_IN = SRC_ACT;
    constexpr bool ALLSTAGE = CO == 128;
    // 256 outputs: the dX waves hold 192 registers of weight planes -- they stage the (smaller) input operand only when MP_ROLES_SPLITSTAGE,
    // the dW waves the dZ operand (or both)
    constexpr bool SPLITSTAGE = !ALLSTAGE && MP_ROLES_SPLITSTAGE;
    // [r5] two planes: the input operand (a third of the staging) is staged by ALL EIGHT waves, one 16-byte piece per thread -- with three
    // products per fp32 product the dX waves wait ~900 of a chunk's 3 800 cycles for the dW + staging waves (tools/roles_timing.sh), and
    // they hold 128 registers of weight planes instead of 192
    constexpr bool BALL = !ALLSTAGE && !SPLITSTAGE && PL == 2 && MP_ROLES_BALL;
    constexpr int NTS = ALLSTAGE ? 512 : 256, NWS = NTS / 64;          // staging threads / waves (per operand)
    constexpr int NWSB = BALL ? 8 : NWS;
    constexpr int NBA = CO / 64, NBB = CI / 64;
    constexpr int PA = DBK * CO / 4 / NTS, PB = DBK * CI / 4 / (NWSB * 64);
    constexpr int KA_STEP = 4 * (NWS / NBA), KB_STEP = 4 * (NWSB / NBB);
    static_assert(PA * KA_STEP == DBK && PB * KB_STEP == DBK && (CO == 128 || CO == 256), "staging covers the chunk");
    constexpr int NST = CO / 32;                                        // k-steps of the dX product
    constexpr int TMW = CO / 64, TNW = 2;                               // 32 x 32 dW tiles per dW wave (waves 2 x 2 over [CO x 128])
    __shared__ __attribute__((aligned(16))) __bf16 hA[2][PL][(CO / 8) * GS];
    __shared__ __attribute__((aligned(16))) __bf16 hB[2][PL][(CI / 8) * GS];
    __shared__ __attribute__((aligned(16))) float sZ[2][DBK * CI];
    __shared__ float red[2][CI];
    __shared__ float4 sKA[5][CO / 4];
    __shared__ float4 sKB[2][CI / 4];
    // (a, e, f) of dZ_l when this kernel is their first consumer: derived straight into the rows of sKA that hold them between chunks
    bn_prologue(DZ.bn, reinterpret_cast<float*>(&sKA[2][0]), CO, 0, CO, blockIdx.x == 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_dx = wave < 4;
    const bool stage_a = ALLSTAGE || !role_dx;                          // this wave stages dZ rows
    const bool stage_b = ALLSTAGE || BALL || (SPLITSTAGE ? role_dx : !role_dx); // ... input rows
    const int p0 = blockIdx.x * p_per_block;
    const int p1 = min(P, p0 + p_per_block);
    const int nchunks = (p1 - p0 + DBK - 1) / DBK;
    if (nchunks <= 0) return;

    // ---- staging (16 lanes x 16 bytes per row: 256-byte global segments) --------------------------------------------------------
    const int sw = ALLSTAGE ? wave : (wave & 3), swb = BALL ? wave : sw;
    const int ca = (sw % NBA) * 64 + 4 * (lane & 15), ka0 = (sw / NBA) * 4 + (lane >> 4);
    const int cb = (swb % NBB) * 64 + 4 * (lane & 15), kb0 = (swb / NBB) * 4 + (lane >> 4);
    if (stage_a) {      // per-channel constants of the staging arithmetic wait in LDS between chunks
        sKA[0][ca >> 2] = ld4(DZ.s + ca);
        sKA[1][ca >> 2] = ld4(DZ.t + ca);
        if (DZ.bn.slots == nullptr) { sKA[2][ca >> 2] = ld4(DZ.a + ca); sKA[3][ca >> 2] = ld4(DZ.e + ca); sKA[4][ca >> 2] = ld4(DZ.f + ca); }
    }
    if (stage_b) {
        ChanConst kb;
        load_consts<MODE_IN>(IN, cb, kb);
        sKB[0][cb >> 2] = kb.s; sKB[1][cb >> 2] = kb.t;
    }
    // dZ rows of a chunk.  Pooled layer (the host guarantees K = 2^kshift >= 16, so a chunk lies inside ONE group): the pooled gradient
    // and the arg-max of the thread's four channels are the same for all its rows -- loaded once per chunk, not once per row
    constexpr bool POOLED = MODE_DZ == SRC_DZ_POOLED;
    struct RSetA {
        Raw4<MODE_DZ> a[POOLED ? 1 : PA];       // pooled: a[0] carries (g, ak) and the chunk's first member index
        float4 z[POOLED ? PA : 1];
    };
    struct RSetB { Raw4<MODE_IN> b[PB]; };
    auto gload_a = [&](int pk, RSetA& rs) {
        if constexpr (POOLED) {
            const unsigned off = ((unsigned)pk >> DZ.kshift) * (unsigned)CO + (unsigned)ca;
            rs.a[0].g = ld4(DZ.g + off);
            rs.a[0].ak = *reinterpret_cast<const int4*>(DZ.argk + off);
            rs.a[0].kk = pk & (DZ.K - 1);
#pragma unroll
            for (int ps = 0; ps < PA; ++ps) {
                const int p = pk + ka0 + ps * KA_STEP;
                rs.z[ps] = ld4(DZ.x + (size_t)((unsigned)(p < p1 ? p : p0) * (unsigned)CO + (unsigned)ca));
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < PA; ++ps) raw_load<MODE_DZ>(DZ, p1, pk + ka0 + ps * KA_STEP, ca, rs.a[ps]);
        }
    };
    auto gload_b = [&](int pk, RSetB& rs) {
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) raw_load<MODE_IN>(IN, p1, pk + kb0 + ps * KB_STEP, cb, rs.b[ps]);
    };
    auto sstore_a = [&](int buf, int pk, RSetA& rs) {
        ChanConst ka;
        ka.s = sKA[0][ca >> 2]; ka.t = sKA[1][ca >> 2]; ka.a = sKA[2][ca >> 2]; ka.e = sKA[3][ca >> 2]; ka.f = sKA[4][ca >> 2];
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            float4 dz;
            if constexpr (POOLED) {
                const int kk = rs.a[0].kk + ka0 + ps * KA_STEP;
                const float4 z = rs.z[ps], g = rs.a[0].g;
                const int4 ak = rs.a[0].ak;
                dz.x = xf1<MODE_DZ>(z.x, ak.x == kk ? g.x : 0.0f, ka.s.x, ka.t.x, ka.a.x, ka.e.x, ka.f.x);
                dz.y = xf1<MODE_DZ>(z.y, ak.y == kk ? g.y : 0.0f, ka.s.y, ka.t.y, ka.a.y, ka.e.y, ka.f.y);
                dz.z = xf1<MODE_DZ>(z.z, ak.z == kk ? g.z : 0.0f, ka.s.z, ka.t.z, ka.a.z, ka.e.z, ka.f.z);
                dz.w = xf1<MODE_DZ>(z.w, ak.w == kk ? g.w : 0.0f, ka.s.w, ka.t.w, ka.a.w, ka.e.w, ka.f.w);
                if (pk + ka0 + ps * KA_STEP >= p1) dz = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                dz = finish<MODE_DZ>(rs.a[ps], ka);
            }
            const Split4 sp = splitn<PL>(dz);
            const int o = (ca >> 3) * GS + ((ka0 + ps * KA_STEP) ^ kswz(ca >> 3)) * 8 + (ca & 7);
            *reinterpret_cast<bf16x4*>(&hA[buf][0][o]) = sp.h;
            *reinterpret_cast<bf16x4*>(&hA[buf][1][o]) = sp.m;
            if constexpr (PL == 3) *reinterpret_cast<bf16x4*>(&hA[buf][2][o]) = sp.l;
        }
    };
    auto sstore_b = [&](int buf, RSetB& rs) {
        ChanConst kb;
        kb.s = sKB[0][cb >> 2]; kb.t = sKB[1][cb >> 2];
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) {
            const Split4 sp = splitn<PL>(finish<MODE_IN>(rs.b[ps], kb));
            const int oh = (cb >> 3) * GS + ((kb0 + ps * KB_STEP) ^ kswz(cb >> 3)) * 8 + (cb & 7);
            *reinterpret_cast<bf16x4*>(&hB[buf][0][oh]) = sp.h;
            *reinterpret_cast<bf16x4*>(&hB[buf][1][oh]) = sp.m;
            if constexpr (PL == 3) *reinterpret_cast<bf16x4*>(&hB[buf][2][oh]) = sp.l;
            *reinterpret_cast<float4*>(&sZ[buf][(kb0 + ps * KB_STEP) * CI + cb]) = rs.b[ps].ok ? rs.b[ps].z : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    __syncthreads();                                   // the constants are in LDS

    if (role_dx) {
        // ================= waves 0..3: G_{l-1} chunk [16 x 128] = dZ [16 x CO] * W_l [CO x 128], 32 columns (two 16 x 16 tiles) per wave
        if constexpr (MP_ROLES_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        const int l15 = lane & 15, kq = lane >> 4;
        const int xcol0 = wave * 32;
        bf16x8 wsp[2][NST][PL];                        // lane (col, kq) holds W[32 st + 8 kq .. + 7][col] as (h, m, l) planes
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const float* wp = W + (size_t)(32 * st + 8 * kq) * CI + xcol0 + 16 * h + l15;
                const Split4 lo = splitn<PL>(make_float4(wp[0], wp[CI], wp[2 * CI], wp[3 * CI]));
                const Split4 hi = splitn<PL>(make_float4(wp[4 * CI], wp[5 * CI], wp[6 * CI], wp[7 * CI]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    wsp[h][st][0][i] = lo.h[i]; wsp[h][st][0][4 + i] = hi.h[i];
                    wsp[h][st][1][i] = lo.m[i]; wsp[h][st][1][4 + i] = hi.m[i];
                    if constexpr (PL == 3) { wsp[h][st][2][i] = lo.l[i]; wsp[h][st][2][4 + i] = hi.l[i]; }
                }
            }
        float spx[2], tpx[2];
        f2 sx1[2], sx2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = xcol0 + 16 * h + l15;
            spx[h] = IN.s[col];
            tpx[h] = IN.t[col];
            sx1[h] = f2{0.0f, 0.0f};
            sx2[h] = f2{0.0f, 0.0f};
        }
        const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(G + (size_t)p0 * CI, 0, (p1 - p0) * CI * 4, 0x00020000);
        int goff = ((4 * kq) * CI + xcol0 + l15) * 4;
        RSetA ra_;
        RSetB rb_;
        if (ALLSTAGE) { gload_a(p0, ra_); sstore_a(0, p0, ra_); }
        if (ALLSTAGE || SPLITSTAGE || BALL) { gload_b(p0, rb_); sstore_b(0, rb_); }
        __syncthreads();
        RT_DECL;
        for (int kc = 0; kc < nchunks; ++kc) {
            const int cur = kc & 1;
            RT_MARK(5);
            if (ALLSTAGE && kc + 1 < nchunks) gload_a(p0 + (kc + 1) * DBK, ra_);
            if ((ALLSTAGE || SPLITSTAGE || BALL) && kc + 1 < nchunks) gload_b(p0 + (kc + 1) * DBK, rb_);
            f32x4 ax[2], cx[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) { ax[h] = f32x4{0.f, 0.f, 0.f, 0.f}; cx[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            const int ao = kq * GS + (l15 ^ kswz(kq)) * 8;
            bf16x8 af[2][PL];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao]);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) {
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl) af[(st + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(&hA[cur][pl][ao + 4 * (st + 1) * GS]);
                }
                const bf16x8 ah = af[st & 1][0], am = af[st & 1][1], al = af[st & 1][PL - 1];
                if constexpr (PL == 2 && MP_ROLES_MFMA_ORDER == 1) {       // (four accumulators: the two products into cx[h] stand three instructions apart)
#pragma unroll
                    for (int h = 0; h < 2; ++h) cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][0], cx[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h) ax[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][0], ax[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h) cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][1], cx[h], 0, 0, 0);
                } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (PL == 3) cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wsp[h][st][0], cx[h], 0, 0, 0);
                    ax[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][0], ax[h], 0, 0, 0);
                    if constexpr (PL == 3) {
                        cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][PL - 1], cx[h], 0, 0, 0);
                        cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][1], cx[h], 0, 0, 0);
                    }
                    cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wsp[h][st][0], cx[h], 0, 0, 0);
                    cx[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wsp[h][st][1], cx[h], 0, 0, 0);
                }
                }
            }
            RT_MARK(0);
            // (BALL: this wave's piece of the next input chunk goes to LDS BEFORE the epilogue's stores are issued -- behind them the wait
            // for its load also waited for their acknowledgements: 945 cycles for one 16-byte piece, tools/roles_timing.sh)
            if constexpr (BALL && MP_ROLES_BEARLY) { if (kc + 1 < nchunks) sstore_b(cur ^ 1, rb_); }
            RT_MARK(4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ax[h] += cx[h];
                const float* zr = sZ[cur] + (4 * kq) * CI + xcol0 + 16 * h + l15;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ax[h][i]), grsrc, goff + 16 * h * 4, i * CI * 4, MP_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ax[h][i + 1]), grsrc, goff + 16 * h * 4, (i + 1) * CI * 4, MP_STORE_AUX);
                    const f2 zp = {zr[i * CI], zr[(i + 1) * CI]};
                    const f2 y = zp * f2{spx[h], spx[h]} + f2{tpx[h], tpx[h]};
                    const f2 dy = {y.x > 0.0f ? ax[h][i] : 0.0f, y.y > 0.0f ? ax[h][i + 1] : 0.0f};
                    sx1[h] += dy;
                    sx2[h] += dy * zp;
                }
            }
            goff += DBK * CI * 4;
            RT_MARK(1);
            if (ALLSTAGE && kc + 1 < nchunks) sstore_a(cur ^ 1, p0 + (kc + 1) * DBK, ra_);
            if ((ALLSTAGE || SPLITSTAGE || (BALL && !MP_ROLES_BEARLY)) && kc + 1 < nchunks) sstore_b(cur ^ 1, rb_);
            RT_MARK(4);
            __syncthreads();
            RT_MARK(2);
        }
        RT_FLUSH(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {       // BatchNorm-backward partial sums of layer l-1: the four 16-lane row groups of the wave
            float s1x = sx1[h].x + sx1[h].y, s2x = sx2[h].x + sx2[h].y;
            s1x += __shfl_xor(s1x, 16, 64); s1x += __shfl_xor(s1x, 32, 64);
            s2x += __shfl_xor(s2x, 16, 64); s2x += __shfl_xor(s2x, 32, 64);
            if (lane < 16) {
                red[0][xcol0 + 16 * h + lane] = s1x;
                red[1][xcol0 + 16 * h + lane] = s2x;
            }
        }
    } else {
        // ================= waves 4..7: dW [CO x 128] += dZ^T * act(Z_{l-1}), one k-step of 16 positions per chunk, TMW x 2 tiles per wave
        if constexpr (MP_ROLES_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        const int w = wave - 4, l31 = lane & 31;
        const int wrow0 = (w >> 1) * (CO / 2), wcol0 = (w & 1) * 64;
        f32x16 accW[TMW][TNW];
#pragma unroll
        for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[mi][ni][r] = 0.0f;
        constexpr bool PD2R = MP_ROLES_PD2 && !ALLSTAGE && !SPLITSTAGE;     // two register sets, the loop unrolled by two (as bwd_fused_kernel's PD2)
        RSetA ra0, ra1;
        RSetB rb0, rb1;
        auto gload = [&](int pk, RSetA& ra_, RSetB& rb_) { gload_a(pk, ra_); if (!SPLITSTAGE) gload_b(pk, rb_); };
        auto sstore = [&](int buf, int pk, RSetA& ra_, RSetB& rb_) { sstore_a(buf, pk, ra_); if (!SPLITSTAGE) sstore_b(buf, rb_); };
        gload(p0, ra0, rb0);
        sstore(0, p0, ra0, rb0);
        if constexpr (PD2R) {
            if (nchunks > 1) gload(p0 + DBK, ra0, rb0);
            if (nchunks > 2) gload(p0 + 2 * DBK, ra1, rb1);
        }
        __syncthreads();
        RT_DECL;
        auto body = [&](const int kc, RSetA& ra_, RSetB& rb_) {        // (ra_, rb_) hold chunk kc + 1
            const int cur = kc & 1;
            RT_MARK(5);
            if (!PD2R && kc + 1 < nchunks) gload(p0 + (kc + 1) * DBK, ra_, rb_);
            RT_MARK(4);
            if constexpr (PL == 2) {      // [r5] two planes: h*m, h*h, m*h
                bf16x8 fb[2][TNW], fa[TMW];
#pragma unroll
                for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) fb[pl][ni] = tr_frag_packed<GS, true>(hB[cur][pl], 0, wcol0 + ni * 32);
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA[cur][0], 0, wrow0 + mi * 32);
                if constexpr ((MP_TR_FENCE >> 2) & 1) tr_fence();
#pragma unroll
                for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                    for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA[cur][1], 0, wrow0 + mi * 32);
                if constexpr ((MP_TR_FENCE >> 2) & 1) tr_fence();
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[0][ni], accW[mi][ni], 0, 0, 0);
            } else {
            // fragments in the order they are consumed (one dZ plane live at a time): l*h, h*l, h*m, h*h, m*m, m*h
            bf16x8 fb[3][TNW], fa[TMW];
#pragma unroll
            for (int ni = 0; ni < TNW; ++ni) fb[0][ni] = tr_frag_packed<GS, true>(hB[cur][0], 0, wcol0 + ni * 32);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA[cur][PL - 1], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 2) & 1) tr_fence();
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[0][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA[cur][0], 0, wrow0 + mi * 32);
#pragma unroll
            for (int pl = 2; pl >= 1; --pl)
#pragma unroll
                for (int ni = 0; ni < TNW; ++ni) fb[pl][ni] = tr_frag_packed<GS, true>(hB[cur][pl], 0, wcol0 + ni * 32);
            if constexpr ((MP_TR_FENCE >> 2) & 1) tr_fence();
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < TMW; ++mi) fa[mi] = tr_frag_packed<GS, true>(hA[cur][1], 0, wrow0 + mi * 32);
            if constexpr ((MP_TR_FENCE >> 2) & 1) tr_fence();
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TMW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TNW; ++ni) accW[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[pl][ni], accW[mi][ni], 0, 0, 0);
            }
            RT_MARK(0);
            if (kc + 1 < nchunks) sstore(cur ^ 1, p0 + (kc + 1) * DBK, ra_, rb_);
            if (PD2R && kc + 3 < nchunks) gload(p0 + (kc + 3) * DBK, ra_, rb_);
            RT_MARK(1);
            __syncthreads();
            RT_MARK(2);
        };
        if constexpr (PD2R) {
            for (int kc = 0; kc < nchunks; kc += 2) {
                body(kc, ra0, rb0);
                if (kc + 1 < nchunks) body(kc + 1, ra1, rb1);
            }
        } else {
            for (int kc = 0; kc < nchunks; ++kc) body(kc, ra0, rb0);
        }
        RT_FLUSH(1);
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
    __syncthreads();
    for (int e = tid; e < 2 * CI; e += 512) {
        const int st = e / CI, c = e - st * CI;
        if (partials.slots) atomicAdd(partials.slots + ((size_t)(blockIdx.x & (BN_NS - 1)) * 2 + st) * CI + c, (double)red[st][c]);
        else partials.rows[((size_t)blockIdx.x * 2 + st) * CI + c] = red[st][c];
    }
}

}  // namespace

#ifdef MP_ROLES_TIMING
extern "C" int mp_debug_roles_times(unsigned long long* host_out)      // [2][8] cycle sums since the last call (timing builds only)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_roles_t), sizeof z) != hipSuccess) return MP_ELAUNCH;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_roles_t), z, sizeof z) == hipSuccess ? MP_OK : MP_ELAUNCH;
}
#endif

// One fused backward layer (dX + dW + the BatchNorm-backward sums of layer l - 1): picks the kernel for (pooled, Co, Ci) and the arithmetic
// (bf16 one plane / fp32 as two or three bf16 planes / fp32 MFMA), launches it, checks the launch.  sa_mlp.hip: mp_sa_mlp_bwd_* (the
// caller has ruled the shape in: Ci in {64, 128}, Co in {64, 128} or 256 x 128).  by_rc: the algorithmic bytes when the input layer is a
// recomputed first layer (rc_in).
int mp_bwd_fused_launch(int pooled_, int rc_in_, int bf16_, int split_, int npl, int Co, int Ci, const void* dz_, const void* in_, int64_t P, int ppb,
                        const float* W, float* dW, float* G, const void* partials_, double fl, double by, double by_rc, hipStream_t stream)
{
    const PosOperand& DZ = *static_cast<const PosOperand*>(dz_);
    const PosOperand& IN = *static_cast<const PosOperand*>(in_);
    const BnOut& partials = *static_cast<const BnOut*>(partials_);
    const bool pooled = pooled_ != 0, rc_in = rc_in_ != 0, bf16 = bf16_ != 0, split = split_ != 0;
    // [r5] bwd_fused_kernel: chunk-interleaved workgroups (negative p_per_block) while byte offsets into the operands fit 31 bits
    static const bool bf_il = [] { const char* e = getenv("MP_BF_IL"); return !e || atoi(e) != 0; }();
    const int ppb_k = (bf_il && (uint64_t)P * (uint64_t)(Co > Ci ? Co : Ci) * 4u < (1ull << 31)) ? -ppb : ppb;
    const unsigned gx = (unsigned)((P + ppb - 1) / ppb);
            char tg[64];
            snprintf(tg, sizeof tg, bf16 ? "bwd_fused_bf16_kernel<%d, %d, %d>" : "bwd_fused_kernel<%d, %d, %d>", pooled ? 3 : 2, Co, Ci);
#define MP_FUSED(MODE, CO_, CI_)                                                                                              \
    if (bf16)                                                                                                                 \
        MP_LAUNCH(tg, fl, by, (bwd_fused_kernel<MODE, CO_, CI_, SRC_ACT, true, true>), dim3(gx), dim3(bwd_fused_threads(CO_, CI_, true, true)), 0, stream, DZ, IN, (int)P, \
                  ppb_k, W, dW, G, partials);                                                           \
    else if (split && npl == 2)                                                                                     \
        MP_LAUNCH(tg, fl, by, (bwd_fused_kernel<MODE, CO_, CI_, SRC_ACT, true, false, 2>), dim3(gx), dim3(bwd_fused_threads(CO_, CI_, true, false)), 0, stream, DZ, IN, (int)P, \
                  ppb_k, W, dW, G, partials);                                                           \
    else if (split)                                                                                                      \
        MP_LAUNCH(tg, fl, by, (bwd_fused_kernel<MODE, CO_, CI_, SRC_ACT, true>), dim3(gx), dim3(bwd_fused_threads(CO_, CI_, true, false)), 0, stream, DZ, IN, (int)P, \
                  ppb_k, W, dW, G, partials);                                                           \
    else                                                                                                                      \
        MP_LAUNCH(tg, fl, by, (bwd_fused_kernel<MODE, CO_, CI_>), dim3(gx), dim3(bwd_fused_threads(CO_, CI_, false, false)), 0, stream, DZ, IN, (int)P, ppb_k, W, \
                  dW, G, partials)
            if (rc_in) {   // (never the pooled layer: n_layers >= 3)
                snprintf(tg, sizeof tg, bf16 ? "bwd_fused_bf16_kernel<2, %d, 64, 4>" : "bwd_fused_kernel<2, %d, 64, 4>", Co);
                if (bf16 && Co == 64)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 64, 64, SRC_ACT_RC, true, true>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (bf16)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 128, 64, SRC_ACT_RC, true, true>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (Co == 64 && split && npl == 2)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 64, 64, SRC_ACT_RC, true, false, 2>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (Co == 64 && split)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 64, 64, SRC_ACT_RC, true>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (Co == 64)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 64, 64, SRC_ACT_RC>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (split && npl == 2)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 128, 64, SRC_ACT_RC, true, false, 2>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else if (split)
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 128, 64, SRC_ACT_RC, true>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
                else
                    MP_LAUNCH(tg, fl, by_rc, (bwd_fused_kernel<SRC_DZ, 128, 64, SRC_ACT_RC>), dim3(gx), dim3(256), 0, stream, DZ, IN,
                              (int)P, ppb_k, W, dW, G, partials);
            } else if (!bf16 && split && Ci == 128 && Co == 256 && (!pooled || (DZ.kshift >= 4 && ppb % 16 == 0))) {
                // [r3] the 256-output layer: the two products on different waves (bwd_roles_kernel)
                snprintf(tg, sizeof tg, "bwd_roles_kernel<%d, %d>", pooled ? 3 : 2, Co);
                if (pooled && npl == 2) MP_LAUNCH(tg, fl, by, (bwd_roles_kernel<SRC_DZ_POOLED, 256, 2>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, W, dW, G, partials);
                else if (pooled) MP_LAUNCH(tg, fl, by, (bwd_roles_kernel<SRC_DZ_POOLED, 256>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, W, dW, G, partials);
                else if (npl == 2) MP_LAUNCH(tg, fl, by, (bwd_roles_kernel<SRC_DZ, 256, 2>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, W, dW, G, partials);
                else MP_LAUNCH(tg, fl, by, (bwd_roles_kernel<SRC_DZ, 256>), dim3(gx), dim3(512), 0, stream, DZ, IN, (int)P, ppb, W, dW, G, partials);
            } else if (Co == 256) {
                if (pooled) { MP_FUSED(SRC_DZ_POOLED, 256, 128); } else { MP_FUSED(SRC_DZ, 256, 128); }
            } else if (pooled) {
                if (Co == 64 && Ci == 64) { MP_FUSED(SRC_DZ_POOLED, 64, 64); }
                else if (Co == 128 && Ci == 64) { MP_FUSED(SRC_DZ_POOLED, 128, 64); }
                else if (Co == 64 && Ci == 128) { MP_FUSED(SRC_DZ_POOLED, 64, 128); }
                else { MP_FUSED(SRC_DZ_POOLED, 128, 128); }
            } else {
                if (Co == 64 && Ci == 64) { MP_FUSED(SRC_DZ, 64, 64); }
                else if (Co == 128 && Ci == 64) { MP_FUSED(SRC_DZ, 128, 64); }
                else if (Co == 64 && Ci == 128) { MP_FUSED(SRC_DZ, 64, 128); }
                else { MP_FUSED(SRC_DZ, 128, 128); }
            }
#undef MP_FUSED
    MP_CHECK_LAUNCH();
    return MP_OK;
}
