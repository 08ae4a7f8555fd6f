// The regression heads' blocks over a skinny batch, one launch each way:
//   forward   y = dropout(relu(bn(x W^T + b)))   (models/pointnet2_cls_ssg.py:309-327: self.dropout(F.relu(self.bn1(self.fc1(x)))))
//             or the plain Linear (fc3 / fc_normals / sm_fc3, :311, :327, :336)
//   backward  dz = BatchNorm + ReLU + dropout backward of grad_y, dgamma, dbeta, and grad_x = dz W
//
// B <= 32 rows ([r5] B <= 64: four 16-row tiles, MT = 4) against 1024 x 1024 .. 11988 x 1024 weights: every layer is one pass over W, and the libraries' GEMMs for a 32-row
// batch pick 16 x 32 tiles on 64 workgroups (11-12 us for 4 MB; 20 us for 49 MB).  Here the batch is the 32-row side of
// v_mfma_f32_16x16x4_f32 tiles (fp32 operands straight from memory into the matrix cores: no staging arithmetic at all), a workgroup
// owns 16 output columns and its eight waves an eighth of K each, every operand byte of a wave requested up front (one memory latency
// per workgroup), the eight partial tiles meet in LDS.  A workgroup then holds all 32 rows of its columns, which is everything the
// BatchNorm1d of the block needs: statistics, running statistics, ReLU and the dropout mask (loss_tail.hip's counter-based one) in
// the epilogue.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HL_WAVES = 8;
constexpr int HL_THREADS = HL_WAVES * 64;

struct HeadBn {
    int on, training;
    float momentum, eps, drop_p;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* save_mean;
    float* save_rstd;
    const long long* rng;
    int layer;
};

__device__ __forceinline__ float drop_keep(unsigned long long key, int r, int C, int c, float drop_p, float keep_scale, float v)
{   // the mask of loss_tail.hip's bn_relu_rows_kernel: same hash, same 24 bits
    unsigned long long z = key + 0x9E3779B97F4A7C15ull * ((unsigned long long)r * (unsigned long long)C + (unsigned long long)c + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return u >= drop_p ? v * keep_scale : 0.0f;
}

// grid: ceil(O / 16) workgroups of 512 threads; I = 128 NJ.  z (may be NULL without BatchNorm): the Linear's output, y: the block's.
// Operand path: the MFMA wants lane (row l16, k-quarter q), i.e. neighbouring lanes on DIFFERENT rows of W / x (4 KB apart: one cache
// line per lane, 64 tag look-ups per load instruction -- measured: 5 us of address processing per workgroup).  So a wave requests its
// [16 rows][k range] tiles row-contiguous (16 lanes per 256 bytes of a row), all of them up front, and turns each through a private
// 4 KB LDS tile into the fragment layout (row stride CH + 4: both directions conflict-free, no barrier: a wave's LDS operations are
// ordered).
// One problem of a launch: y = [dropout(relu(bn(] x W^T + b [)))].  A launch carries up to two (same B and I): the column tiles of the
// second follow the first's -- two blocks of the two head branches (fc1 / sm_fc1 on the global feature, fc2 / sm_fc2 on their outputs),
// or two plain Linears fed by one activation (fc3 / fc_normals).
struct HeadFwd {
    const float* x;
    const float* W;
    const float* bias;
    float* z;
    float* y;
    int O;
    HeadBn bn;
};

// [r5] MT: 16-row tiles of the batch -- 2 (B <= 32) or 4 (B <= 64: the reference's own batch size, configs/maskplanner/cuboids_v2.yaml:12).
template <int NJ, int MT = 2>
__global__ __launch_bounds__(HL_THREADS) void head_fwd_kernel(HeadFwd pa, HeadFwd pb, int B, int I, int tiles_a)
{
    const bool second = (int)blockIdx.x >= tiles_a;
    const float* __restrict__ x = second ? pb.x : pa.x;
    const float* __restrict__ W = second ? pb.W : pa.W;
    const float* __restrict__ bias = second ? pb.bias : pa.bias;
    float* __restrict__ z = second ? pb.z : pa.z;
    float* __restrict__ y = second ? pb.y : pa.y;
    const int O = second ? pb.O : pa.O;
    const HeadBn bn = second ? pb.bn : pa.bn;
    constexpr int KW = 16 * NJ;                 // k range of a wave
    constexpr int CH = KW < 64 ? KW : 64;       // k per pass through the LDS tile
    constexpr int NCH = KW / CH;
    constexpr int LPR = CH / 4, RPI = 64 / LPR, NI = 16 / RPI;     // lanes per row, rows per load instruction, instructions per 16-row tile
    constexpr int NF = CH / 16;                 // fragments (float4 per lane) per tile and pass
    __shared__ __attribute__((aligned(16))) float tr[HL_WAVES][16][CH + 4];
    constexpr int RB = 16 * MT, EPT = MT / 2;        // rows of the batch tile; epilogue elements per thread (rows b, b + 32)
    __shared__ float part[HL_WAVES][MT][4][64];
    __shared__ float zt[RB][17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, q = lane >> 4;
    const int o0 = ((int)blockIdx.x - (second ? tiles_a : 0)) * 16;
    const int lr = lane / LPR, lk = 4 * (lane % LPR);          // row / first k of this lane inside one load instruction
    const int k0 = wave * KW + lk;
    // the epilogue's per-column constants ride along with the operand requests (asked for after the products they would add a second
    // memory latency to the workgroup's life)
    const int b = tid >> 4, oc = tid & 15;          // epilogue: one element per thread, row b, column o0 + oc
    const int o = o0 + oc;
    const bool colok = o < O;
    const float bias_v = (bias != nullptr && colok) ? bias[o] : 0.0f;
    float ga = 1.0f, be = 0.0f, rm = 0.0f, rv = 1.0f;
    unsigned long long drop_key = 0ull;
    const bool drop = bn.on && bn.rng != nullptr && bn.drop_p > 0.0f;
    if (bn.on && colok) {
        if (bn.gamma) ga = bn.gamma[o];
        if (bn.beta) be = bn.beta[o];
        if (bn.running_mean) { rm = bn.running_mean[o]; rv = bn.running_var[o]; }
    }
    if (drop) drop_key = (unsigned long long)bn.rng[0] + 0xD1B54A32D192ED03ull * (unsigned long long)bn.rng[1] + ((unsigned long long)(unsigned)bn.layer << 48);
    f32x4 rw[NCH][NI], rx[MT][NCH][NI];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int row = n * RPI + lr;
            rw[c][n] = *reinterpret_cast<const f32x4*>(W + (size_t)min(o0 + row, O - 1) * I + k0 + c * CH);
        }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int row = n * RPI + lr;
#pragma unroll
            for (int t = 0; t < MT; ++t) rx[t][c][n] = *reinterpret_cast<const f32x4*>(x + (size_t)min(16 * t + row, B - 1) * I + k0 + c * CH);
        }
    __builtin_amdgcn_sched_barrier(0);       // every operand byte requested before the first product: one memory latency per workgroup
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float (*tl)[CH + 4] = tr[wave];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 wv[NF], xv[MT][NF];
#pragma unroll
        for (int n = 0; n < NI; ++n) *reinterpret_cast<f32x4*>(&tl[n * RPI + lr][lk]) = rw[c][n];
#pragma unroll
        for (int j = 0; j < NF; ++j) wv[j] = *reinterpret_cast<const f32x4*>(&tl[l16][16 * j + 4 * q]);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int n = 0; n < NI; ++n) *reinterpret_cast<f32x4*>(&tl[n * RPI + lr][lk]) = rx[t][c][n];
#pragma unroll
            for (int j = 0; j < NF; ++j) xv[t][j] = *reinterpret_cast<const f32x4*>(&tl[l16][16 * j + 4 * q]);
        }
        // k-step (c, j, e) multiplies elements [row][wave KW + c CH + 16 j + 4 q + e] of both operands
#pragma unroll
        for (int j = 0; j < NF; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t][j][e], wv[j][e], acc[t], 0, 0, 0);
            }
        }
    }
    // accumulator register r of lane (l16, q): row 4 q + r of the 16-row tile, column l16
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    float zv[EPT];
    bool live[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int bb = b + 32 * u;
        live[u] = colok && bb < B;
        const int t = bb >> 4, qq = (bb & 15) >> 2, r = bb & 3, ln = qq * 16 + oc;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < HL_WAVES; ++w) s += part[w][t][r][ln];
        zv[u] = s + bias_v;
        zt[bb][oc] = zv[u];
    }
    if (!bn.on) {
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (live[u]) y[(size_t)(b + 32 * u) * O + o] = zv[u];
        return;
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u)
        if (live[u] && z != nullptr) z[(size_t)(b + 32 * u) * O + o] = zv[u];
    __syncthreads();
    float mean, rstd;
    if (bn.training) {
        float col[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) col[r] = zt[r][oc];        // (fixed trip count: the reads are issued together)
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < RB; ++r) s += r < B ? col[r] : 0.0f;
        mean = s / (float)B;
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < RB; ++r) { const float d = col[r] - mean; v += r < B ? d * d : 0.0f; }
        const float var = v / (float)B;
        rstd = 1.0f / sqrtf(var + bn.eps);
        if (b == 0 && colok && bn.running_mean != nullptr) {
            bn.running_mean[o] = (1.0f - bn.momentum) * rm + bn.momentum * mean;
            bn.running_var[o] = (1.0f - bn.momentum) * rv + bn.momentum * (B > 1 ? v / (float)(B - 1) : var);
        }
    } else {
        mean = rm;
        rstd = 1.0f / sqrtf(rv + bn.eps);
    }
    if (b == 0 && colok) { bn.save_mean[o] = mean; bn.save_rstd[o] = rstd; }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        if (!live[u]) continue;
        const int bb = b + 32 * u;
        float v = (zv[u] - mean) * rstd * ga + be;
        v = v > 0.0f ? v : 0.0f;
        if (drop) v = drop_keep(drop_key, bb, O, o, bn.drop_p, 1.0f / (1.0f - bn.drop_p), v);
        y[(size_t)bb * O + o] = v;
    }
}

// ---- backward of a block:  dz = bn_relu_drop_bwd(grad_y),  grad_x += dz W  -------------------------------------------------------
// grid (I / 64, ceil(O / 256)): a workgroup owns 64 input columns and 256 rows of W (a wave: 32 of them, requested up front).  It forms
// the dz of those 256 columns itself, every array read as coalesced float4 rows (thread = four columns x four batch rows, the column
// sums of the BatchNorm backward through LDS), keeps it in LDS as the A operand, and the workgroups of column tile 0 store it (with
// dgamma / dbeta).  MFMA roles: m = batch row (two 16-row tiles), n = input column, k = o; a lane's float4 of W holds four input
// columns: tile e of the four takes columns {i0 + 4 n + e}.  The eight waves' partial tiles meet in LDS, the row slices
// of W in grad_x (global atomics; one slice: plain stores).
constexpr int HB_OS = 256;
constexpr int HB_LD = HB_OS + 4;      // row stride of the dz image: lane (row l16, k q) reads bank (4 l16 + q) -- conflict-free

struct HeadBwd {
    const float* grad_y;
    const float* y;
    const float* z;
    const float* W;
    const float* gamma;
    const float* save_mean;
    const float* save_rstd;
    float* dz;
    float* dgamma;
    float* dbeta;
    float* gx;
    int O, training;
    float keep_scale;
};

// Up to two problems per launch (same B and I): the row slices of the second follow the first's in grid.y.  Their grad_x may be ONE buffer
// (fc1 / sm_fc1 both read the global feature: no fan-out add) -- `atomic`: more than one slice adds into some grad_x.
template <int MT>       // 16-row tiles of the batch: 2 (B <= 32) or 4 (B <= 64)
__global__ __launch_bounds__(HL_THREADS) void head_bwd_kernel(HeadBwd pa, HeadBwd pb, int B, int I, int slices_a, int atomic)
{
    constexpr int RB = 16 * MT, NR = MT * 2;          // batch rows of the tile; rows per thread while dz is formed (bq + 8 i)
    const bool second = (int)blockIdx.y >= slices_a;
    const float* __restrict__ grad_y = second ? pb.grad_y : pa.grad_y;
    const float* __restrict__ y = second ? pb.y : pa.y;
    const float* __restrict__ zin = second ? pb.z : pa.z;
    const float* __restrict__ W = second ? pb.W : pa.W;
    const float* __restrict__ gamma = second ? pb.gamma : pa.gamma;
    const float* __restrict__ save_mean = second ? pb.save_mean : pa.save_mean;
    const float* __restrict__ save_rstd = second ? pb.save_rstd : pa.save_rstd;
    float* __restrict__ dz = second ? pb.dz : pa.dz;
    float* __restrict__ dgamma = second ? pb.dgamma : pa.dgamma;
    float* __restrict__ dbeta = second ? pb.dbeta : pa.dbeta;
    float* __restrict__ gx = second ? pb.gx : pa.gx;
    const int O = second ? pb.O : pa.O, training = second ? pb.training : pa.training;
    const float keep_scale = second ? pb.keep_scale : pa.keep_scale;
    // LDS: the dz image + the column-sum exchange (49 KB) while dz is formed and multiplied, then -- behind a barrier -- the eight waves'
    // partial tiles (64 KB) over the same bytes.  (ds_add_f32 into one shared tile instead: ~200 cycles per instruction, 23 us.)
    // (MT = 4: the image is 65 KB, the partial tiles go through it in two rounds of 64 KB)
    __shared__ __attribute__((aligned(16))) float smem[MT == 2 ? 16384 : RB * HB_LD + HL_WAVES * HB_OS * 2];
    float (*dzs)[HB_LD] = reinterpret_cast<float (*)[HB_LD]>(smem);                          // [RB][HB_LD]
    float (*red)[HB_OS][2] = reinterpret_cast<float (*)[HB_OS][2]>(smem + RB * HB_LD);       // [HL_WAVES][HB_OS][2]
    float (*part)[2][4][4][64] = reinterpret_cast<float (*)[2][4][4][64]>(smem);             // [HL_WAVES][2][4][4][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l16 = lane & 15, q = lane >> 4;
    const int i0 = blockIdx.x * 64, os = ((int)blockIdx.y - (second ? slices_a : 0)) * HB_OS;
    const int ow = os + wave * 32;
    // B operand: lane (n = l16, k = q) of step s: W[ow + 4 s + q][i0 + 4 l16 .. + 3]
    f32x4 wv[8];
    {
        // (rows past O: clamped -- their dz columns are zeros; no branch around a load: the eight requests must overlap)
#pragma unroll
        for (int s = 0; s < 8; ++s) wv[s] = *reinterpret_cast<const f32x4*>(W + (size_t)min(ow + 4 * s + q, O - 1) * I + i0 + 4 * l16);
    }
    __builtin_amdgcn_sched_barrier(0);       // the weight rows are under way while dz is formed
    // ---- dz of columns os .. os + 255: thread (c4, bq) = columns os + 4 c4 .. + 3, rows bq, bq + 8, bq + 16, bq + 24
    {
        const int c4 = lane, bq = wave;
        const int oc = os + 4 * c4;
        const bool colok = oc < O;             // (O % 4 == 0)
        f32x4 dy[NR], xh[NR];
        f32x4 mean = {0.0f, 0.0f, 0.0f, 0.0f}, rstd = mean, ga = {1.0f, 1.0f, 1.0f, 1.0f};
        if (colok) {
            mean = *reinterpret_cast<const f32x4*>(save_mean + oc);
            rstd = *reinterpret_cast<const f32x4*>(save_rstd + oc);
            if (gamma) ga = *reinterpret_cast<const f32x4*>(gamma + oc);
        }
        f32x4 db = {0.0f, 0.0f, 0.0f, 0.0f}, dg = db;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int b = bq + 8 * i;
            const bool ok = colok && b < B;
            const size_t e = ok ? (size_t)b * O + oc : 0;
            const f32x4 gv = *reinterpret_cast<const f32x4*>(grad_y + e), yv = *reinterpret_cast<const f32x4*>(y + e),
                        zv = *reinterpret_cast<const f32x4*>(zin + e);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dy[i][c] = (ok && yv[c] > 0.0f) ? gv[c] * keep_scale : 0.0f;
                xh[i][c] = ok ? (zv[c] - mean[c]) * rstd[c] : 0.0f;
                db[c] += dy[i][c];
                dg[c] += dy[i][c] * xh[i][c];
            }
        }
        float* rp = &red[bq][4 * c4][0];
        reinterpret_cast<f32x4*>(rp)[0] = f32x4{db[0], dg[0], db[1], dg[1]};
        reinterpret_cast<f32x4*>(rp)[1] = f32x4{db[2], dg[2], db[3], dg[3]};
        __syncthreads();
        db = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dg = db;
#pragma unroll
        for (int w = 0; w < HL_WAVES; ++w) {
            const f32x4 p0 = reinterpret_cast<const f32x4*>(&red[w][4 * c4][0])[0], p1 = reinterpret_cast<const f32x4*>(&red[w][4 * c4][0])[1];
            db[0] += p0[0]; dg[0] += p0[1]; db[1] += p0[2]; dg[1] += p0[3];
            db[2] += p1[0]; dg[2] += p1[1]; db[3] += p1[2]; dg[3] += p1[3];
        }
        const float inv = 1.0f / (float)B;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int b = bq + 8 * i;
            f32x4 d;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                d[c] = training ? ga[c] * rstd[c] * (dy[i][c] - db[c] * inv - xh[i][c] * dg[c] * inv) : ga[c] * rstd[c] * dy[i][c];
            if (!(colok && b < B)) d = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(&dzs[b][4 * c4]) = d;
            if (blockIdx.x == 0 && colok && b < B) *reinterpret_cast<f32x4*>(dz + (size_t)b * O + oc) = d;
        }
        if (blockIdx.x == 0 && colok && bq == 0) {
            if (dbeta) *reinterpret_cast<f32x4*>(dbeta + oc) = db;
            if (dgamma) *reinterpret_cast<f32x4*>(dgamma + oc) = dg;
        }
    }
    __syncthreads();
    f32x4 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float d[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) d[t] = dzs[16 * t + l16][wave * 32 + 4 * s + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[t], wv[s][e], acc[t][e], 0, 0, 0);
        }
    }
    // the partial tiles of 32 batch rows at a time through the image's bytes
#pragma unroll
    for (int hh = 0; hh < MT / 2; ++hh) {
        __syncthreads();              // every wave has read its dz fragments (/ the previous round's partial tiles)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave][t][e][r][lane] = acc[2 * hh + t][e][r];
        __syncthreads();
        // 32 rows x 64 columns, four per thread: thread -> (row b, columns i0 + 4 n .. + 3), n = tid & 15
        const int bl = tid >> 4, n = tid & 15, b = 32 * hh + bl;
        if (b < B) {
            const int t = bl >> 4, qq = (bl & 15) >> 2, r = bl & 3, ln = qq * 16 + n;
            float* dst = gx + (size_t)b * I + i0 + 4 * n;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int w = 0; w < HL_WAVES; ++w)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += part[w][t][e][r][ln];
            if (!atomic) *reinterpret_cast<f32x4*>(dst) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, v[e]);
            }
        }
    }
}

}  // namespace

extern "C" int mp_head_block_supported(int64_t B, int64_t I, int64_t O)
{
    if (B < 1 || B > 64 || O < 1 || O > (1 << 24)) return 0;
    return I == 128 || I == 256 || I == 512 || I == 1024 || I == 2048;
}

static int head_fwd_launch(const HeadFwd& a, const HeadFwd* b, int64_t B, int64_t I, mp_stream_t stream_)
{
    const int tiles_a = (a.O + 15) / 16;
    const dim3 grid((unsigned)(tiles_a + (b ? (b->O + 15) / 16 : 0)));
    hipStream_t stream = mp_stream(stream_);
    const double Ot = (double)a.O + (b ? (double)b->O : 0.0);
    const double flops = 2.0 * (double)B * (double)I * Ot, bytes = 4.0 * ((double)I * Ot + (double)B * ((double)I + 2.0 * Ot));
    const HeadFwd pb = b ? *b : a;
#define HL_FWD(NJ)                                                                                                                          \
    if (B <= 32) MP_LAUNCH("head_fwd_kernel", flops, bytes, (head_fwd_kernel<NJ, 2>), grid, dim3(HL_THREADS), 0, stream, a, pb, (int)B, (int)I, \
                           b ? tiles_a : (1 << 30));                                                                                         \
    else MP_LAUNCH("head_fwd_kernel", flops, bytes, (head_fwd_kernel<NJ, 4>), grid, dim3(HL_THREADS), 0, stream, a, pb, (int)B, (int)I,    \
                   b ? tiles_a : (1 << 30))
    switch (I) {
        case 128: HL_FWD(1); break;
        case 256: HL_FWD(2); break;
        case 512: HL_FWD(4); break;
        case 1024: HL_FWD(8); break;
        default: HL_FWD(16); break;
    }
#undef HL_FWD
    MP_CHECK_LAUNCH();
    return MP_OK;
}

static int head_fwd_set(const mp_head_block_t& h, int64_t B, int64_t I, HeadFwd& out)
{
    if (!mp_head_block_supported(B, I, h.O)) return MP_EUNSUPPORTED;
    if (!h.x || !h.weight || !h.y || h.drop_p < 0.0 || h.drop_p >= 1.0) return MP_EINVAL;
    if (h.bn && (!h.save_mean || !h.save_rstd || (!h.training && (!h.running_mean || !h.running_var)))) return MP_EINVAL;
    out = HeadFwd{h.x, h.weight, h.bias, h.z, h.y, (int)h.O,
                  HeadBn{h.bn, h.training, (float)h.momentum, (float)h.eps, (float)h.drop_p, h.gamma, h.beta, h.running_mean, h.running_var,
                         h.save_mean, h.save_rstd, reinterpret_cast<const long long*>(h.rng), h.layer}};
    return MP_OK;
}

extern "C" int mp_head_blocks_fwd_f32(int n, const mp_head_block_t* blocks, int64_t B, int64_t I, mp_stream_t stream_)
{
    if (n < 1 || n > 2 || !blocks) return MP_EINVAL;
    HeadFwd s[2];
    for (int k = 0; k < n; ++k)
        if (int rc = head_fwd_set(blocks[k], B, I, s[k])) return rc;
    return head_fwd_launch(s[0], n == 2 ? &s[1] : nullptr, B, I, stream_);
}

extern "C" int mp_head_block_fwd_f32(const float* x, const float* weight, const float* bias, int64_t B, int64_t I, int64_t O, int bn,
                                     int training, double momentum, double eps, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float* z, float* y, float* save_mean, float* save_rstd,
                                     double drop_p, const int64_t* rng, int layer, mp_stream_t stream_)
{
    mp_head_block_t h{};
    h.x = x; h.weight = weight; h.bias = bias; h.O = O; h.bn = bn; h.training = training; h.momentum = momentum; h.eps = eps;
    h.gamma = gamma; h.beta = beta; h.running_mean = running_mean; h.running_var = running_var; h.z = z; h.y = y;
    h.save_mean = save_mean; h.save_rstd = save_rstd; h.drop_p = drop_p; h.rng = rng; h.layer = layer;
    return mp_head_blocks_fwd_f32(1, &h, B, I, stream_);
}

extern "C" int mp_head_linear2_fwd_f32(const float* x, int64_t B, int64_t I, const float* w1, const float* b1, int64_t O1, float* y1,
                                       const float* w2, const float* b2, int64_t O2, float* y2, mp_stream_t stream_)
{
    mp_head_block_t h[2] = {};
    h[0].x = x; h[0].weight = w1; h[0].bias = b1; h[0].O = O1; h[0].y = y1;
    h[1].x = x; h[1].weight = w2; h[1].bias = b2; h[1].O = O2; h[1].y = y2;
    return mp_head_blocks_fwd_f32(w2 ? 2 : 1, h, B, I, stream_);
}

// mp_head_block_bwd_slices(O) > 1: the row slices of W add their tiles into grad_x with atomics (summation order not fixed); the call
// clears grad_x first unless it lies in the armed zero arena
extern "C" int mp_head_block_bwd_slices(int64_t O)
{
    return (int)((O + HB_OS - 1) / HB_OS);
}

extern "C" int mp_head_blocks_bwd_f32(int n, const mp_head_block_t* blocks, int64_t B, int64_t I, mp_stream_t stream_)
{
    if (n < 1 || n > 2 || !blocks) return MP_EINVAL;
    if (B < 1 || B > 64 || I < 64 || I % 64 != 0) return MP_EUNSUPPORTED;
    HeadBwd s[2];
    int slices[2] = {0, 0};
    double Ot = 0.0;
    for (int k = 0; k < n; ++k) {
        const mp_head_block_t& h = blocks[k];
        if (h.O < 4 || h.O % 4 != 0 || h.O > 4096) return MP_EUNSUPPORTED;
        if (!h.grad_y || !h.y || !h.z || !h.weight || !h.save_mean || !h.save_rstd || !h.dz || !h.grad_x || h.drop_p < 0.0 || h.drop_p >= 1.0)
            return MP_EINVAL;
        s[k] = HeadBwd{h.grad_y, h.y, h.z, h.weight, h.gamma, h.save_mean, h.save_rstd, h.dz, h.grad_gamma, h.grad_beta, h.grad_x, (int)h.O,
                       h.training, (float)(1.0 / (1.0 - h.drop_p))};
        slices[k] = mp_head_block_bwd_slices(h.O);
        Ot += (double)h.O;
    }
    const bool shared = n == 2 && blocks[0].grad_x == blocks[1].grad_x;
    const int atomic = (slices[0] > 1 || slices[1] > 1 || shared) ? 1 : 0;
    hipStream_t stream = mp_stream(stream_);
    if (atomic) {
        if (!mp::zero_async(blocks[0].grad_x, (size_t)(B * I), stream)) return MP_ELAUNCH;     // (nothing to do inside an armed zero arena)
        if (n == 2 && !shared && !mp::zero_async(blocks[1].grad_x, (size_t)(B * I), stream)) return MP_ELAUNCH;
    }
    const dim3 grid((unsigned)(I / 64), (unsigned)(slices[0] + slices[1]));
    const double flops = 2.0 * (double)B * (double)I * Ot, bytes = 4.0 * ((double)I * Ot + (double)B * ((double)I + 4.0 * Ot));
    if (B <= 32)
        MP_LAUNCH("head_bwd_kernel", flops, bytes, head_bwd_kernel<2>, grid, dim3(HL_THREADS), 0, stream, s[0], n == 2 ? s[1] : s[0], (int)B, (int)I,
                  n == 2 ? slices[0] : (1 << 30), atomic);
    else
        MP_LAUNCH("head_bwd_kernel", flops, bytes, head_bwd_kernel<4>, grid, dim3(HL_THREADS), 0, stream, s[0], n == 2 ? s[1] : s[0], (int)B, (int)I,
                  n == 2 ? slices[0] : (1 << 30), atomic);
    MP_CHECK_LAUNCH();
    return MP_OK;
}

extern "C" int mp_head_block_bwd_f32(const float* grad_y, const float* y, const float* z, const float* weight, int64_t B, int64_t I,
                                     int64_t O, int training, const float* gamma, const float* save_mean, const float* save_rstd,
                                     double drop_p, float* dz, float* grad_gamma, float* grad_beta, float* grad_x, mp_stream_t stream_)
{
    mp_head_block_t h{};
    h.grad_y = grad_y; h.y = const_cast<float*>(y); h.z = const_cast<float*>(z); h.weight = weight; h.O = O; h.training = training; h.gamma = gamma;
    h.save_mean = const_cast<float*>(save_mean); h.save_rstd = const_cast<float*>(save_rstd); h.drop_p = drop_p; h.dz = dz;
    h.grad_gamma = grad_gamma; h.grad_beta = grad_beta; h.grad_x = grad_x;
    return mp_head_blocks_bwd_f32(1, &h, B, I, stream_);
}
