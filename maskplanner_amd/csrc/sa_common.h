// Shared device helpers of the set-abstraction MLP kernels (sa_mlp.hip, sa_stream16.hip): operand descriptors, the BatchNorm "sites",
// the staging algebra (BN + ReLU / dZ), bf16 plane splitting, the K-packed LDS image and its fragment reads.  [r5] split out of sa_mlp.hip.
// Everything lives in an anonymous namespace: each translation unit gets its own copy (the kernels of one file only launch each other's
// structs inside that file; cross-file launch helpers take pointers to the structs, whose layout this header fixes).
#pragma once
#include <cstdio>
#include <type_traits>

#include "common.h"

namespace {


using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef MP_STORE_AUX
#define MP_STORE_AUX 0      // cache-policy bits of the raw Z / G buffer stores ([r2] same-box sweep of 0 / nt / sc0 / sc0+sc1: 0 is best)
#endif
#ifndef MP_BK
#define MP_BK 32
#endif
constexpr int THREADS = 256;

enum SrcMode { SRC_ID = 0, SRC_ACT = 1, SRC_DZ = 2, SRC_DZ_POOLED = 3, SRC_ACT_RC = 4, SRC_DZ_RC = 5 };
constexpr bool is_id(int m) { return m == SRC_ID; }
// *_RC ("recompute"): the raw Z of this operand is not in memory -- it is the first layer of a level with a 4-channel input
// (xyz + pad), z[p][c] = X0[p][0:4] . W0[c][0:4], four FMAs per element: cheaper to recompute from the 16-byte input row than
// to write [P, C] floats once and read them back three times (next layer forward, next layer backward, its own dW).
constexpr bool is_rc(int m) { return m == SRC_ACT_RC || m == SRC_DZ_RC; }
constexpr bool is_dz(int m) { return m == SRC_DZ || m == SRC_DZ_POOLED || m == SRC_DZ_RC; }

// ---- BatchNorm statistics without a finalize launch of their own (train mode, per-replica statistics) ---------------------------
// A layer's per-channel sums -- (sum z, sum z^2) forward, (sum dy, sum dy z) backward -- used to leave their producer as one fp32
// partial row per workgroup and be reduced by bn_{fwd,bwd}_finalize_kernel: 18 launches of ~5 us per training step, each of them a
// dependency bubble between two large kernels.  Now ("sites"):
//  * the PRODUCER adds its fp64 sums into one of BN_NS slot rows with atomics (slot = workgroup index mod BN_NS: 8 rows to reduce
//    instead of 512, a few dozen atomics per address);
//  * the FIRST kernel that consumes the constants derives them in its prologue: every workgroup reduces the BN_NS rows of the
//    channels it needs (L2 hits, one round trip), converts them (fp64, the finalize kernels' algebra) and keeps the result in LDS,
//    where load_consts() reads it; workgroup 0 also writes the global arrays later kernels read (scale / shift for the backward
//    pass, a / e / f for the second kernel of a dW + dX pair), updates the running statistics and writes dgamma / dbeta / dbias;
//  * nobody counts readers: workgroup 0 of a kernel zeroes the slot rows of the site the PREVIOUS kernel of the call consumed
//    (that kernel has completed), and the first kernel of a call zeroes the two sites the previous calls left behind (the last
//    forward site and the first-layer backward site).  At any time the only non-zero rows are ones that were already consumed.
// The rows live in caller-owned, persistent, zero-initialised memory (mp_mlp_layer_t::bn_state).  Statistics equal the finalize
// kernels' up to the order of fp64 additions.
constexpr int BN_NS = 8;
constexpr int BN_POOL_CMAX = 1024;   // widest pooled layer whose select kernel derives its own constants
// mp_mlp_layer_t::bn_state of a layer with C outputs, in doubles: [forward slots BN_NS * 2 * C | backward slots BN_NS * 2 * C]
// (MP_BN_STATE_DOUBLES in the header)
inline double* bn_fwd_slots(double* st, int C) { (void)C; return st; }
inline double* bn_bwd_slots(double* st, int C) { return st + (size_t)BN_NS * 2 * C; }
struct BnSite {
    double* slots;            // [BN_NS][2][C]; NULL: off (the constants come from a finalize launch as before)
    int C;
    int kind;                 // 1: forward statistics -> scale, shift (+ mean, rstd, running stats); 2: backward sums -> a, e, f (+ dgamma, dbeta, dbias)
    double invP, unbias, momentum, eps;
    const float* gamma;
    const float* beta;
    const float* bias;
    float* running_mean;
    float* running_var;
    float* mean;              // kind 1: out; kind 2: in
    float* rstd;
    float* o0;                // kind 1: scale, shift; kind 2: a, e, f (global copies, written by workgroup 0)
    float* o1;
    float* o2;
    float* dgamma;
    float* dbeta;
    float* dbias;
    double* z0;               // slot rows of the site the previous kernel consumed: zeroed by workgroup 0
    int n0;
};
// where a producer's per-workgroup sums go: one fp32 partial row per workgroup (legacy: finalize kernels) or a site's slot rows
struct BnOut {
    float* rows;
    double* slots;
    double* z0;               // first kernel of a call: the two sites earlier calls left behind (zeroed by workgroup 0)
    int n0;
    double* z1;
    int n1;
};
__device__ __forceinline__ int bn_linear_tid() { return (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x; }
__device__ __forceinline__ void bn_zero_rows(double* z, int n)
{
    const int nt = blockDim.x * blockDim.y * blockDim.z;
    if (z) for (int i = bn_linear_tid(); i < n; i += nt) z[i] = 0.0;
}
// producers call this once (any place): the zeroing duty of the first kernel of a call
__device__ __forceinline__ void bn_zero(const BnOut& o)
{
    if ((blockIdx.x | blockIdx.y | blockIdx.z) != 0) return;
    bn_zero_rows(o.z0, o.n0);
    bn_zero_rows(o.z1, o.n1);
}
__device__ __forceinline__ void bn_emit(const BnOut& o, int C, unsigned blk, int c, double s1, double s2)
{
    if (o.slots) {
        double* q = o.slots + (size_t)(blk & (BN_NS - 1)) * 2 * (unsigned)C;
        atomicAdd(q + c, s1);
        atomicAdd(q + C + c, s2);
    } else {
        o.rows[((size_t)blk * 2 + 0) * C + c] = (float)s1;
        o.rows[((size_t)blk * 2 + 1) * C + c] = (float)s2;
    }
}
// Consumer prologue: every thread of every workgroup calls it before load_consts(); contains one barrier.  The constants of channels
// [c0, c0 + cn) go to lds[k * ld + (c - c0)] (k = 0, 1: scale, shift; or 0, 1, 2: a, e, f).  `writer`: this workgroup also writes
// the global arrays and does the once-only duties for its channels (one workgroup per channel range must be the writer).
__device__ __forceinline__ void bn_prologue(const BnSite& b, float* lds, int ld, int c0, int cn, bool writer)
{
    if (b.slots == nullptr) return;
    const int nt = blockDim.x * blockDim.y * blockDim.z;
    const int tid = bn_linear_tid();
    const int C = b.C;
    if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0) bn_zero_rows(b.z0, b.n0);
    for (int i = tid; i < cn; i += nt) {
        const int c = c0 + i;
        if (c >= C) break;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int s = 0; s < BN_NS; ++s) {
            s1 += b.slots[(size_t)(s * 2 + 0) * C + c];
            s2 += b.slots[(size_t)(s * 2 + 1) * C + c];
        }
        if (b.kind == 1) {      // bn_fwd_finalize_kernel, training branch
            const double mean = s1 * b.invP;
            double var = s2 * b.invP - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + b.eps);
            const float sc = (float)((double)b.gamma[c] * rstd);
            const float sh = (float)((double)b.beta[c] - mean * (double)sc);
            lds[i] = sc;
            lds[ld + i] = sh;
            if (writer) {
                b.o0[c] = sc;
                b.o1[c] = sh;
                b.mean[c] = (float)mean;
                b.rstd[c] = (float)rstd;
                if (b.running_mean) {
                    const float bi = b.bias ? b.bias[c] : 0.0f;
                    b.running_mean[c] = (float)((1.0 - b.momentum) * (double)b.running_mean[c] + b.momentum * (mean + (double)bi));
                    b.running_var[c] = (float)((1.0 - b.momentum) * (double)b.running_var[c] + b.momentum * var * b.unbias);
                }
            }
        } else {                // bn_bwd_finalize_kernel, training branch
            const double mu = b.mean[c], rs = b.rstd[c], g = b.gamma[c];
            const double dbe = s1, dga = rs * (s2 - mu * s1);
            const double a = g * rs, c1 = dbe * b.invP, c2 = dga * b.invP;
            const float fa = (float)a, fe = (float)(-a * c2 * rs), ff = (float)(-a * c1 + a * c2 * rs * mu);
            lds[i] = fa;
            lds[ld + i] = fe;
            lds[2 * ld + i] = ff;
            if (writer) {
                b.o0[c] = fa;
                b.o1[c] = fe;
                b.o2[c] = ff;
                b.dbeta[c] = (float)dbe;
                b.dgamma[c] = (float)dga;
                if (b.dbias) b.dbias[c] = 0.0f;
            }
        }
    }
    __syncthreads();
}
// a site whose first consumer has no prologue (tiled forward GEMMs, the unfused pool): the same algebra as a launch of its own
__global__ __launch_bounds__(256) void bn_site_finalize_kernel(BnSite b)
{
    __shared__ float scratch[3 * 256];
    const int c0 = blockIdx.x * 256;
    bn_prologue(b, scratch, 256, c0, 256, true);
}

// A positions-major operand: rows = positions, columns = channels (contiguous).  Every channel count is a multiple
// of 4 (checked on the host; the Python layer zero-pads 3 -> 4, 131 -> 132, 259 -> 260), so a float4 of channels is
// either entirely inside or entirely outside and every global access is an aligned 16-byte one.  Element offsets
// fit 32 bits (P * C < 2^31, checked on the host).
struct PosOperand {
    const float* x;      // X (SRC_ID) or raw Z [P, C]
    const float* g;      // SRC_DZ: G [P, C];  SRC_DZ_POOLED: pooled grad (relu-masked) [P/K, C]
    const int* argk;     // SRC_DZ_POOLED: arg-max position inside the group [P/K, C]
    const float* s;      // activation scale / shift of THIS tensor's layer [C]
    const float* t;
    const float* a;      // dZ constants [C]: dz = a*dy + e*z + f
    const float* e;
    const float* f;
    int C;
    int K;               // group size (pooled)
    int kshift;          // log2(K) when K is a power of two (every sampled level), else -1: position -> (group, member) by shift / mask
    const float* rx;     // *_RC: the level's input rows X0 [P, 4]
    const float* rw;     // *_RC: the first layer's weight W0 [C, 4]
    BnSite bn;           // the kernel is the FIRST consumer of this operand's constants (s, t or a, e, f): derive them in its prologue
};

// Per-channel constants of 4 consecutive channels, loaded ONCE per thread and tile (not per element).
struct ChanConst {
    float4 s, t, a, e, f;
    float4 w[4];         // *_RC: W0 rows of the thread's 4 channels
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// bf16 ACTIVATION STORAGE (the bf16 variant of BASELINE configs[4], chains that run entirely on the position-stream kernels): the raw
// pre-BatchNorm activations Z_l and the activation gradients G_l live in memory as bf16 -- half the bytes of the kernels that are
// HBM-bound on them.  A buffer keeps its `const float*` type in the operand structs; `h` says how to read it.  The BatchNorm sums are
// taken from the fp32 accumulators BEFORE the rounding; everything downstream (the next layer, the backward pass) sees the rounded value.
__device__ __forceinline__ float4 ld4h(const float* base, size_t elem)      // 4 consecutive bf16 elements (8 bytes) -> fp32
{
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <bool H>
__device__ __forceinline__ float4 ldz4(const float* base, size_t elem) { if constexpr (H) return ld4h(base, elem); else return ld4(base + elem); }
__device__ __forceinline__ float4 ldz4(const float* base, size_t elem, int h) { return h ? ld4h(base, elem) : ld4(base + elem); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)            // two fp32 -> one dword of bf16 (nearest even), lo in the low half
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<const unsigned*>(&v);
}
__device__ __forceinline__ void st4h(float* base, size_t elem, const float4& v)   // 4 consecutive elements as bf16 (8 bytes)
{
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + elem) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}
__device__ __forceinline__ float comp(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// lds != NULL and o.bn on: the constants this kernel derived itself (bn_prologue) -- (scale, shift) of an activation operand or
// (a, e, f) of a dZ operand -- are read from lds[k * ld + (c - c0)]; everything else from the global arrays as before.
template <int MODE>
__device__ __forceinline__ void load_consts(const PosOperand& o, int c, ChanConst& k, const float* lds = nullptr, int ld = 0, int c0 = 0)
{
    const int cc = c < o.C ? c : 0;  // clamped: out-of-range channels are zeroed by the `ok` flag of their data
    const bool own = lds != nullptr && o.bn.slots != nullptr;
    const int cl = c < o.C ? c - c0 : 0;
    if constexpr (!is_id(MODE)) {
        if (!is_dz(MODE) && own) {
            k.s = *reinterpret_cast<const float4*>(lds + cl);
            k.t = *reinterpret_cast<const float4*>(lds + ld + cl);
        } else {
            k.s = ld4(o.s + cc);
            k.t = ld4(o.t + cc);
        }
    }
    if constexpr (is_dz(MODE)) {
        if (own) {
            k.a = *reinterpret_cast<const float4*>(lds + cl);
            k.e = *reinterpret_cast<const float4*>(lds + ld + cl);
            k.f = *reinterpret_cast<const float4*>(lds + 2 * ld + cl);
        } else {
            k.a = ld4(o.a + cc);
            k.e = ld4(o.e + cc);
            k.f = ld4(o.f + cc);
        }
    }
    if constexpr (is_rc(MODE)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) k.w[j] = ld4(o.rw + (size_t)(cc + j) * 4);
    }
}

// Staging is split in two so that global-load latency hides under the MFMAs of the current K chunk:
//   raw_load : issues the (unconditional, address-clamped) loads of 4 channels of one row, no arithmetic;
//   finish   : BN / ReLU / dZ algebra on those registers, executed when the tile is written to LDS (after the MFMAs).
template <int MODE>
struct Raw4 {
    float4 z;
    float4 g;
    int4 ak;
    int kk;   // position inside its group (pooled)
    bool ok;
};

template <int MODE, bool H = false>       // H: this operand's stored Z (and dense G) are bf16 in memory (ld4h)
__device__ __forceinline__ void raw_load(const PosOperand& o, int P, int p, int c, Raw4<MODE>& r)
{
    r.ok = (p < P) && (c < o.C);
    const int pp = r.ok ? p : 0, cc = r.ok ? c : 0;
    if constexpr (is_rc(MODE)) r.z = ld4(o.rx + (size_t)pp * 4);   // the input row; raw_z() turns it into 4 channels of z
    else r.z = ldz4<H>(o.x, (size_t)((unsigned)pp * (unsigned)o.C + (unsigned)cc));
    if constexpr (MODE == SRC_DZ || MODE == SRC_DZ_RC) {
        r.g = ldz4<H>(o.g, (size_t)((unsigned)pp * (unsigned)o.C + (unsigned)cc));
    } else if constexpr (MODE == SRC_DZ_POOLED) {
        unsigned grp;
        if (o.kshift >= 0) { grp = (unsigned)pp >> o.kshift; r.kk = pp & (o.K - 1); }      // (a division by a run-time K is ~20 VALU instructions)
        else { grp = (unsigned)pp / (unsigned)o.K; r.kk = pp - (int)(grp * (unsigned)o.K); }
        const size_t off = (size_t)(grp * (unsigned)o.C + (unsigned)cc);
        r.g = ld4(o.g + off);
        r.ak = *reinterpret_cast<const int4*>(o.argk + off);
    }
}

template <int MODE>
__device__ __forceinline__ float xf1(float z, float g, float s, float t, float a, float e, float f)
{
    if constexpr (is_id(MODE)) {
        return z;
    } else if constexpr (MODE == SRC_ACT || MODE == SRC_ACT_RC) {
        const float y = z * s + t;
        return y > 0.0f ? y : 0.0f;
    } else if constexpr (MODE == SRC_DZ_POOLED) {
        // g is the pooled gradient at the group's arg-max member and 0 elsewhere, already masked by [pooled output > 0]
        // (pool_bwd_prep_kernel) -- and the pooled output IS relu(z * s + t) of that member, so the ReLU mask is in g
        return a * g + (e * z + f);
    } else {
        const float y = z * s + t;
        const float dy = y > 0.0f ? g : 0.0f;
        return a * dy + (e * z + f);
    }
}

// the k-ordered FMA chain of the MFMA kernels (which start from a zero accumulator): bit-identical to a stored Z
__device__ __forceinline__ float dot4_rc(const float4& x, const float4& w)
{
    return __builtin_fmaf(x.w, w.w, __builtin_fmaf(x.z, w.z, __builtin_fmaf(x.y, w.y, x.x * w.x)));
}

// raw pre-BatchNorm z of the operand's 4 channels
template <int MODE>
__device__ __forceinline__ float4 raw_z(const Raw4<MODE>& r, const ChanConst& k)
{
    if constexpr (is_rc(MODE)) return make_float4(dot4_rc(r.z, k.w[0]), dot4_rc(r.z, k.w[1]), dot4_rc(r.z, k.w[2]), dot4_rc(r.z, k.w[3]));
    else return r.z;
}

template <int MODE>
__device__ __forceinline__ float4 finish(const Raw4<MODE>& r, const ChanConst& k)
{
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 z = raw_z<MODE>(r, k);
    if constexpr (MODE == SRC_DZ || MODE == SRC_DZ_RC) g = r.g;
    if constexpr (MODE == SRC_DZ_POOLED) {
        g.x = r.ak.x == r.kk ? r.g.x : 0.0f;
        g.y = r.ak.y == r.kk ? r.g.y : 0.0f;
        g.z = r.ak.z == r.kk ? r.g.z : 0.0f;
        g.w = r.ak.w == r.kk ? r.g.w : 0.0f;
    }
    float4 o;
    o.x = xf1<MODE>(z.x, g.x, k.s.x, k.t.x, k.a.x, k.e.x, k.f.x);
    o.y = xf1<MODE>(z.y, g.y, k.s.y, k.t.y, k.a.y, k.e.y, k.f.y);
    o.z = xf1<MODE>(z.z, g.z, k.s.z, k.t.z, k.a.z, k.e.z, k.f.z);
    o.w = xf1<MODE>(z.w, g.w, k.s.w, k.t.w, k.a.w, k.e.w, k.f.w);
    if (!r.ok) o = make_float4(0.f, 0.f, 0.f, 0.f);
    return o;
}

// bf16 variant of the VALU-side products (factorised / recomputed first layers): an operand rounded to bf16 (nearest even) when `on`
__device__ __forceinline__ float rb16(float x, int on) { return on ? (float)(__bf16)x : x; }
__device__ __forceinline__ float4 rb16(const float4& v, int on) { return make_float4(rb16(v.x, on), rb16(v.y, on), rb16(v.z, on), rb16(v.w, on)); }

// Plain matrix rows (weights): row-major [R, C] with C % 4 == 0, no transform, zero outside.
__device__ __forceinline__ float4 ld4_plain(const float* m, int R, int C, int ld, int r, int c)
{
    const bool ok = r < R && c < C;   // C = valid columns, ld = row stride (both multiples of 4)
    const float4 v = ld4(m + (size_t)((unsigned)(ok ? r : 0) * (unsigned)ld + (unsigned)(ok ? c : 0)));
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- MFMA chunk: acc += A_tile(BM x BK) * B_tile(BK x BN) for this wave's TM x TN sub-tiles ------------------
template <bool A_KROW, bool B_KROW, int LDA, int LDB, int TM, int TN, int KC>
__device__ __forceinline__ void mma_chunk(const float* sA, const float* sB, int wrow0, int wcol0, f32x16 (&acc)[TM][TN])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
        const int k = kk + hi;
        float a[TM], b[TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int r = wrow0 + mi * 32 + l31;
            a[mi] = A_KROW ? sA[k * LDA + r] : sA[r * LDA + k];
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int c = wcol0 + ni * 32 + l31;
            b[ni] = B_KROW ? sB[k * LDB + c] : sB[c * LDB + k];
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}

// ---- bf16 operands (BASELINE configs[4]: "bf16 MFMA grouped-MLP") ------------------------------------------------------
// The same GEMMs with both operands rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) as they are staged into LDS and
// v_mfma_f32_32x32x16_bf16 accumulating in fp32: 16x the fp32 MFMA rate.  Everything around the contraction stays fp32: the
// stored raw activations Z, BatchNorm / ReLU / dZ algebra (applied BEFORE the rounding, while staging), the per-column sums,
// the pool, dW accumulation.  LDS tiles are straight images of the memory layout -- [row][k] where k is contiguous in memory
// (fragment = one ds_read_b128 of 8 consecutive k), [k][col] where the columns are (fragment = two ds_read_b64_tr_b16
// hardware-transposed reads of 4 k each) -- so staging is the same coalesced float4 traffic as the fp32 path.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v)
{
    bf16x4 r = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return r;
}

// row stride (in elements) of a [k][col] bf16 tile read with ds_read_b64_tr_b16: >= n, 8-byte aligned rows, and the four rows
// of a transposed block land in four disjoint 16-dword bank windows (stride/2 = 16 or 48 mod 64)
constexpr int tr_ld(int n) { return (n % 128 <= 32) ? (n / 128 * 128 + 32) : (n % 128 <= 96 ? n / 128 * 128 + 96 : n / 128 * 128 + 160); }

// Fragment of a 32x32x16 MFMA operand from a [k][col] tile: lane (c = lane & 31, h = lane >> 5) gets rows k0 + 8h .. + 7 of
// column c0 + c.  Per 16-lane group one ds_read_b64_tr_b16 takes a 4-row x 16-column block and hands lane i column i;
// lane 4q + p supplies the address of row q, columns 4p .. 4p+3.  EXEC must be full (it is: no divergence in the main loops).
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int k0, int c0)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = i >> 2, pp = i & 3, nh = (lane >> 4) & 1, h = lane >> 5;
    const __bf16* p = tile + (k0 + 8 * h + q) * ld + c0 + 16 * nh + 4 * pp;
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 4 * ld));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 contraction on the bf16 matrix cores ("split" kernels): x = h + m + l with three bf16 numbers (8 + 8 + 8 significant
// bits: the differences x - h and (x - h) - m are exact in fp32, so only l is rounded, at 2^-25 |x|), and
//   x * w  =  h_x h_w + (h_x m_w + m_x h_w) + (m_x m_w + h_x l_w + l_x h_w)  +  O(2^-24 |x w|)
// -- the six products of order <= 2^-16, each EXACT in the fp32 accumulate of v_mfma_f32_32x32x16_bf16.  What is dropped
// (m l, l m, l l) is of the size of ONE fp32 rounding of the product, i.e. the result carries the same error as the fp32 FMA
// chain it replaces (tests/test_gpu_split.py measures both against fp64).  Six bf16 MFMAs of K = 16 take 6 x 8 passes where
// the eight v_mfma_f32_32x32x2_f32 they replace take 8 x 16 -- and, unlike the fp32 MFMA, they run beside the VALU instead
// of on it.
// ---------------------------------------------------------------------------------------------------------------
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

// [r5] Two planes (h, m) for the BACKWARD contractions: x = h + m + O(2^-18 |x|) (h rounded to nearest: |x - h| <= 2^-9 |x|, m the bf16 of the
// exact difference), x * w = h_x h_w + (h_x m_w + m_x h_w) + O(2^-17 |x w|): three plane products instead of six, two thirds of the staging's split
// arithmetic and LDS writes, two thirds of the weight fragments.  The result carries ~1e-5 of relative error per product -- random in sign,
// averaged over the 64 ... 262 144 terms of a sum -- where the three-plane form carries 6e-8: used for gradients only (the forward pass, whose
// outputs north_star holds to 1e-5, keeps six products); gate: tests/test_gpu_routing.py (every parameter gradient within twice the fp32
// oracle's distance from the fp64 evaluation).  MP_BWD_PLANES=3 restores the three-plane backward (read on every call).
__device__ __forceinline__ Split4 split2(const float4& v)
{
    Split4 r;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)x[i];
        r.h[i] = h;
        r.m[i] = (__bf16)(x[i] - (float)h);
    }
    r.l = r.m;
    return r;
}
template <int NPL>
__device__ __forceinline__ Split4 splitn(const float4& v) { if constexpr (NPL == 2) return split2(v); else return split3(v); }

// K-packed bf16 tile for the split backward kernel: element (row, c) of a [rows][C] chunk sits at (c / 8) * GS + row * 8 + c % 8
// (GS = rows * 8 + 32 halves: 16-byte groups of 8 channels, rows of one group contiguous, groups 16 dwords apart mod 64 banks:
// the transposed read's 16 lanes -- 4 rows x 2 half-groups x 2 groups -- then cover 32 distinct banks, 32 lanes all 64).
// One image serves both contractions that read the chunk:
//   * as A[row][k = c] of a 16x16x32 MFMA: lane (row, kq) reads the 16 bytes of group 4*st + kq -- 16 lanes = 256 contiguous bytes;
//   * as [k = row][col = c] of a 32x32x16 MFMA through ds_read_b64_tr_b16: lane 4q + p supplies row q, columns 4p .. 4p+3.
// [r3] Swizzled form (SWZ; 16-row chunks, GS = 128 halves: no pad): row r of group g sits at row r ^ S[g & 3], S = {0, 12, 4, 8} (+ low bits, below).
// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) -- NOT 16 consecutive lanes -- so the
// A-fragment read above (lane (row, kq) -> group 4 st + kq) met rows {0-3, 12-15} of one group and rows {4-11} of the next in ONE
// LDS cycle: with groups 16 dwords apart those overlap on 16 banks (the 40-49 % SQ_LDS_BANK_CONFLICT of the fused backward kernels
// in profiles/r02_sq_counters.md).  With groups a whole 64-bank line apart and the XOR, both reads are conflict-free: the b128
// lane groups see 16 distinct rows (S[0] ^ S[1] = S[2] ^ S[3] = 12 maps {4..11} onto itself), and the transposed read's four groups
// x four rows land in four disjoint 16-dword windows (the S values are the four multiples of 4).
#ifndef MP_KSWZ
#define MP_KSWZ 1
#endif
// The low bits of S (S = {0, 14, 4, 10}) do not move a row out of its 4-row window; they spread the staging writes (ds_write_b64: 16
// consecutive lanes per LDS cycle, 32 banks) of the four groups over distinct residues of row mod 8.  (S depends on g & 3 only: the
// lane part of every read address stays one loop-invariant register; a fifth bit for g >> 2 cost address registers the 256-output
// kernel does not have.)
__device__ __forceinline__ constexpr int kswz(int g) { return (int)((0xA4E0u >> ((g & 3) * 4)) & 15u); }
template <int GS, bool SWZ = false>
__device__ __forceinline__ bf16x8 tr_frag_packed(const __bf16* tile, int k0, int c0)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = i >> 2, pp = i & 3, nh = (lane >> 4) & 1, h = lane >> 5;
    const int g = (c0 >> 3) + 2 * nh + (pp >> 1), r = k0 + 8 * h + q;          // (k0 % 8 == 0: r + 4 == r ^ 4)
    const int pr = SWZ ? (r ^ kswz(g)) : r;
    const __bf16* p = tile + g * GS + pr * 8 + 4 * (pp & 1);
    const __bf16* p2 = SWZ ? tile + g * GS + (pr ^ 4) * 8 + 4 * (pp & 1) : p + 4 * 8;
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p2));
    bf16x8 r8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r8;
}

// [r4] Every transposed fragment requested so far has ARRIVED behind this point.  Found with the 32-position one-plane kernel: the
// compiler's partial `s_waitcnt lgkmcnt(N)` in front of the MFMAs that consume ds_read_b64_tr_b16 results -- correct if LDS operations
// complete in issue order -- let stale fragment registers into the dW product whenever other waves' staging writes kept the LDS busy
// (run-to-run varying weight gradients; a full lgkmcnt(0) here removes it, NOTEBOOK.md).  MP_TR_FENCE: bit 0 the one-plane dW product,
// bit 1 the three-plane dW product of bwd_fused_kernel (a fence behind each fragment batch), bit 2 the dW waves of bwd_roles_kernel, bit 3 the tiled GEMM kernels' chunk products (mma_chunk_bf16 / mma_chunk_split with a
// transposed operand).
#ifndef MP_TR_FENCE
#define MP_TR_FENCE 15      // every consumer of transposed fragments (no measurable cost: headline 2.157 vs 2.149 ms median over three alternations;
                            // the tiled GEMMs of the group_all level 322 vs 317 us)
#endif
__device__ __forceinline__ void tr_fence()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
}

// acc += A * B over one K chunk.  A_TR / B_TR: the operand's tile is [k][row or col] (transposed reads) instead of [row][k].
template <bool A_TR, bool B_TR, int LDA, int LDB, int TM, int TN, int KC>
__device__ __forceinline__ void mma_chunk_bf16(const __bf16* sA, const __bf16* sB, int wrow0, int wcol0, f32x16 (&acc)[TM][TN])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            if constexpr (A_TR) a[mi] = tr_frag(sA, LDA, ks, wrow0 + mi * 32);
            else a[mi] = *reinterpret_cast<const bf16x8*>(sA + (wrow0 + mi * 32 + l31) * LDA + ks + 8 * hi);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if constexpr (B_TR) b[ni] = tr_frag(sB, LDB, ks, wcol0 + ni * 32);
            else b[ni] = *reinterpret_cast<const bf16x8*>(sB + (wcol0 + ni * 32 + l31) * LDB + ks + 8 * hi);
        }
        if constexpr ((A_TR || B_TR) && ((MP_TR_FENCE >> 3) & 1)) tr_fence();
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}

// The six plane products of the split scheme over one K chunk of the tiled kernels: sA / sB point at plane 0, planes are PSA / PSB
// elements apart.  One A plane is live at a time (l, then h, then m), the three B planes stay.
// [r6] NPL = 2: the two-plane form of the backward contractions (split2: h, m; three products m h, h m, h h -- smallest first), as the
// position-stream backward kernels have used since r5; two thirds of the fragment reads, half the MFMAs.
template <bool A_TR, bool B_TR, int LDA, int LDB, int TM, int TN, int KC, int PSA, int PSB, int NPL = 3>
__device__ __forceinline__ void mma_chunk_split(const __bf16* sA, const __bf16* sB, int wrow0, int wcol0, f32x16 (&acc)[TM][TN])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    auto afrag = [&](int pl, int ks, int mi) -> bf16x8 {
        if constexpr (A_TR) return tr_frag(sA + pl * PSA, LDA, ks, wrow0 + mi * 32);
        else return *reinterpret_cast<const bf16x8*>(sA + pl * PSA + (wrow0 + mi * 32 + l31) * LDA + ks + 8 * hi);
    };
    auto bfrag = [&](int pl, int ks, int ni) -> bf16x8 {
        if constexpr (B_TR) return tr_frag(sB + pl * PSB, LDB, ks, wcol0 + ni * 32);
        else return *reinterpret_cast<const bf16x8*>(sB + pl * PSB + (wcol0 + ni * 32 + l31) * LDB + ks + 8 * hi);
    };
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
        bf16x8 a[TM], b[NPL][TN];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b[pl][ni] = bfrag(pl, ks, ni);
        // three planes: A plane l meets B plane h; h meets h, m, l; m meets h, m.  Two planes: m meets h; h meets h, m.
        constexpr int APL[3] = {NPL == 3 ? 2 : 1, 0, 1}, NB[3] = {1, NPL, 2};
#pragma unroll
        for (int s = 0; s < NPL; ++s) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) a[mi] = afrag(APL[s], ks, mi);
            if constexpr ((A_TR || B_TR) && ((MP_TR_FENCE >> 3) & 1)) tr_fence();
#pragma unroll
            for (int pl = NB[s] - 1; pl >= 0; --pl)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[pl][ni], acc[mi][ni], 0, 0, 0);
        }
    }
}

// The same chunk product with the operand fetch software-pipelined: the fragments of k-step s+1 are requested from LDS before
// the MFMAs of step s are issued.  (Left to itself the compiler emits "ds_read; s_waitcnt lgkmcnt(0); MFMA x TM*TN" per step,
// i.e. every group of MFMAs waits out a full LDS round trip -- visible as ~60 % MFMA utilisation of the fused backward
// kernels in profiles/r02_mfma_util.md.)  Same products in the same k order: bit-identical results.
template <bool A_KROW, bool B_KROW, int LDA, int LDB, int TM, int TN, int KC>
__device__ __forceinline__ void mma_chunk_pipelined(const float* sA, const float* sB, int wrow0, int wcol0, f32x16 (&acc)[TM][TN])
{
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, hi = lane >> 5;
    float a[2][TM], b[2][TN];
    auto fetch = [&](int kk, float (&av)[TM], float (&bv)[TN]) {
        const int k = kk + hi;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int r = wrow0 + mi * 32 + l31;
            av[mi] = A_KROW ? sA[k * LDA + r] : sA[r * LDA + k];
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int c = wcol0 + ni * 32 + l31;
            bv[ni] = B_KROW ? sB[k * LDB + c] : sB[c * LDB + k];
        }
    };
    fetch(0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
        const int cur = (kk >> 1) & 1;
        if (kk + 2 < KC) fetch(kk + 2, a[cur ^ 1], b[cur ^ 1]);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
    }
}

// accumulator register r of this lane -> row inside a 32-row MFMA tile
__device__ __forceinline__ int acc_row_in_tile(int r) { return (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x & 63) >> 5); }

// EPI_SQ_POOL (last forward layer, group size K in {32, 64, 128}): besides Z and the BatchNorm sums the epilogue
// reduces every group of K rows to (max, argmax, min, argmin) of the RAW z per channel -- the max-pool commutes with
// the monotone map z -> relu(z*scale + shift), max for scale >= 0, min for scale < 0 -- so the pooled output needs no
// second pass over Z once the batch statistics are known (pool_select_kernel).
enum Epi { EPI_NONE = 0, EPI_SQ = 1, EPI_DY = 2, EPI_SQ_POOL = 3 };      // epilogues of pos_gemm_kernel (sa_gemm.hip)
struct PoolOut {
    float* vmax;  // [P/K, N]
    float* vmin;
    int* imax;
    int* imin;
    int K;
};

}  // namespace

// sa_stream16.hip ([r5] one-plane bf16 position-stream kernels): 1 = launched, 0 = no kernel for this shape, < 0 = error.  dz / in point at
// PosOperand, partials at BnOut.
int mp_s16_bwd_launch(int pooled, int rc_in, int Co, int Ci, const void* dz, const void* in, int64_t P, int ppb, const float* W, float* dW, float* G,
                      const void* partials, const char* tag, double flops, double bytes, hipStream_t stream);
int mp_s16_fwd_launch(int pool, int rc_in, int Ci, int Co, const void* a, int64_t P, int ppb, const float* W, float* Z, const void* partials,
                      const void* po, const float* gamma, const char* tag, double flops, double bytes, hipStream_t stream);
#ifndef MP_MAPWIDE
#define MP_MAPWIDE 0        // 1: 16 lanes per row for every plane write of the fused backward kernels (A/B builds)
#endif
// sa_gemm.hip ([r6] the tiled GEMM kernels, split out of sa_mlp.hip): mode / epi / prec as the kernels' template arguments (SRC_*, Epi, PREC);
// a / dz / in point at PosOperand, partials at BnOut, po at PoolOut (may be NULL)
int mp_pos_gemm_launch(int mode, int w_krow, int epi, int prec, const void* a, int64_t P, const float* W, int N, int Kd, float* C, const void* partials,
                       const float* zprev, const float* sprev, const float* tprev, hipStream_t stream, int* nblk_out, const void* po, int ldw, int ldc);
int mp_dw_gemm_launch(int mode_dz, int mode_in, int prec, const void* dz, const void* in, int64_t P, float* dW, hipStream_t stream);
int mp_bwd_pair_launch(int mode_dz, int mode_in, int epi, const void* dz, const void* in, int64_t P, const float* W, int N, int Kd, float* G, const void* partials,
                       const float* zprev, const float* sprev, const float* tprev, int ldw, int ldc, float* dW, hipStream_t stream, int* nblk_out, int probe);
int mp_dw_ci4_rc_launch(const void* dz, const void* in, int64_t P, float* dW, int r16, int h16, double flops, double bytes, hipStream_t stream);
// sa_bwd_fused.hip
int mp_bwd_fused_launch(int pooled, int rc_in, int bf16, int split, int npl, int Co, int Ci, const void* dz, const void* in, int64_t P, int ppb,
                        const float* W, float* dW, float* G, const void* partials, double flops, double bytes, double bytes_rc, hipStream_t stream);
