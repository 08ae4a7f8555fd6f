// ABI version and error strings.
#include "common.h"

extern "C" int mp_abi_version(void) { return MP_ABI_VERSION; }

extern "C" const char* mp_error_string(int code)
{
    switch (code) {
        case MP_OK: return "ok";
        case MP_EINVAL: return "invalid argument (null pointer or inconsistent dimension)";
        case MP_EUNSUPPORTED: return "size outside the range the gfx950 kernels are built for";
        case MP_EWORKSPACE: return "workspace too small";
        case MP_ELAUNCH: return "HIP launch failed";
        default: return "unknown error";
    }
}

// ---- kernel profiler (debug/bench facility; process-wide, off by default: with the zero arena's table below the only such state) ----
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>

namespace {
struct ProfRec {
    std::string tag;
    double flops, bytes;
    hipEvent_t a, b;
};
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;

// [r5] Marks: while a stream RECORDS (hipGraph capture), launches whose tag contains a marked substring are bracketed by two one-thread
// kernels that append the device's wall clock (100 MHz) to a ring, so every replay of the graph timestamps them; mp_profiler_read_marks()
// turns the last `last_n` pairs into durations.  bench.py marks the kernels that can be the step's largest: the roofline figure then is the
// kernel's AVERAGE over the K timed replays (no eagerly launched step inside the timed region; ~3 us per marked launch for the two stamps,
// whose own boundaries are inside the measured interval: it overstates the kernel by about that much).  (Event records cannot do this here:
// a plain record on a recording stream is a dependency marker that no replay signals, and hipEventRecordExternal nodes make the recording
// fail over to eager launches on this runtime.)
constexpr int MARK_MAX = 8, MARK_CAP = 256;           // marked launches per recording; samples kept per launch
struct Mark { std::string tag; double flops, bytes; };
std::atomic<int> g_mark_on{0};
std::string g_mark;
std::vector<Mark> g_marks;
unsigned long long* g_mark_dev = nullptr;            // [MARK_MAX][2 + 2 * MARK_CAP]: begin count, end count, begin ring, end ring
int g_mark_open = -1;
__global__ void mp_stamp_kernel(unsigned long long* ctr, unsigned long long* ring)
{
    const unsigned long long i = atomicAdd(ctr, 1ull);
    ring[i % MARK_CAP] = wall_clock64();
}
bool mark_matches(const char* tag)          // g_mark: substrings separated by '|'
{
    size_t a = 0;
    while (a <= g_mark.size()) {
        size_t b = g_mark.find('|', a);
        if (b == std::string::npos) b = g_mark.size();
        if (b > a && strstr(tag, g_mark.substr(a, b - a).c_str())) return true;
        a = b + 1;
    }
    return false;
}
}  // namespace

namespace mp {
bool prof_on() { return g_prof_on.load(std::memory_order_relaxed) != 0 || g_mark_on.load(std::memory_order_relaxed) != 0; }

void prof_begin(const char* tag, double flops, double bytes, hipStream_t stream)
{
    if (g_prof_on.load(std::memory_order_relaxed) == 0) {        // marks only
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_mark_open = -1;
        if (g_mark.empty() || !g_mark_dev || (int)g_marks.size() >= MARK_MAX || !mark_matches(tag) || hipStreamIsCapturing(stream, &st) != hipSuccess ||
            st != hipStreamCaptureStatusActive)
            return;
        g_marks.push_back(Mark{tag, flops, bytes});
        g_mark_open = (int)g_marks.size() - 1;
        unsigned long long* m = g_mark_dev + (size_t)g_mark_open * (2 + 2 * MARK_CAP);
        hipLaunchKernelGGL(mp_stamp_kernel, dim3(1), dim3(1), 0, stream, m, m + 2);
        return;
    }
    ProfRec r{tag, flops, bytes, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(r);
}

void prof_end(hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_on.load(std::memory_order_relaxed) == 0) {
        if (g_mark_open >= 0) {
            unsigned long long* m = g_mark_dev + (size_t)g_mark_open * (2 + 2 * MARK_CAP);
            hipLaunchKernelGGL(mp_stamp_kernel, dim3(1), dim3(1), 0, stream, m + 1, m + 2 + MARK_CAP);
        }
        g_mark_open = -1;
        return;
    }
    if (!g_prof_recs.empty()) (void)hipEventRecord(g_prof_recs.back().b, stream);
}
}  // namespace mp

extern "C" int mp_profiler_enable(int on)
{
    g_prof_on.store(on ? 1 : 0);
    return MP_OK;
}

// tag_substr: which launches to mark while a stream records, substrings separated by '|' (NULL or "": none).  Marks of an earlier call are
// forgotten (graphs recorded with them keep stamping into the same buffer: read them before marking again).  Not while a stream records.
extern "C" int mp_profiler_mark(const char* tag_substr)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_mark = tag_substr ? tag_substr : "";
    g_marks.clear();
    g_mark_open = -1;
    const size_t bytes = (size_t)MARK_MAX * (2 + 2 * MARK_CAP) * sizeof(unsigned long long);
    if (!g_mark.empty()) {
        if (!g_mark_dev && hipMalloc(&g_mark_dev, bytes) != hipSuccess) { g_mark_dev = nullptr; g_mark.clear(); }
        if (g_mark_dev && (hipDeviceSynchronize() != hipSuccess || hipMemset(g_mark_dev, 0, bytes) != hipSuccess)) g_mark.clear();
    }
    g_mark_on.store(g_mark.empty() ? 0 : 1);
    return MP_OK;
}

// The marked launches' durations over their last `last_n` executions (waits for the device), aggregated per tag in mp_profiler_collect's
// format: calls = samples used, total_ms their sum.
extern "C" int mp_profiler_read_marks(char* buf, size_t cap, int last_n)
{
    std::vector<Mark> marks;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        marks = g_marks;
    }
    std::string out;
    if (!marks.empty() && g_mark_dev) {
        const size_t per = 2 + 2 * MARK_CAP;
        std::vector<unsigned long long> host(per * marks.size());
        int dev = 0, khz = 0;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host.data(), g_mark_dev, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
            return MP_ELAUNCH;
        struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
        std::map<std::string, Agg> agg;
        for (size_t m = 0; m < marks.size(); ++m) {
            const unsigned long long* h = host.data() + m * per;
            const unsigned long long n = h[0] < h[1] ? h[0] : h[1];          // completed pairs
            unsigned long long use = n < (unsigned long long)MARK_CAP ? n : MARK_CAP;
            if (last_n > 0 && use > (unsigned long long)last_n) use = last_n;
            Agg& a = agg[marks[m].tag];
            for (unsigned long long i = n - use; i < n; ++i) {
                const unsigned long long t0 = h[2 + i % MARK_CAP], t1 = h[2 + MARK_CAP + i % MARK_CAP];
                if (t1 <= t0) continue;
                a.calls += 1;
                a.ms += (double)(t1 - t0) / (double)khz;
                a.flops += marks[m].flops;
                a.bytes += marks[m].bytes;
            }
        }
        char line[512];
        for (auto& kv : agg) {
            if (!kv.second.calls) continue;
            snprintf(line, sizeof line, "%s\t%ld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.flops, kv.second.bytes);
            out += line;
        }
    }
    if (out.size() + 1 > cap) return MP_EWORKSPACE;
    if (buf) memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// Waits for the recorded events, aggregates per tag and writes lines "tag\tcalls\ttotal_ms\tflops\tbytes\n" into buf.
// Returns the number of bytes written (0 if nothing was recorded), or MP_EWORKSPACE if buf is too small.
extern "C" int mp_profiler_collect(char* buf, size_t cap)
{
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_prof_recs);
    }
    struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : recs) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = agg[r.tag];
            a.calls += 1;
            a.ms += ms;
            a.flops += r.flops;
            a.bytes += r.bytes;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    std::string out;
    char line[512];
    for (auto& kv : agg) {
        snprintf(line, sizeof line, "%s\t%ld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.calls, kv.second.ms,
                 kv.second.flops, kv.second.bytes);
        out += line;
    }
    if (out.size() + 1 > cap) return MP_EWORKSPACE;
    if (buf) memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// ---- zero arena ------------------------------------------------------------------------------------------------------------------
// A training step has half a dozen small outputs that are accumulated with atomics and must start from zero (the scatter targets of
// the nearest-neighbour backward, the input gradients of the wide heads, dA of the factorised first layer): a clear launch each,
// ~5 us apiece on a dependent chain.  The caller may instead carve them out of ONE buffer, clear it with this call at the start of
// the phase, and the library then skips its own clear of every output inside the armed range (same stream only).  Every part of the
// range must be handed out at most once per arming; arming the same stream again (or mp_zero_arena_disarm[_stream]) ends the previous one.
// [r5] One armed range PER STREAM (a table keyed by the stream handle, under a mutex): two training steps on two streams of one process
// do not see each other's arena.  This table and the profiler above are the library's only process-wide state; both are bookkeeping of
// caller-owned resources (no device memory is held).
namespace {
struct ArenaEntry { hipStream_t stream; const char* base; size_t bytes; };
std::mutex g_arena_mu;
std::vector<ArenaEntry> g_arenas;
void arena_set(hipStream_t stream, const char* base, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    for (auto it = g_arenas.begin(); it != g_arenas.end(); ++it)
        if (it->stream == stream) { g_arenas.erase(it); break; }
    if (base && bytes) g_arenas.push_back(ArenaEntry{stream, base, bytes});
}
}  // namespace

namespace mp {
bool zero_arena_covers(const void* p, size_t bytes, hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    const char* q = static_cast<const char*>(p);
    for (const ArenaEntry& e : g_arenas)
        if (e.stream == stream) return q >= e.base && q + bytes <= e.base + e.bytes;
    return false;
}
}  // namespace mp

extern "C" int mp_zero_arena_disarm_stream(mp_stream_t stream_)
{
    arena_set(mp_stream(stream_), nullptr, 0);
    return MP_OK;
}

extern "C" int mp_zero_arena_disarm(void)       // every stream's
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    g_arenas.clear();
    return MP_OK;
}

// [r4] The same launch may advance the step's device-side counters: the num_batches_tracked of every train-mode BatchNorm (int64), the
// dropout step, the dense optimizer's update count (float32) -- each was an elementwise launch of its own on the dependent chain.
namespace {
constexpr int TICK_I64 = 40, TICK_F32 = 8;
struct TickTable {
    long long* i64[TICK_I64];
    float* f32[TICK_F32];
    int n64, n32;
};
__global__ __launch_bounds__(256) void arena_arm_kernel(float* __restrict__ p, size_t n4, TickTable t)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0) {
        const int k = threadIdx.x;
        if (k < t.n64) *t.i64[k] += 1;
        else if (k - t.n64 < t.n32) *t.f32[k - t.n64] += 1.0f;
    }
}
}  // namespace

extern "C" int mp_zero_arena_arm_ticks(void* base, size_t bytes, int n_i64, int64_t* const* counters_i64, int n_f32,
                                       float* const* counters_f32, mp_stream_t stream_)
{
    if (!base || (bytes & 15) || (reinterpret_cast<uintptr_t>(base) & 15) || bytes == 0) return MP_EINVAL;
    if (n_i64 < 0 || n_f32 < 0 || (n_i64 > 0 && !counters_i64) || (n_f32 > 0 && !counters_f32)) return MP_EINVAL;
    if (n_i64 > TICK_I64 || n_f32 > TICK_F32) return MP_EUNSUPPORTED;
    hipStream_t stream = mp_stream(stream_);
    arena_set(stream, nullptr, 0);
    TickTable t{};
    t.n64 = n_i64;
    t.n32 = n_f32;
    for (int k = 0; k < n_i64; ++k) { if (!counters_i64[k]) return MP_EINVAL; t.i64[k] = reinterpret_cast<long long*>(counters_i64[k]); }
    for (int k = 0; k < n_f32; ++k) { if (!counters_f32[k]) return MP_EINVAL; t.f32[k] = counters_f32[k]; }
    const size_t n4 = bytes / 16;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(arena_arm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<float*>(base), n4, t);
    if (hipGetLastError() != hipSuccess) return MP_ELAUNCH;
    arena_set(stream, static_cast<const char*>(base), bytes);
    return MP_OK;
}

extern "C" int mp_zero_arena_arm(void* base, size_t bytes, mp_stream_t stream_)
{
    if (!base || (bytes & 3) || (reinterpret_cast<uintptr_t>(base) & 15)) return MP_EINVAL;
    hipStream_t stream = mp_stream(stream_);
    arena_set(stream, nullptr, 0);
    if (bytes == 0) return MP_OK;
    if (!mp::zero_async(static_cast<float*>(base), bytes / 4, stream)) return MP_ELAUNCH;
    arena_set(stream, static_cast<const char*>(base), bytes);
    return MP_OK;
}
