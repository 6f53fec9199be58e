// Error string + ABI version of libpianobart_hip.so.
#include "pb_common.h"
#include "pb_api_internal.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void pb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* pb_last_error(void) { return g_err; }
extern "C" int pb_abi_version(void) { return PB_ABI_VERSION; }

// Events for the engine's two-stream schedule (device-local producer/consumer ordering between two streams of ONE GPU): created
// without the system-scope fence a default event carries -- the cache write-back / invalidate at every record showed up as a
// ~7 us bubble on the recording stream, ~175 times per training step. mode 0: hipEventDisableTiming only (what torch.cuda.Event()
// gives); 1: + hipEventDisableSystemFence (what the engine uses); 2: + hipEventReleaseToDevice (the runtime rejects 1 and 2 together).
extern "C" int pb_event_create(void** ev, int32_t mode) {
    PB_REQUIRE(ev != nullptr, "pb_event_create: NULL");
    unsigned flags = hipEventDisableTiming;
    if (mode & 1) flags |= hipEventDisableSystemFence;
    if (mode & 2) flags |= hipEventReleaseToDevice;
    hipEvent_t e;
    PB_CHECK_HIP(hipEventCreateWithFlags(&e, flags));
    *ev = (void*)e;
    return 0;
}
extern "C" int pb_event_destroy(void* ev) {
    if (ev) PB_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
extern "C" int pb_event_record(void* ev, void* stream) {
    PB_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int pb_stream_wait_event(void* stream, void* ev) {
    PB_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side of the generate loop: nucleus() of model.py:84-98 for the 8 heads of one position, given the softmax rows
// and the 8 uniform draws np.random.choice would have consumed (RandomState.choice with p: one random_sample per call). Pure host
// arithmetic in numpy's own order and precision (numpy >= 2 scalar rules): float32 left-to-right sums, float32 quotients, the
// candidates' cdf in float64, searchsorted(side='right'). The descending order of EQUAL probabilities is the one thing numpy's
// unstable argsort decides and this does not reproduce: a head whose result could depend on it (a tie among its candidates or
// with the first excluded entry) is reported in *tie_mask and left to the numpy code. The per-position host time of the decode loop
// (in series with the GPU) drops from ~0.2 ms of small numpy calls to a few microseconds.
#include <algorithm>
#include <vector>
extern "C" int pb_nucleus_rows(const float* probs, int32_t width, const int32_t* n, const float* p, const double* u, int32_t heads,
                               int32_t* out, int32_t* tie_mask) {
    PB_REQUIRE(probs && n && p && u && out && tie_mask && heads > 0 && heads <= 32 && width > 0, "pb_nucleus_rows: bad argument");
    *tie_mask = 0;
    std::vector<std::pair<float, int>> v;
    std::vector<float> q;
    for (int h = 0; h < heads; ++h) {
        const float* row = probs + (size_t)h * width;
        const int len = n[h];
        PB_REQUIRE(len > 0 && len <= width, "pb_nucleus_rows: row length %d", len);
        float s = row[0];
        for (int i = 1; i < len; ++i) s = s + row[i];                        // np.cumsum(probs)[-1]
        const volatile float c = s + 1e-5f;                                  // float32 + weak Python float
        v.resize(len);
        for (int i = 0; i < len; ++i) v[i] = {row[i] / c, i};                // probs /= (sum + 1e-5)
        std::sort(v.begin(), v.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
        int k = 1;                                                           // candidates: up to and including the first cumsum > p; none -> top 1
        if (p[h] < 1.0f) {
            float cs = v[0].first;
            int first = cs > p[h] ? 0 : -1;
            for (int i = 1; i < len && first < 0; ++i) { cs = cs + v[i].first; if (cs > p[h]) first = i; }
            k = first < 0 ? 1 : first + 1;
        }
        bool tie = false;
        for (int i = 0; i < k && i + 1 < len; ++i) tie = tie || v[i].first == v[i + 1].first;
        if (tie) { *tie_mask |= 1 << h; out[h] = -1; continue; }
        q.resize(k);
        float qs = v[0].first;
        for (int i = 1; i < k; ++i) qs = qs + v[i].first;                    // np.cumsum(q)[-1]
        double cum = 0.0, last = 0.0;
        for (int i = 0; i < k; ++i) { q[i] = v[i].first / qs; last += (double)q[i]; }
        int idx = k - 1;
        for (int i = 0; i < k; ++i) {                                        // cdf = cumsum(q as f64) / cdf[-1]; first index with cdf > u
            cum += (double)q[i];
            if (cum / last > u[h]) { idx = i; break; }
        }
        out[h] = v[idx].second;
    }
    return 0;
}
