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
