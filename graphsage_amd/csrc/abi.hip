// Version and error plumbing of libgsage.so.
#include "gs_common.h"

namespace gs {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static thread_local LaunchTimer *armed_timer = nullptr;

LaunchTimer *take_armed_timer() {
    LaunchTimer *t = armed_timer;
    armed_timer = nullptr;
    return t;
}

}  // namespace gs

extern "C" int gs_timer_create(void **timer) {
    if (!timer) return gs::fail(GS_EINVAL, "gs_timer_create: null pointer argument");
    gs::LaunchTimer *t = new gs::LaunchTimer{nullptr, nullptr};
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) {
        if (t->start) (void)hipEventDestroy(t->start);
        delete t;
        return gs::fail(GS_ELAUNCH, "gs_timer_create: cannot create events");
    }
    *timer = t;
    return GS_OK;
}

extern "C" int gs_timer_arm(void *timer) {
    gs::armed_timer = static_cast<gs::LaunchTimer *>(timer);     // NULL disarms
    return GS_OK;
}

extern "C" int gs_timer_elapsed_us(void *timer, float *us) {
    if (!timer || !us) return gs::fail(GS_EINVAL, "gs_timer_elapsed_us: null pointer argument");
    gs::LaunchTimer *t = static_cast<gs::LaunchTimer *>(timer);
    hipError_t e = hipEventSynchronize(t->stop);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t->start, t->stop);
    if (e != hipSuccess) return gs::fail(GS_ELAUNCH, "gs_timer_elapsed_us: %s", hipGetErrorString(e));
    *us = ms * 1000.f;
    return GS_OK;
}

extern "C" int gs_timer_destroy(void *timer) {
    if (!timer) return GS_OK;
    gs::LaunchTimer *t = static_cast<gs::LaunchTimer *>(timer);
    if (gs::armed_timer == t) gs::armed_timer = nullptr;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return GS_OK;
}

extern "C" int gs_version(void) { return 7; }

extern "C" const char *gs_last_error(void) { return gs::error_buffer(); }
