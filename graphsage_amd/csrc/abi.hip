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

}  // namespace gs

extern "C" int gs_version(void) { return 1; }

extern "C" const char *gs_last_error(void) { return gs::error_buffer(); }
