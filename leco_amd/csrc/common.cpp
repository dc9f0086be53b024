#include "common.h"

#include <errno.h>

namespace leco {
char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-EIO, "%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}
}  // namespace leco

extern "C" int leco_version(void) { return 100; }
extern "C" const char* leco_last_error(void) { return leco::error_buffer(); }
