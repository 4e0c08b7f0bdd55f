// Library-level entry points: version, thread-local error string, device probe.
#include "common.hpp"
#include <string.h>

namespace blgnn {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace blgnn

extern "C" int blgnn_version(void) { return 100; }   // 0.1.0

extern "C" const char *blgnn_last_error(void) { return blgnn::err_buf(); }

extern "C" int blgnn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
