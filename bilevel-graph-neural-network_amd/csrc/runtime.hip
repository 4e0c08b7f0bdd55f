// Library-level entry points: version, thread-local error string, device probe, CU-masked streams.
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

// A HIP stream restricted to a set of compute units (the engine runs the molecular sweep and the
// interaction level side by side on disjoint CUs).  Created and destroyed through the runtime this
// library is linked against, on the device that is current for the calling thread.
extern "C" int blgnn_stream_create_cu_mask(const uint32_t *mask_words_host, int32_t num_words, void **stream_out) {
    BLGNN_CHECK_ARG(mask_words_host && num_words > 0 && stream_out, "stream_create_cu_mask: bad args");
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)num_words, mask_words_host);
    if (e != hipSuccess || !s) {
        (void)hipGetLastError();
        return blgnn::fail(BLGNN_ERR_LAUNCH, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    }
    *stream_out = (void *)s;
    return BLGNN_OK;
}

extern "C" int blgnn_stream_destroy(void *stream) {
    BLGNN_CHECK_ARG(stream, "stream_destroy: null stream");
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return blgnn::fail(BLGNN_ERR_LAUNCH, "hipStreamDestroy: %s", hipGetErrorString(e));
    }
    return BLGNN_OK;
}
