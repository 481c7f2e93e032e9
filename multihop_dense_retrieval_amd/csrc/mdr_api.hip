// csrc/mdr_api.hip -- error state and version of libmdrhip.so (include/mdr_hip.h).
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <set>
#include <utility>

#include "mdr_common.h"

namespace mdr {

char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    MDR_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return MDR_OK;
    MDR_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, dev});
    return MDR_OK;
}

}  // namespace mdr

extern "C" {
const char* mdr_last_error(void) { return mdr::last_error_buf(); }
const char* mdr_version(void) { return "mdr-hip 0.2 (gfx950)"; }
}
