// errors.hip -- thread-local error string + library identification.
#include "common.hpp"

namespace dagr {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
}  // namespace dagr

extern "C" {
const char *dagr_last_error(void) { return dagr::g_last_error.c_str(); }
int dagr_version(void) { return 100; /* 0.1.0 */ }
int dagr_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        dagr::set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return -1;
    }
    return n;
}
}
