// libnrf.so: version / error reporting.
#include "nrf_common.h"

namespace nrf {
char *last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace nrf

extern "C" int nrf_version(void) { return NRF_VERSION; }
extern "C" const char *nrf_last_error_string(void) { return nrf::last_error_buf(); }
