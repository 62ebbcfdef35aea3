// Status / error plumbing of the C ABI (include/panfusion_hip.h).
#include "pf_common.h"

namespace pf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pf

extern "C" int pf_version(void) { return 100; }
extern "C" const char* pf_last_error_string(void) { return pf::g_err; }
