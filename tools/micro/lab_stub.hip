// Stand-ins for the library plumbing (composite.hip) so one kernel file can be built alone into a diagnostic .so.
#include "../../multimodal_diffusion_amd/csrc/avd_common.h"
namespace avd {
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "avd error %d: %s\n", code, g_err);
    return code;
}
bool g_prof_on = false;
void prof_mark(int, double, hipStream_t, bool) {}
int prof_tag_id(const char*, ...) { return 0; }
int LdsAttr::ensure(const void* kern, int lds_bytes, const char* what) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    return e == hipSuccess ? AVD_OK : set_error(AVD_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}
}  // namespace avd
