// api.hip -- ABI version, thread-local error string, device probe.
#include <stdarg.h>

#include "common.h"

namespace umereg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return UMEREG_ENODEV;
    }
    return UMEREG_OK;
}

}  // namespace umereg

UMEREG_API int umereg_abi_version(void) { return UMEREG_ABI_VERSION; }

UMEREG_API const char* umereg_last_error(void) { return umereg::g_err; }

UMEREG_API int umereg_device_count(char* arch_name, size_t arch_name_len)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    if (arch_name && arch_name_len) {
        arch_name[0] = 0;
        if (n > 0) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, 0) == hipSuccess) {
                strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
                arch_name[arch_name_len - 1] = 0;
            }
        }
    }
    return n;
}
