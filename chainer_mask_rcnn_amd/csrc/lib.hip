// Library-level entry points: error string, ABI version, device info.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace mrcnn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mrcnn

extern "C" const char *mrcnn_last_error(void) { return mrcnn::g_err; }

extern "C" int mrcnn_abi_version(void) { return 1; }

extern "C" int mrcnn_device_info(int *n_cu, char *name, int name_len)
{
    int dev = 0;
    MRCNN_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MRCNN_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return 0;
}
