// dhts_common.hip -- library-level entry points of the C ABI (include/dhts.h).
#include <hip/hip_runtime.h>

#include "../../include/dhts.h"

extern "C" {

int dhts_version(void) { return DHTS_VERSION; }

int dhts_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DHTS_E_NO_DEVICE;
    return n;
}

}  // extern "C"
