// ABI version, last-error diagnostics.
#include <hip/hip_runtime.h>
#include "../../include/effdet_hip.h"
thread_local int effdet_last_hip_error = 0;
extern "C" int effdet_abi_version(void) { return 1; }
extern "C" const char* effdet_last_error(void) {
    return hipGetErrorString((hipError_t)effdet_last_hip_error);
}
