// ABI version, last-error diagnostics, the device-side failure word.
#include "common.h"
thread_local int effdet_last_hip_error = 0;
extern "C" int effdet_abi_version(void) { return 1; }

// A kernel that detects a failure on the device (today: a wave of mbconv_wide.hip whose poll of an LDS arrival counter ran
// out of spins) cannot return an error code: it ORs a bit into this word instead.  The word lives in host-coherent pinned memory
// that the device writes directly, so the host reads it without a copy or a synchronisation: every effdet_* entry point that
// launches a kernel looks at it after its launch (effdet_check_launch) and returns EFFDET_ELAUNCH (-5) once it is set - that is,
// the first call AFTER the failing kernel has run reports it (and every later one, until effdet_device_error(1) clears it).
__attribute__((visibility("hidden"))) volatile int* effdet_err_host = nullptr;
__attribute__((visibility("hidden"))) int* effdet_err_dev = nullptr;
static bool effdet_err_tried = false;

// device pointer of the word, or null when it cannot be set up (e.g. first use inside a stream capture): kernels skip the
// store then
__attribute__((visibility("hidden"))) int* effdet_device_error_word() {
    if (!effdet_err_tried) {
        effdet_err_tried = true;
        int* h = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && h) {
            *h = 0;
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) { effdet_err_host = h; effdet_err_dev = reinterpret_cast<int*>(d); }
            else (void)hipHostFree(h);
        }
        (void)hipGetLastError();
    }
    return effdet_err_dev;
}

// Bits of the device-side failure word (0 = none); clear != 0 resets it.  Bit 0: mbconv_wide hand-off timeout.
extern "C" int effdet_device_error(int clear) {
    if (!effdet_err_host) return 0;
    const int v = *effdet_err_host;
    if (clear) *effdet_err_host = 0;
    return v;
}

extern "C" const char* effdet_last_error(void) {
    if (effdet_last_hip_error == EFFDET_DEVICE_ERROR_CODE)
        return "a kernel reported a device-side failure (effdet_device_error(): bit 0 = MBConv X-ring hand-off timed out; its output is invalid)";
    return hipGetErrorString((hipError_t)effdet_last_hip_error);
}
