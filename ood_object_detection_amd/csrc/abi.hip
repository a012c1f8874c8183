// ABI version + compile-time check that the definitions match include/effdet_hip.h.
#include "../../include/effdet_hip.h"
extern "C" int effdet_abi_version(void) { return 1; }
