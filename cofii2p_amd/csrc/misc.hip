// ABI identification of libcofi_hip.so.
#include "common.h"

extern "C" int cofi_abi_version(void) { return COFI_ABI_VERSION; }
extern "C" const char *cofi_target_arch(void) { return "gfx950"; }
