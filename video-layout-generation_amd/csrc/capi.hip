// Library identification entry points of libvlg_hip.so (see include/vlg_hip.h).
#include "common.h"

extern "C" int vlg_abi_version(void) { return 1; }
extern "C" const char* vlg_build_arch(void) { return "gfx950"; }
