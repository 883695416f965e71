// lib_info.hip -- identification entry points of libnerficg_hip.so
#include "common.h"

extern "C" {
int nrc_abi_version(void) { return 1; }
const char* nrc_build_info(void) { return "libnerficg_hip gfx950 (MI355X, CDNA4) hipcc " __VERSION__; }
}
