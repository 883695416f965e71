// lib_info.hip -- identification entry points of libnerficg_hip.so
#include "common.h"

static thread_local int g_last_hip_error = 0;
void nrc_set_last_hip_error(int e) { g_last_hip_error = e; }

extern "C" {
int nrc_abi_version(void) { return 1; }
const char* nrc_build_info(void) { return "libnerficg_hip gfx950 (MI355X, CDNA4) hipcc " __VERSION__; }
int nrc_ngp_tile_width(void) { return NRC_TILE_W; }
int nrc_ngp_tile_height(void) { return NRC_TILE_H; }
const char* nrc_last_error(void) { return hipGetErrorString((hipError_t)g_last_hip_error); }
}
