// lib_info.hip -- identification entry points of libnerficg_hip.so
#include "common.h"

static thread_local int g_last_hip_error = 0;
void nrc_set_last_hip_error(int e) { g_last_hip_error = e; }

// ---- stage timer -----------------------------------------------------------------------------------------------------------------------
// One process-wide list of (event, name) marks, filled by NRC_STAGE while armed.  Not thread-safe by design: a measurement harness arms it,
// runs ONE call sequence on one stream and reads it back.
#include <string.h>

#include <vector>
int g_nrc_stage_timer_armed = 0;
namespace {
struct StageMark { hipEvent_t ev; const char* name; };
std::vector<StageMark> g_marks;
std::vector<hipEvent_t> g_event_pool;
size_t g_capacity = 0;
}  // namespace
void nrc_stage_mark(hipStream_t s, const char* name) {
    if (g_marks.size() >= g_capacity) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;   // never inside a recording
    if (g_event_pool.size() <= g_marks.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_event_pool.push_back(e);
    }
    hipEvent_t e = g_event_pool[g_marks.size()];
    if (hipEventRecord(e, s) != hipSuccess) return;
    g_marks.push_back({e, name});
}

extern "C" {
int nrc_stage_timer_begin(int32_t capacity) {
    if (capacity < 1 || capacity > (1 << 20)) return NRC_ERR_INVALID;
    g_marks.clear();
    g_marks.reserve((size_t)capacity);
    g_capacity = (size_t)capacity;
    g_nrc_stage_timer_armed = 1;
    return NRC_OK;
}
int nrc_stage_timer_end(int32_t max_stages, char* names, float* ms, int32_t* count) {
    g_nrc_stage_timer_armed = 0;
    if (max_stages < 0 || !count || (max_stages > 0 && (!names || !ms))) return NRC_ERR_INVALID;
    int n = 0;
    if (!g_marks.empty() && hipEventSynchronize(g_marks.back().ev) != hipSuccess) { g_marks.clear(); return NRC_ERR_LAUNCH; }
    for (size_t i = 1; i < g_marks.size() && n < max_stages; i++) {
        if (!g_marks[i].name) continue;   // an entry point's opening mark: the interval in front of it is host time between two calls
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_marks[i - 1].ev, g_marks[i].ev) != hipSuccess) continue;
        strncpy(names + (size_t)n * 32, g_marks[i].name, 31);
        names[(size_t)n * 32 + 31] = 0;
        ms[n++] = t;
    }
    *count = n;
    g_marks.clear();
    return NRC_OK;
}
int nrc_abi_version(void) { return NRC_ABI_VERSION; }
const char* nrc_build_info(void) { return "libnerficg_hip gfx950 (MI355X, CDNA4) hipcc " __VERSION__; }
int nrc_ngp_tile_width(void) { return NRC_TILE_W; }
int nrc_ngp_tile_height(void) { return NRC_TILE_H; }
const char* nrc_last_error(void) { return hipGetErrorString((hipError_t)g_last_hip_error); }
}
