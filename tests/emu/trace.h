// TEST INFRASTRUCTURE (debugging aid, never part of a normal build): value tracing at one pixel for hunting the first intermediate that differs between
// the device sources (run in the CPU emulation) and the oracle. tests/emu/autotrace.py instruments copies of both sources with TRACE(...) lines
// after every float / float2 / float3 / float4 declaration of a line range; NRD_TRACE_X / NRD_TRACE_Y select the pixel.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
inline thread_local int g_nrdTrace = 0;
inline int NrdTraceCoord(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : -1;
}
inline void NrdTraceAt(int px, int py) {
    static const int tx = NrdTraceCoord("NRD_TRACE_X"), ty = NrdTraceCoord("NRD_TRACE_Y");
    g_nrdTrace = px == tx && py == ty;
}
inline void NrdTraceValue(const char* tag, const char* comp, float v) {
    if (!g_nrdTrace)
        return;
    uint32_t u;
    memcpy(&u, &v, 4);
    fprintf(stderr, "TRACE %s %s%s %08x %.9g\n", NRD_TRACE_SIDE, tag, comp, u, v);
}
#define TRACE(tag, v) NrdTraceValue(tag, "", (float)(v))
#define TRACE2(tag, v) (NrdTraceValue(tag, ".x", (v).x), NrdTraceValue(tag, ".y", (v).y))
#define TRACE3(tag, v) (NrdTraceValue(tag, ".x", (v).x), NrdTraceValue(tag, ".y", (v).y), NrdTraceValue(tag, ".z", (v).z))
#define TRACE4(tag, v) (NrdTraceValue(tag, ".x", (v).x), NrdTraceValue(tag, ".y", (v).y), NrdTraceValue(tag, ".z", (v).z), NrdTraceValue(tag, ".w", (v).w))
#define TRACE_AT(px, py) NrdTraceAt(px, py)
