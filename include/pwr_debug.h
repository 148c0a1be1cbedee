/* pwr_debug.h -- debugging aids of libpwr_hip_dbg.so, the DEBUG build of the library (-DPWR_DEBUG_BUILD, tools/build_debug.py).
 * NOT part of the product ABI: the shipped libpwr_hip.so exports none of these, reads no experiment environment variable, and has
 * one configuration.  tools/dbglib.py loads the debug build for the measurement scripts under tools/. */
#ifndef PWR_DEBUG_H_
#define PWR_DEBUG_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* dst = src (bytes % 16 == 0) iff *flag != 0, decided on the device (tools/race_hunt.py) */
int pwr_debug_copy_if(const int* flag, const void* src, void* dst, size_t bytes, void* stream);
/* per-workgroup phase time stamps of the 3x3 patch conv (8 x int64 each); NULL = off */
void pwr_debug_set_stamps(void* stamps);
/* phase experiment of the 3x3 patch conv: workgroups in an odd wave slot sleep ~cycles before staging their patch; 0 = off */
void pwr_debug_set_delay(int cycles);
/* arena layout as text lines "offset bytes tag"; returns the size needed */
size_t pwr_engine_layout(void* engine, char* buf, size_t cap);
/* 0 (the product's mode since round 4): the caller's stream waits for the side streams after the last segment only; 1: after every
 * segment (rounds 2 - 3) */
void pwr_engine_set_join(void* engine, int each_segment);
/* per-scope time of the chain (the caller's stream): set_timing(1) puts an event at every change of network part ("stem", "s0.hg3",
 * "s1.heads", ...) in the launch lists of the following forward / backward calls; timing_report synchronises, writes lines
 * "phase<TAB>scope<TAB>total ms<TAB>intervals", clears the collection and returns the size needed (tools/step_breakdown.py) */
void pwr_engine_set_timing(void* engine, int on);
size_t pwr_engine_timing_report(void* engine, char* buf, size_t cap);
#ifdef __cplusplus
}
#endif
#endif
