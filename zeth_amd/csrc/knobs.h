#pragma once
#include <stdlib.h>

// Tuning, measurement and test knobs (ZG_MSM_*, ZG_PROVE_*, ZG_NTT_*, ZG_TEST_FAIL_LANE, ...) are read from the
// environment only by builds with -DZG_DEBUG_KNOBS (`make dbg` -> build/dbg/libzeth_gpu.so, used by tools/ and by the one
// test that injects a failure); the shipped library never looks at them.  The one setting it does read, ZG_PK_TABLE_MIN,
// is documented in include/zeth_gpu.h.
inline const char* zg_knob(const char* name) {
#ifdef ZG_DEBUG_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Breadcrumbs (debug builds only): the last library step each queueing thread of a proof entered, readable by a watchdog
// while that thread is blocked inside a runtime call (zg_debug_pk_progress, tools/diag_prio.py).  Slot 0-3: the lanes'
// threads, 4: the calling thread.
#ifdef ZG_DEBUG_KNOBS
extern const char* volatile zg_crumbs[8];
extern thread_local int zg_crumb_slot;
#define ZG_CRUMB(text) (zg_crumbs[zg_crumb_slot & 7] = (text))
#define ZG_CRUMB_SLOT(i) (zg_crumb_slot = (i))
#else
#define ZG_CRUMB(text) ((void)0)
#define ZG_CRUMB_SLOT(i) ((void)0)
#endif


// Issue priority of a kernel's waves among the waves that share its SIMD (s_setprio, 0-3; the hardware default is 0).
// Short latency-bound kernels (scans, unit tables, the reductions' chains) and the kernels of a proof's longest chain
// (constraint evaluation, transforms) raise theirs: inside a proof they share CUs with the accumulation kernels of the
// other evaluations, whose waves otherwise take three issue slots in four.  This is per wave and inside the CU -- it has
// nothing to do with stream priorities (prover.cuh), which stall the runtime.
#ifndef ZG_WAVE_PRIO_ON
#define ZG_WAVE_PRIO_ON 1
#endif
#if ZG_WAVE_PRIO_ON && defined(__HIP_DEVICE_COMPILE__)
#define ZG_WAVE_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define ZG_WAVE_PRIO(n) ((void)0)
#endif
