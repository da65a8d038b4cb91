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
