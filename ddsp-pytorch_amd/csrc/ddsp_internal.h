// Internal to libddsp_hip.so: optional per-kernel HIP-event timing (include/ddsp_hip.h: ddsp_profile_*).
#pragma once
#include <hip/hip_runtime.h>

namespace ddsp_prof {
enum KernelId { PREP = 0, TOTALS = 1, SCAN = 2, SYNTH = 3, NOISE = 4 };
// Record an event pair around one launch on `s` when profiling is enabled (no-ops otherwise).
int begin(int kernel_id, hipStream_t s);
void end(int slot, hipStream_t s);
}  // namespace ddsp_prof
