// Internal to libddsp_hip.so: optional per-kernel HIP-event timing (include/ddsp_hip.h: ddsp_profile_*).
#pragma once
#include <hip/hip_runtime.h>

// DDSP_TEST_HOOKS=1 at load time (ddsp_capi.hip): the process-global *_set_* hooks are live; otherwise they refuse.
bool ddsp_hooks_on();

namespace ddsp_prof {
enum KernelId { PREP = 0, TOTALS = 1, SCAN = 2, SYNTH = 3, NOISE = 4, NOISE_IR = 5 };
// Record an event pair around one launch on `s` when profiling is enabled (no-ops otherwise).
int begin(int kernel_id, hipStream_t s);
void end(int slot, hipStream_t s);
}  // namespace ddsp_prof

// One-time (per device of this process) opt-in of a kernel to more than 64 KiB of dynamic LDS.
// `done` is the caller's static per-device table.  Benign if two threads race: the attribute is idempotent.
inline hipError_t ddsp_allow_big_lds(const void *fn, bool (&done)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 63;
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done[dev] = true;
    return e;
}
