// C-ABI odds and ends of libddsp_hip.so (see include/ddsp_hip.h): ABI version, per-kernel event timing.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "ddsp_hip.h"
#include "ddsp_internal.h"

#include <stdlib.h>

extern "C" int ddsp_hip_abi_version(void) { return DDSP_HIP_ABI_VERSION; }

namespace {
// read ONCE, when the shared object is loaded: a later setenv cannot arm the hooks of a running process
const bool g_hooks_on = [] { const char *e = getenv("DDSP_TEST_HOOKS"); return e && e[0] == '1' && e[1] == 0; }();
}  // namespace
bool ddsp_hooks_on() { return g_hooks_on; }
extern "C" int ddsp_test_hooks_enabled(void) { return g_hooks_on ? 1 : 0; }

namespace {
struct Record { hipEvent_t t0, t1; int kernel_id; };
std::mutex g_mu;
std::vector<Record> g_pool;  // created by ddsp_profile_enable
int g_used = 0;
bool g_on = false;
unsigned g_select = ~0u;     // ddsp_profile_select: bit i = record kernel id i
}  // namespace

namespace ddsp_prof {
int begin(int kernel_id, hipStream_t s)
{
    if (!g_on) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!((g_select >> (kernel_id & 31)) & 1u)) return -1;
    if (g_used >= (int)g_pool.size()) return -1;
    const int slot = g_used++;
    g_pool[slot].kernel_id = kernel_id;
    (void)hipEventRecord(g_pool[slot].t0, s);
    return slot;
}
void end(int slot, hipStream_t s)
{
    if (slot < 0) return;
    (void)hipEventRecord(g_pool[slot].t1, s);
}
}  // namespace ddsp_prof

extern "C" int ddsp_profile_enable(int capacity)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_pool) { (void)hipEventDestroy(r.t0); (void)hipEventDestroy(r.t1); }
    g_pool.clear();
    g_used = 0;
    g_on = false;
    if (capacity <= 0) return 0;
    g_pool.resize(capacity);
    for (auto &r : g_pool) {
        hipError_t e = hipEventCreate(&r.t0);
        if (e == hipSuccess) e = hipEventCreate(&r.t1);
        if (e != hipSuccess) { g_pool.clear(); return (int)e; }
    }
    g_on = true;
    return 0;
}

extern "C" int ddsp_profile_select(unsigned kernel_mask)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_select = kernel_mask ? kernel_mask : ~0u;
    return 0;
}

extern "C" int ddsp_profile_read(int *kernel_ids, float *ms, int cap)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (int i = 0; i < g_used && n < cap; ++i) {
        if (hipEventSynchronize(g_pool[i].t1) != hipSuccess) break;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, g_pool[i].t0, g_pool[i].t1) != hipSuccess) break;
        kernel_ids[n] = g_pool[i].kernel_id;
        ms[n] = t;
        ++n;
    }
    g_used = 0;
    return n;
}
