// Recurrent part of the controller's GRU for MI355X (gfx950) -- SURVEY §8f next rows 2-4: the
// control network of model/autoencoder/decoder.py:60-65,91 is `nn.GRU(2*width, units, layers, batch_first=True)`;
// its recurrence is 500 strictly sequential steps per 4 s clip, and the stock MIOpen path spends ~110 us per step
// (forward + backward) in a few dozen tiny launches: 85 % of the whole training step (profiles/r01_train_step*.json).
//
// Here the recurrence is ONE persistent launch per direction:
//   * the input projection gi = x W_ih^T + b_ih for all (b,t) stays a library GEMM (caller side);
//   * the batch rows are split into NG independent GROUPS (no data ever crosses groups); the hidden units of a group
//     are split over NW = ceil(Hd/16) workgroups of 256 threads, one per CU, NG*NW <= #CUs.  Workgroup blockIdx.x
//     belongs to group blockIdx.x % NGpad (NGpad a multiple of 8): with the round-robin dispatch over the 8 XCDs
//     all members of a group share one XCD (speed only -- every hand-off is agent scope, correct for any placement;
//     the test hook ddsp_gru_set_mode(1) deals each group over all XCDs and the results stay bitwise the same);
//   * a workgroup keeps its slice of W_hh (16 units x 3 gates x Hd <= 96 KB) in REGISTERS for the whole sequence:
//     thread (unit ul = tid/16, slice ks = tid%16) holds 3 x Hd/16 weights; per step it multiplies them with the
//     group's h_{t-1} rows (LDS, conflict-free 16-byte reads, broadcast over the units) and the 16 slices are summed
//     with four DPP steps inside one DPP row;
//   * h_t travels between the workgroups of a group as 8-byte {epoch, value} granules (the data is the flag):
//     relaxed agent-scope stores (sc1, write-through) and relaxed agent-scope polling loads, double-buffered by step
//     parity; no fence, no separate flag, two workgroup barriers per step.
// Every spin is bounded by wall-clock time (2 s): on a timeout the workgroup records a status word, poisons its
// outputs with NaN and leaves, and the others follow, so the grid always drains.
//
// Arithmetic is fp32 in ATen's CPU order (RNN.cpp gru cell): r = s(gi_r + gh_r), z = s(gi_z + gh_z),
// n = tanh(gi_n + r*gh_n), h' = (h - n)*z + n, with gh = W_hh h + b_hh.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_osc_common.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int kUnits = 16;        // hidden units per workgroup (threadIdx.x >> 4)
constexpr int kMaxRows = 64;      // batch rows per group (LDS: 64 x 512 x 4 B = 128 KB forward)
constexpr int kMaxRowsBwd = 16;   // backward stages 3 payloads per row
// Batch rows per register tile RT (2 or 4) and row sets NRS (1 or 2 x 256 threads): see launch_gru.
constexpr long kSpinTicks = 200000000L;  // 2 s of the 100 MHz wall clock
constexpr long kSpinTicksFaultTest = 2000000L;   // 20 ms when the fault-injection hook is armed (tests)

enum { GRU_OK = 0, GRU_TIMEOUT = 1 };

struct GruParams {
    const float *gi;      // [B,T,3Hd]  x W_ih^T + b_ih
    const float *w_hh;    // [3Hd,Hd]
    const float *b_hh;    // [3Hd] nullable
    const float *h0;      // [B,Hd] nullable (zeros)
    float *y;             // [B,T,Hd]   h_t
    float *hT;            // [B,Hd]
    float *gates;         // [B,T,3Hd]  r|z|n, nullable (inference) / input of the backward
    float *hn;            // [B,T,Hd]   W_hn h_{t-1} + b_hn, nullable / input of the backward
    const float *dy;      // backward: [B,T,Hd]
    const float *dhT;     // backward: [B,Hd] nullable
    float *d_gi;          // backward: [B,T,3Hd]
    float *d_gh;          // backward: [B,T,3Hd]
    float *dh0;           // backward: [B,Hd]
    gu64 *xchg;           // granules
    gu32 *status;         // 0 ok / GRU_TIMEOUT
    int B, T, Hd;
    int NG, NGpad, NW, BL;  // groups, padded group count (blockIdx modulus), workgroups per group, rows per group
    int lowp;             // 1: the products run on the matrix cores in bf16 (autocast callers), fp32 accumulate
    int io16;             // with lowp, backward: d_gi / d_gh are bf16 arrays -- what the autocast GEMMs behind the recurrence consume
    long spin_ticks;      // bound of every spin, in ticks of the 100 MHz wall clock
    int fault_step;       // test hook (ddsp_gru_set_mode(2)): workgroup 0 withholds its publishes from this step on; -1 = off
};

// Gate non-linearities on the hardware exp2 / rcp (1 ulp each): |error| <= 2e-7, on the critical path of every step.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ unsigned long long pack_granule(unsigned epoch, float v)
{
    return ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
}

// Polls `rows` granule rows of width Hd (row stride HP granules) until every tag equals `epoch`; values go to LDS.
// A thread owns columns tid and tid + NT (NT = workgroup size) of every row; rows are polled RB at a time (2 RB independent loads in
// flight per lane), and a batch whose tags all matched is not polled again.  Returns false (wave-uniform) on
// timeout / abort.
template <int RB, bool FIRST_LIGHT, typename DT = float>
__device__ __forceinline__ bool sweep_rows(gu64 *src, DT *dst, int rows, int Hd, int HP, unsigned epoch, gu32 *status, int NT,
                                           long spin_ticks, int DS = 0)
{
    if (DS == 0) DS = HP;                                 // destination row stride in LDS (the granule rows are HP apart)
    const long t0 = wall_clock64();
    const int kc[2] = {(int)threadIdx.x, (int)threadIdx.x + NT};
    const int nb = (rows + RB - 1) / RB;                  // <= 16 batches
    unsigned todo = (1u << nb) - 1u;
    // First light: a full pass moves rows x Hd granules per workgroup through the L2 and takes about as long as the
    // hand-off itself, so a pass that starts a little too early costs a whole extra pass (measured: +2 us per
    // backward step).  Each lane first watches ONE granule -- its own column of the last row, the last one its
    // publisher stores -- with cheap passes, and only then reads (and checks the tag of) everything.  Used where a pass is
    // heavy (backward, three payloads per row); the forward's 8-load passes are cheaper than the extra round trip.
    if (FIRST_LIGHT && kc[0] < Hd) {
        gu64 *sentinel = src + (size_t)(rows - 1) * HP + kc[0];
        for (unsigned pass = 0;; ++pass) {
            const unsigned long long x = __hip_atomic_load(sentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == epoch)) break;
            if ((pass & 63) == 63) {
                if (wall_clock64() - t0 > spin_ticks) return false;
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    for (unsigned pass = 0;; ++pass) {
        for (int bi = 0; bi < nb; ++bi) {
            if (!((todo >> bi) & 1u)) continue;           // wave-uniform
            unsigned long long x[RB][2];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = bi * RB + r;
                    x[r][c] = (unsigned long long)epoch << 32;
                    if (row < rows && kc[c] < Hd)
                        x[r][c] = __hip_atomic_load(src + (size_t)row * HP + kc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            bool ok = true;   // lanes / rows outside the problem carry the expected tag already
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) ok = ok && (unsigned)(x[r][c] >> 32) == epoch;
            if (__all(ok)) {   // the whole batch arrived: store it without per-granule predication
                todo &= ~(1u << bi);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int row = bi * RB + r;
                    if (row < rows) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (kc[c] < Hd) dst[row * DS + kc[c]] = (DT)__uint_as_float((unsigned)x[r][c]);
                    }
                }
            }
        }
        if (todo == 0u) return true;
        if ((pass & 63) == 63) {
            if (wall_clock64() - t0 > spin_ticks) return false;
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// One granule store: relaxed, agent scope (sc1, written through), visible to a poll from any XCD.
__device__ __forceinline__ void publish(gu64 *dst, unsigned epoch, float v)
{
    __hip_atomic_store(dst, pack_granule(epoch, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- packed granules of the bf16 matrix-core variants ----------------------------------------------------------------
// The matrix cores consume bf16 operands, so the hand-off carries bf16: the polls' volume through the L2 -- what a step's time
// grows with (measured: +0.27 us per batch row forward, +0.54 backward with one fp32 value per granule) -- shrinks 2x / 3x.
//   PACK2  {epoch:32 | v1:16 | v0:16}            forward: h of batch rows 2r and 2r+1, one unit
//   PACK3  {epoch:16 | v2:16 | v1:16 | v0:16}    backward: the three gate-gradient payloads of one (row, unit); epochs < 2^16
__device__ __forceinline__ unsigned bf16_bits(float v)
{
    const __bf16 b = (__bf16)v;     // round to nearest even, as the fragments' conversion did before
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void publish2(gu64 *dst, unsigned epoch, float v0, float v1)
{
    const unsigned long long g = ((unsigned long long)epoch << 32) | ((unsigned long long)bf16_bits(v1) << 16) | (unsigned long long)bf16_bits(v0);
    __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish3(gu64 *dst, unsigned epoch, float v0, float v1, float v2)
{
    const unsigned long long g = ((unsigned long long)(epoch & 0xffffu) << 48) | ((unsigned long long)bf16_bits(v2) << 32) |
                                 ((unsigned long long)bf16_bits(v1) << 16) | (unsigned long long)bf16_bits(v0);
    __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (Tried on top, same-process A/B at 512 units: the backward WITHOUT first light 2.72 -> 3.05 us per step at batch 32, and two sets of polls
//  in flight half a round trip apart 1.83 -> 2.24 forward / 2.72 -> 3.29 backward -- every extra poll slows the L2 for the publishers;
//  first light on ONE granule per publisher instead of one per column: polled by every wavefront 2.65 -> 3.0, polled by wavefront 0 alone
//  + a barrier 2.65 -> 2.66 at batch 32 (2.31 -> 2.05 at batch 8): not taken.)
// sweep_rows for packed granules: `rows` granule rows of width Hd (stride HP); PACK values of each granule go to the bf16 LDS image at
// dst[(row * PACK + i) * DS + column] (PACK2: image rows 2r, 2r+1 = batch rows; PACK3: image rows 3r + gate).
template <int RB, bool FIRST_LIGHT, int PACK>
__device__ __forceinline__ bool sweep_packed(gu64 *src, unsigned short *dst, int rows, int Hd, int HP, unsigned epoch, gu32 *status,
                                             long spin_ticks, int DS)
{
    constexpr int NT = 256;
    constexpr int TAG_SHIFT = PACK == 3 ? 48 : 32;
    const unsigned want = PACK == 3 ? (epoch & 0xffffu) : epoch;
    const long t0 = wall_clock64();
    const int kc[2] = {(int)threadIdx.x, (int)threadIdx.x + NT};
    const int nb = (rows + RB - 1) / RB;
    unsigned todo = (1u << nb) - 1u;
    if (FIRST_LIGHT && kc[0] < Hd) {
        gu64 *sentinel = src + (size_t)(rows - 1) * HP + kc[0];
        for (unsigned pass = 0;; ++pass) {
            const unsigned long long x = __hip_atomic_load(sentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> TAG_SHIFT) == want)) break;
            if ((pass & 63) == 63) {
                if (wall_clock64() - t0 > spin_ticks) return false;
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    for (unsigned pass = 0;; ++pass) {
        for (int bi = 0; bi < nb; ++bi) {
            if (!((todo >> bi) & 1u)) continue;           // wave-uniform
            unsigned long long x[RB][2];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int row = bi * RB + r;
                    x[r][c] = (unsigned long long)want << TAG_SHIFT;
                    if (row < rows && kc[c] < Hd)
                        x[r][c] = __hip_atomic_load(src + (size_t)row * HP + kc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            bool ok = true;
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) ok = ok && (unsigned)(x[r][c] >> TAG_SHIFT) == want;
            if (__all(ok)) {
                todo &= ~(1u << bi);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int row = bi * RB + r;
                    if (row < rows) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (kc[c] < Hd) {
#pragma unroll
                                for (int i = 0; i < PACK; ++i) dst[(row * PACK + i) * DS + kc[c]] = (unsigned short)(x[r][c] >> (16 * i));
                            }
                    }
                }
            }
        }
        if (todo == 0u) return true;
        if ((pass & 63) == 63) {
            if (wall_clock64() - t0 > spin_ticks) return false;
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- forward ------------------------------------------------------------------------------------------------
template <int KP, int RT, int NRS>
__global__ void __launch_bounds__(256 * NRS, 1) gru_fwd_kernel(GruParams p)
{
    extern __shared__ __attribute__((aligned(16))) float h_s[];  // [BLpad][HP] + 1 word (failure flag)
    constexpr int HP = 16 * KP;
    const int group = blockIdx.x % p.NGpad, member = blockIdx.x / p.NGpad;
    if (group >= p.NG) return;
    const int row0 = group * p.BL;
    const int nrows = min(p.BL, p.B - row0);
    if (nrows <= 0) return;
    const int BLpad = (p.BL + RT - 1) & ~(RT - 1);
    int *fail_s = reinterpret_cast<int *>(h_s + BLpad * HP);
    const int rs = threadIdx.x >> 8, lt = threadIdx.x & 255;   // row set, thread within the set
    const int ks = lt & 15, ul = lt >> 4;
    const int u = member * kUnits + ul;
    const int Hd = p.Hd;
    const bool unit_ok = u < Hd;

    // this thread's slice of W_hh: gate g, unit u, columns k = 64 i + 4 ks + c
    float w[3][KP];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < KP; ++e) {
            const int k = (e >> 2) * 64 + ks * 4 + (e & 3);
            w[g][e] = (unit_ok && k < Hd) ? p.w_hh[((size_t)g * Hd + u) * Hd + k] : 0.0f;
        }
    float bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = (unit_ok && p.b_hh) ? p.b_hh[g * Hd + u] : 0.0f;

    for (int i = threadIdx.x; i < BLpad * HP; i += 256 * NRS) {
        const int bl = i / HP, k = i - bl * HP;
        h_s[i] = (bl < nrows && k < Hd && p.h0) ? p.h0[(size_t)(row0 + bl) * Hd + k] : 0.0f;
    }
    if (threadIdx.x == 0) *fail_s = 0;
    __syncthreads();

    gu64 *xg = p.xchg + (size_t)group * 2 * p.BL * HP;  // [2][BL][HP]
    const bool gate_lane = unit_ok && ks < RT;
    const size_t G3 = (size_t)3 * Hd;

    // input-projection terms of the first row tile, fetched one step ahead (HBM latency off the critical path)
    float pre[3] = {0.0f, 0.0f, 0.0f};
    const bool mine0 = gate_lane && rs * RT + ks < nrows;
    if (mine0) {
        const size_t bt = (size_t)(row0 + rs * RT + ks) * p.T;
#pragma unroll
        for (int g = 0; g < 3; ++g) pre[g] = p.gi[bt * G3 + g * Hd + u];
    }

    const int fault_from = (p.fault_step >= 0 && blockIdx.x == 0) ? p.fault_step : 0x7fffffff;   // test hook: this workgroup stops publishing
    int t_reached = 0;                                             // steps [0, t_reached) were completed by this workgroup
    for (int t = 0; t < p.T; ++t) {
        if (t > 0) {
            const bool ok = sweep_rows<4, false>(xg + (size_t)((t - 1) & 1) * p.BL * HP, h_s, nrows, Hd, HP, (unsigned)t, p.status, 256 * NRS,
                                                 p.spin_ticks);
            if (!ok && (threadIdx.x & 63) == 0) *fail_s = 1;
        }
        __syncthreads();
        if (*fail_s) break;
        t_reached = t + 1;
        for (int bt0 = rs * RT; bt0 < nrows; bt0 += RT * NRS) {
            const int bl = bt0 + ks;
            const bool mine = gate_lane && bl < nrows;
            const size_t bt = ((size_t)(row0 + bl) * p.T + t);
            float gir = pre[0], giz = pre[1], gin = pre[2];
            if (bt0 == rs * RT) {
                if (mine && t + 1 < p.T) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) pre[g] = p.gi[(bt + 1) * G3 + g * Hd + u];
                }
            } else if (mine) {
                gir = p.gi[bt * G3 + u];
                giz = p.gi[bt * G3 + Hd + u];
                gin = p.gi[bt * G3 + 2 * Hd + u];
            }
            float acc[RT][3];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0f;
#pragma unroll
            for (int i = 0; i < KP / 4; ++i) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const float4 hv = *reinterpret_cast<const float4 *>(h_s + (bt0 + r) * HP + i * 64 + ks * 4);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[r][g] = __fmaf_rn(hv.x, w[g][4 * i + 0], acc[r][g]);
                        acc[r][g] = __fmaf_rn(hv.y, w[g][4 * i + 1], acc[r][g]);
                        acc[r][g] = __fmaf_rn(hv.z, w[g][4 * i + 2], acc[r][g]);
                        acc[r][g] = __fmaf_rn(hv.w, w[g][4 * i + 3], acc[r][g]);
                    }
                }
            }
            float sr = 0.0f, sz = 0.0f, sn = 0.0f;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[r][g] = ddsp_osc::group_sum(acc[r][g], 4);  // the 16 slices = one DPP row
                if (ks == r) { sr = acc[r][0]; sz = acc[r][1]; sn = acc[r][2]; }
            }
            if (mine) {
                const float hp = h_s[bl * HP + u];
                const float ghn = sn + bh[2];
                const float r = sigmoidf_(gir + (sr + bh[0]));
                const float z = sigmoidf_(giz + (sz + bh[1]));
                const float n = tanhf_(__fmaf_rn(r, ghn, gin));
                const float hnew = __fmaf_rn(hp - n, z, n);
                if (t < fault_from) publish(xg + ((size_t)(t & 1) * p.BL + bl) * HP + u, (unsigned)t + 1u, hnew);
                p.y[bt * Hd + u] = hnew;
                if (p.gates) {
                    p.gates[bt * G3 + u] = r;
                    p.gates[bt * G3 + Hd + u] = z;
                    p.gates[bt * G3 + 2 * Hd + u] = n;
                }
                if (p.hn) p.hn[bt * Hd + u] = ghn;
                if (t == p.T - 1) p.hT[(size_t)(row0 + bl) * Hd + u] = hnew;
            }
        }
        __syncthreads();  // h_s is rewritten by the next sweep
    }
    if (*fail_s) {
        // Loud failure: everything this workgroup owns of the steps it did not complete becomes NaN (the buffers come from
        // torch.empty: stale memory would otherwise flow into the weight gradients as finite garbage).
        if (threadIdx.x == 0) __hip_atomic_store(p.status, (unsigned)GRU_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nan = __builtin_nanf("");
        if (unit_ok && rs == 0)
            for (int bl = ks; bl < nrows; bl += 16) {
                p.hT[(size_t)(row0 + bl) * Hd + u] = nan;
                for (int t = t_reached; t < p.T; ++t) {
                    const size_t bt = (size_t)(row0 + bl) * p.T + t;
                    p.y[bt * Hd + u] = nan;
                    if (p.gates) {
                        p.gates[bt * G3 + u] = nan;
                        p.gates[bt * G3 + Hd + u] = nan;
                        p.gates[bt * G3 + 2 * Hd + u] = nan;
                    }
                    if (p.hn) p.hn[bt * Hd + u] = nan;
                }
            }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------
// Same decomposition, transposed: the workgroup owns 16 COLUMNS k of dh; thread (k = tid/16, slice us = tid%16)
// holds W_hh[g*Hd + u'][k] for its source units u' = 64 i + 4 us + c.  Per step (t = T-1 .. 0) the gate lane of
// (row, k) turns dh_t into the three pre-activation gradients, publishes them, and after the group-wide exchange
// dh_{t-1}[k] = dh_t[k] z_t[k] + sum_u' (dr W_hr + dz W_hz + d(hn) W_hn)[u',k]  (+ dy_{t-1}[k] at the next step).
template <int KP, int RT, int NRS>
__global__ void __launch_bounds__(256 * NRS, 1) gru_bwd_kernel(GruParams p)
{
    extern __shared__ __attribute__((aligned(16))) float d_s[];  // [BLpad][3][HP] + 1 word
    constexpr int HP = 16 * KP;
    const int group = blockIdx.x % p.NGpad, member = blockIdx.x / p.NGpad;
    if (group >= p.NG) return;
    const int row0 = group * p.BL;
    const int nrows = min(p.BL, p.B - row0);
    if (nrows <= 0) return;
    const int BLpad = (p.BL + RT - 1) & ~(RT - 1);
    int *fail_s = reinterpret_cast<int *>(d_s + BLpad * 3 * HP);
    const int rs = threadIdx.x >> 8, lt = threadIdx.x & 255;   // row set, thread within the set
    const int us = lt & 15, kl = lt >> 4;
    const int k = member * kUnits + kl;
    const int Hd = p.Hd;
    const bool col_ok = k < Hd;

    float w[3][KP];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < KP; ++e) {
            const int uu = (e >> 2) * 64 + us * 4 + (e & 3);
            w[g][e] = (col_ok && uu < Hd) ? p.w_hh[((size_t)g * Hd + uu) * Hd + k] : 0.0f;
        }
    for (int i = threadIdx.x; i < BLpad * 3 * HP; i += 256 * NRS) d_s[i] = 0.0f;
    if (threadIdx.x == 0) *fail_s = 0;
    __syncthreads();

    gu64 *xg = p.xchg + (size_t)group * 2 * p.BL * 3 * HP;  // [2][BL][3][HP]
    const bool gate_lane = col_ok && us < RT;
    const size_t G3 = (size_t)3 * Hd;
    // running dh for the (row, k) pairs this lane is the gate lane of: rows us, us + 4, ... (register array, Q of them)
    constexpr int Q = (RT == 2 && NRS == 1) ? 1 : kMaxRowsBwd / (RT * NRS);   // row tiles per row set: tile q of set rs = q * NRS + rs
    float carry[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bl = (q * NRS + rs) * RT + us;
        carry[q] = (gate_lane && bl < nrows && p.dhT) ? p.dhT[(size_t)(row0 + bl) * Hd + k] : 0.0f;
    }

    // Everything the gate gradients need besides dh itself is linear in dh: the five factors (and dy) of step t are
    // prepared one step ahead, so the critical path of a step is one add and five multiplies before the publish.
    float dyv[Q], f_r[Q], f_z[Q], f_hn[Q], f_n[Q], f_dir[Q];
    float raw[Q][6];  // r, z, n, W_hn h + b_hn, h_{t-1}, dy of the NEXT step: loaded a whole step before they are used
    auto fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int bl = (q * NRS + rs) * RT + us;
            if (gate_lane && bl < nrows) {
                const size_t bt = (size_t)(row0 + bl) * p.T + t;
                raw[q][0] = p.gates[bt * G3 + k];
                raw[q][1] = p.gates[bt * G3 + Hd + k];
                raw[q][2] = p.gates[bt * G3 + 2 * Hd + k];
                raw[q][3] = p.hn[bt * Hd + k];
                raw[q][4] = (t > 0) ? p.y[(bt - 1) * Hd + k] : (p.h0 ? p.h0[(size_t)(row0 + bl) * Hd + k] : 0.0f);
                raw[q][5] = p.dy[bt * Hd + k];
            }
        }
    };
    auto derive = [&]() {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float r = raw[q][0], z = raw[q][1], n = raw[q][2], ghn = raw[q][3], hp = raw[q][4];
            dyv[q] = raw[q][5];
            f_n[q] = (1.0f - z) * (1.0f - n * n);        // d(n pre-activation) / dh
            f_r[q] = (f_n[q] * ghn) * (r * (1.0f - r));  // d(r pre-activation) / dh
            f_z[q] = (hp - n) * (z * (1.0f - z));        // d(z pre-activation) / dh
            f_hn[q] = f_n[q] * r;                        // d(W_hn h + b_hn) / dh
            f_dir[q] = z;                                // direct path h_{t-1} -> h_t
        }
    };
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int e = 0; e < 6; ++e) raw[q][e] = 0.0f;
    fetch(p.T - 1);
    derive();
    if (p.T > 1) fetch(p.T - 2);

    const int fault_from = (p.fault_step >= 0 && blockIdx.x == 0) ? p.fault_step : 0x7fffffff;   // test hook: this workgroup stops publishing
    int s_reached = 0;                                             // steps [0, s_reached) (t = T-1-s) wrote their gradients
    for (int s = 0; s < p.T; ++s) {
        const int t = p.T - 1 - s;
        const unsigned epoch = (unsigned)s + 1u;
        gu64 *slot = xg + (size_t)(s & 1) * p.BL * 3 * HP;
        float direct[Q];
        // 1. gate gradients of the owned (row, k) pairs; publish
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int bl = (q * NRS + rs) * RT + us;
            direct[q] = 0.0f;
            if (gate_lane && bl < nrows) {
                const size_t bt = (size_t)(row0 + bl) * p.T + t;
                const float dh = dyv[q] + carry[q];
                const float dr_pre = dh * f_r[q], dz_pre = dh * f_z[q], dhn = dh * f_hn[q], dn_pre = dh * f_n[q];
                direct[q] = dh * f_dir[q];
                gu64 *gdst = slot + (size_t)bl * 3 * HP + k;
                if (s < fault_from) {
                    publish(gdst, epoch, dr_pre);
                    publish(gdst + HP, epoch, dz_pre);
                    publish(gdst + 2 * HP, epoch, dhn);
                }
                p.d_gi[bt * G3 + k] = dr_pre;
                p.d_gi[bt * G3 + Hd + k] = dz_pre;
                p.d_gi[bt * G3 + 2 * Hd + k] = dn_pre;
                p.d_gh[bt * G3 + k] = dr_pre;
                p.d_gh[bt * G3 + Hd + k] = dz_pre;
                p.d_gh[bt * G3 + 2 * Hd + k] = dhn;
            }
        }
        s_reached = s + 1;
        // 2. the group's gate gradients -> LDS  (rows of 3 payloads: treated as 3*nrows rows of width Hd)
        {
            const bool ok = sweep_rows<12, NRS == 1>(slot, d_s, 3 * nrows, Hd, HP, epoch, p.status, 256 * NRS, p.spin_ticks);
            if (!ok && (threadIdx.x & 63) == 0) *fail_s = 1;
        }
        __syncthreads();
        if (*fail_s) break;
        // factors of the next step from what was fetched a whole step ago, then the fetch for the step after it
        derive();
        if (t > 1) fetch(t - 2);
        // 3. dh_{t-1}[k] = direct + sum over source units
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int bt0 = (q * NRS + rs) * RT;
            if (bt0 < nrows) {
                float acc[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < KP / 4; ++i)
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            const float4 dv = *reinterpret_cast<const float4 *>(d_s + ((bt0 + r) * 3 + g) * HP + i * 64 + us * 4);
                            acc[r] = __fmaf_rn(dv.x, w[g][4 * i + 0], acc[r]);
                            acc[r] = __fmaf_rn(dv.y, w[g][4 * i + 1], acc[r]);
                            acc[r] = __fmaf_rn(dv.z, w[g][4 * i + 2], acc[r]);
                            acc[r] = __fmaf_rn(dv.w, w[g][4 * i + 3], acc[r]);
                        }
                float mine = 0.0f;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    acc[r] = ddsp_osc::group_sum(acc[r], 4);
                    if (us == r) mine = acc[r];
                }
                carry[q] = direct[q] + mine;
            }
        }
        __syncthreads();  // d_s is rewritten by the next sweep
    }
    if (*fail_s) {
        // Loud failure: dh0 and the gate gradients of every step this workgroup did not reach become NaN, so the weight
        // gradients (d_gh^T h, sum d_gh, d_gi through autograd) are NaN instead of stale memory.
        if (threadIdx.x == 0) __hip_atomic_store(p.status, (unsigned)GRU_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nan = __builtin_nanf("");
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            carry[q] = nan;
            const int bl = (q * NRS + rs) * RT + us;
            if (gate_lane && bl < nrows)
                for (int s = s_reached; s < p.T; ++s) {
                    const size_t bt = (size_t)(row0 + bl) * p.T + (p.T - 1 - s);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        p.d_gi[bt * G3 + g * Hd + k] = nan;
                        p.d_gh[bt * G3 + g * Hd + k] = nan;
                    }
                }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int bl = (q * NRS + rs) * RT + us;
        if (gate_lane && bl < nrows) p.dh0[(size_t)(row0 + bl) * Hd + k] = carry[q];
    }
}

// ---- bf16 matrix-core variants (torch.autocast callers: train/train.py:50 `precision=16`) -----------------------------
// Same decomposition, hand-off protocol, gate arithmetic (fp32) and failure handling as above; only the products
// h_{t-1} W^T (forward) and (dr, dz, dhn) W (backward) change: `v_mfma_f32_16x16x32_bf16`, fp32 accumulation, weights held
// as bf16 B-operand fragments in registers for the whole sequence, the group's rows (<= 16) as the A operand converted from
// the fp32 LDS image on the fly.  The K range is dealt over the workgroup's four wavefronts (one per SIMD), their partial
// tiles meet in LDS, and thread (row, unit) = (tid / 16, tid % 16) does the gate math.  A step's arithmetic drops from
// ~1.5 us of VALU work to a dozen MFMAs per wavefront, so the step is left with its hand-off latency.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8_t to_bf16x8(float4 a, float4 b)
{
    bf16x8_t r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}

// d_gi / d_gh as the caller's GEMMs want them (GruParams::io16): bf16 arrays behind the float pointers, or fp32.
// (The forward's gi stays fp32: read as bf16 -- even with the conversion deferred to the step that uses the value -- the forward
//  kernel measured 0.87 -> 1.05 ms at the training shape; the cast pass in front of it costs 0.02 ms.)
__device__ __forceinline__ void store_gate_grad(const GruParams &p, float *base, size_t i, float v)
{
    if (p.io16) reinterpret_cast<__bf16 *>(base)[i] = (__bf16)v;
    else base[i] = v;
}

constexpr int kMfmaRows = 16;     // rows of one MFMA tile = the most batch rows a group may hold in these variants

// Forward: the tile is turned round -- M = (unit, gate) packed four to a unit (r, z, n, pad), N = the group's rows -- so that one
// wavefront owns 4 units over the WHOLE K range and each lane ends up with the three pre-activations of one (row, unit) pair
// in its own accumulator registers: no partial tiles, no second barrier between the product and the publish.
template <int KP>
__global__ void __launch_bounds__(256, 1) gru_fwd_mfma_kernel(GruParams p)
{
    // [16][HS] h_{t-1} as the matrix cores take it (bf16: the poller converts each value once, a B fragment is one 16-byte read) | failure flag
    extern __shared__ __attribute__((aligned(16))) __bf16 h_bf[];
    constexpr int HP = 16 * KP, HS = HP + 8, NKS = HP / 32;
    int *fail_s = reinterpret_cast<int *>(h_bf + kMfmaRows * HS);
    const int group = blockIdx.x % p.NGpad, member = blockIdx.x / p.NGpad;
    if (group >= p.NG) return;
    const int row0 = group * p.BL;
    const int nrows = min(p.BL, p.B - row0);
    if (nrows <= 0) return;
    const int Hd = p.Hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ln = lane & 15, kb = lane >> 4;
    const size_t G3 = (size_t)3 * Hd;

    // A fragments (weights): row m = ln = 4 uq + g of the wavefront's tile, unit = member*16 + 4 wave + uq, gate g (3 = padding)
    bf16x8_t wa[NKS];
    {
        const int uq = ln >> 2, g = ln & 3;
        const int u = member * kUnits + 4 * wave + uq;
#pragma unroll
        for (int s = 0; s < NKS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 32 * s + 8 * kb + i;
                const float v = (g < 3 && u < Hd && k < Hd) ? p.w_hh[((size_t)g * Hd + u) * Hd + k] : 0.0f;
                wa[s][i] = (__bf16)v;
            }
    }
    // this lane's (row, unit): D column = ln = batch row, D rows 4 kb + i = (unit 4 wave + kb, gate i)
    const int gr = ln;
    const int gu = member * kUnits + 4 * wave + kb;
    const bool gate = gr < nrows && gu < Hd;
    float bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bh[g] = (gate && p.b_hh) ? p.b_hh[g * Hd + gu] : 0.0f;

    for (int i = threadIdx.x; i < kMfmaRows * HS; i += 256) {
        const int bl = i / HS, k = i - bl * HS;
        h_bf[i] = (__bf16)((bl < nrows && k < Hd && p.h0) ? p.h0[(size_t)(row0 + bl) * Hd + k] : 0.0f);
    }
    float hp = (gate && p.h0) ? p.h0[(size_t)(row0 + gr) * Hd + gu] : 0.0f;   // this lane's own h_{t-1}, fp32 (the blend needs it unrounded)
    if (threadIdx.x == 0) *fail_s = 0;
    __syncthreads();

    const int PRW = (p.BL + 1) / 2;                      // granule rows: batch rows 2r and 2r+1 share a granule (PACK2)
    gu64 *xg = p.xchg + (size_t)group * 2 * PRW * HP;    // [2][PRW][HP]
    float pre[3] = {0.0f, 0.0f, 0.0f};
    if (gate) {
        const size_t bt = (size_t)(row0 + gr) * p.T;
#pragma unroll
        for (int g = 0; g < 3; ++g) pre[g] = p.gi[bt * G3 + g * Hd + gu];
    }
    const int fault_from = (p.fault_step >= 0 && blockIdx.x == 0) ? p.fault_step : 0x7fffffff;
    int t_reached = 0;
    for (int t = 0; t < p.T; ++t) {
        if (t > 0) {
            const bool ok = sweep_packed<4, false, 2>(xg + (size_t)((t - 1) & 1) * PRW * HP, reinterpret_cast<unsigned short *>(h_bf), (nrows + 1) / 2,
                                                      Hd, HP, (unsigned)t, p.status, p.spin_ticks, HS);
            if (!ok && (threadIdx.x & 63) == 0) *fail_s = 1;
        }
        __syncthreads();
        if (*fail_s) break;
        t_reached = t + 1;
        const float gir = pre[0], giz = pre[1], gin = pre[2];
        if (gate && t + 1 < p.T) {
            const size_t bt1 = (size_t)(row0 + gr) * p.T + t + 1;
#pragma unroll
            for (int g = 0; g < 3; ++g) pre[g] = p.gi[bt1 * G3 + g * Hd + gu];
        }
        // B fragments (h_{t-1}, column = batch row ln), two accumulation chains
        // (rows >= nrows of h_bf stay zero from the start: the loads are unconditional, so that all of them are in flight
        //  together instead of one exec-masked, waited-for pair per k-step)
        bf16x8_t hb[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) hb[s] = *reinterpret_cast<const bf16x8_t *>(h_bf + ln * HS + 32 * s + 8 * kb);
        f32x4_t acc0 = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s], hb[s], acc1, 0, 0, 0);
            else       acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s], hb[s], acc0, 0, 0, 0);
        }
        __syncthreads();   // every wavefront has read h_bf: the next sweep may overwrite it
        float hnew = 0.0f;
        if (gate) {
            const float sr = acc0[0] + acc1[0], sz = acc0[1] + acc1[1], sn = acc0[2] + acc1[2];
            const size_t bt = (size_t)(row0 + gr) * p.T + t;
            const float ghn = sn + bh[2];
            const float r = sigmoidf_(gir + (sr + bh[0]));
            const float z = sigmoidf_(giz + (sz + bh[1]));
            const float n = tanhf_(__fmaf_rn(r, ghn, gin));
            hnew = __fmaf_rn(hp - n, z, n);
            hp = hnew;
            p.y[bt * Hd + gu] = hnew;
            if (p.gates) {
                p.gates[bt * G3 + gu] = r;
                p.gates[bt * G3 + Hd + gu] = z;
                p.gates[bt * G3 + 2 * Hd + gu] = n;
            }
            if (p.hn) p.hn[bt * Hd + gu] = ghn;
            if (t == p.T - 1) p.hT[(size_t)(row0 + gr) * Hd + gu] = hnew;
        }
        // hand-off: the lane of the even batch row takes its right-hand neighbour's value (row gr + 1, same unit; 0 past the last
        // row) with one DPP move and publishes the pair
        {
            const float nb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, hnew), 0x101 /* row_shl:1 */, 0xf, 0xf, true));
            if (gate && (gr & 1) == 0 && t < fault_from)
                publish2(xg + ((size_t)(t & 1) * PRW + (gr >> 1)) * HP + gu, (unsigned)t + 1u, hnew, nb);
        }
    }
    __syncthreads();
    if (*fail_s) {
        if (threadIdx.x == 0) __hip_atomic_store(p.status, (unsigned)GRU_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nan = __builtin_nanf("");
        if (gate) {
            p.hT[(size_t)(row0 + gr) * Hd + gu] = nan;
            for (int t = t_reached; t < p.T; ++t) {
                const size_t bt = (size_t)(row0 + gr) * p.T + t;
                p.y[bt * Hd + gu] = nan;
                if (p.gates) {
                    p.gates[bt * G3 + gu] = nan;
                    p.gates[bt * G3 + Hd + gu] = nan;
                    p.gates[bt * G3 + 2 * Hd + gu] = nan;
                }
                if (p.hn) p.hn[bt * Hd + gu] = nan;
            }
        }
    }
}

template <int KP>
__global__ void __launch_bounds__(256, 1) gru_bwd_mfma_kernel(GruParams p)
{
    extern __shared__ __attribute__((aligned(16))) __bf16 d_bf[];   // [16][3][HS] gate gradients (bf16, as the matrix cores take them) | red [4][16][16] fp32 | failure flag
    constexpr int HP = 16 * KP, HS = HP + 8, NKS = 3 * HP / 32, KSW = (NKS + 3) / 4;
    float *red = reinterpret_cast<float *>(d_bf + kMfmaRows * 3 * HS);
    int *fail_s = reinterpret_cast<int *>(red + 4 * 256);
    const int group = blockIdx.x % p.NGpad, member = blockIdx.x / p.NGpad;
    if (group >= p.NG) return;
    const int row0 = group * p.BL;
    const int nrows = min(p.BL, p.B - row0);
    if (nrows <= 0) return;
    const int Hd = p.Hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ln = lane & 15, kb = lane >> 4;
    const size_t G3 = (size_t)3 * Hd;

    // B fragments over the K axis (gate g, source unit u'): k-step s = wave + 4 si covers g = s / (HP/32), u' = 32 (s % (HP/32)) + 8 kb + i;
    // B[k][col ln] = W_hh[g][u'][column member*16 + ln]
    bf16x8_t wb[KSW];
    {
        const int kcol = member * kUnits + ln;
#pragma unroll
        for (int si = 0; si < KSW; ++si) {
            const int s = wave + 4 * si;
            const int g = s / (HP / 32), sb = s - g * (HP / 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int uu = 32 * sb + 8 * kb + i;
                const float v = (s < NKS && kcol < Hd && uu < Hd) ? p.w_hh[((size_t)g * Hd + uu) * Hd + kcol] : 0.0f;
                wb[si][i] = (__bf16)v;
            }
        }
    }
    for (int i = threadIdx.x; i < kMfmaRows * 3 * HS; i += 256) d_bf[i] = (__bf16)0.0f;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) red[i] = 0.0f;
    if (threadIdx.x == 0) *fail_s = 0;
    __syncthreads();

    gu64 *xg = p.xchg + (size_t)group * 2 * p.BL * HP;  // [2][BL][HP]: one PACK3 granule per (row, unit)
    // gate thread (row gr, column gk)
    const int gr = threadIdx.x >> 4, gkl = threadIdx.x & 15;
    const int gk = member * kUnits + gkl;
    const bool gate = gr < nrows && gk < Hd;
    float carry = (gate && p.dhT) ? p.dhT[(size_t)(row0 + gr) * Hd + gk] : 0.0f;
    float direct = 0.0f;
    float dyv = 0.0f, f_r = 0.0f, f_z = 0.0f, f_hn = 0.0f, f_n = 0.0f, f_dir = 0.0f;
    float raw[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto fetch = [&](int t) {
        if (gate) {
            const size_t bt = (size_t)(row0 + gr) * p.T + t;
            raw[0] = p.gates[bt * G3 + gk];
            raw[1] = p.gates[bt * G3 + Hd + gk];
            raw[2] = p.gates[bt * G3 + 2 * Hd + gk];
            raw[3] = p.hn[bt * Hd + gk];
            raw[4] = (t > 0) ? p.y[(bt - 1) * Hd + gk] : (p.h0 ? p.h0[(size_t)(row0 + gr) * Hd + gk] : 0.0f);
            raw[5] = p.dy[bt * Hd + gk];
        }
    };
    auto derive = [&]() {
        const float r = raw[0], z = raw[1], n = raw[2], ghn = raw[3], hp = raw[4];
        dyv = raw[5];
        f_n = (1.0f - z) * (1.0f - n * n);
        f_r = (f_n * ghn) * (r * (1.0f - r));
        f_z = (hp - n) * (z * (1.0f - z));
        f_hn = f_n * r;
        f_dir = z;
    };
    fetch(p.T - 1);
    derive();
    if (p.T > 1) fetch(p.T - 2);

    const int fault_from = (p.fault_step >= 0 && blockIdx.x == 0) ? p.fault_step : 0x7fffffff;
    int s_reached = 0;
    for (int s = 0; s < p.T; ++s) {
        const int t = p.T - 1 - s;
        const unsigned epoch = (unsigned)s + 1u;
        gu64 *slot = xg + (size_t)(s & 1) * p.BL * HP;
        // 1. dh of this step = dy + (what the previous step's product left in `red`) ; gate gradients; publish
        if (gate) {
            if (s > 0)
                carry = direct + ((red[(0 * 16 + gr) * 16 + gkl] + red[(1 * 16 + gr) * 16 + gkl]) +
                                  (red[(2 * 16 + gr) * 16 + gkl] + red[(3 * 16 + gr) * 16 + gkl]));
            const size_t bt = (size_t)(row0 + gr) * p.T + t;
            const float dh = dyv + carry;
            const float dr_pre = dh * f_r, dz_pre = dh * f_z, dhn = dh * f_hn, dn_pre = dh * f_n;
            direct = dh * f_dir;
            if (s < fault_from) publish3(slot + (size_t)gr * HP + gk, epoch, dr_pre, dz_pre, dhn);
            store_gate_grad(p, p.d_gi, bt * G3 + gk, dr_pre);
            store_gate_grad(p, p.d_gi, bt * G3 + Hd + gk, dz_pre);
            store_gate_grad(p, p.d_gi, bt * G3 + 2 * Hd + gk, dn_pre);
            store_gate_grad(p, p.d_gh, bt * G3 + gk, dr_pre);
            store_gate_grad(p, p.d_gh, bt * G3 + Hd + gk, dz_pre);
            store_gate_grad(p, p.d_gh, bt * G3 + 2 * Hd + gk, dhn);
        }
        s_reached = s + 1;
        // 2. the group's gate gradients -> LDS (3 * nrows rows of width Hd)
        {
            const bool ok = sweep_packed<4, true, 3>(slot, reinterpret_cast<unsigned short *>(d_bf), nrows, Hd, HP, epoch, p.status, p.spin_ticks, HS);
            if (!ok && (threadIdx.x & 63) == 0) *fail_s = 1;
        }
        __syncthreads();
        if (*fail_s) break;
        derive();
        if (t > 1) fetch(t - 2);
        // 3. partial tiles of dh_{t-1} = sum over (gate, source unit)
        f32x4_t acc = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int si = 0; si < KSW; ++si) {
            const int ks = wave + 4 * si;
            if (ks < NKS) {   // wave-uniform
                const int g = ks / (HP / 32), sb = ks - g * (HP / 32);
                // (rows >= nrows of d_bf stay zero from the start: unconditional loads, all in flight together)
                const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(d_bf + (ln * 3 + g) * HS + 32 * sb + 8 * kb);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wb[si], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * kb + i) * 16 + ln] = acc[i];
        __syncthreads();   // `red` complete (read at the top of the next step); d_bf free for the next sweep
    }
    if (*fail_s) {
        if (threadIdx.x == 0) __hip_atomic_store(p.status, (unsigned)GRU_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nan = __builtin_nanf("");
        if (gate) {
            p.dh0[(size_t)(row0 + gr) * Hd + gk] = nan;
            for (int s = s_reached; s < p.T; ++s) {
                const size_t bt = (size_t)(row0 + gr) * p.T + (p.T - 1 - s);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    store_gate_grad(p, p.d_gi, bt * G3 + g * Hd + gk, nan);
                    store_gate_grad(p, p.d_gh, bt * G3 + g * Hd + gk, nan);
                }
            }
        }
        return;
    }
    if (gate) {
        const float last = direct + ((red[(0 * 16 + gr) * 16 + gkl] + red[(1 * 16 + gr) * 16 + gkl]) +
                                     (red[(2 * 16 + gr) * 16 + gkl] + red[(3 * 16 + gr) * 16 + gkl]));
        p.dh0[(size_t)(row0 + gr) * Hd + gk] = last;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
struct GruPlan { int KP, HP, NW, NG, NGpad, BL; };
std::atomic<int> g_gru_mode{0};        // ddsp_gru_set_mode: bit 0 = spread placement, bit 1 = fault injection (tests)
std::atomic<int> g_gru_fault_step{0};  // with bit 1: workgroup 0 withholds its publishes from this step on

// Co-residency guard, part 2: two persistent launches that each want one workgroup per CU must not run at the same time
// on one device (two streams of a process could each get part of the CUs and starve each other into the timeout).  Every
// launch of this process therefore waits for the previous recurrence launch on the same device (an event wait on the
// launching stream -- device-side ordering, the host never blocks) and records its own completion.  Launches into a
// stream that is being captured are ordered by the graph itself and skip the event (waiting on an event recorded
// outside the capture is not capturable).  Other PROCESSES sharing the GPU are outside this guard: see DESIGN.md §9a.
struct DeviceGate {
    std::mutex mu;
    hipEvent_t last = nullptr;
    bool armed = false;
};
DeviceGate g_gate[64];

// Groups / rows per group for a [B, Hd] problem on a device with `cus` compute units; false if it does not fit.
// `spread` (test hook): an odd blockIdx modulus NG | 1, which deals every group's workgroups over all XCDs.
bool plan_gru(int B, int Hd, int cus, int max_rows, bool spread, GruPlan *pl)
{
    if (Hd > 512) return false;
    pl->KP = Hd <= 64 ? 4 : (Hd <= 128 ? 8 : (Hd <= 256 ? 16 : 32));
    pl->HP = 16 * pl->KP;
    pl->NW = (Hd + kUnits - 1) / kUnits;
    int slots = cus / pl->NW;          // groups that can be co-resident, one workgroup per CU (two per CU measured
                                       // slower: the waiting workgroup's polls slow its neighbour and the L2)
    if (spread) slots -= 1;            // room for the padding group of the odd modulus
    else slots -= slots % 8;           // blockIdx modulus is a multiple of 8 (XCD alignment)
    if (slots < (spread ? 1 : 8)) return false;
    pl->NG = B < slots ? B : slots;
    pl->BL = (B + pl->NG - 1) / pl->NG;
    pl->NG = (B + pl->BL - 1) / pl->BL;
    pl->NGpad = spread ? (pl->NG | 1) : ((pl->NG + 7) & ~7);
    return pl->BL <= max_rows;
}

int device_cus(int *cus)
{
    static int cached[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (!cached[dev & 63]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        cached[dev & 63] = n;
    }
    *cus = cached[dev & 63];
    return 0;
}

size_t xchg_bytes(const GruPlan &pl, int payloads) { return (size_t)pl.NG * 2 * pl.BL * payloads * pl.HP * sizeof(unsigned long long); }

// Co-residency guard, part 1: every workgroup of the grid spins on its peers, so the whole grid has to be resident at
// once.  The grid is sized one workgroup per CU; what the runtime says a CU admits of this kernel (registers, LDS,
// threads) x the CU count must cover it, otherwise the launch is refused (DDSP_ERANGE) instead of stalling into the
// timeout.  Queried once per kernel instantiation, LDS size class and device.
template <typename K>
hipError_t check_resident(K kernel, int threads, size_t lds, unsigned grid, int (&cache)[64])
{
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (device_cus(&cus)) return hipErrorInvalidDevice;
    int &per_cu = cache[dev & 63];
    if (per_cu == 0) {
        int n = 0;
        // asked with the largest LDS footprint any plan of this instantiation uses, so that the answer can be cached
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds);
        if (e != hipSuccess) return e;
        per_cu = n > 0 ? n : -1;
    }
    if (per_cu < 0 || (unsigned long long)per_cu * (unsigned)cus < grid) return hipErrorCooperativeLaunchTooLarge;
    return hipSuccess;
}

template <int KP, int RT, int NRS>
hipError_t launch_fwd(const GruParams &p, size_t lds, hipStream_t s)
{
    static bool attr[64] = {};
    static int resident[64] = {};
    hipError_t e = ddsp_allow_big_lds((const void *)gru_fwd_kernel<KP, RT, NRS>, attr);
    if (e != hipSuccess) return e;
    e = check_resident(gru_fwd_kernel<KP, RT, NRS>, 256 * NRS, sizeof(float) * ((size_t)kMaxRows * 16 * KP + 4),
                       (unsigned)(p.NGpad * p.NW), resident);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gru_fwd_kernel<KP, RT, NRS>), dim3((unsigned)(p.NGpad * p.NW)), dim3(256 * NRS), lds, s, p);
    return hipGetLastError();
}

template <int KP, int RT, int NRS>
hipError_t launch_bwd(const GruParams &p, size_t lds, hipStream_t s)
{
    static bool attr[64] = {};
    static int resident[64] = {};
    hipError_t e = ddsp_allow_big_lds((const void *)gru_bwd_kernel<KP, RT, NRS>, attr);
    if (e != hipSuccess) return e;
    e = check_resident(gru_bwd_kernel<KP, RT, NRS>, 256 * NRS, sizeof(float) * ((size_t)kMaxRowsBwd * 3 * 16 * KP + 4),
                       (unsigned)(p.NGpad * p.NW), resident);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gru_bwd_kernel<KP, RT, NRS>), dim3((unsigned)(p.NGpad * p.NW)), dim3(256 * NRS), lds, s, p);
    return hipGetLastError();
}

template <int KP>
hipError_t launch_mfma(const GruParams &p, bool backward, hipStream_t s)
{
    constexpr int HS = 16 * KP + 8;     // row stride of the bf16 image in LDS
    static bool attr[2][64] = {};
    static int resident[2][64] = {};
    const size_t lds = backward ? sizeof(__bf16) * (size_t)kMfmaRows * 3 * HS + sizeof(float) * (4 * 256 + 4)
                                : sizeof(__bf16) * (size_t)kMfmaRows * HS + sizeof(float) * 4;
    const unsigned grid = (unsigned)(p.NGpad * p.NW);
    if (backward) {
        hipError_t e = ddsp_allow_big_lds((const void *)gru_bwd_mfma_kernel<KP>, attr[1]);
        if (e != hipSuccess) return e;
        e = check_resident(gru_bwd_mfma_kernel<KP>, 256, lds, grid, resident[1]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gru_bwd_mfma_kernel<KP>), dim3(grid), dim3(256), lds, s, p);
    } else {
        hipError_t e = ddsp_allow_big_lds((const void *)gru_fwd_mfma_kernel<KP>, attr[0]);
        if (e != hipSuccess) return e;
        e = check_resident(gru_fwd_mfma_kernel<KP>, 256, lds, grid, resident[0]);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((gru_fwd_mfma_kernel<KP>), dim3(grid), dim3(256), lds, s, p);
    }
    return hipGetLastError();
}

// Shapes of a workgroup, chosen by same-box A/B (tools/ab_gru.sh): rows per register tile RT and row sets NRS (256 threads
// each, holding the same weights, taking every other tile).  One wavefront per SIMD issues a VALU instruction every 4+
// cycles, so the forward's 384 FMAs + reductions of a 4-row tile (1.45 us of a 2.4 us step by in-kernel timers) run faster
// as two 2-row tiles on two wavefronts per SIMD; the backward prefers one set with a 4-row tile up to 4 rows per group.
template <int KP>
hipError_t launch_gru(const GruParams &p, bool backward, int RT, int NRS, size_t lds, hipStream_t s)
{
    if (!backward) return NRS == 1 ? launch_fwd<KP, 2, 1>(p, lds, s) : launch_fwd<KP, 2, 2>(p, lds, s);
    if (RT == 4) return launch_bwd<KP, 4, 1>(p, lds, s);
    return NRS == 1 ? launch_bwd<KP, 2, 1>(p, lds, s) : launch_bwd<KP, 2, 2>(p, lds, s);
}

int run_gru(GruParams &p, void *scratch, bool backward, hipStream_t s)
{
    int cus = 0;
    const int rc = device_cus(&cus);
    if (rc) return rc;
    GruPlan pl;
    const int mode = g_gru_mode.load(std::memory_order_relaxed);   // test hooks: one snapshot per launch
    const int max_rows = p.lowp ? kMfmaRows : (backward ? kMaxRowsBwd : kMaxRows);
    if (!plan_gru(p.B, p.Hd, cus, max_rows, (mode & 1) != 0, &pl)) return DDSP_ERANGE;
    p.NG = pl.NG; p.NGpad = pl.NGpad; p.NW = pl.NW; p.BL = pl.BL;
    const int payloads = backward ? 3 : 1;
    const bool inject = (mode & 2) != 0;
    p.spin_ticks = inject ? kSpinTicksFaultTest : kSpinTicks;
    p.fault_step = inject ? g_gru_fault_step.load(std::memory_order_relaxed) : -1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    e = hipStreamIsCapturing(s, &capturing);
    if (e != hipSuccess) return (int)e;
    DeviceGate &gate = g_gate[dev & 63];
    std::unique_lock<std::mutex> lock(gate.mu, std::defer_lock);
    if (capturing == hipStreamCaptureStatusNone) {
        lock.lock();      // launch order == event order, also with several host threads
        if (!gate.last) {
            e = hipEventCreateWithFlags(&gate.last, hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
        if (gate.armed) {
            e = hipStreamWaitEvent(s, gate.last, 0);
            if (e != hipSuccess) return (int)e;
        }
    }
    // scratch: [status: 256 B][granules]; every polled word is zeroed before EVERY launch (epochs restart at 1)
    p.status = (gu32 *)scratch;
    p.xchg = (gu64 *)((char *)scratch + 256);
    e = hipMemsetAsync(scratch, 0, 256 + xchg_bytes(pl, payloads), s);
    if (e != hipSuccess) return (int)e;
    // Measured (tools/microbench/gru_lowp_time.py, 512 units, us per step fp32 -> bf16): forward 2.4 -> 2.0 at batch 32, 4.3 -> 2.8 at 64,
    // 8.1 -> 5.9 at 128; backward 4.8 -> 3.5, 7.5 -> 5.1, 16.0 -> 10.6.  With ONE row per group (batch <= 8, the live callback) a step
    // is pure hand-off latency and the forward ties or loses (1.75 -> 1.76 at batch 8, 2.3 -> 2.6 at batch 1): the fp32 kernel, which
    // is at least as accurate, is taken there.
    // (the backward's packed granules carry 16-bit epochs: sequences of 65 536 steps and more take the fp32 kernels)
    const bool use_mfma = p.lowp && (backward ? p.T < 65536 : pl.BL >= 2);
    if (p.io16 && !use_mfma) return DDSP_ERANGE;      // (the fp32 kernels write fp32 arrays only)
    if (use_mfma) {
        switch (pl.KP) {
            case 4: e = launch_mfma<4>(p, backward, s); break;
            case 8: e = launch_mfma<8>(p, backward, s); break;
            case 16: e = launch_mfma<16>(p, backward, s); break;
            default: e = launch_mfma<32>(p, backward, s); break;
        }
    } else {
        const int RT = (backward && pl.BL > 2 && pl.BL <= 4) ? 4 : 2;
        const int NRS = (pl.BL <= 2 || RT == 4) ? 1 : 2;
        const int BLpad = (pl.BL + RT - 1) & ~(RT - 1);
        const size_t lds = sizeof(float) * ((size_t)BLpad * payloads * pl.HP + 4);
        switch (pl.KP) {
            case 4: e = launch_gru<4>(p, backward, RT, NRS, lds, s); break;
            case 8: e = launch_gru<8>(p, backward, RT, NRS, lds, s); break;
            case 16: e = launch_gru<16>(p, backward, RT, NRS, lds, s); break;
            default: e = launch_gru<32>(p, backward, RT, NRS, lds, s); break;
        }
    }
    if (e == hipErrorCooperativeLaunchTooLarge) return DDSP_ERANGE;   // the grid cannot be resident at once on this device
    if (e == hipSuccess && capturing == hipStreamCaptureStatusNone) {
        e = hipEventRecord(gate.last, s);
        gate.armed = e == hipSuccess;
    }
    return (int)e;
}

}  // namespace

extern "C" size_t ddsp_gru_scratch_bytes(int B, int Hd)
{
    if (B <= 0 || Hd <= 0 || Hd > 512) return 0;
    // upper bound over every plan: (B + one group's padding) rows x 3 payloads x 2 parities x 512 granules
    return 256 + (size_t)(B + kMaxRows) * 2 * 3 * 512 * sizeof(unsigned long long);
}

extern "C" int ddsp_gru_max_batch(int Hd, int backward)
{
    int cus = 0;
    if (device_cus(&cus)) return 0;
    GruPlan pl;
    if (!plan_gru(1, Hd, cus, 1, false, &pl)) return 0;
    int slots = cus / pl.NW;
    slots -= slots % 8;
    if (backward & 2) return (slots - 1) * kMfmaRows;           // bit 1: the bf16 matrix-core variants (<= 16 rows per group)
    return (slots - 1) * ((backward & 1) ? kMaxRowsBwd : kMaxRows);   // also valid in the spread test mode
}

extern "C" int ddsp_gru_forward(const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *y, float *hT,
                                float *gates, float *hn, void *scratch, int B, int T, int Hd, void *stream)
{
    if (B == 0) return 0;
    if (!gi || !w_hh || !y || !hT || !scratch || B < 0 || T <= 0 || Hd <= 0) return DDSP_EINVAL;
    if ((gates == nullptr) != (hn == nullptr)) return DDSP_EINVAL;
    GruParams p = {};
    p.gi = gi; p.w_hh = w_hh; p.b_hh = b_hh; p.h0 = h0; p.y = y; p.hT = hT; p.gates = gates; p.hn = hn;
    p.B = B; p.T = T; p.Hd = Hd;
    return run_gru(p, scratch, false, (hipStream_t)stream);
}

extern "C" int ddsp_gru_backward(const float *dy, const float *dhT, const float *w_hh, const float *h0, const float *y,
                                 const float *gates, const float *hn, float *d_gi, float *d_gh, float *dh0, void *scratch,
                                 int B, int T, int Hd, void *stream)
{
    if (B == 0) return 0;
    if (!dy || !w_hh || !y || !gates || !hn || !d_gi || !d_gh || !dh0 || !scratch || B < 0 || T <= 0 || Hd <= 0) return DDSP_EINVAL;
    GruParams p = {};
    p.dy = dy; p.dhT = dhT; p.w_hh = w_hh; p.h0 = h0; p.y = const_cast<float *>(y);
    p.gates = const_cast<float *>(gates); p.hn = const_cast<float *>(hn);
    p.d_gi = d_gi; p.d_gh = d_gh; p.dh0 = dh0;
    p.B = B; p.T = T; p.Hd = Hd;
    return run_gru(p, scratch, true, (hipStream_t)stream);
}

extern "C" int ddsp_gru_forward_bf16(const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *y, float *hT,
                                     float *gates, float *hn, void *scratch, int B, int T, int Hd, void *stream)
{
    if (B == 0) return 0;
    if (!gi || !w_hh || !y || !hT || !scratch || B < 0 || T <= 0 || Hd <= 0) return DDSP_EINVAL;
    if ((gates == nullptr) != (hn == nullptr)) return DDSP_EINVAL;
    GruParams p = {};
    p.gi = gi; p.w_hh = w_hh; p.b_hh = b_hh; p.h0 = h0; p.y = y; p.hT = hT; p.gates = gates; p.hn = hn;
    p.B = B; p.T = T; p.Hd = Hd; p.lowp = 1;
    return run_gru(p, scratch, false, (hipStream_t)stream);
}

extern "C" int ddsp_gru_backward_bf16(const float *dy, const float *dhT, const float *w_hh, const float *h0, const float *y,
                                      const float *gates, const float *hn, void *d_gi, void *d_gh, float *dh0, void *scratch,
                                      int B, int T, int Hd, int io_type, void *stream)
{
    if (B == 0) return 0;
    if (!dy || !w_hh || !y || !gates || !hn || !d_gi || !d_gh || !dh0 || !scratch || B < 0 || T <= 0 || Hd <= 0) return DDSP_EINVAL;
    if (io_type != 0 && io_type != DDSP_IO_BF16) return DDSP_EINVAL;
    GruParams p = {};
    p.io16 = io_type == DDSP_IO_BF16;
    p.dy = dy; p.dhT = dhT; p.w_hh = w_hh; p.h0 = h0; p.y = const_cast<float *>(y);
    p.gates = const_cast<float *>(gates); p.hn = const_cast<float *>(hn);
    p.d_gi = (float *)d_gi; p.d_gh = (float *)d_gh; p.dh0 = dh0;
    p.B = B; p.T = T; p.Hd = Hd; p.lowp = 1;
    return run_gru(p, scratch, true, (hipStream_t)stream);
}

extern "C" int ddsp_gru_set_mode(int mode)
{
    if (mode < 0 || mode > 3) return DDSP_ERANGE;
    if (mode != 0 && !ddsp_hooks_on()) return DDSP_EPERM;
    g_gru_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" int ddsp_gru_set_fault_step(int step)
{
    if (step < 0) return DDSP_ERANGE;
    if (!ddsp_hooks_on()) return DDSP_EPERM;
    g_gru_fault_step.store(step, std::memory_order_relaxed);
    return 0;
}

extern "C" int ddsp_gru_status(const void *scratch, int *status_host)
{
    if (!scratch || !status_host) return DDSP_EINVAL;
    unsigned v = 0;
    const hipError_t e = hipMemcpy(&v, scratch, sizeof(v), hipMemcpyDeviceToHost);
    *status_host = (int)v;
    return (int)e;
}
