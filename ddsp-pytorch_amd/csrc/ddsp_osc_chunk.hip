// Harmonic oscillator bank, CHUNKED form (round 4) -- the production path for power-of-two hops >= 64 with
// 4, 8 or 16 lanes per row group.  Same arithmetic as ddsp_osc.hip (model/ddsp/harmonic_oscillator.py:24-62,
// SURVEY.md App. A), different decomposition of the time axis:
//
//   * The row's samples are cut into CHUNKS of Lc samples (a multiple of 32, >= hop), chosen on the host so that
//     every (row block, chunk) task is resident at once: ONE round of wavefronts, no tail of partly filled rounds.
//   * A wavefront = one chunk index of 64/G consecutive batch rows (G lanes per row, K harmonics per lane, as in the
//     frame kernels); every lane group is at the same sample offset, so loop bounds are wave-uniform.  The wavefronts of a
//     SIMD take turns by rotating their priority on the wall clock (take_turn): without that the oldest one runs ahead and
//     the last one finishes alone.
//   * Inside a chunk the lanes walk SEGMENTS: segment s = samples [s*hop - hop/2, s*hop + hop/2) is the stretch over
//     which F.interpolate (:52-55) uses the ONE bracketing pair (s-1, s) with weight (2n+1)/(2 hop), n = 0..hop-1
//     (clamped at both clip ends).  Crossing into the next segment costs one row of increments and amplitudes;
//     the fp64 accumulators, the older row and everything else stay in registers.  Per-frame work of the frame
//     kernels that is gone: the start-phase loads, the range check's three extra rows, the second segment load.
//   * Per-sample work shared by a lane's harmonics shrinks too: the loudness factor is applied once per output
//     sample in the flush, not once per lane and sample.
//
//   * Silent harmonics (above Nyquist, :31-32) are skipped per chunk: pass 1 records, per (row, chunk), the highest
//     harmonic slot that is audible anywhere in the chunk; a wavefront walks 1/8, 1/4, 1/2, 3/4 or all of the K slots,
//     and pass 2 orders the rows of every chunk index so that rows which stop at the same slot share a wavefront.
//
// Launches: osc_chunk_totals_kernel (rows w / amp, chunk totals, highest audible slot per row and chunk),
// osc_chunk_scan_kernel (exclusive scan of the chunk totals along the row, flag reset), osc_chunk_synth_kernel
// (audio; a wavefront that has to decline its chunk -- phases beyond the fast modulo's exact range, negative or NaN
// increments -- walks it a second time with the exact modulo: DDSP_CHUNK_INLINE_REPAIR).
//
// Compile with -ffp-contract=off: every rounding point below is part of the parity contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <stdlib.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_osc_common.h"

using namespace ddsp_osc;

namespace {

#ifndef DDSP_TURN_EVERY
#define DDSP_TURN_EVERY 8    // synth walk: samples between two looks at the clock (divides 32)
#endif
constexpr int kRow = 256 + 4;          // staging row: one float per thread of the workgroup, padded
constexpr float kReuseMaxInc = 4.8f;   // quotient reuse: r = P - q*2pi32 stays exact while |r| < 8, i.e. increments < 8 - pi

#define DDSP_STAGE_END() __builtin_amdgcn_sched_barrier(0)
#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// F.interpolate weights inside a segment (App. A item 4 for a power-of-two hop): sample n has w1 = (2n+1)/(2 hop), exact, and
// advances by exactly 1/hop per sample; segment 0 has its source index clamped to 0: w1 = 0 throughout.  (Built by the
// scalar unit instead -- 25 dependent SALU instructions per sample -- the walk ran 13 % slower: gpurun_out r04a.)
__device__ __forceinline__ void segment_lambda(const OscParams &p, int n, bool clamp0, float &lam, float &dlam)
{
    lam = clamp0 ? 0.0f : (float)(2 * n + 1) * p.inv2R;
    dlam = clamp0 ? 0.0f : 2.0f * p.inv2R;
}

// Fair sharing of a SIMD among its resident wavefronts.  The instruction arbiter serves the highest user priority first and
// the OLDEST wavefront among equals: three equal, always-ready wavefronts on a SIMD then finish at 0.55 / 0.77 / 1.0 of the
// run (measured: tools/microbench/osc_stamps.py), and the last one walks alone at half the SIMD's throughput.  With ONE round of
// resident wavefronts nothing backfills, so the wavefronts take turns instead: priority level (slot + epoch) mod 3 with `slot`
// the hardware wave slot on the SIMD (HW_ID[3:0]: 0, 1, 2 when three are resident) and `epoch` = the 100 MHz wall clock in
// units of about a tenth of the kernel's expected run time (host: turn_shift), which every wavefront of the SIMD reads alike --
// the three levels are always all different, each wavefront holds each of them a third of the time, and all of them reach the
// end together.  (Rotating with a wavefront's own progress instead drifts into equal levels, where age decides again: slot 0
// still finished 25 % early.)
__device__ __forceinline__ int wave_slot()
{
    return __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));   // HW_REG_HW_ID, bits 3:0 = wave slot on the SIMD
}
__device__ __forceinline__ void take_turn(int slot, int nres, int shift)
{
    const unsigned epoch = (unsigned)(__builtin_amdgcn_s_memrealtime() >> shift);
    const unsigned turn = epoch + (unsigned)slot;
    switch (nres >= 3 ? turn % 3u : (nres == 2 ? (turn & 1u) : 0u)) {   // (constant moduli; s_setprio takes an immediate)
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(2); break;
    }
}

// Element `idx` of a scratch array through a 32-bit byte offset (chunked_eligible keeps B*T*H below 2^29 elements): the
// address is uniform base + per-lane 32-bit offset, which costs one register per access instead of a 64-bit pair -- with
// 64-bit per-slot addresses the compiler spilled them and reloaded each one behind a full s_waitcnt at every segment.
__device__ __forceinline__ float ldf(const float *base, unsigned idx)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ double ldd(const double *base, unsigned idx)
{
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 8u));
}

template <int K>
struct ChunkState {
    double acc[K];
    float x0[K], x1[K];  // increments of the segment's bracketing rows
    float a1[K], da[K];  // amplitude of the NEWER row and (newer - older): A(n) = a1 - w0 * da, so that crossing into the next
                         // segment needs the new row only (the older amplitude is the a1 already held)   (synth only)
};

struct Task {
    int wt, rb, c;       // wave task, row block, chunk
    int b, j;            // this lane's batch row (clamped) and position in its row group
    bool active;         // row < B
    int i, i_end;        // absolute sample range of the chunk
    int s, n;            // current segment and offset inside it
};

__device__ __forceinline__ Task decode_task(const OscParams &p, int wt)
{
    Task k;
    k.wt = wt;
    k.rb = wt / p.NC;
    k.c = wt - k.rb * p.NC;
    const int lane = threadIdx.x & 63;
    k.j = lane & ((1 << p.logG) - 1);
    k.b = k.rb * (64 >> p.logG) + (lane >> p.logG);
    k.active = k.b < p.B;
    if (!k.active) k.b = p.B - 1;   // keep the lanes alive: loads are clamped, stores masked
    k.i = k.c * p.Lc;
    k.i_end = min(k.i + p.Lc, p.T * p.R);
    k.s = (k.i + (p.R >> 1)) >> p.lgR;
    k.n = (k.i + (p.R >> 1)) & (p.R - 1);
    return k;
}

// bracketing rows of segment s: (s-1, s) clamped to the clip; segment 0 keeps neighbour 1 (weight 0: 0*inf / 0*NaN only)
__device__ __forceinline__ void segment_rows(int s, int T, int &r0, int &r1)
{
    r0 = s == 0 ? 0 : s - 1;
    r1 = s == 0 ? min(1, T - 1) : min(s, T - 1);
}

// ---- production walk: samples [n_beg, n_end) of the current segment, the lane's first KL harmonic slots ---------
// NS samples per iteration (a stage then covers NS*KL independent instructions); QKEEP: the modulo's quotient is
// computed on even samples and reused on the odd ones (increments < kReuseMaxInc, checked by the caller).
// Every lane parks its partial sum in LDS; after each 32nd sample the G lanes of a row group each sum the partials of
// 32/G samples, apply the loudness and store: one whole 128-byte line per row.
template <int K, int KL, int NS, int QMODE>
__device__ __forceinline__ void walk_synth(const OscParams &p, ChunkState<K> &st, float *ystage, float *yrow, int j,
                                           bool active, int i_abs, int n_beg, int n_end, bool clamp0, float L0, float L1, int slot)
{
    const int G = 1 << p.logG, per = 32 >> p.logG;
    float lam, dlam;
    segment_lambda(p, n_beg, clamp0, lam, dlam);
    float mal = 1.0f - lam;   // w0
    float *ycol = ystage + threadIdx.x;
    const float *yblk = ystage + (threadIdx.x & ~(G - 1));
    static_assert(QMODE == 0 || (QMODE == 1 && NS == 1) || (QMODE == 2 && NS == 2), "quotient reuse: across iterations or inside a pair");
    float qk[QMODE == 1 ? KL : 1];
    // (pieces are multiples of 32 samples; the turn-taking and the flush sit BETWEEN runs of 8 samples so that the sample body
    // stays one basic block -- with a branch inside it the compiler re-interleaved the stages: 4.15 instead of 3.8 cycles per
    // VALU instruction)
    constexpr int RUN = DDSP_TURN_EVERY < NS ? NS : DDSP_TURN_EVERY;
    for (int nb = n_beg; nb < n_end; nb += RUN) {
    take_turn(slot, p.nres, p.turn_shift);
    for (int n = nb; n < nb + RUN; n += NS) {
        float w0[NS], w1[NS];
        w1[0] = lam;
        w0[0] = mal;
#pragma unroll
        for (int e = 1; e < NS; ++e) {   // both weights advance by exactly 1/hop (dyadic rationals: no rounding)
            w1[e] = w1[e - 1] + dlam;
            w0[e] = w0[e - 1] - dlam;
        }
        float v[NS][KL];
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = w1[e] * st.x1[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = __fmaf_rn(w0[e], st.x0[m], v[e][m]);   // fl32(fma(w0,x[i0],fl32(w1*x[i1])))
        DDSP_STAGE_END();
        double d[NS][KL];
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) d[e][m] = (double)v[e][m];
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) {
                st.acc[m] += d[e][m];                                                     // :41 double accumulator
                d[e][m] = st.acc[m];
            }
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = (float)d[e][m];                        // ... rounded to fp32 per sample
        DDSP_STAGE_END();
        // P - q*2pi32 is exact in fp32 for q = rint(P/2pi32) < 2^21; nearest multiple instead of floor
        float q[NS][KL];
        const bool fresh = QMODE != 1 || (n & 1) == 0;   // wave-uniform
        if (fresh) {
            constexpr int NQ = QMODE == 2 ? 1 : NS;
#pragma unroll
            for (int e = 0; e < NQ; ++e)
#pragma unroll
                for (int m = 0; m < KL; ++m) q[e][m] = __fmaf_rn(v[e][m], kInvTwoPi32, kRoundMagic);
            DDSP_STAGE_END();
#pragma unroll
            for (int e = 0; e < NQ; ++e)
#pragma unroll
                for (int m = 0; m < KL; ++m) q[e][m] = q[e][m] - kRoundMagic;
            DDSP_STAGE_END();
            if (QMODE == 1) {
#pragma unroll
                for (int m = 0; m < KL; ++m) qk[m] = q[0][m];
            }
            if (QMODE == 2) {
#pragma unroll
                for (int m = 0; m < KL; ++m) q[NS - 1][m] = q[0][m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < KL; ++m) q[0][m] = qk[m];
        }
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = __fmaf_rn(-q[e][m], kTwoPi32, v[e][m]);  // :42
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = v[e][m] * kRevPerRad;
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) v[e][m] = __builtin_amdgcn_sinf(v[e][m]);         // v_sin_f32 (revolutions)
        DDSP_STAGE_END();
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int m = 0; m < KL; ++m) q[e][m] = __fmaf_rn(-w0[e], st.da[m], st.a1[m]);
        DDSP_STAGE_END();
        float s0[NS], s1[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            s0[e] = 0.0f;
            s1[e] = 0.0f;
#pragma unroll
            for (int m = 0; m < KL; ++m) {
                if (m & 1) s1[e] = __fmaf_rn(q[e][m], v[e][m], s1[e]); else s0[e] = __fmaf_rn(q[e][m], v[e][m], s0[e]);  // :48-49
            }
        }
        DDSP_STAGE_END();
        const int ia = i_abs + (n - n_beg);
#pragma unroll
        for (int e = 0; e < NS; ++e) ycol[((ia + e) & 31) * kRow] = s0[e] + s1[e];
        lam = w1[NS - 1] + dlam;     // (advanced after their last use: no copies)
        mal = w0[NS - 1] - dlam;
    }
        const int ia_end = i_abs + (nb + RUN - n_beg);   // absolute index one past the run
        if ((ia_end & 31) == 0) {
            DDSP_WAVE_ORDER();
            // 32 samples x G partials per row group = 32 floats per lane whatever G is: float4 number f of lane j holds
            // lanes 4*(f mod G/4).. of sample j*per + f / (G/4)
            const int lgq = p.logG - 2;
            float t[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const float4 t4 = *reinterpret_cast<const float4 *>(yblk + (j * per + (f >> lgq)) * kRow + ((f & ((1 << lgq) - 1)) << 2));
                t[f] = (t4.x + t4.y) + (t4.z + t4.w);
            }
            const int nblk = nb + RUN - 32;      // segment offset of the block's first sample
            // loudness of a sample: fma(w0, L0, fl32(w1*L1)) like every other upsampled control (:46)
            auto loud = [&](int u) {
                const int nl = nblk + j * per + u;
                float lw1 = (float)(2 * nl + 1) * p.inv2R, lw0 = 1.0f - lw1;
                if (clamp0) { lw1 = 0.0f; lw0 = 1.0f; }
                return __fmaf_rn(lw0, L0, lw1 * L1);
            };
            float *dst = yrow + (ia_end - 32) + j * per;
            if (p.logG == 2) {
                float o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = t[u] * loud(u);
                if (active) {
                    reinterpret_cast<float4 *>(dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
                    reinterpret_cast<float4 *>(dst)[1] = make_float4(o[4], o[5], o[6], o[7]);
                }
            } else if (p.logG == 3) {
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = (t[2 * u] + t[2 * u + 1]) * loud(u);
                if (active) reinterpret_cast<float4 *>(dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                float o[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) o[u] = ((t[4 * u] + t[4 * u + 1]) + (t[4 * u + 2] + t[4 * u + 3])) * loud(u);
                if (active) reinterpret_cast<float2 *>(dst)[0] = make_float2(o[0], o[1]);
            }
            DDSP_WAVE_ORDER();
        }
    }
}

// Reference-exact walk of the same samples: libm fmodf modulo, per-sample cross-lane sum, direct stores.
template <int K>
__device__ __forceinline__ void walk_synth_exact(const OscParams &p, ChunkState<K> &st, float *yrow, int j, bool active,
                                                 int i_abs, int n_beg, int n_end, bool clamp0, float L0, float L1)
{
    float lam, dlam;
    segment_lambda(p, n_beg, clamp0, lam, dlam);
    for (int n = n_beg; n < n_end; ++n) {
        const float w1 = lam, w0 = 1.0f - lam;
        lam += dlam;
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const float inc = __fmaf_rn(w0, st.x0[m], w1 * st.x1[m]);
            st.acc[m] += (double)inc;
            const float r = remainder_two_pi((float)st.acc[m]);   // :42, exact
            const float sn = __builtin_amdgcn_sinf(r * kRevPerRad);
            sum = __fmaf_rn(__fmaf_rn(-w0, st.da[m], st.a1[m]), sn, sum);
        }
        sum = group_sum(sum, p.logG);
        if (j == 0 && active) yrow[i_abs + (n - n_beg)] = __fmaf_rn(w0, L0, w1 * L1) * sum;
    }
}

// frame totals' chain only (increment, fp64 accumulate) over all K slots
template <int K>
__device__ __forceinline__ void walk_totals(const OscParams &p, double (&acc)[K], const float (&x0)[K], const float (&x1)[K],
                                            int n_beg, int n_end, bool clamp0, int slot)
{
    float lam, dlam;
    segment_lambda(p, n_beg, clamp0, lam, dlam);
    float mal = 1.0f - lam;
    for (int nb = n_beg; nb < n_end; nb += 16) {
    take_turn(slot, p.nres, p.turn_shift);
    for (int n = nb; n < nb + 16; ++n) {
        const float w1 = lam, w0 = mal;
        lam += dlam;
        mal -= dlam;
        float v[K];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < K; ++m) v[m] = w1 * x1[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < K; ++m) v[m] = __fmaf_rn(w0, x0[m], v[m]);
        DDSP_STAGE_END();
        double d[K];
#pragma unroll
        for (int m = 0; m < K; ++m) d[m] = (double)v[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < K; ++m) acc[m] += d[m];
        DDSP_STAGE_END();
    }
    }
}

// OR over the G lanes of a row group
__device__ __forceinline__ unsigned group_or(unsigned v, int logG)
{
    for (int o = 1; o < (1 << logG); o <<= 1) v |= (unsigned)__shfl_xor((int)v, o);
    return v;
}

// ---- pass 1: rows, chunk totals, highest audible slot ----------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256, K <= 13 ? 3 : 1) osc_chunk_totals_kernel(OscParams p)
{
    const int wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wt >= p.RB * p.NC) return;
    Task k = decode_task(p, wt);
    const int G = 1 << p.logG;
    const long rowbase = (long)k.b * p.T;
    const int i_beg = k.i;
    const int slot = wave_slot();
    unsigned nz = 0u;   // this lane's slots with a non-zero amplitude (NaN counts) at any row the chunk interpolates from

    // row r of this lane's batch row: increments (:26-35) and masked, normalised amplitudes (:31-33); stored by the
    // chunk in which the row first becomes the NEWER row of a segment (every row exactly once)
    auto make_row = [&](int r, float (&w)[K]) {
        const float fb = p.f0[rowbase + r];
        const unsigned crow = ((unsigned)k.b * p.T + r) * p.H;
        float a0[K];
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) a0[m] = ldf(p.c, crow + min(k.j + m * G, p.H - 1));   // unconditional, inside the caller's tensor
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            const bool ok = h < p.H;
            const float hz = (float)(h + 1) * fb;
            a0[m] = (ok && !(hz > p.nyquist)) ? a0[m] : 0.0f;     // :31-32 strict >, integer Nyquist
            s += a0[m];
        }
        s = group_sum(s, p.logG);                                  // any summation order: App. A item 2
        const float rs = 1.0f / s;                                 // 0 * inf = NaN for an all-masked frame (:33)
        const int seg_start = max(r * p.R - (p.R >> 1), 0);
        const bool own = k.active && seg_start >= i_beg && seg_start < k.i_end;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            const bool ok = h < p.H;
            w[m] = ok ? frame_increment(h, fb, p.sr) : 0.0f;
            const float amp = ok ? a0[m] * rs : 0.0f;
            if (amp != 0.0f) nz |= 1u << m;
            if (ok && own) {
                p.w[crow + h] = w[m];
                p.amp[crow + h] = amp;
            }
        }
    };

    float x0[K], x1[K];
    int r0, r1;
    segment_rows(k.s, p.T, r0, r1);
    make_row(r0, x0);
    make_row(r1, x1);
    double acc[K];
#pragma unroll
    for (int m = 0; m < K; ++m) acc[m] = 0.0;
    while (true) {
        const int n_end = min(p.R, k.n + (k.i_end - k.i));
        walk_totals<K>(p, acc, x0, x1, k.n, n_end, k.s == 0, slot);
        k.i += n_end - k.n;
        if (k.i >= k.i_end) break;
        ++k.s; k.n = 0;
        if (k.s >= 2) {
#pragma unroll
            for (int m = 0; m < K; ++m) x0[m] = x1[m];
            if (k.s <= p.T - 1) make_row(k.s, x1);
        }
    }
    nz = group_or(nz, p.logG);
    if (k.active) {
        double *cp = p.ctot + ((long)k.b * p.NC + k.c) * p.H;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            if (h < p.H) cp[h] = acc[m];
        }
        if (k.j == 0) p.rlive[(long)k.b * p.NC + k.c] = nz ? 32 - __builtin_clz(nz) : 0;
    }
}

// ---- pass 2: exclusive scan of the chunk totals along the row; rows ordered by audible slots; flag reset --------------
// The first nscan_blocks workgroups scan the columns (exact fp64 sums, so the order of the additions is free); the wavefronts
// of the others take one chunk index each: perm[c][.] = the batch rows ordered by the class of their highest
// audible slot in chunk c (all K slots first, then 3/4, 1/2, 1/4, 1/8), so that the rows a synth wavefront walks together
// stop at the same slot; entries past B are -1.
__global__ void __launch_bounds__(512) osc_chunk_scan_kernel(OscParams p, int nscan_blocks, int Q)
{
    __shared__ double seg_tot[8][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *p.redo_flag = 0;
        p.frame_flag[2] = 0;   // not a frame-form scratch (ddsp_osc_backward checks kFrameScratchTag here)
    }
    if ((int)blockIdx.x < nscan_blocks) {
        // 64 columns per workgroup, coalesced along h; the chunk range is cut into Q <= 8 segments, one per wavefront: first the
        // segment totals (independent loads), then each segment's exclusive scan starting from the totals before it
        const long ncol = (long)p.B * p.H;
        const long idx = (long)blockIdx.x * 64 + lane;
        const bool live = idx < ncol && q < Q;
        const int b = live ? (int)(idx / p.H) : 0, h = live ? (int)(idx - (long)b * p.H) : 0;
        double *col = p.ctot + (long)b * p.NC * p.H + h;
        const int per = (p.NC + Q - 1) / Q, c_beg = q * per, c_end = min(c_beg + per, p.NC);
        if (per <= 32) {
            // the whole segment stays in registers between the two steps: one read, one write
            double v[32];
            double tot = 0.0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                v[i] = (live && c_beg + i < c_end) ? col[(long)(c_beg + i) * p.H] : 0.0;
                tot += v[i];
            }
            seg_tot[q][lane] = tot;
            __syncthreads();
            double run = 0.0;
            for (int qq = 0; qq < q; ++qq) run += seg_tot[qq][lane];
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (live && c_beg + i < c_end) {
                    col[(long)(c_beg + i) * p.H] = run;
                    run += v[i];
                }
            return;
        }
        double tot = 0.0;
        if (live) {
            for (int s0 = c_beg; s0 < c_end; s0 += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (s0 + i < c_end) ? col[(long)(s0 + i) * p.H] : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) tot += v[i];
            }
        }
        seg_tot[q][lane] = tot;
        __syncthreads();
        if (live) {
            double run = 0.0;
            for (int qq = 0; qq < q; ++qq) run += seg_tot[qq][lane];
            for (int s0 = c_beg; s0 < c_end; s0 += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (s0 + i < c_end) ? col[(long)(s0 + i) * p.H] : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (s0 + i < c_end) {
                        col[(long)(s0 + i) * p.H] = run;
                        run += v[i];
                    }
            }
        }
        return;
    }
    const long wv = (long)(blockIdx.x - nscan_blocks) * (blockDim.x >> 6) + q;
    const int c = (int)wv;
    if (c >= p.NC) return;
    const int Bpad = p.RB * (64 >> p.logG);
    int *out = p.perm + (long)c * Bpad;
    const int K = p.K;
    const int lim[4] = {(3 * K + 3) / 4, (K + 1) / 2, (K + 3) / 4, (K + 7) / 8};   // osc_chunk_synth_kernel: KT, KH, KQ, KE
    auto cls_of = [&](int ml) {   // 0 = walks every slot ... 4 = an eighth; -1 = no such row
        if (ml < 0) return -1;
        int q = 0;
        while (q < 4 && ml <= lim[q]) ++q;
        return q;
    };
    int base[5], cnt[5] = {0, 0, 0, 0, 0};
    for (int b0 = 0; b0 < p.B; b0 += 64) {
        const int b = b0 + lane;
        const int cls = cls_of(b < p.B ? p.rlive[(long)b * p.NC + c] : -1);
        for (int q = 0; q < 5; ++q) cnt[q] += __popcll(__ballot(cls == q));
    }
    base[0] = 0;
    for (int q = 1; q < 5; ++q) base[q] = base[q - 1] + cnt[q - 1];
    for (int b0 = 0; b0 < p.B; b0 += 64) {
        const int b = b0 + lane;
        const int cls = cls_of(b < p.B ? p.rlive[(long)b * p.NC + c] : -1);
        for (int q = 0; q < 5; ++q) {
            const unsigned long long mask = __ballot(cls == q);
            if (cls == q) out[base[q] + __popcll(mask & ((1ull << lane) - 1ull))] = b;
            base[q] += __popcll(mask);
        }
    }
    for (int b = p.B + lane; b < Bpad; b += 64) out[b] = -1;
}

#ifdef DDSP_CHUNK_STAMPS
// tuning builds only (tools/build_variant.sh ... -DDDSP_CHUNK_STAMPS): per wave task {start, end} of the synth walk on the
// 100 MHz wall clock, HW_ID, XCC_ID -- read back with ddsp_osc_read_stamps
__device__ long g_stamps[16384 * 4];
extern "C" int ddsp_osc_read_stamps(long *host, int ntasks)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long) * 4 * (size_t)(ntasks < 16384 ? ntasks : 16384));
}
#endif

// ---- pass 3: synthesis --------------------------------------------------------------------------------------------
// DDSP_CHUNK_INLINE_REPAIR (default): a wavefront of the fast kernel that has to decline its chunk (phases beyond the fast modulo's
// exact range, negative or NaN increments) walks it again at once with the exact modulo -- same registers, no LDS -- instead of
// flagging it for a fourth launch (<EXACT>, <= 256 workgroups that returned on a clear flag): the fast path's registers and
// kernel time are unchanged (same-box A/B), the call is one launch shorter (cfg2 0.251 -> 0.249 ms per step).  0 restores the launch.
#ifndef DDSP_CHUNK_INLINE_REPAIR
#define DDSP_CHUNK_INLINE_REPAIR 1
#endif
template <int K, bool EXACT>
__global__ void __launch_bounds__(256, (!EXACT && K <= 13) ? 3 : 1) osc_chunk_synth_kernel(OscParams p)
{
    extern __shared__ float ystage[];   // [32][kRow] (fast kernel only)
    constexpr int KE = (K + 7) / 8, KQ = (K + 3) / 4, KH = (K + 1) / 2, KT = (3 * K + 3) / 4;   // walk lengths below K
    const int ntasks = p.RB * p.NC;
    int wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (EXACT && *p.redo_flag == 0) return;
    for (; wt < ntasks; wt += gridDim.x * 4) {
        if (EXACT && p.redo[wt] == 0) continue;
#ifdef DDSP_CHUNK_STAMPS
        const long stamp0 = wall_clock64();
#endif
        if (!EXACT && wt == 0 && (threadIdx.x & 63) == 0) clock_stamp(p.redo_flag, 0);
        bool exact = EXACT;      // wave-uniform; turns true for a second pass over a declined chunk (DDSP_CHUNK_INLINE_REPAIR)
      for (;;) {
        Task k = decode_task(p, wt);
        {   // the rows of this chunk index in the order of pass 2 (rows that stop at the same slot share a wavefront)
            const int idx = k.rb * (64 >> p.logG) + ((threadIdx.x & 63) >> p.logG);
            const int b = p.perm[(long)k.c * (p.RB * (64 >> p.logG)) + idx];
            k.active = b >= 0;
            k.b = k.active ? b : p.perm[(long)k.c * (p.RB * (64 >> p.logG))];
        }
        const int G = 1 << p.logG;
        const long rowbase = (long)k.b * p.T;
        float *yrow = p.y + (long)k.b * p.T * p.R;
        ChunkState<K> st;
        const int slot = exact ? 0 : wave_slot();
        bool bad = false;   // per lane: increments negative / NaN, phases beyond the fast modulo's range
        // issues them back to back instead of one exec-masked branch per element)
        // (padded slots, h >= H, read whatever follows inside the scratch buffer and are zeroed by a select: every load is
        // unconditional, so the compiler issues them back to back instead of one exec-masked branch per element)
        const unsigned cbase = ((unsigned)k.b * p.NC + k.c) * p.H + k.j;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const double v = ldd(p.ctot, cbase + (unsigned)(m * G));
            st.acc[m] = (k.j + m * G < p.H) ? v : 0.0;
            bad = bad || !(st.acc[m] >= 0.0);
        }
        // slots above the highest audible one of the wavefront's rows are silent for the whole chunk and their phase feeds
        // nothing else (the next chunk starts from the scanned totals): walk 1/8, 1/4, 1/2, 3/4 or all of the K slots
        int mlive = K;
        if (!exact) {
            mlive = p.rlive[(long)k.b * p.NC + k.c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mlive = max(mlive, __shfl_xor(mlive, o));
            mlive = __builtin_amdgcn_readfirstlane(mlive);
        }
        int r0, r1;
        segment_rows(k.s, p.T, r0, r1);
        float L0, L1;
        auto load_rows = [&](int ra, int rb2, bool first) {
            const unsigned o0 = ((unsigned)k.b * p.T + ra) * p.H + k.j, o1 = ((unsigned)k.b * p.T + rb2) * p.H + k.j;
            float t0[K], t1[K], u0[K], u1[K];
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const unsigned g = (unsigned)(m * G);
                t0[m] = first ? ldf(p.w, o0 + g) : 0.0f;
                t1[m] = ldf(p.w, o1 + g);
                u0[m] = first ? ldf(p.amp, o0 + g) : 0.0f;
                u1[m] = ldf(p.amp, o1 + g);
            }
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const bool ok = k.j + m * G < p.H;
                if (first) st.x0[m] = ok ? t0[m] : 0.0f;
                st.x1[m] = ok ? t1[m] : 0.0f;
                const float older = first ? (ok ? u0[m] : 0.0f) : st.a1[m];   // (after the first segment: the row already held)
                st.a1[m] = ok ? u1[m] : 0.0f;
                st.da[m] = st.a1[m] - older;
            }
            if (first) L0 = p.a[rowbase + ra]; else L0 = L1;
            L1 = p.a[rowbase + rb2];
        };
        load_rows(r0, r1, true);
#ifdef DDSP_CHUNK_STAMPS
        long stamp_walk = 0;
        if (DDSP_CHUNK_STAMPS == 2) {   // (forces the first rows to have arrived: the prologue's length)
            float chk = 0.0f;
#pragma unroll
            for (int m = 0; m < K; ++m) chk += st.x1[m] + st.da[m];
            if (__any(chk == 123456.0f)) bad = true;
            stamp_walk = wall_clock64();
        }
#endif
        while (true) {
            const int n_end = min(p.R, k.n + (k.i_end - k.i));
            const bool clamp0 = k.s == 0;
            bool big = false;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                bad = bad || !(st.x0[m] >= 0.0f) || !(st.x1[m] >= 0.0f);
                big = big || !(st.x0[m] < kReuseMaxInc) || !(st.x1[m] < kReuseMaxInc);
            }
            if (exact) {
                walk_synth_exact<K>(p, st, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
            } else if (mlive <= KE && KE < KQ) {
                walk_synth<K, KE, 4, 0>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            } else if (mlive <= KQ && KQ < KH) {
                walk_synth<K, KQ, 4, 0>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            } else if (mlive <= KH && KH < KT) {
                walk_synth<K, KH, 2, 0>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            } else if (mlive <= KT && KT < K) {
                walk_synth<K, KT, 1, 0>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            } else if (!__any(big)) {
                walk_synth<K, K, 1, 1>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            } else {
                walk_synth<K, K, 1, 0>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1, slot);
            }
            k.i += n_end - k.n;
            if (k.i >= k.i_end) break;
            ++k.s; k.n = 0;
            if (k.s >= 2) {
#pragma unroll
                for (int m = 0; m < K; ++m) st.x0[m] = st.x1[m];
                segment_rows(k.s, p.T, r0, r1);
                load_rows(r0, r1, false);
            }
        }
        if (!exact) {
#pragma unroll
            for (int m = 0; m < K; ++m) bad = bad || !(st.acc[m] < (double)kFastPhaseLimit);
            const bool redo = __any(bad);
            if (DDSP_CHUNK_INLINE_REPAIR && redo) { exact = true; continue; }
            if ((threadIdx.x & 63) == 0) {
                p.redo[wt] = redo ? 1 : 0;
                if (redo) atomicOr(p.redo_flag, 1);
                if (wt == 0) clock_stamp(p.redo_flag, 1);
            }
#ifdef DDSP_CHUNK_STAMPS
            if ((threadIdx.x & 63) == 0 && wt < 16384) {
                g_stamps[wt * 4 + 0] = stamp0;
                g_stamps[wt * 4 + 1] = wall_clock64();
                g_stamps[wt * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
                g_stamps[wt * 4 + 3] = DDSP_CHUNK_STAMPS == 2 ? stamp_walk : (long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID, or the first walk's start
            }
#endif
        } else if (DDSP_CHUNK_INLINE_REPAIR && !EXACT && (threadIdx.x & 63) == 0) {   // the second pass of a declined chunk is done
            p.redo[wt] = 0;
            if (wt == 0) clock_stamp(p.redo_flag, 1);
        }
        break;
      }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct Residency { int cus, wg_per_cu; };

template <int K>
hipError_t synth_residency(Residency *out)
{
    static std::mutex mu;
    static Residency cache[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 63;
    std::lock_guard<std::mutex> lk(mu);
    if (cache[dev].cus == 0) {
        int cus = 0, nb = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, osc_chunk_synth_kernel<K, false>, 256, sizeof(float) * 32 * kRow);
        if (e != hipSuccess) return e;
        if (cus <= 0 || nb <= 0) return hipErrorInvalidValue;
        cache[dev].cus = cus;
        cache[dev].wg_per_cu = nb;
    }
    *out = cache[dev];
    return hipSuccess;
}

}  // namespace

namespace ddsp_osc {

std::atomic<int> g_chunk_any_batch{0};   // ddsp_osc_set_path(2): chunked form whatever the batch's fill of the row blocks (tests)

// Chunk length.  A compute unit takes whole workgroups (4 wavefronts, one per SIMD), at most `wg_per_cu` at a time, and a
// SIMD's throughput is about the same with 2 or 3 resident wavefronts (half of it with 1): the run time is the number of
// workgroups the fullest unit gets times one chunk's walk.  Preferred: every (row block, chunk) task resident at once and the
// same number of workgroups on every unit; c0 = fixed cost of a chunk in samples' worth of walking (prologue loads, first rows).
void pick_chunks(int T, int R, int RB, int cus, int wg_per_cu, int *Lc_out, int *NC_out)
{
    const long N = (long)T * R;
    const double c0 = 40.0;
    double best = 1e300;
    int bestLc = (int)N, bestNC = 1;
    int lastLc = -1;
    // (more chunks per row than ~16 workgroups per compute unit never pay: bounds the search for long clips of few rows)
    const long nc_cap = (64L * cus) / (RB > 0 ? RB : 1) + 2;
    for (int nc = 1; nc <= T && nc <= nc_cap; ++nc) {   // chunks are at least one hop long
        long Lc = (N + nc - 1) / nc;
        Lc = (Lc + 31) & ~31L;
        if (Lc < R) Lc = R;
        if (Lc == lastLc) continue;
        lastLc = (int)Lc;
        const long NC = (N + Lc - 1) / Lc;
        const long wgs = (NC * RB + 3) / 4;
        const long per_cu = (wgs + cus - 1) / cus;
        const double alone = per_cu == 1 ? 1.9 : (per_cu == 2 ? 1.03 : 1.0);   // a lone wavefront gets half of the SIMD
        const double cost = (double)per_cu * alone * ((double)Lc + c0);
        if (cost < best) {
            best = cost;
            bestLc = (int)Lc;
            bestNC = (int)NC;
        }
        if (Lc == R) break;
    }
    (void)wg_per_cu;
    *Lc_out = bestLc;
    *NC_out = bestNC;
}

bool chunked_eligible(const OscParams &p)
{
    if (!(p.pow2 && p.R >= 64 && p.R <= 8192 && p.logG >= 2 && p.logG <= 4 && !p.live_in && !p.live_out && !p.dbg_phi &&
          (long)p.B * p.T * p.H < (1L << 29)))   // 32-bit byte offsets into the scratch arrays
        return false;
    // a wavefront takes 64/G ROWS: a batch that fills its last row block badly (3 rows of 8, 9 of 16) idles those lanes for the
    // whole walk, which the frame kernels (consecutive frames of one row per wavefront) do not -- they are ~9 % slower when full
    const int gpw = 64 >> p.logG, rb = (p.B + gpw - 1) / gpw;
    return (long)p.B * 100 >= (long)rb * gpw * 88 || g_chunk_any_batch.load(std::memory_order_relaxed);
}

// The chunked layout keeps the frame layout's first parts (w | amp | fp64 region: ctot [B,NC,H] fits where loc [B,T,H] sits,
// chunks being at least one hop long) and appends its small arrays BEHIND the frame layout's flag words, so that it can clear
// the frame-form tag there (a recycled buffer must not pass for a frame-form scratch in ddsp_osc_backward).
size_t chunk_scratch_bytes(int B, int T, int H)
{
    return frame_scratch_bytes(B, T, H) + 2 * align256((size_t)B * T * sizeof(int)) +
           align256((size_t)T * ((size_t)B + 16) * sizeof(int)) + 256;
}

template <int K>
hipError_t chunk_geometry(OscParams &p, Residency *res_out)
{
    Residency res;
    hipError_t e = synth_residency<K>(&res);
    if (e != hipSuccess) return e;
    const int GPW = 64 >> p.logG;
    p.RB = (p.B + GPW - 1) / GPW;
    p.lgR = 0;
    while ((1 << p.lgR) < p.R) ++p.lgR;
    p.inv2R = 0.5f / (float)p.R;
    pick_chunks(p.T, p.R, p.RB, res.cus, res.wg_per_cu, &p.Lc, &p.NC);
    if (ddsp_hooks_on()) {   // tuning experiments only (DDSP_TEST_HOOKS=1): force the chunk length
        const char *e = getenv("DDSP_OSC_CHUNK_LEN");
        const int v = e ? atoi(e) : 0;
        if (v >= p.R && v % 32 == 0) {
            p.Lc = v;
            p.NC = (int)(((long)p.T * p.R + v - 1) / v);
        }
    }
    p.nres = res.wg_per_cu < 3 ? (res.wg_per_cu < 1 ? 1 : res.wg_per_cu) : 3;
    {   // turn-taking epoch = 1/12 .. 1/6 of the synth kernel's run (a chunk walk costs ~0.056 us per sample and harmonic slot); the
        // totals kernel, a third as long, takes the same epoch: sweeps in profiles/r04_wave_fairness.txt (shorter epochs leave a
        // low-priority wavefront more of each epoch before it looks at the clock again; longer ones too few rotations)
        const double ticks = (double)p.Lc * (double)p.K * 5.6;   // 100 MHz ticks
        int sh = 8;
        while (sh < 16 && (double)(1 << (sh + 1)) <= ticks / 6.0) ++sh;
        p.turn_shift = sh;
        if (ddsp_hooks_on()) {   // tuning experiments only
            const char *e = getenv("DDSP_OSC_TURN_SHIFT");
            const int v = e ? atoi(e) : 0;
            if (v >= 6 && v <= 20) p.turn_shift = v;
        }
    }
    *res_out = res;
    return hipSuccess;
}

void carve_chunk_scratch(OscParams &p, void *scratch)
{
    // scratch: w | amp | ctot [B,NC,H] ... (frame layout's flag words) | rlive [B,NC] | perm [NC, RB*64/G] | redo [RB*NC] | flag
    const size_t n = (size_t)p.B * p.T * p.H;
    char *base = (char *)scratch;
    p.frame_flag = p.redo_flag;   // where setup_params put the frame layout's flag words
    p.w = (float *)base;
    p.amp = (float *)(base + align256(n * sizeof(float)));
    p.ctot = (double *)(base + 2 * align256(n * sizeof(float)));
    p.rlive = (int *)(base + frame_scratch_bytes(p.B, p.T, p.H));
    p.perm = (int *)((char *)p.rlive + align256((size_t)p.B * p.NC * sizeof(int)));
    p.redo = (int *)((char *)p.perm + align256((size_t)p.NC * p.RB * (64 >> p.logG) * sizeof(int)));
    p.redo_flag = (int *)((char *)p.redo + align256((size_t)p.RB * p.NC * sizeof(int)));
}

template <int K>
hipError_t launch_chunked(OscParams p, void *scratch, hipStream_t s)
{
    Residency res;
    hipError_t e = chunk_geometry<K>(p, &res);
    if (e != hipSuccess) return e;
    carve_chunk_scratch(p, scratch);

    const long tasks = (long)p.RB * p.NC;
    const unsigned grid = (unsigned)((tasks + 3) / 4);
    int slot = ddsp_prof::begin(ddsp_prof::TOTALS, s);
    hipLaunchKernelGGL((osc_chunk_totals_kernel<K>), dim3(grid), dim3(256), 0, s, p);
    ddsp_prof::end(slot, s);
    slot = ddsp_prof::begin(ddsp_prof::SCAN, s);
    const long nscan_blocks = ((long)p.B * p.H + 63) / 64;
    const int Q = p.NC > 128 ? 8 : (p.NC > 64 ? 4 : (p.NC > 32 ? 2 : 1));   // <= 32 chunks per wavefront up to 256 chunks: the register form
    hipLaunchKernelGGL(osc_chunk_scan_kernel, dim3((unsigned)(nscan_blocks + (p.NC + Q - 1) / Q)), dim3(64 * Q), 0, s, p, (int)nscan_blocks, Q);   // one wavefront per segment
    ddsp_prof::end(slot, s);
    slot = ddsp_prof::begin(ddsp_prof::SYNTH, s);
    hipLaunchKernelGGL((osc_chunk_synth_kernel<K, false>), dim3(grid), dim3(256), sizeof(float) * 32 * kRow, s, p);
    ddsp_prof::end(slot, s);
    if (!DDSP_CHUNK_INLINE_REPAIR) {
        const unsigned rgrid = grid < 256u ? grid : 256u;
        hipLaunchKernelGGL((osc_chunk_synth_kernel<K, true>), dim3(rgrid), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

hipError_t chunk_geometry_k(OscParams &p, int *cus, int *wg_per_cu)
{
    Residency res = {};
    hipError_t e = hipErrorInvalidValue;
    switch (p.K) {
#define DDSP_CASE(KK) case KK: e = chunk_geometry<KK>(p, &res); break;
        DDSP_CASE(4) DDSP_CASE(8) DDSP_CASE(12) DDSP_CASE(13) DDSP_CASE(15) DDSP_CASE(16) DDSP_CASE(20) DDSP_CASE(23) DDSP_CASE(25)
#undef DDSP_CASE
        default: break;
    }
    *cus = res.cus;
    *wg_per_cu = res.wg_per_cu;
    return e;
}

// the flag words of a chunked launch inside `scratch` (p from setup_params on that scratch)
const int *chunk_flag_words(OscParams p)
{
    int cus = 0, wgs = 0;
    if (chunk_geometry_k(p, &cus, &wgs) != hipSuccess) return p.redo_flag;
    carve_chunk_scratch(p, (void *)p.w);
    return p.redo_flag;
}

hipError_t launch_chunked_k(const OscParams &p, void *scratch, hipStream_t s)
{
    switch (p.K) {
#define DDSP_CASE(KK) case KK: return launch_chunked<KK>(p, scratch, s);
        DDSP_CASE(4) DDSP_CASE(8) DDSP_CASE(12) DDSP_CASE(13) DDSP_CASE(15) DDSP_CASE(16) DDSP_CASE(20) DDSP_CASE(23) DDSP_CASE(25)
#undef DDSP_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ddsp_osc
