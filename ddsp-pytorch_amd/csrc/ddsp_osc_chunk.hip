// Harmonic oscillator bank, CHUNKED form (round 4) -- the production path for power-of-two hops >= 64 with
// 4, 8 or 16 lanes per row group.  Same arithmetic as ddsp_osc.hip (model/ddsp/harmonic_oscillator.py:24-62,
// SURVEY.md App. A), different decomposition of the time axis:
//
//   * The row's samples are cut into CHUNKS of Lc samples (a multiple of 32, >= hop), chosen on the host so that
//     every (row block, chunk) task is resident at once: ONE round of wavefronts, no tail of partly filled rounds.
//   * A wavefront = one chunk index of 64/G consecutive batch rows (G lanes per row, K harmonics per lane, as in the
//     frame kernels); every lane group is at the same sample offset, so loop bounds and interpolation weights are
//     wave-uniform and live in scalar registers.
//   * Inside a chunk the lanes walk SEGMENTS: segment s = samples [s*hop - hop/2, s*hop + hop/2) is the stretch over
//     which F.interpolate (:52-55) uses the ONE bracketing pair (s-1, s) with weight (2n+1)/(2 hop), n = 0..hop-1
//     (clamped at both clip ends).  Crossing into the next segment costs one row of increments and amplitudes;
//     the fp64 accumulators, the older row and everything else stay in registers.  Per-frame work of the frame
//     kernels that is gone: the start-phase loads, the range check's three extra rows, the second segment load.
//   * Per-sample work shared by a lane's harmonics is (almost) gone too: the weights are built by the scalar unit,
//     the loudness factor is applied once per output sample in the flush, not once per lane.
//
// Launches: osc_chunk_totals_kernel (rows w / amp, chunk totals, per-piece "live slots" class and piece totals),
// osc_chunk_scan_kernel (exclusive scan of the chunk totals along the row, flag reset), osc_chunk_synth_kernel
// (audio), osc_chunk_synth_kernel<EXACT> (a <= 256-workgroup grid that returns at once unless a wavefront of the
// fast kernel declined its chunk: phases beyond the fast modulo's exact range, negative or NaN increments).
//
// Compile with -ffp-contract=off: every rounding point below is part of the parity contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_osc_common.h"

using namespace ddsp_osc;

namespace {

constexpr int kRow = 256 + 4;          // staging row: one float per thread of the workgroup, padded
constexpr float kReuseMaxInc = 4.8f;   // quotient reuse: r = P - q*2pi32 stays exact while |r| < 8, i.e. increments < 8 - pi

#define DDSP_STAGE_END() __builtin_amdgcn_sched_barrier(0)
#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// bits of v / 2^lg for odd v, 1 <= v < 2^lg <= 2^14 -- exact; integer operations on wave-uniform values (scalar unit)
__device__ __forceinline__ unsigned dyadic_bits(unsigned v, int lg)
{
    const int top = 31 - __builtin_clz(v);
    return ((unsigned)(126 - lg + top) << 23) + (v << (23 - top));
}

// F.interpolate weights of sample n of a segment (App. A item 4 for a power-of-two hop): w1 = (2n+1)/(2 hop), w0 = 1 - w1
// = (2(hop-1-n)+1)/(2 hop), both exact.  Segment 0 has its source index clamped to 0 (w1 = 0, w0 = 1): `keep` = 0 and
// `one` = bits of 1.0f there, ~0 and 0 elsewhere -- integer selects, so that the weights never leave the scalar registers.
struct SegW { unsigned keep, one; };
__device__ __forceinline__ SegW seg_w(bool clamp0)
{
    SegW g;
    g.keep = clamp0 ? 0u : ~0u;
    g.one = clamp0 ? 0x3f800000u : 0u;
    return g;
}
__device__ __forceinline__ void segment_weights(int n, int R, int lgR, SegW g, float &w0, float &w1)
{
    w1 = __uint_as_float(dyadic_bits(2u * (unsigned)n + 1u, lgR + 1) & g.keep);
    w0 = __uint_as_float((dyadic_bits(2u * (unsigned)(R - 1 - n) + 1u, lgR + 1) & g.keep) | g.one);
}

template <int K>
struct ChunkState {
    double acc[K];
    float x0[K], x1[K];  // increments of the segment's bracketing rows
    float a0[K], da[K];  // amplitude of the older row and (newer - older)   (synth only)
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
template <int K, int KL, int NS, bool QKEEP>
__device__ __forceinline__ void walk_synth(const OscParams &p, ChunkState<K> &st, float *ystage, float *yrow, int j,
                                           bool active, int i_abs, int n_beg, int n_end, bool clamp0, float L0, float L1)
{
    const int G = 1 << p.logG, per = 32 >> p.logG;
    const SegW sw = seg_w(clamp0);
    float *ycol = ystage + threadIdx.x;
    const float *yblk = ystage + (threadIdx.x & ~(G - 1));
    float qk[QKEEP ? KL : 1];
    for (int n = n_beg; n < n_end; n += NS) {
        float w0[NS], w1[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) segment_weights(n + e, p.R, p.lgR, sw, w0[e], w1[e]);
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
        const bool fresh = !QKEEP || (n & 1) == 0;   // wave-uniform
        if (fresh) {
#pragma unroll
            for (int e = 0; e < NS; ++e)
#pragma unroll
                for (int m = 0; m < KL; ++m) q[e][m] = __fmaf_rn(v[e][m], kInvTwoPi32, kRoundMagic);
            DDSP_STAGE_END();
#pragma unroll
            for (int e = 0; e < NS; ++e)
#pragma unroll
                for (int m = 0; m < KL; ++m) q[e][m] = q[e][m] - kRoundMagic;
            DDSP_STAGE_END();
            if (QKEEP) {
#pragma unroll
                for (int m = 0; m < KL; ++m) qk[m] = q[0][m];
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
            for (int m = 0; m < KL; ++m) q[e][m] = __fmaf_rn(w1[e], st.da[m], st.a0[m]);
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
        if (((ia + NS - 1) & 31) == 31) {
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
            const int nblk = n + NS - 32;        // segment offset of the block's first sample
            // loudness of a sample: fma(w0, L0, fl32(w1*L1)) like every other upsampled control (:46)
            auto loud = [&](int u) {
                const int nl = nblk + j * per + u;
                float lw1 = (float)(2 * nl + 1) * p.inv2R, lw0 = 1.0f - lw1;
                if (clamp0) { lw1 = 0.0f; lw0 = 1.0f; }
                return __fmaf_rn(lw0, L0, lw1 * L1);
            };
            float *dst = yrow + (ia + NS - 32) + j * per;
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
    const SegW sw = seg_w(clamp0);
    for (int n = n_beg; n < n_end; ++n) {
        float w0, w1;
        segment_weights(n, p.R, p.lgR, sw, w0, w1);
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const float inc = __fmaf_rn(w0, st.x0[m], w1 * st.x1[m]);
            st.acc[m] += (double)inc;
            const float r = remainder_two_pi((float)st.acc[m]);   // :42, exact
            const float sn = __builtin_amdgcn_sinf(r * kRevPerRad);
            sum = __fmaf_rn(__fmaf_rn(w1, st.da[m], st.a0[m]), sn, sum);
        }
        sum = group_sum(sum, p.logG);
        if (j == 0 && active) yrow[i_abs + (n - n_beg)] = __fmaf_rn(w0, L0, w1 * L1) * sum;
    }
}

// frame totals' chain only (increment, fp64 accumulate) over all K slots
template <int K>
__device__ __forceinline__ void walk_totals(const OscParams &p, double (&acc)[K], const float (&x0)[K], const float (&x1)[K],
                                            int n_beg, int n_end, bool clamp0)
{
    const SegW sw = seg_w(clamp0);
    for (int n = n_beg; n < n_end; ++n) {
        float w0, w1;
        segment_weights(n, p.R, p.lgR, sw, w0, w1);
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

// "live slots" class of a piece: 0 = the wavefront's audible harmonics sit in the first quarter of the slots, 1 = first
// half, 2 = anywhere.  A slot is audible if any lane's amplitude at either bracketing row is not exactly zero (NaN counts);
// nz0 / nz1: this lane's per-slot "amplitude != 0" bits of the two rows.
template <int K>
__device__ __forceinline__ int live_class(unsigned nz0, unsigned nz1)
{
    constexpr int KQ = (K + 3) / 4, KH = (K + 1) / 2;
    unsigned any = nz0 | nz1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) any |= (unsigned)__shfl_xor((int)any, o);
    const int mlive = any ? 32 - __builtin_clz(any) : 0;
    return mlive <= KQ ? 0 : (mlive <= KH ? 1 : 2);
}

// ---- pass 1: rows, chunk totals, piece classes / totals ---------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256, K <= 13 ? 3 : 1) osc_chunk_totals_kernel(OscParams p)
{
    const int wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wt >= p.RB * p.NC) return;
    Task k = decode_task(p, wt);
    const int G = 1 << p.logG;
    const long rowbase = (long)k.b * p.T;
    const int i_beg = k.i;

    // row r of this lane's batch row: increments (:26-35) and masked, normalised amplitudes (:31-33); stored by the
    // chunk in which the row first becomes the NEWER row of a segment (every row exactly once)
    auto make_row = [&](int r, float (&w)[K], unsigned &nz) {
        const float fb = p.f0[rowbase + r];
        const float *crow = p.c + (rowbase + r) * p.H;
        float a0[K];
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            const bool ok = h < p.H;
            const float hz = (float)(h + 1) * fb;
            a0[m] = (ok && !(hz > p.nyquist)) ? crow[h] : 0.0f;   // :31-32 strict >, integer Nyquist
            s += a0[m];
        }
        s = group_sum(s, p.logG);                                  // any summation order: App. A item 2
        const float rs = 1.0f / s;                                 // 0 * inf = NaN for an all-masked frame (:33)
        const int seg_start = max(r * p.R - (p.R >> 1), 0);
        const bool own = k.active && seg_start >= i_beg && seg_start < k.i_end;
        nz = 0u;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            const bool ok = h < p.H;
            w[m] = ok ? frame_increment(h, fb, p.sr) : 0.0f;
            const float amp = ok ? a0[m] * rs : 0.0f;
            if (amp != 0.0f) nz |= 1u << m;                        // NaN counts as audible
            if (ok && own) {
                p.w[(rowbase + r) * p.H + h] = w[m];
                p.amp[(rowbase + r) * p.H + h] = amp;
            }
        }
    };

    float x0[K], x1[K];
    unsigned nz0, nz1;
    int r0, r1;
    segment_rows(k.s, p.T, r0, r1);
    make_row(r0, x0, nz0);
    make_row(r1, x1, nz1);
    double ctot[K];
#pragma unroll
    for (int m = 0; m < K; ++m) ctot[m] = 0.0;
    int piece = 0;
    while (true) {
        const int n_end = min(p.R, k.n + (k.i_end - k.i));
        const int cls = live_class<K>(nz0, nz1);
        if ((threadIdx.x & 63) == 0) p.klive[(long)wt * p.P + piece] = cls;
        double acc[K];
#pragma unroll
        for (int m = 0; m < K; ++m) acc[m] = 0.0;
        walk_totals<K>(p, acc, x0, x1, k.n, n_end, k.s == 0);
#pragma unroll
        for (int m = 0; m < K; ++m) ctot[m] += acc[m];
        if (cls < 2 && k.active) {
            // the synth kernel skips this piece's silent slots and advances their accumulators by these totals
            double *tp = p.tot + ((long)k.b * (p.T + 1 + p.NC) + (k.s + k.c)) * p.H;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const int h = k.j + m * G;
                if (h < p.H) tp[h] = acc[m];
            }
        }
        k.i += n_end - k.n;
        if (k.i >= k.i_end) break;
        ++k.s; ++piece; k.n = 0;
        if (k.s >= 2) {
#pragma unroll
            for (int m = 0; m < K; ++m) x0[m] = x1[m];
            nz0 = nz1;
            if (k.s <= p.T - 1) make_row(k.s, x1, nz1);
        }
    }
    if (k.active) {
        double *cp = p.ctot + ((long)k.b * p.NC + k.c) * p.H;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            if (h < p.H) cp[h] = ctot[m];
        }
    }
}

// ---- pass 2: exclusive scan of the chunk totals along the row (B*H columns, NC steps; exact fp64 sums) ----------
__global__ void __launch_bounds__(256) osc_chunk_scan_kernel(OscParams p)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx == 0) *p.redo_flag = 0;
    if (idx >= (long)p.B * p.H) return;
    const int b = (int)(idx / p.H), h = (int)(idx - (long)b * p.H);
    double *col = p.ctot + (long)b * p.NC * p.H + h;
    double run = 0.0;
    for (int s0 = 0; s0 < p.NC; s0 += 16) {   // sixteen independent loads per round trip; the additions keep their order
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (s0 + i < p.NC) ? col[(long)(s0 + i) * p.H] : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (s0 + i < p.NC) {
                col[(long)(s0 + i) * p.H] = run;
                run += v[i];
            }
    }
}

// ---- pass 3: synthesis --------------------------------------------------------------------------------------------
template <int K, bool EXACT>
__global__ void __launch_bounds__(256, (!EXACT && K <= 13) ? 3 : 1) osc_chunk_synth_kernel(OscParams p)
{
    extern __shared__ float ystage[];   // [32][kRow] (fast kernel only)
    constexpr int KQ = (K + 3) / 4, KH = (K + 1) / 2;
    const int ntasks = p.RB * p.NC;
    int wt = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (EXACT && *p.redo_flag == 0) return;
    for (; wt < ntasks; wt += gridDim.x * 4) {
        if (EXACT && p.redo[wt] == 0) continue;
        Task k = decode_task(p, wt);
        const int G = 1 << p.logG;
        const long rowbase = (long)k.b * p.T;
        float *yrow = p.y + (long)k.b * p.T * p.R;
        ChunkState<K> st;
        bool bad = false;   // per lane: increments negative / NaN, phases beyond the fast modulo's range
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = k.j + m * G;
            st.acc[m] = (h < p.H) ? p.ctot[((long)k.b * p.NC + k.c) * p.H + h] : 0.0;
        }
        int r0, r1;
        segment_rows(k.s, p.T, r0, r1);
        float L0, L1;
        auto load_rows = [&](int ra, int rb2, bool first) {
            const float *w1row = p.w + (rowbase + rb2) * p.H;
            const float *a0row = p.amp + (rowbase + ra) * p.H;
            const float *a1row = p.amp + (rowbase + rb2) * p.H;
            const float *w0row = p.w + (rowbase + ra) * p.H;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const int h = k.j + m * G;
                const bool ok = h < p.H;
                if (first) st.x0[m] = ok ? w0row[h] : 0.0f;
                st.x1[m] = ok ? w1row[h] : 0.0f;
                const float u0 = ok ? a0row[h] : 0.0f;
                const float u1 = ok ? a1row[h] : 0.0f;
                st.a0[m] = u0;
                st.da[m] = u1 - u0;
            }
            L0 = p.a[rowbase + ra];
            L1 = p.a[rowbase + rb2];
        };
        load_rows(r0, r1, true);
        int piece = 0;
        while (true) {
            const int n_end = min(p.R, k.n + (k.i_end - k.i));
            const bool clamp0 = k.s == 0;
            bool big = false;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                bad = bad || !(st.x0[m] >= 0.0f) || !(st.x1[m] >= 0.0f);
                big = big || !(st.x0[m] < kReuseMaxInc) || !(st.x1[m] < kReuseMaxInc);
            }
            if (EXACT) {
                walk_synth_exact<K>(p, st, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
            } else {
                const int cls = __builtin_amdgcn_readfirstlane(p.klive[(long)wt * p.P + piece]);
                if (cls == 0 && KQ < K) {
                    walk_synth<K, KQ, 4, false>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
                } else if (cls <= 1 && KH < K) {
                    walk_synth<K, KH, 2, false>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
                } else if (!__any(big)) {
                    walk_synth<K, K, 1, true>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
                } else {
                    walk_synth<K, K, 1, false>(p, st, ystage, yrow, k.j, k.active, k.i, k.n, n_end, clamp0, L0, L1);
                }
                if (cls < 2) {
                    // silent slots were not walked: advance their accumulators by the piece totals of pass 1
                    const int kl = (cls == 0 && KQ < K) ? KQ : ((KH < K) ? KH : K);
                    const double *tp = p.tot + ((long)k.b * (p.T + 1 + p.NC) + (k.s + k.c)) * p.H;
#pragma unroll
                    for (int m = 0; m < K; ++m) {
                        const int h = k.j + m * G;
                        if (m >= kl && h < p.H) st.acc[m] += tp[h];
                    }
                }
            }
            k.i += n_end - k.n;
            if (k.i >= k.i_end) break;
            ++k.s; ++piece; k.n = 0;
            if (k.s >= 2) {
#pragma unroll
                for (int m = 0; m < K; ++m) st.x0[m] = st.x1[m];
                segment_rows(k.s, p.T, r0, r1);
                load_rows(r0, r1, false);
            }
        }
        if (!EXACT) {
#pragma unroll
            for (int m = 0; m < K; ++m) bad = bad || !(st.acc[m] < (double)kFastPhaseLimit);
            const bool redo = __any(bad);
            if ((threadIdx.x & 63) == 0) {
                p.redo[wt] = redo ? 1 : 0;
                if (redo) atomicOr(p.redo_flag, 1);
            }
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

// Chunk length for `slots` resident wavefronts: every (row block, chunk) task resident at once when the problem allows it
// (one round: the run time is one chunk's walk), otherwise the number of chunks per row that wastes least in the last round.
// c0 = fixed cost of a chunk in samples' worth of walking (prologue loads, first rows).
void pick_chunks(int T, int R, int RB, long slots, int *Lc_out, int *NC_out)
{
    const long N = (long)T * R;
    const double c0 = 24.0;
    double best = 1e300;
    int bestLc = (int)N, bestNC = 1;
    const int ncmax = T;   // chunks are at least one hop long
    int lastLc = -1;
    for (int nc = 1; nc <= ncmax; ++nc) {
        long Lc = (N + nc - 1) / nc;
        Lc = (Lc + 31) & ~31L;
        if (Lc < R) Lc = R;
        if (Lc == lastLc) continue;
        lastLc = (int)Lc;
        const long NC = (N + Lc - 1) / Lc;
        const long tasks = NC * RB;
        const long rounds = (tasks + slots - 1) / slots;
        const double cost = (double)rounds * ((double)Lc + c0);
        if (cost < best) {
            best = cost;
            bestLc = (int)Lc;
            bestNC = (int)NC;
        }
        if (Lc == R) break;
    }
    *Lc_out = bestLc;
    *NC_out = bestNC;
}

bool chunked_eligible(const OscParams &p)
{
    return p.pow2 && p.R >= 64 && p.R <= 8192 && p.logG >= 2 && p.logG <= 4 && !p.live_in && !p.live_out && !p.dbg_phi;
}

size_t chunk_scratch_bytes(int B, int T, int H)
{
    const size_t n = (size_t)B * T * H;
    return 2 * align256(n * sizeof(float)) + align256(n * sizeof(double)) +
           align256((size_t)B * (2 * (size_t)T + 1) * H * sizeof(double)) + align256((size_t)B * (4 * (size_t)T + 4) * sizeof(int)) +
           align256((size_t)B * T * sizeof(int)) + 256;
}

template <int K>
hipError_t launch_chunked(OscParams p, void *scratch, hipStream_t s)
{
    Residency res;
    hipError_t e = synth_residency<K>(&res);
    if (e != hipSuccess) return e;
    const int GPW = 64 >> p.logG;
    p.RB = (p.B + GPW - 1) / GPW;
    p.lgR = 0;
    while ((1 << p.lgR) < p.R) ++p.lgR;
    p.inv2R = 0.5f / (float)p.R;
    const long slots = (long)res.cus * res.wg_per_cu * 4;
    pick_chunks(p.T, p.R, p.RB, slots, &p.Lc, &p.NC);
    p.P = p.Lc / p.R + 2;
    // scratch: w | amp | ctot [B,NC,H] | tot [B,T+1+NC,H] | klive [RB*NC*P] | redo [RB*NC] | flag
    const size_t n = (size_t)p.B * p.T * p.H;
    char *base = (char *)scratch;
    p.w = (float *)base;
    p.amp = (float *)(base + align256(n * sizeof(float)));
    p.ctot = (double *)(base + 2 * align256(n * sizeof(float)));
    p.tot = (double *)((char *)p.ctot + align256((size_t)p.B * p.NC * p.H * sizeof(double)));
    p.klive = (int *)((char *)p.tot + align256((size_t)p.B * (p.T + 1 + p.NC) * p.H * sizeof(double)));
    p.redo = (int *)((char *)p.klive + align256((size_t)p.RB * p.NC * p.P * sizeof(int)));
    p.redo_flag = (int *)((char *)p.redo + align256((size_t)p.RB * p.NC * sizeof(int)));

    const long tasks = (long)p.RB * p.NC;
    const unsigned grid = (unsigned)((tasks + 3) / 4);
    int slot = ddsp_prof::begin(ddsp_prof::TOTALS, s);
    hipLaunchKernelGGL((osc_chunk_totals_kernel<K>), dim3(grid), dim3(256), 0, s, p);
    ddsp_prof::end(slot, s);
    slot = ddsp_prof::begin(ddsp_prof::SCAN, s);
    hipLaunchKernelGGL(osc_chunk_scan_kernel, dim3((unsigned)(((long)p.B * p.H + 255) / 256)), dim3(256), 0, s, p);
    ddsp_prof::end(slot, s);
    slot = ddsp_prof::begin(ddsp_prof::SYNTH, s);
    hipLaunchKernelGGL((osc_chunk_synth_kernel<K, false>), dim3(grid), dim3(256), sizeof(float) * 32 * kRow, s, p);
    ddsp_prof::end(slot, s);
    const unsigned rgrid = grid < 256u ? grid : 256u;
    hipLaunchKernelGGL((osc_chunk_synth_kernel<K, true>), dim3(rgrid), dim3(256), 0, s, p);
    return hipGetLastError();
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
