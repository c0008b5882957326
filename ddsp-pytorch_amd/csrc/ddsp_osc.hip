// Harmonic oscillator bank for MI355X (gfx950) -- replaces the torch-op chain of
// model/ddsp/harmonic_oscillator.py:24-75 (prepare_harmonics, rescale, generate_phases,
// generate_signal, forward, live).  Written from the arithmetic spec in SURVEY.md App. A.
//
// Decomposition, three launches (+ one that normally exits at once):
//   osc_totals_kernel   per (b,t): frame-rate increments w = fl32(fl32(k*f0*2pi)/sr) recomputed from f0 for the
//                       three bracketing frames, masked/normalised amplitudes of frame t (:26-35), the exact fp64 sum
//                       of the frame's `hop` upsampled increments (:36,:41), and -- through LDS -- the exclusive scan
//                       of those sums over the workgroup's 256/G consecutive frames (a "superblock") + its total
//   osc_supscan_kernel  exclusive scan of the superblock totals along t (tiny)
//   osc_synth_kernel    per (b,t): start phase = superblock prefix + local prefix; re-walk the frame sample by sample:
//                       increment, fp64 accumulate, round to fp32, modulo 2pi32, sin, amplitude, sum over harmonics (:41-49)
//   osc_synth_kernel<EXACT>  bit-exact modulo / live state / debug phases; also redoes the wavefronts the fast kernel
//                       declined (phases outside the fast modulo's exact range) -- exits immediately otherwise
//
// Work mapping of the frame kernels: a GROUP of G = 2^logG adjacent lanes owns one (b,t) frame; each lane keeps
// K harmonics (k = j + G*m) entirely in registers -- fp64 accumulator, the two bracketing frame-rate increments
// and amplitudes -- and walks the frame's samples sequentially, so the phase recurrence needs NO cross-lane scan;
// the only cross-lane traffic is a log2(G)-step DPP sum per output sample.  64/G frames per wavefront.
//
// Compile with -ffp-contract=off: every rounding point below is part of the parity contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_osc_common.h"

using namespace ddsp_osc;

namespace {

// Production walk over samples [n_beg, n_end) of frame t, written stage by stage over the lane's K harmonics
// so that the K independent dependency chains interleave.
// KL <= K: only the lane's first KL harmonic slots are walked (the others are silent in this frame, see
// osc_synth_kernel).  NS consecutive samples are advanced per iteration: a stage then covers NS*KL independent
// instructions, which keeps the short walks (KL = K/4, K/2) from stalling on the latency of the previous stage;
// only the fp64 accumulate is carried from sample to sample.  (segment lengths must be multiples of NS)
template <int K, int MODE, bool POW2, int KL = K, int NS = 1, bool QREUSE = false>
__device__ __forceinline__ void walk_fast(const OscParams &p, FrameState<K> &st, int b, int t, int j, bool active,
                                          int i0, float L0, float L1, int n_beg, int n_end)
{
    extern __shared__ float ystage[];  // [32][256 + 4] per-lane partial outputs (synth kernels launched with stage_out)
    const float i0f = (float)i0;
    float *yframe = p.y + ((long)b * p.T + t) * p.R;
    // POW2 (hop a power of two, clip <= 2^23 samples): the interpolation weight is an exact dyadic rational that
    // advances by exactly 1/hop per sample (0 while the source index is clamped at the clip start), so it is
    // carried incrementally -- bit-identical to the reference's expression -- and the loop bounds are wave-uniform.
    float lam = 0.0f, dlam = 0.0f;
    if (POW2) {
        float w0s;
        upsample_weights(p.scale, t * p.R + n_beg, i0f, w0s, lam);
        dlam = (t == 0 && n_beg == 0) ? 0.0f : p.scale;
    }
    // Each stage is one instruction TYPE over the lane's harmonics; the scheduling barriers keep the stages
    // apart: runs of same-type VALU instructions issue ~10 % faster on gfx950 than the interleaved chains
    // (tools/microbench/valu_rates.hip: "chain staged" vs "osc chain").
#define DDSP_STAGE_END() __builtin_amdgcn_sched_barrier(0)
    // LDS operations of one wavefront execute in order, so the staging buffer needs no s_waitcnt between lane 0's
    // writes and the group's reads -- only the compiler must not reorder them.  (A wavefront-scope fence here also
    // waits for the global stores of the previous flush: measured 2x slowdown of the short walks.)
#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
    // QKEEP: full-width synthesis walks (one sample per iteration) compute the modulo's quotient on every other sample only
    constexpr bool QKEEP = QREUSE && (MODE == MODE_SYNTH) && POW2 && NS == 1;
    float qk[QKEEP ? KL : 1];
    auto one_step = [&](int n, const bool fresh) {
        float w0[NS], w1[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            if (POW2) {
                w1[e] = lam;
                w0[e] = 1.0f - lam;
                lam += dlam;
            } else {
                upsample_weights(p.scale, t * p.R + n + e, i0f, w0[e], w1[e]);
            }
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
            for (int m = 0; m < KL; ++m) v[e][m] = __fmaf_rn(w0[e], st.x0[m], v[e][m]);  // fl32(fma(w0,x[i0],fl32(w1*x[i1])))
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
                st.acc[m] += d[e][m];                                                // :41 double accumulator
                d[e][m] = st.acc[m];
            }
        DDSP_STAGE_END();
        if (MODE == MODE_SYNTH) {
#pragma unroll
            for (int e = 0; e < NS; ++e)
#pragma unroll
                for (int m = 0; m < KL; ++m) v[e][m] = (float)d[e][m];               // ... rounded to fp32 per sample
            DDSP_STAGE_END();
            // P - q*2pi32 is exact in fp32 for q = rint(P/2pi32 +- 0.25) < 2^21: r in (-4.8, 4.8).  Taking the
            // nearest multiple instead of the floor changes sin(r) by <= |2pi32 - 2pi| = 1.75e-7.
            float q[NS][KL];
            if (fresh) {   // wave-uniform
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
                // The previous sample's quotient: below Nyquist the phase advanced by at most pi since then, so
                // r = P - q*2pi32 stays in (-pi, 2 pi] -- still a multiple of 2^-21 below 8, i.e. exact -- and v_sin_f32
                // takes any argument within +-256 revolutions.  Saves the two rounding instructions every other sample.
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
                for (int m = 0; m < KL; ++m) v[e][m] = __builtin_amdgcn_sinf(v[e][m]);  // v_sin_f32 (revolutions)
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
            if (POW2 && p.stage_out) {
                // Output path without cross-lane work in the sample loop: every lane parks L*(its partial sum) in LDS
                // ([sample & 31][thread], padded rows); every 32 samples the G lanes of a group each sum the G partials
                // of 32/G samples and store them -- one whole 128-byte line per group (a 4-byte store per sample makes
                // the L2 allocate, and fetch, each output line long before it is completely written).
                constexpr int kRow = 256 + 4;
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    ystage[((n + e) & 31) * kRow + threadIdx.x] = __fmaf_rn(w0[e], L0, w1[e] * L1) * (s0[e] + s1[e]);
                if (((n + NS - 1) & 31) == 31) {
                    DDSP_WAVE_ORDER();
                    const int G = 1 << p.logG, per = 32 >> p.logG;   // samples per lane: 8, 4 or 2 (G = 4, 8, 16)
                    const int gbase = threadIdx.x & ~(G - 1);
                    float o[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        o[q] = 0.0f;
                        if (q < per) {
                            const float *row = ystage + (j * per + q) * kRow + gbase;
                            for (int g4 = 0; g4 < G; g4 += 4) {
                                const float4 t4 = *reinterpret_cast<const float4 *>(row + g4);
                                o[q] += (t4.x + t4.y) + (t4.z + t4.w);
                            }
                        }
                    }
                    float *dst = yframe + (n + NS - 32) + j * per;
                    if (active) {
                        if (per == 8) {
                            reinterpret_cast<float4 *>(dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
                            reinterpret_cast<float4 *>(dst)[1] = make_float4(o[4], o[5], o[6], o[7]);
                        } else if (per == 4) {
                            reinterpret_cast<float4 *>(dst)[0] = make_float4(o[0], o[1], o[2], o[3]);
                        } else {
                            reinterpret_cast<float2 *>(dst)[0] = make_float2(o[0], o[1]);
                        }
                    }
                    DDSP_WAVE_ORDER();
                }
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e) {
                    const float sum = group_sum(s0[e] + s1[e], p.logG);
                    if (j == 0 && active) yframe[n + e] = __fmaf_rn(w0[e], L0, w1[e] * L1) * sum;
                }
            }
        }
    };
    // (one copy of the body, the choice is a scalar branch: two unrolled copies cost 30 more VGPRs and a wavefront per SIMD)
    for (int n = n_beg; n < n_end; n += NS) one_step(n, !QKEEP || ((n - n_beg) & 1) == 0);
#undef DDSP_STAGE_END
#undef DDSP_WAVE_ORDER
}

// Reference-exact walk: libm fmodf modulo, live offsets (:70), live state and debug phase outputs.
template <int K, int MODE, bool FAST_MOD = false>
__device__ __forceinline__ void walk_exact(const OscParams &p, FrameState<K> &st, const float (&lp)[K], int b, int t,
                                           int j, bool active, int i0, float L0, float L1, int n_beg, int n_end)
{
    const int G = 1 << p.logG;
    const float i0f = (float)i0;
    const long N = (long)p.T * p.R;
    for (int n = n_beg; n < n_end; ++n) {
        const int i = t * p.R + n;
        float w0, w1;
        upsample_weights(p.scale, i, i0f, w0, w1);
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            float inc = __fmaf_rn(w0, st.x0[m], w1 * st.x1[m]);
            inc = (i == 0) ? inc + lp[m] : inc;                   // :70 (lp is zero unless live and b == 0)
            st.acc[m] += (double)inc;
            if (MODE == MODE_SYNTH) {
                const float P = (float)st.acc[m];
                float r;
                if (FAST_MOD) {
                    // live calls inside the fast modulo's range: exact remainder only for the state that is kept (:72)
                    const float q = __fmaf_rn(P, kInvTwoPi32, kRoundMagic) - kRoundMagic;
                    r = __fmaf_rn(-q, kTwoPi32, P);
                    if (p.live_out && active && b == 0 && i == N - 1 && h < p.H) p.live_out[h] = remainder_two_pi(P);
                } else {
                    r = remainder_two_pi(P);                      // :42, exact
                    if (p.dbg_phi && active && h < p.H) p.dbg_phi[((long)b * N + i) * p.H + h] = r;
                    if (p.live_out && active && b == 0 && i == N - 1 && h < p.H) p.live_out[h] = r;  // :72
                }
                const float s = __builtin_amdgcn_sinf(r * kRevPerRad);
                const float A = __fmaf_rn(w1, st.da[m], st.a0[m]);
                sum = __fmaf_rn(A, s, sum);
            }
        }
        if (MODE == MODE_SYNTH) {
            sum = group_sum(sum, p.logG);
            const float L = __fmaf_rn(w0, L0, w1 * L1);
            if (j == 0 && active) p.y[(long)b * N + i] = L * sum;
        }
    }
}

// ---- pass 1: frame totals + superblock-local exclusive scan ----------------------------------------
// One workgroup = one superblock = 256/G consecutive frames of ONE batch row (grid = B * NSB).
template <int K, bool LIVE, bool POW2>
__global__ void __launch_bounds__(256) osc_totals_kernel(OscParams p)
{
    extern __shared__ double tot_s[];  // [FPB][H]
    const int G = 1 << p.logG, FPB = 256 >> p.logG;
    const int blk = (int)xcd_block(blockIdx.x, gridDim.x);
    const int b = blk / p.NSB, sb = blk - b * p.NSB;
    const int fl = threadIdx.x >> p.logG, j = threadIdx.x & (G - 1);
    int t = sb * FPB + fl;
    const bool active = t < p.T;
    if (!active) t = p.T - 1;
    const int ia = max(t - 1, 0), ib = t;
    const long rowbase = (long)b * p.T;
    const float fb = p.f0[rowbase + ib];

    FrameState<K> st;
    float xb[K], xc[K], lp[K];
    // Increments of the superblock's frames (+ one halo row on each side) go through LDS so that every
    // (frame, harmonic) costs ONE true division; the three bracketing rows are then read back.
    float *w_s = reinterpret_cast<float *>(tot_s + (size_t)FPB * p.H);  // [(FPB + 2)][H]
    {
        const float *crow = p.c + (rowbase + t) * p.H;
        float a0[K];
        float s = 0.0f;
        bool silent = false;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            const bool ok = h < p.H;
            const float hz = (float)(h + 1) * fb;
            a0[m] = (ok && !(hz > p.nyquist)) ? crow[h] : 0.0f;   // :31-32 strict >, integer Nyquist
            s += a0[m];
            silent = silent || (ok && a0[m] == 0.0f);
        }
        // (read first: one atomic per batch instead of one per wavefront on the same address; a stale read only repeats it)
        if (__any(silent && active) && (threadIdx.x & 63) == 0 &&
            __hip_atomic_load(p.redo_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            atomicOr(p.redo_flag + 1, 1);
        s = group_sum(s, p.logG);                                 // any summation order: App. A item 2
        const float rs = 1.0f / s;                                // amp = a0 * (1/s): <= 1 ulp from a0/s (:33); 0 * inf = NaN
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            if (h < p.H) {
                const float wv = frame_increment(h, fb, p.sr);
                w_s[(fl + 1) * p.H + h] = wv;
                if (active) {
                    p.amp[(rowbase + t) * p.H + h] = a0[m] * rs;
                    p.w[(rowbase + t) * p.H + h] = wv;
                }
            }
        }
        for (int e = threadIdx.x; e < 2 * p.H; e += 256) {        // halo rows: frame before / after the superblock
            const int side = e >= p.H, h = e - side * p.H;
            const int row = side ? min(sb * FPB + FPB, p.T - 1) : max(sb * FPB - 1, 0);
            w_s[(side ? FPB + 1 : 0) * p.H + h] = frame_increment(h, p.f0[rowbase + row], p.sr);
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        const bool ok = h < p.H;
        st.acc[m] = 0.0;
        st.x0[m] = ok ? w_s[fl * p.H + h] : 0.0f;
        xc[m] = ok ? w_s[(fl + (p.R == 1 ? 1 : 2)) * p.H + h] : 0.0f;   // hop 1: identity upsampling
        // first half of the hop: frames (t-1, t); frame 0 clamps its source index to 0 but still pairs it with frame 1
        // (weight 0 -- visible only through 0*inf / 0*NaN)
        st.x1[m] = (t == 0) ? xc[m] : (ok ? w_s[(fl + 1) * p.H + h] : 0.0f);
        xb[m] = ok ? w_s[(fl + 1) * p.H + h] : 0.0f;
        lp[m] = (LIVE && ok && b == 0 && p.live_in) ? p.live_in[h] : 0.0f;
    }
    const int split = split_index(t, p.R, p.scale);
    if (LIVE) {
        walk_exact<K, MODE_TOTALS>(p, st, lp, b, t, j, active, ia, 0.0f, 0.0f, 0, split);
    } else if (POW2) {
        walk_fast<K, MODE_TOTALS, true>(p, st, b, t, j, active, ia, 0.0f, 0.0f, 0, p.R >> 1);
    } else {
        walk_fast<K, MODE_TOTALS, false>(p, st, b, t, j, active, ia, 0.0f, 0.0f, 0, split);
    }
#pragma unroll
    for (int m = 0; m < K; ++m) { st.x0[m] = xb[m]; st.x1[m] = xc[m]; }
    if (LIVE) {
        walk_exact<K, MODE_TOTALS>(p, st, lp, b, t, j, active, ib, 0.0f, 0.0f, split, p.R);
    } else if (POW2) {
        walk_fast<K, MODE_TOTALS, true>(p, st, b, t, j, active, ib, 0.0f, 0.0f, p.R >> 1, p.R);
    } else {
        walk_fast<K, MODE_TOTALS, false>(p, st, b, t, j, active, ib, 0.0f, 0.0f, split, p.R);
    }
    // exclusive scan over the superblock's frames, one thread per harmonic column (exact: fp64 sums of fp32 values)
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        if (h < p.H) tot_s[fl * p.H + h] = active ? st.acc[m] : 0.0;
    }
    __syncthreads();
    const int nvalid = min(FPB, p.T - sb * FPB);
    for (int h = threadIdx.x; h < p.H; h += 256) {
        double run = 0.0;
        double *dst = p.loc + (rowbase + (long)sb * FPB) * p.H + h;
        for (int f = 0; f < nvalid; ++f) {
            const double v = tot_s[f * p.H + h];
            dst[(long)f * p.H] = run;
            run += v;
        }
        // (one superblock per row: its exclusive prefix is 0 and the scan launch is skipped)
        p.sup[((long)b * p.NSB + sb) * p.H + h] = p.NSB == 1 ? 0.0 : run;
    }
    if (p.NSB == 1 && blockIdx.x == 0 && threadIdx.x == 0) p.redo_flag[2] = kFrameScratchTag;   // (otherwise the scan kernel tags)
}

// ---- pass 2: exclusive scan of the superblock totals along t (B*H independent columns, NSB steps) -------
__global__ void __launch_bounds__(256) osc_supscan_kernel(OscParams p)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx == 0) p.redo_flag[2] = kFrameScratchTag;   // this scratch can serve ddsp_osc_backward
    if (idx >= (long)p.B * p.H) return;
    const int b = (int)(idx / p.H), h = (int)(idx - (long)b * p.H);
    double *col = p.sup + (long)b * p.NSB * p.H + h;
    double run = 0.0;
    // sixteen totals per round trip: read first, then write (one load -> store -> load chain per superblock was 0.6 us each:
    // 11 us for the 16 superblocks of a 4 s clip); the additions keep their order, so the sums keep their bits
    for (int s0 = 0; s0 < p.NSB; s0 += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (s0 + i < p.NSB) ? col[(long)(s0 + i) * p.H] : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (s0 + i < p.NSB) {
                col[(long)(s0 + i) * p.H] = run;
                run += v[i];
            }
    }
}

// ---- pass 3: synthesis ------------------------------------------------------------------------------
// SKIP: the variant that stops at the highest audible harmonic slot; the totals kernel raises redo_flag[1] when the
// batch has any silent (masked / zero-amplitude) harmonic, and exactly one of the SKIP / non-SKIP launches runs
// (the other exits on the flag), so an all-audible batch pays nothing for the extra code.
template <int K, int VARIANT, bool POW2, bool SKIP>
__global__ void __launch_bounds__(256, (SKIP && K <= 13) ? 3 : 1) osc_synth_kernel(OscParams p)
{
    if (VARIANT == VAR_EXACT && !p.force_exact && *p.redo_flag == 0) return;
    if (VARIANT == VAR_FAST && POW2 && p.R >= 8 && (p.redo_flag[1] != 0) != SKIP) return;
    const bool probe = VARIANT == VAR_FAST && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    if (probe) clock_stamp(p.redo_flag, 0);
    const int G = 1 << p.logG;
    const long gid = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int j = threadIdx.x & (G - 1);
    long f = gid >> p.logG;
    const long nframes = (long)p.B * p.T;
    const bool active = f < nframes;
    if (!active) f = nframes - 1;  // keep the lanes alive for the cross-lane sums; their stores are masked
    const int b = (int)(f / p.T);
    const int t = (int)(f - (long)b * p.T);
    const int ia = max(t - 1, 0), ib = t, ic = (p.R == 1) ? t : min(t + 1, p.T - 1);  // hop 1: F.interpolate copies (no neighbour term)
    const int sb = t / (256 >> p.logG);

    FrameState<K> st;
    bool fast = true;
    const float *wb = p.w + (long)b * p.T * p.H;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        st.acc[m] = 0.0;
        if (h < p.H) {
            st.acc[m] = p.loc[((long)b * p.T + t) * p.H + h] + p.sup[((long)b * p.NSB + sb) * p.H + h];
            // The fast modulo needs 0 <= P < kFastPhaseLimit over the whole frame: increments of the three
            // bracketing frames non-negative (phases then grow monotonically) and the end-of-frame bound small.
            const float xa = wb[(long)ia * p.H + h], xb = wb[(long)ib * p.H + h], xc = wb[(long)ic * p.H + h];
            const float xmax = fmaxf(fmaxf(xa, xb), xc);
            const float bound = (float)st.acc[m] + (float)p.R * xmax * 1.0001f;
            // (xmax: a quotient reused for the next sample leaves |r| <= pi + increment; v_sin_f32 wants |r| < 256 * 2 pi)
            fast = fast && (xa >= 0.0f) && (xb >= 0.0f) && (xc >= 0.0f) && (st.acc[m] >= 0.0) && (bound < kFastPhaseLimit) && (xmax < 1024.0f);
        }
    }
    fast = __all(fast);  // wave-uniform
    if (VARIANT == VAR_FAST && !fast) {
        if ((threadIdx.x & 63) == 0) atomicOr(p.redo_flag, 1);  // the EXACT kernel that follows redoes this wavefront
        return;
    }
    if (VARIANT == VAR_EXACT && fast && !p.force_exact) return;  // already written by the FAST kernel
    const int split = split_index(t, p.R, p.scale);
    float L0, L1;
    if (VARIANT == VAR_FAST) {
        // Harmonics whose amplitude is exactly zero at all three bracketing frames (above Nyquist: :31-32) are silent
        // for the whole frame and their phase is not needed either (every frame starts from the scanned totals), so
        // the walk stops at the highest slot that is audible anywhere in the wavefront: 1/4, 1/2 or all of K.
        constexpr int KQ = (K + 3) / 4, KH = (K + 1) / 2;
        int mlive = K;
        if (SKIP) {
            // from f0 (three loads) instead of the 3 x K amplitudes themselves (39 loads and their addresses: the registers this
            // variant spilled): a harmonic above Nyquist at all three bracketing frames has amplitude exactly zero there (:31-32).
            // Zero amplitudes BELOW Nyquist are simply walked; NaN f0 compares false -> kept; an all-masked frame (NaN amplitudes,
            // :33) still reaches the output through slot 0, which every walk includes.
            mlive = 0;
            const long rowbase = (long)b * p.T;
            const float fmin3 = fminf(fminf(p.f0[rowbase + ia], p.f0[rowbase + ib]), p.f0[rowbase + ic]);
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const int h = j + m * G;
                const bool keep = h < p.H && !((float)(h + 1) * fmin3 > p.nyquist);
                if (__any(keep)) mlive = m + 1;
            }
        }
#define DDSP_WALK2(KL, NS)                                                                                   \
        do {                                                                                                 \
            load_synth_segment<K>(p, st, b, j, ia, t == 0 ? ic : ib, L0, L1);   /* frame 0 clamps to source 0 but keeps neighbour 1 */                                              \
            if (POW2) walk_fast<K, MODE_SYNTH, true, KL, NS, !SKIP>(p, st, b, t, j, active, ia, L0, L1, 0, p.R >> 1);   \
            else      walk_fast<K, MODE_SYNTH, false, KL, 1>(p, st, b, t, j, active, ia, L0, L1, 0, split);  \
            {   /* (opaque copies: otherwise the compiler forms the second segment's row addresses before the first walk and    \
                   spills the 64-bit pairs across it -- the 10 spilled registers of round 3's SKIP variant) */                 \
                int ib2 = ib, ic2 = ic;                                                                      \
                asm volatile("" : "+v"(ib2), "+v"(ic2));                                                     \
                load_synth_segment<K>(p, st, b, j, ib2, ic2, L0, L1);                                        \
            }                                                                                                \
            if (POW2) walk_fast<K, MODE_SYNTH, true, KL, NS, !SKIP>(p, st, b, t, j, active, ib, L0, L1, p.R >> 1, p.R); \
            else      walk_fast<K, MODE_SYNTH, false, KL, 1>(p, st, b, t, j, active, ib, L0, L1, split, p.R); \
        } while (0)
        // short walks advance 4 / 2 samples per iteration (the SKIP kernel is only launched for hop >= 8)
        if (SKIP && mlive <= KQ) DDSP_WALK2(KQ, 4);
        else if (SKIP && mlive <= KH) DDSP_WALK2(KH, 2);
        else DDSP_WALK2(K, 1);
#undef DDSP_WALK2
        if (probe) clock_stamp(p.redo_flag, 1);
    } else {
        float lp[K];
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            lp[m] = (h < p.H && b == 0 && p.live_in) ? p.live_in[h] : 0.0f;
        }
        if (fast && !p.dbg_phi) {
            load_synth_segment<K>(p, st, b, j, ia, t == 0 ? ic : ib, L0, L1);   /* frame 0 clamps to source 0 but keeps neighbour 1 */
            walk_exact<K, MODE_SYNTH, true>(p, st, lp, b, t, j, active, ia, L0, L1, 0, split);
            load_synth_segment<K>(p, st, b, j, ib, ic, L0, L1);
            walk_exact<K, MODE_SYNTH, true>(p, st, lp, b, t, j, active, ib, L0, L1, split, p.R);
        } else {
            load_synth_segment<K>(p, st, b, j, ia, t == 0 ? ic : ib, L0, L1);   /* frame 0 clamps to source 0 but keeps neighbour 1 */
            walk_exact<K, MODE_SYNTH>(p, st, lp, b, t, j, active, ia, L0, L1, 0, split);
            load_synth_segment<K>(p, st, b, j, ib, ic, L0, L1);
            walk_exact<K, MODE_SYNTH>(p, st, lp, b, t, j, active, ib, L0, L1, split, p.R);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
template <int K>
hipError_t launch_frames(const OscParams &p, hipStream_t s)
{
    const long lanes = ((long)p.B * p.T) << p.logG;
    const unsigned grid = (unsigned)((lanes + 255) / 256);
    const bool live = p.live_in || p.live_out;
    const size_t lds = (sizeof(double) * (size_t)(256 >> p.logG) + sizeof(float) * (size_t)((256 >> p.logG) + 2)) * p.H;
    static bool lds_attr_set[3][64] = {};  // superblocks of 256 short frames need more than the default 64 KiB
    const void *fns[3] = {(const void *)osc_totals_kernel<K, true, false>, (const void *)osc_totals_kernel<K, false, false>,
                          (const void *)osc_totals_kernel<K, false, true>};
    for (int i = 0; i < 3; ++i) {
        const hipError_t e = ddsp_allow_big_lds(fns[i], lds_attr_set[i]);
        if (e != hipSuccess) return e;
    }
    const dim3 tgrid((unsigned)(p.B * p.NSB)), blk(256);
    if (!live && !p.force_exact) {
        // [0] redo-exact, [1] batch has silent harmonics: only the FAST synth kernels read them (live / debug calls go straight
        // to the EXACT kernel: no reset, one launch fewer in the real-time callback)
        hipError_t me = hipMemsetAsync(p.redo_flag, 0, 2 * sizeof(int), s);
        if (me != hipSuccess) return me;
    }
    int slot = ddsp_prof::begin(ddsp_prof::TOTALS, s);
    if (p.live_in) {
        hipLaunchKernelGGL((osc_totals_kernel<K, true, false>), tgrid, blk, lds, s, p);
    } else if (p.pow2) {
        hipLaunchKernelGGL((osc_totals_kernel<K, false, true>), tgrid, blk, lds, s, p);
    } else {
        hipLaunchKernelGGL((osc_totals_kernel<K, false, false>), tgrid, blk, lds, s, p);
    }
    ddsp_prof::end(slot, s);
    if (p.NSB > 1) {
        slot = ddsp_prof::begin(ddsp_prof::SCAN, s);
        hipLaunchKernelGGL(osc_supscan_kernel, dim3((unsigned)(((long)p.B * p.H + 255) / 256)), dim3(256), 0, s, p);
        ddsp_prof::end(slot, s);
    }
    slot = ddsp_prof::begin(ddsp_prof::SYNTH, s);
    if (!live && !p.force_exact) {
        if (p.pow2) {
            const size_t ylds = p.stage_out ? sizeof(float) * 32 * (256 + 4) : 0;
            hipLaunchKernelGGL((osc_synth_kernel<K, VAR_FAST, true, false>), dim3(grid), blk, ylds, s, p);
            if (p.R >= 8) hipLaunchKernelGGL((osc_synth_kernel<K, VAR_FAST, true, true>), dim3(grid), blk, ylds, s, p);
        } else {
            hipLaunchKernelGGL((osc_synth_kernel<K, VAR_FAST, false, false>), dim3(grid), blk, 0, s, p);
        }
    }
    ddsp_prof::end(slot, s);
    // exits at once unless a wavefront of the FAST kernel raised redo_flag (or exactness is forced)
    hipLaunchKernelGGL((osc_synth_kernel<K, VAR_EXACT, false, false>), dim3(grid), blk, 0, s, p);
    return hipGetLastError();
}

}  // namespace

namespace ddsp_osc {


// Harmonics per lane K (compile-time, register resident) and lanes per frame G = 2^logG with G*K >= H.
// Cost model (measured on MI355X): padded harmonic slots are pure waste; the per-sample work
// shared by a lane's harmonics (weights, cross-lane sum, store) is amortised over K; K >= 20 leaves only two
// wavefronts per SIMD (>= 200 VGPRs), which costs about 10 %.
const int kKs[] = {4, 8, 12, 13, 15, 16, 20, 23, 25};

double tiling_cost(int H, int K, int logG, long frames)
{
    const double waste = (double)((1 << logG) * K) / (double)H;
    const double shared = 1.0 + (0.5 + 0.08 * logG) / (double)K;
    const double occupancy = K >= 20 ? 1.10 : (K >= 15 ? 1.03 : 1.0);
    // small problems (the real-time path: a handful of frames) cannot fill 256 CUs x 4 SIMDs x 2 wavefronts:
    // there the run time is one wavefront's walk, proportional to K, so fewer harmonics per lane win
    const double waves = (double)(frames << logG) / 64.0;
    const double fill = waves >= 2048.0 ? 1.0 : (waves < 1.0 ? 1.0 : waves) / 2048.0;
    return waste * shared * occupancy / fill;
}

std::atomic<int> g_forced_k{0};  // ddsp_osc_set_tiling: 0 = automatic (a test / tuning hook; read once per launch)
std::atomic<int> g_path{0};      // ddsp_osc_set_path: 0 = automatic, 1 = frame kernels only

bool pick_tiling(int H, long frames, Tiling *out)
{
    double best = 1e30;
    bool found = false;
    const int forced = g_forced_k.load(std::memory_order_relaxed);
    for (int K : kKs) {
        if (forced && K != forced) continue;
        const int lanes = (H + K - 1) / K;
        int logG = 0;
        while ((1 << logG) < lanes) ++logG;
        if (logG > 6) continue;
        const double cost = tiling_cost(H, K, logG, frames);
        if (cost < best) {
            best = cost;
            out->K = K;
            out->logG = logG;
            found = true;
        }
    }
    return found;
}

size_t frame_scratch_bytes(int B, int T, int H)
{
    const size_t n = (size_t)B * T * H;
    return 2 * align256(n * sizeof(float)) + align256(n * sizeof(double)) + align256(sup_elems(B, T, H) * sizeof(double)) + 256;
}

bool setup_params(OscParams &p, void *scratch, int B, int T, int H, int hop, int sample_rate)
{
    Tiling tl;
    if (!pick_tiling(H, (long)B * T, &tl)) return false;
    const size_t n = (size_t)B * T * H;
    char *base = (char *)scratch;
    p.w = (float *)base;
    p.amp = (float *)(base + align256(n * sizeof(float)));
    p.loc = (double *)(base + 2 * align256(n * sizeof(float)));
    p.sup = (double *)((char *)p.loc + align256(n * sizeof(double)));
    p.redo_flag = (int *)((char *)p.sup + align256(sup_elems(B, T, H) * sizeof(double)));
    p.B = B; p.T = T; p.H = H; p.R = hop;
    p.K = tl.K;
    p.logG = tl.logG;
    p.NSB = (T + (256 >> tl.logG) - 1) / (256 >> tl.logG);
    p.scale = (float)(1.0 / (double)hop);
    p.nyquist = (float)(sample_rate / 2);
    p.sr = (float)sample_rate;
    p.pow2 = ((hop & (hop - 1)) == 0 && hop >= 2 && (long)T * hop <= (1L << 23)) ? 1 : 0;
    p.stage_out = (p.pow2 && hop >= 64 && tl.logG >= 2 && tl.logG <= 4) ? 1 : 0;
    return true;
}

}  // namespace ddsp_osc

extern "C" int ddsp_osc_set_tiling(int harmonics_per_lane)
{
    if (harmonics_per_lane != 0) {
        if (!ddsp_hooks_on()) return DDSP_EPERM;
        bool known = false;
        for (int K : kKs) known = known || (K == harmonics_per_lane);
        if (!known) return DDSP_ERANGE;
    }
    g_forced_k.store(harmonics_per_lane, std::memory_order_relaxed);
    return 0;
}

extern "C" int ddsp_osc_set_path(int path)
{
    if (path != 0 && !ddsp_hooks_on()) return DDSP_EPERM;
    if (path < 0 || path > 2) return DDSP_ERANGE;
    g_path.store(path == 1 ? 1 : 0, std::memory_order_relaxed);
    g_chunk_any_batch.store(path == 2 ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

extern "C" int ddsp_osc_plan(int B, int T, int H, int hop, int sample_rate, int *out, int cap)
{
    if (!out || cap < 8 || B <= 0 || T <= 0 || H <= 0 || hop <= 0 || sample_rate <= 0) return DDSP_EINVAL;
    OscParams p = {};
    char dummy[1] = {};
    if (!setup_params(p, dummy, B, T, H, hop, sample_rate)) return DDSP_ERANGE;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = p.K;
    out[1] = 1 << p.logG;
    if (g_path.load(std::memory_order_relaxed) == 0 && chunked_eligible(p)) {
        int cus = 0, wgs = 0;
        const hipError_t e = chunk_geometry_k(p, &cus, &wgs);
        if (e != hipSuccess) return (int)e;
        out[2] = 1; out[3] = p.Lc; out[4] = p.NC; out[5] = p.RB; out[6] = cus; out[7] = wgs;
    }
    return 0;
}

extern "C" int ddsp_osc_clock(const void *scratch, int B, int T, int H, int hop, int sample_rate, double *ghz, void *stream)
{
    if (!scratch || !ghz || B <= 0 || T <= 0 || H <= 0 || hop <= 0 || sample_rate <= 0) return DDSP_EINVAL;
    OscParams p = {};
    if (!setup_params(p, const_cast<void *>(scratch), B, T, H, hop, sample_rate)) return DDSP_ERANGE;
    const int *flag = p.redo_flag;
    if (g_path.load(std::memory_order_relaxed) == 0 && chunked_eligible(p)) flag = chunk_flag_words(p);
    unsigned long long w[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(w, flag + 16, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    *ghz = (w[3] > w[1] && w[2] > w[0]) ? (double)(w[2] - w[0]) / (double)(w[3] - w[1]) * 0.1 : 0.0;
    return 0;
}

extern "C" size_t ddsp_osc_scratch_bytes(int B, int T, int H)
{
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    return chunk_scratch_bytes(B, T, H);   // = the frame layout + the chunked form's small arrays behind it
}

extern "C" int ddsp_osc_forward(const float *f0, const float *c, const float *a, float *y, void *scratch,
                                const float *live_in, float *live_out, float *dbg_phi, int B, int T, int H, int hop,
                                int sample_rate, void *stream)
{
    return ddsp_osc_forward_ex(f0, c, a, y, scratch, live_in, live_out, dbg_phi, B, T, H, hop, sample_rate, 0u, stream);
}

extern "C" int ddsp_osc_forward_ex(const float *f0, const float *c, const float *a, float *y, void *scratch,
                                   const float *live_in, float *live_out, float *dbg_phi, int B, int T, int H, int hop,
                                   int sample_rate, unsigned flags, void *stream)
{
    if (B == 0) return 0;
    if (!f0 || !c || !a || !y || !scratch || B < 0 || T <= 0 || H <= 0 || hop <= 0 || sample_rate <= 0) return DDSP_EINVAL;
    if (live_in && live_in == live_out) return DDSP_EINVAL;
    if ((long)T * hop >= (1L << 24) || (long)B * T >= (1L << 31) / 64) return DDSP_ERANGE;

    OscParams p = {};
    if (!setup_params(p, scratch, B, T, H, hop, sample_rate)) return DDSP_ERANGE;
    p.f0 = f0; p.c = c; p.a = a; p.y = y;
    p.live_in = live_in; p.live_out = live_out; p.dbg_phi = dbg_phi;
    p.force_exact = (dbg_phi != nullptr || live_in != nullptr || live_out != nullptr) ? 1 : 0;

    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (!(flags & DDSP_OSC_KEEP_FRAME_SCRATCH) && g_path.load(std::memory_order_relaxed) == 0 && chunked_eligible(p))
        return (int)launch_chunked_k(p, scratch, s);
    switch (p.K) {
#define DDSP_CASE(KK) case KK: e = launch_frames<KK>(p, s); break;
        DDSP_CASE(4) DDSP_CASE(8) DDSP_CASE(12) DDSP_CASE(13) DDSP_CASE(15) DDSP_CASE(16) DDSP_CASE(20) DDSP_CASE(23) DDSP_CASE(25)
#undef DDSP_CASE
        default: return DDSP_ERANGE;
    }
    return (int)e;
}
