// Backward of the harmonic oscillator bank w.r.t. the harmonic amplitudes `c` and the loudness `a`
// (what autograd derives for model/ddsp/harmonic_oscillator.py:24-62; `f0` carries no gradient:
// decoder.py:105 feeds the dataset's f0, so nothing flows through the phase scan).
//
//   y[i] = L[i] * sum_k A[i,k] * sin(phi[i,k]),   L = upsample(a),  A = upsample(amp),  amp = mask(c) / sum_k mask(c)
//
//   d/dA[i,k]   = g[i] * L[i] * sin(phi[i,k])            d/dL[i] = g[i] * sum_k A[i,k] sin(phi[i,k])
//   d/damp[t,k] = sum_i (w0[i] [i0==t] + w1[i] [i1==t]) * d/dA[i,k]          (transpose of F.interpolate)
//   d/dc[t,k]   = mask ? 0 : (d/damp[t,k] - sum_k' d/damp[t,k'] amp[t,k']) / S[t]
//
// osc_bwd_kernel re-walks every frame exactly like the forward synth kernel (same phases, recomputed
// from the forward's scratch: w, amp, loc, sup) and accumulates, per lane and harmonic, the partial
// d/damp aimed at frames t-1, t, t+1; osc_bwd_finish_kernel gathers the three partials per row
// (deterministic: no atomics), applies the normalisation Jacobian and the mask.
// Like the forward it is specialised for power-of-two hops (incremental weights, wave-uniform loop bounds) and
// skips the harmonic slots that are above Nyquist at all three bracketing frames (1/4, 1/2 or all of K walked).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_osc_common.h"

using namespace ddsp_osc;

namespace {

__device__ __noinline__ float remainder_two_pi_call(float p) { return remainder_two_pi(p); }

// One segment (samples [n_beg,n_end) of frame t, interpolating frames i0 -> i1).  plo/phi collect
// sum w0*g*L*sin and sum w1*g*L*sin per harmonic; galo/gahi the same for the loudness.
// POW2 (power-of-two hop): the interpolation weight advances by exactly 1/hop per sample and the segment bounds are
// wave-uniform, as in the forward (ddsp_osc.hip: walk_fast).  KL <= K: only the lane's first KL harmonic slots are walked --
// the others are above Nyquist at all three bracketing frames, so their gradient is zero whatever the walk would give.
template <int K, bool EXACT, bool POW2, int KL = K>
__device__ __forceinline__ void walk_bwd(const OscParams &p, FrameState<K> &st, float (&plo)[K], float (&phi)[K], float &galo,
                                         float &gahi, const float *g_lds, const float *g_glb, int t, int i0, float L0, float L1,
                                         int n_beg, int n_end)
{
    const float i0f = (float)i0;
    float lam = 0.0f, dlam = 0.0f;
    if (POW2) {
        float w0s;
        upsample_weights(p.scale, t * p.R + n_beg, i0f, w0s, lam);
        dlam = (t == 0 && n_beg == 0) ? 0.0f : p.scale;
    }
    // stage-ordered like the forward synth walk (ddsp_osc.hip: walk_fast): one instruction type at a time over the K harmonics
#define DDSP_STAGE_END() __builtin_amdgcn_sched_barrier(0)
    float qk[KL];
    for (int n = n_beg; n < n_end; ++n) {
        float w0, w1;
        if (POW2) {
            w1 = lam;
            w0 = 1.0f - lam;
            lam += dlam;
        } else {
            upsample_weights(p.scale, t * p.R + n, i0f, w0, w1);
        }
        const float gi = g_lds ? g_lds[n] : g_glb[n];
        const float gl = gi * __fmaf_rn(w0, L0, w1 * L1);
        float v[KL];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = w1 * st.x1[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = __fmaf_rn(w0, st.x0[m], v[m]);
        DDSP_STAGE_END();
        double d[KL];
#pragma unroll
        for (int m = 0; m < KL; ++m) d[m] = (double)v[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) st.acc[m] += d[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = (float)st.acc[m];
        DDSP_STAGE_END();
        if (EXACT) {
#pragma unroll
            for (int m = 0; m < KL; ++m) v[m] = remainder_two_pi_call(v[m]);
        } else {
            // the modulo's quotient is computed on every other sample and reused for the next one (ddsp_osc.hip: walk_fast)
            if (!POW2 || ((n - n_beg) & 1) == 0) {   // wave-uniform
#pragma unroll
                for (int m = 0; m < KL; ++m) qk[m] = __fmaf_rn(v[m], kInvTwoPi32, kRoundMagic);
                DDSP_STAGE_END();
#pragma unroll
                for (int m = 0; m < KL; ++m) qk[m] = qk[m] - kRoundMagic;
                DDSP_STAGE_END();
            }
#pragma unroll
            for (int m = 0; m < KL; ++m) v[m] = __fmaf_rn(-qk[m], kTwoPi32, v[m]);
        }
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = v[m] * kRevPerRad;
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = __builtin_amdgcn_sinf(v[m]);
        DDSP_STAGE_END();
        float A[KL];
#pragma unroll
        for (int m = 0; m < KL; ++m) A[m] = __fmaf_rn(w1, st.da[m], st.a0[m]);
        DDSP_STAGE_END();
        float u0 = 0.0f, u1 = 0.0f;
#pragma unroll
        for (int m = 0; m < KL; ++m) {
            if (m & 1) u1 = __fmaf_rn(A[m], v[m], u1); else u0 = __fmaf_rn(A[m], v[m], u0);
        }
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) v[m] = gl * v[m];
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) plo[m] = __fmaf_rn(w0, v[m], plo[m]);
        DDSP_STAGE_END();
#pragma unroll
        for (int m = 0; m < KL; ++m) phi[m] = __fmaf_rn(w1, v[m], phi[m]);
        DDSP_STAGE_END();
        const float gu = gi * group_sum(u0 + u1, p.logG);
        galo = __fmaf_rn(w0, gu, galo);
        gahi = __fmaf_rn(w1, gu, gahi);
    }
#undef DDSP_STAGE_END
}

template <int K, bool POW2>
__global__ void __launch_bounds__(256) osc_bwd_kernel(OscParams p, int use_lds)
{
    extern __shared__ float g_s[];  // [FPB][R] tile of grad_y (when it fits)
    if (p.redo_flag[2] != kFrameScratchTag) return;   // not a frame-form scratch: the finish kernel poisons the gradients
    const int G = 1 << p.logG, FPB = 256 >> p.logG;
    const unsigned blk = xcd_block(blockIdx.x, gridDim.x);
    const long gid = (long)blk * 256 + threadIdx.x;
    const int j = threadIdx.x & (G - 1), fl = threadIdx.x >> p.logG;
    long f = gid >> p.logG;
    const long nframes = (long)p.B * p.T;
    const bool active = f < nframes;
    if (use_lds) {
        const long f0 = (long)blk * FPB;
        const long total = min((long)FPB, nframes - f0) * p.R;
        for (long e = threadIdx.x; e < total; e += 256) g_s[e] = p.grad_y[f0 * p.R + e];
        __syncthreads();
    }
    if (!active) f = nframes - 1;
    const int b = (int)(f / p.T);
    const int t = (int)(f - (long)b * p.T);
    const int ia = max(t - 1, 0), ib = t, ic = (p.R == 1) ? t : min(t + 1, p.T - 1);  // hop 1: F.interpolate copies (no neighbour term)
    const int sb = t / FPB;

    FrameState<K> st;
    bool fast = true;
    const float *wb = p.w + (long)b * p.T * p.H;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        st.acc[m] = 0.0;
        if (h < p.H) {
            st.acc[m] = p.loc[((long)b * p.T + t) * p.H + h] + p.sup[((long)b * p.NSB + sb) * p.H + h];
            const float xa = wb[(long)ia * p.H + h], xb = wb[(long)ib * p.H + h], xc = wb[(long)ic * p.H + h];
            const float xmax = fmaxf(fmaxf(xa, xb), xc);
            const float bound = (float)st.acc[m] + (float)p.R * xmax * 1.0001f;
            fast = fast && (xa >= 0.0f) && (xb >= 0.0f) && (xc >= 0.0f) && (st.acc[m] >= 0.0) && (bound < kFastPhaseLimit) && (xmax < 1024.0f);
        }
    }
    fast = __all(fast);
    const int split = POW2 ? (p.R >> 1) : split_index(t, p.R, p.scale);
    const float *g_lds = use_lds ? g_s + (active ? fl : 0) * p.R : nullptr;
    const float *g_glb = p.grad_y + f * p.R;
    float pm1[K], p0[K], pp1[K];
#pragma unroll
    for (int m = 0; m < K; ++m) pm1[m] = p0[m] = pp1[m] = 0.0f;
    float ga_m1 = 0.0f, ga_0 = 0.0f, ga_p1 = 0.0f;
    float L0, L1;
    // Harmonics above Nyquist at all three bracketing frames (:31-32, strict >) get a zero gradient at each of them and their
    // phase feeds nothing else: the walk stops at the highest slot that is below Nyquist anywhere in the wavefront.
    constexpr int KQ = (K + 3) / 4, KH = (K + 1) / 2;
    int mlive = 0;
    {
        const long rowbase = (long)b * p.T;
        const float fmin3 = fminf(fminf(p.f0[rowbase + ia], p.f0[rowbase + ib]), p.f0[rowbase + ic]);
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            // NaN f0 compares false -> kept; `!(hz > nyquist)` at the smallest of the three f0 <=> unmasked at one of them
            const bool keep = h < p.H && !((float)(h + 1) * fmin3 > p.nyquist);
            if (__any(keep)) mlive = m + 1;
        }
    }
#define DDSP_BWD2(KL)                                                                                                     \
    do {                                                                                                                  \
        load_synth_segment<K>(p, st, b, j, ia, t == 0 ? ic : ib, L0, L1);   /* frame 0 clamps to source 0, keeps neighbour 1 */ \
        if (fast) walk_bwd<K, false, POW2, KL>(p, st, pm1, p0, ga_m1, ga_0, g_lds, g_glb, t, ia, L0, L1, 0, split);        \
        else      walk_bwd<K, true, POW2, KL>(p, st, pm1, p0, ga_m1, ga_0, g_lds, g_glb, t, ia, L0, L1, 0, split);         \
        load_synth_segment<K>(p, st, b, j, ib, ic, L0, L1);                                                               \
        if (fast) walk_bwd<K, false, POW2, KL>(p, st, p0, pp1, ga_0, ga_p1, g_lds, g_glb, t, ib, L0, L1, split, p.R);      \
        else      walk_bwd<K, true, POW2, KL>(p, st, p0, pp1, ga_0, ga_p1, g_lds, g_glb, t, ib, L0, L1, split, p.R);       \
    } while (0)
    if (mlive <= KQ) DDSP_BWD2(KQ);
    else if (mlive <= KH) DDSP_BWD2(KH);
    else DDSP_BWD2(K);
#undef DDSP_BWD2
    if (!active) return;
    float *pc = p.part_c + (((long)b * p.T + t) * 3) * p.H;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        if (h < p.H) {
            pc[h] = pm1[m];
            pc[p.H + h] = p0[m];
            pc[2 * p.H + h] = pp1[m];
        }
    }
    if (j == 0) {
        float *pa = p.part_a + ((long)b * p.T + t) * 3;
        pa[0] = ga_m1; pa[1] = ga_0; pa[2] = ga_p1;
    }
}

// One wavefront per (b,t) row: gather the partials aimed at this row, then the Jacobian of mask + normalise.
__global__ void __launch_bounds__(256) osc_bwd_finish_kernel(OscParams p)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.B * p.T) return;
    if (p.redo_flag[2] != kFrameScratchTag) {
        // the forward was not asked to keep the frame-form scratch (DDSP_OSC_KEEP_FRAME_SCRATCH): fail loudly
        const float nan = __builtin_nanf("");
        for (int h = lane; h < p.H; h += 64) p.grad_c[row * p.H + h] = nan;
        if (lane == 0) p.grad_a[row] = nan;
        return;
    }
    const int b = (int)(row / p.T), t = (int)(row - (long)b * p.T);
    const long base = (long)b * p.T;
    // sources: slot 1 of frame t; slot 0 of frame t+1 (its row t+1-1); slot 2 of frame t-1; the clamped slots at the clip ends
    const float *s_mid = p.part_c + ((base + t) * 3 + 1) * p.H;
    const float *s_next = (t + 1 < p.T) ? p.part_c + ((base + t + 1) * 3 + 0) * p.H : nullptr;
    const float *s_prev = (t > 0) ? p.part_c + ((base + t - 1) * 3 + 2) * p.H : nullptr;
    const float *s_head = (t == 0) ? p.part_c + ((base + 0) * 3 + 0) * p.H : nullptr;
    const float *s_tail = (t == p.T - 1) ? p.part_c + ((base + t) * 3 + 2) * p.H : nullptr;
    const float f = p.f0[row];
    const float *crow = p.c + row * p.H;
    const float *arow = p.amp + row * p.H;
    float s = 0.0f, dot = 0.0f;
    for (int h = lane; h < p.H; h += 64) {
        const float hz = (float)(h + 1) * f;
        s += (hz > p.nyquist) ? 0.0f : crow[h];
        float g = s_mid[h];
        if (s_next) g += s_next[h];
        if (s_prev) g += s_prev[h];
        if (s_head) g += s_head[h];
        if (s_tail) g += s_tail[h];
        dot = __fmaf_rn(g, arow[h], dot);
    }
    s = wave_sum(s);
    dot = wave_sum(dot);
    for (int h = lane; h < p.H; h += 64) {
        const float hz = (float)(h + 1) * f;
        float g = s_mid[h];
        if (s_next) g += s_next[h];
        if (s_prev) g += s_prev[h];
        if (s_head) g += s_head[h];
        if (s_tail) g += s_tail[h];
        p.grad_c[row * p.H + h] = (hz > p.nyquist) ? 0.0f : (g - dot) / s;
    }
    if (lane == 0) {
        float ga = p.part_a[(base + t) * 3 + 1];
        if (t + 1 < p.T) ga += p.part_a[(base + t + 1) * 3 + 0];
        if (t > 0) ga += p.part_a[(base + t - 1) * 3 + 2];
        if (t == 0) ga += p.part_a[(base + 0) * 3 + 0];
        if (t == p.T - 1) ga += p.part_a[(base + t) * 3 + 2];
        p.grad_a[row] = ga;
    }
}

template <int K>
hipError_t launch_bwd(const OscParams &p, hipStream_t s)
{
    const long lanes = ((long)p.B * p.T) << p.logG;
    const unsigned grid = (unsigned)((lanes + 255) / 256);
    const size_t lds = sizeof(float) * (size_t)(256 >> p.logG) * p.R;
    const int use_lds = lds <= 64 * 1024;
    if (p.pow2) hipLaunchKernelGGL((osc_bwd_kernel<K, true>), dim3(grid), dim3(256), use_lds ? lds : 0, s, p, use_lds);
    else        hipLaunchKernelGGL((osc_bwd_kernel<K, false>), dim3(grid), dim3(256), use_lds ? lds : 0, s, p, use_lds);
    const long rows = (long)p.B * p.T;
    hipLaunchKernelGGL(osc_bwd_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace

extern "C" size_t ddsp_osc_backward_scratch_bytes(int B, int T, int H)
{
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    return align256((size_t)B * T * 3 * H * sizeof(float)) + align256((size_t)B * T * 3 * sizeof(float));
}

extern "C" int ddsp_osc_backward(const float *grad_y, const float *f0, const float *c, const float *a, const void *fwd_scratch,
                                 void *bwd_scratch, float *grad_c, float *grad_a, int B, int T, int H, int hop,
                                 int sample_rate, void *stream)
{
    if (B == 0) return 0;
    if (!grad_y || !f0 || !c || !a || !fwd_scratch || !bwd_scratch || !grad_c || !grad_a || B < 0 || T <= 0 || H <= 0 ||
        hop <= 0 || sample_rate <= 0)
        return DDSP_EINVAL;
    if ((long)T * hop >= (1L << 24) || (long)B * T >= (1L << 31) / 64) return DDSP_ERANGE;
    OscParams p = {};
    if (!setup_params(p, const_cast<void *>(fwd_scratch), B, T, H, hop, sample_rate)) return DDSP_ERANGE;
    p.f0 = f0; p.c = c; p.a = a;
    p.grad_y = grad_y; p.grad_c = grad_c; p.grad_a = grad_a;
    p.part_c = (float *)bwd_scratch;
    p.part_a = (float *)((char *)bwd_scratch + align256((size_t)B * T * 3 * H * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    switch (p.K) {
#define DDSP_CASE(KK) case KK: e = launch_bwd<KK>(p, s); break;
        DDSP_CASE(4) DDSP_CASE(8) DDSP_CASE(12) DDSP_CASE(13) DDSP_CASE(15) DDSP_CASE(16) DDSP_CASE(20) DDSP_CASE(23) DDSP_CASE(25)
#undef DDSP_CASE
        default: return DDSP_ERANGE;
    }
    return (int)e;
}
