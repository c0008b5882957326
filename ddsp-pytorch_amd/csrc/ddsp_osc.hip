// Harmonic oscillator bank for MI355X (gfx950) -- replaces the torch-op chain of
// model/ddsp/harmonic_oscillator.py:24-75 (prepare_harmonics, rescale, generate_phases,
// generate_signal, forward, live).  Written from the arithmetic spec in SURVEY.md App. A.
//
// Decomposition:
//   osc_prep_kernel    frame rate: w[b,t,k] = fl32(fl32(k*f0*2pi)/sr), amp = masked c / sum      (:26-35)
//   osc_frame_kernel<MODE_TOTALS>  per (b,t): exact fp64 sum of the hop upsampled increments      (:36,:41)
//   osc_scan_kernel    exclusive scan of those totals along t -> phase accumulator at frame start (:41)
//   osc_frame_kernel<MODE_SYNTH>   per (b,t): re-walk the frame sample by sample                  (:41-49)
//
// Work mapping of the two frame kernels: a GROUP of G = 2^logG adjacent lanes owns one (b,t) frame;
// each lane keeps K harmonics (k = j + G*m) entirely in registers -- fp64 accumulator, the two
// bracketing frame-rate increments and amplitudes -- and walks the frame's samples sequentially, so
// the phase recurrence needs NO cross-lane scan; the only cross-lane traffic is a log2(G)-step DPP
// sum per output sample.  64/G frames per wavefront, 256/G per workgroup.
//
// Compile with -ffp-contract=off: every rounding point below is part of the parity contract.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"

namespace {

constexpr float kTwoPi32 = 6.2831854820251465f;     // fl32(2*pi): the modulus the reference uses (:34,:42)
constexpr float kInvTwoPi32 = 0.15915493667125702f; // fl32(1/fl32(2*pi))
constexpr float kRevPerRad = 0.15915494309189535f;  // 1/(2*pi) for v_sin_f32 (argument in revolutions)
constexpr float kFastPhaseLimit = 1.0e7f;           // fast modulo is exact while P/2pi32 < 2^21
constexpr float kRoundMagic = 12582912.0f;          // 1.5*2^23: (x + magic) - magic = rint(x) for |x| < 2^22

struct OscParams {
    const float *f0, *c, *a;
    float *y;
    float *w, *amp;   // scratch [B,T,H]
    double *ph0;      // scratch [B,T,H]: frame totals, then (in place) exclusive scan
    const float *live_in;
    float *live_out;
    float *dbg_phi;
    int *redo_flag;   // scratch: set by the FAST synth kernel when a wavefront needs the EXACT one
    int B, T, H, R;
    int logG;
    int force_exact;
    float scale;      // fl32(1/R): F.interpolate's source-index scale
    float nyquist;    // float(sample_rate // 2)
    float sr;         // float(sample_rate)
};

// ---- cross-lane helpers -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the G = 2^logG lanes of a group (groups are G-aligned).  Result valid in every lane for
// G <= 16 and at least in the group's first lane for G = 32, 64.
__device__ __forceinline__ float group_sum(float v, int logG)
{
    if (logG >= 1) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if (logG >= 2) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if (logG >= 3) v += dpp_mov<0x141>(v);  // row_half_mirror
    if (logG >= 4) v += dpp_mov<0x140>(v);  // row_mirror
    if (logG >= 5) v += __shfl_xor(v, 16);
    if (logG >= 6) v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// torch CPU `%=` on floats (aten remainder): fmod, then shifted into the divisor's sign.
__device__ __forceinline__ float remainder_two_pi(float p)
{
    float r = fmodf(p, kTwoPi32);
    if (r != 0.0f && r < 0.0f) r += kTwoPi32;
    return r;
}

// ---- frame-rate preparation (harmonic_oscillator.py:26-35) ---------------------------------
// One wavefront per (b,t) row.
__global__ void __launch_bounds__(256) osc_prep_kernel(OscParams p)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.redo_flag = 0;
    if (row >= (long)p.B * p.T) return;
    const float f = p.f0[row];
    const float *crow = p.c + row * p.H;
    float s = 0.0f;
    for (int h = lane; h < p.H; h += 64) {
        const float hz = (float)(h + 1) * f;
        s += (hz > p.nyquist) ? 0.0f : crow[h];
    }
    s = wave_sum(s);
    for (int h = lane; h < p.H; h += 64) {
        const float hz = (float)(h + 1) * f;                 // :26-29
        const float a0 = (hz > p.nyquist) ? 0.0f : crow[h];  // :31-32 (strict >, integer Nyquist)
        p.amp[row * p.H + h] = a0 / s;                       // :33 (0/0 = NaN when all masked)
        const float rad = hz * kTwoPi32;                     // :34
        p.w[row * p.H + h] = rad / p.sr;                     // :35 true division
    }
}

// ---- exclusive scan of the frame totals along t ----------------------------------------------
// One workgroup per (b, 64-harmonic tile); kScanWaves wavefronts split the T axis.
constexpr int kScanWaves = 8;
__global__ void __launch_bounds__(64 * kScanWaves) osc_scan_kernel(OscParams p)
{
    __shared__ double part[kScanWaves][64];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int tiles = (p.H + 63) >> 6;
    const int b = blockIdx.x / tiles;
    const int h = (blockIdx.x - b * tiles) * 64 + lane;
    const bool ok = h < p.H;
    const int chunk = (p.T + kScanWaves - 1) / kScanWaves;
    const int t0 = min(wv * chunk, p.T), t1 = min(t0 + chunk, p.T);
    double *col = p.ph0 + (long)b * p.T * p.H + (ok ? h : 0);
    double s = 0.0;
    if (ok)
        for (int t = t0; t < t1; ++t) s += col[(long)t * p.H];
    part[wv][lane] = s;
    __syncthreads();
    double run = 0.0;
    for (int v = 0; v < wv; ++v) run += part[v][lane];
    if (ok)
        for (int t = t0; t < t1; ++t) {
            const double v = col[(long)t * p.H];
            col[(long)t * p.H] = run;
            run += v;
        }
}

// ---- frame kernels -----------------------------------------------------------------------------
enum { MODE_TOTALS = 0, MODE_SYNTH = 1 };
// FAST: production path.  EXACT: bit-exact modulo (libm fmodf), live state and debug outputs; it also
// repairs the frames the FAST synth kernel declined (phases outside the fast modulo's exact range).
enum { VAR_FAST = 0, VAR_EXACT = 1 };

template <int K>
struct FrameState {
    double acc[K];
    float x0[K], x1[K];  // frame-rate increments at the bracketing frames i0, i1
    float a0[K], da[K];  // amplitude at i0 and (amp[i1] - amp[i0])
};

// First sample index n in [0,R) of frame t whose interpolation source index is >= t, i.e. where
// F.interpolate switches from frames (t-1,t) to (t,t+1).  Evaluated with the reference's own fp32
// expression so that it is right for every hop (R/2 for even hops in exact arithmetic).
__device__ __forceinline__ int split_index(int t, int R, float scale)
{
    int m = R >> 1;
    if (t == 0) return m;  // both halves clamp to frame 0: any split gives identical results
    const float tf = (float)t;
    const int base = t * R;
    while (m > 0 && __fmaf_rn(scale, (float)(base + m - 1) + 0.5f, -0.5f) >= tf) --m;
    while (m < R && __fmaf_rn(scale, (float)(base + m) + 0.5f, -0.5f) < tf) ++m;
    return m;
}

template <int K, int MODE>
__device__ __forceinline__ void load_segment(const OscParams &p, FrameState<K> &st, int b, int j, int i0, int i1,
                                             float &L0, float &L1)
{
    const int G = 1 << p.logG;
    const long rowbase = (long)b * p.T;
    const float *w0row = p.w + (rowbase + i0) * p.H;
    const float *w1row = p.w + (rowbase + i1) * p.H;
    const float *a0row = p.amp + (rowbase + i0) * p.H;
    const float *a1row = p.amp + (rowbase + i1) * p.H;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        const bool ok = h < p.H;
        st.x0[m] = ok ? w0row[h] : 0.0f;
        st.x1[m] = ok ? w1row[h] : 0.0f;
        if (MODE == MODE_SYNTH) {
            const float u0 = ok ? a0row[h] : 0.0f;
            const float u1 = ok ? a1row[h] : 0.0f;
            st.a0[m] = u0;
            st.da[m] = u1 - u0;
        }
    }
    L0 = L1 = 0.0f;
    if (MODE == MODE_SYNTH) {
        L0 = p.a[rowbase + i0];
        L1 = p.a[rowbase + i1];
    }
}

// F.interpolate(linear, align_corners=False) weights of output sample i against source frame i0: App. A item 4
__device__ __forceinline__ void upsample_weights(float scale, int i, float i0f, float &w0, float &w1)
{
    float src = __fmaf_rn(scale, (float)i + 0.5f, -0.5f);
    src = fmaxf(src, 0.0f);
    w1 = fminf(fmaxf(src - i0f, 0.0f), 1.0f);
    w0 = 1.0f - w1;
}

// Production walk over samples [n_beg, n_end) of frame t, written stage by stage over the lane's K harmonics
// so that the K independent dependency chains interleave.
template <int K, int MODE>
__device__ __forceinline__ void walk_fast(const OscParams &p, FrameState<K> &st, int b, int t, int j, bool active,
                                          int i0, int i1, int n_beg, int n_end)
{
    float L0, L1;
    load_segment<K, MODE>(p, st, b, j, i0, i1, L0, L1);
    const float i0f = (float)i0;
    float *yrow = p.y + (long)b * p.T * p.R;
    for (int n = n_beg; n < n_end; ++n) {
        const int i = t * p.R + n;
        float w0, w1;
        upsample_weights(p.scale, i, i0f, w0, w1);
        float v[K];
#pragma unroll
        for (int m = 0; m < K; ++m) v[m] = __fmaf_rn(w0, st.x0[m], w1 * st.x1[m]);  // fl32(fma(w0,x[i0],fl32(w1*x[i1])))
#pragma unroll
        for (int m = 0; m < K; ++m) st.acc[m] += (double)v[m];                        // :41 double accumulator
        if (MODE == MODE_SYNTH) {
#pragma unroll
            for (int m = 0; m < K; ++m) v[m] = (float)st.acc[m];                      // ... rounded to fp32 per sample
#pragma unroll
            for (int m = 0; m < K; ++m) {
                // P - q*2pi32 is exact in fp32 for q = rint(P/2pi32 +- 0.25) < 2^21: r in (-4.8, 4.8).  Taking the
                // nearest multiple instead of the floor changes sin(r) by <= |2pi32 - 2pi| = 1.75e-7.
                const float q = __fmaf_rn(v[m], kInvTwoPi32, kRoundMagic) - kRoundMagic;
                v[m] = __fmaf_rn(-q, kTwoPi32, v[m]);                                 // :42
            }
#pragma unroll
            for (int m = 0; m < K; ++m) v[m] = __builtin_amdgcn_sinf(v[m] * kRevPerRad);  // v_sin_f32 (revolutions)
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int m = 0; m < K; ++m) {
                const float A = __fmaf_rn(w1, st.da[m], st.a0[m]);
                if (m & 1) s1 = __fmaf_rn(A, v[m], s1); else s0 = __fmaf_rn(A, v[m], s0);  // :48-49
            }
            const float sum = group_sum(s0 + s1, p.logG);
            const float L = __fmaf_rn(w0, L0, w1 * L1);
            if (j == 0 && active) yrow[i] = L * sum;
        }
    }
}

// Reference-exact walk: libm fmodf modulo, live offsets (:70), live state and debug phase outputs.
template <int K, int MODE>
__device__ __forceinline__ void walk_exact(const OscParams &p, FrameState<K> &st, const float (&lp)[K], int b, int t,
                                           int j, bool active, int i0, int i1, int n_beg, int n_end)
{
    const int G = 1 << p.logG;
    float L0, L1;
    load_segment<K, MODE>(p, st, b, j, i0, i1, L0, L1);
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
                const float r = remainder_two_pi(P);              // :42, exact
                if (p.dbg_phi && active && h < p.H) p.dbg_phi[((long)b * N + i) * p.H + h] = r;
                if (p.live_out && active && b == 0 && i == N - 1 && h < p.H) p.live_out[h] = r;  // :72
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

template <int K, int MODE, int VARIANT>
__global__ void __launch_bounds__(256) osc_frame_kernel(OscParams p)
{
    if (MODE == MODE_SYNTH && VARIANT == VAR_EXACT && !p.force_exact && *p.redo_flag == 0) return;
    const int G = 1 << p.logG;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int j = threadIdx.x & (G - 1);
    long f = gid >> p.logG;
    const long nframes = (long)p.B * p.T;
    const bool active = f < nframes;
    if (!active) f = nframes - 1;  // keep the lanes alive for the cross-lane sums; their stores are masked
    const int b = (int)(f / p.T);
    const int t = (int)(f - (long)b * p.T);
    const int ia = max(t - 1, 0), ib = t, ic = min(t + 1, p.T - 1);

    FrameState<K> st;
    bool fast = true;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        st.acc[m] = (MODE == MODE_SYNTH && h < p.H) ? p.ph0[((long)b * p.T + t) * p.H + h] : 0.0;
    }
    if (MODE == MODE_SYNTH) {
        // The fast modulo needs 0 <= P < kFastPhaseLimit over the whole frame: increments of the three
        // bracketing frames non-negative (phases then grow monotonically) and the end-of-frame bound small.
        const float *wb = p.w + (long)b * p.T * p.H;
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            if (h < p.H) {
                const float xa = wb[(long)ia * p.H + h], xb = wb[(long)ib * p.H + h], xc = wb[(long)ic * p.H + h];
                const float bound = (float)st.acc[m] + (float)p.R * fmaxf(fmaxf(xa, xb), xc) * 1.0001f;
                fast = fast && (xa >= 0.0f) && (xb >= 0.0f) && (xc >= 0.0f) && (st.acc[m] >= 0.0) && (bound < kFastPhaseLimit);
            }
        }
        fast = __all(fast);  // wave-uniform
        if (VARIANT == VAR_FAST && !fast) {
            if ((threadIdx.x & 63) == 0) atomicOr(p.redo_flag, 1);  // the EXACT kernel that follows redoes this wavefront
            return;
        }
        if (VARIANT == VAR_EXACT && fast && !p.force_exact) return;  // already written by the FAST kernel
    }
    const int split = split_index(t, p.R, p.scale);
    if (VARIANT == VAR_FAST) {
        walk_fast<K, MODE>(p, st, b, t, j, active, ia, ib, 0, split);
        walk_fast<K, MODE>(p, st, b, t, j, active, ib, ic, split, p.R);
    } else {
        float lp[K];
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            lp[m] = (h < p.H && b == 0 && p.live_in) ? p.live_in[h] : 0.0f;
        }
        walk_exact<K, MODE>(p, st, lp, b, t, j, active, ia, ib, 0, split);
        walk_exact<K, MODE>(p, st, lp, b, t, j, active, ib, ic, split, p.R);
    }
    if (MODE == MODE_TOTALS && active) {
#pragma unroll
        for (int m = 0; m < K; ++m) {
            const int h = j + m * G;
            if (h < p.H) p.ph0[((long)b * p.T + t) * p.H + h] = st.acc[m];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
struct Tiling { int K, logG; };

// Harmonics per lane K (compile-time, register resident) and lanes per frame G = 2^logG with
// G*K >= H, minimising padded (idle) harmonic slots; ties go to the larger K.
const int kKs[] = {4, 8, 12, 13, 15, 16, 20, 23, 25};

int g_forced_k = 0;  // ddsp_osc_set_tiling: 0 = automatic

bool pick_tiling(int H, Tiling *out)
{
    double best = 1e30;
    bool found = false;
    for (int K : kKs) {
        if (g_forced_k && K != g_forced_k) continue;
        const int lanes = (H + K - 1) / K;
        int logG = 0;
        while ((1 << logG) < lanes) ++logG;
        if (logG > 6) continue;
        const double waste = (double)((1 << logG) * K) / (double)H;
        if (waste <= best + 1e-12) {
            best = waste;
            out->K = K;
            out->logG = logG;
            found = true;
        }
    }
    return found;
}

template <int K>
hipError_t launch_frames(const OscParams &p, hipStream_t s)
{
    const long lanes = ((long)p.B * p.T) << p.logG;
    const unsigned grid = (unsigned)((lanes + 255) / 256);
    const bool live = p.live_in || p.live_out;
    int slot = ddsp_prof::begin(ddsp_prof::TOTALS, s);
    if (p.live_in) {
        hipLaunchKernelGGL((osc_frame_kernel<K, MODE_TOTALS, VAR_EXACT>), dim3(grid), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((osc_frame_kernel<K, MODE_TOTALS, VAR_FAST>), dim3(grid), dim3(256), 0, s, p);
    }
    ddsp_prof::end(slot, s);
    const int tiles = (p.H + 63) / 64;
    slot = ddsp_prof::begin(ddsp_prof::SCAN, s);
    hipLaunchKernelGGL(osc_scan_kernel, dim3((unsigned)(p.B * tiles)), dim3(64 * kScanWaves), 0, s, p);
    ddsp_prof::end(slot, s);
    slot = ddsp_prof::begin(ddsp_prof::SYNTH, s);
    if (!live && !p.force_exact)
        hipLaunchKernelGGL((osc_frame_kernel<K, MODE_SYNTH, VAR_FAST>), dim3(grid), dim3(256), 0, s, p);
    ddsp_prof::end(slot, s);
    // exits at once unless a wavefront of the FAST kernel raised redo_flag (or exactness is forced)
    hipLaunchKernelGGL((osc_frame_kernel<K, MODE_SYNTH, VAR_EXACT>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

extern "C" int ddsp_osc_set_tiling(int harmonics_per_lane)
{
    if (harmonics_per_lane != 0) {
        bool known = false;
        for (int K : kKs) known = known || (K == harmonics_per_lane);
        if (!known) return DDSP_ERANGE;
    }
    g_forced_k = harmonics_per_lane;
    return 0;
}

extern "C" size_t ddsp_osc_scratch_bytes(int B, int T, int H)
{
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    const size_t n = (size_t)B * T * H;
    return 2 * align256(n * sizeof(float)) + align256(n * sizeof(double)) + 256;
}

extern "C" int ddsp_osc_forward(const float *f0, const float *c, const float *a, float *y, void *scratch,
                                const float *live_in, float *live_out, float *dbg_phi, int B, int T, int H, int hop,
                                int sample_rate, void *stream)
{
    if (B == 0) return 0;
    if (!f0 || !c || !a || !y || !scratch || B < 0 || T <= 0 || H <= 0 || hop <= 0 || sample_rate <= 0) return DDSP_EINVAL;
    if (live_in && live_in == live_out) return DDSP_EINVAL;
    if ((long)T * hop >= (1L << 24) || (long)B * T >= (1L << 31) / 64) return DDSP_ERANGE;
    Tiling tl;
    if (!pick_tiling(H, &tl)) return DDSP_ERANGE;

    OscParams p;
    p.f0 = f0; p.c = c; p.a = a; p.y = y;
    const size_t n = (size_t)B * T * H;
    char *base = (char *)scratch;
    p.w = (float *)base;
    p.amp = (float *)(base + align256(n * sizeof(float)));
    p.ph0 = (double *)(base + 2 * align256(n * sizeof(float)));
    p.redo_flag = (int *)(base + 2 * align256(n * sizeof(float)) + align256(n * sizeof(double)));
    p.live_in = live_in; p.live_out = live_out; p.dbg_phi = dbg_phi;
    p.B = B; p.T = T; p.H = H; p.R = hop;
    p.logG = tl.logG;
    p.force_exact = (dbg_phi != nullptr || live_in != nullptr || live_out != nullptr) ? 1 : 0;
    p.scale = (float)(1.0 / (double)hop);
    p.nyquist = (float)(sample_rate / 2);
    p.sr = (float)sample_rate;

    hipStream_t s = (hipStream_t)stream;
    const long rows = (long)B * T;
    const int slot = ddsp_prof::begin(ddsp_prof::PREP, s);
    hipLaunchKernelGGL(osc_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
    ddsp_prof::end(slot, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    switch (tl.K) {
#define DDSP_CASE(KK) case KK: e = launch_frames<KK>(p, s); break;
        DDSP_CASE(4) DDSP_CASE(8) DDSP_CASE(12) DDSP_CASE(13) DDSP_CASE(15) DDSP_CASE(16) DDSP_CASE(20) DDSP_CASE(23) DDSP_CASE(25)
#undef DDSP_CASE
        default: return DDSP_ERANGE;
    }
    return (int)e;
}
