// One scale of the multi-scale spectral loss (loss/mss_loss.py:11-33) as ONE kernel from the two waveforms to the loss
// partials and the gradient frames -- framing, the transforms themselves, the power spectra, both L1 terms, d loss / d spectrum
// and the transform back -- instead of torch.stft's pad / frame / window / library FFT / copies on both signals and, for the
// backward, the zero-filled full-spectrum library transform autograd derives for an rfft.
//
// Round 3: WAVEFRONT-PRIVATE transforms (ddsp_wave_fft.h, the organisation of ddsp_noise_fft.hip): a wavefront owns its frames
// from the waveform to the gradient frames and shares nothing, so there is no workgroup barrier (the round-2 Stockham form, a
// 256-thread workgroup per 1024+ points, had 6-18 per transform and lost 26-41 % of its LDS cycles to bank conflicts).
//   * n_fft = 64 ... 1024: frames 2q and 2q+1 of one batch row are packed as one complex sequence z = a w + i b w (reflect
//     padding, window w: torch.stft center=True semantics), likewise the target's; pairing a signal with itself keeps both
//     halves of a packed transform at the same magnitude, and pairing inside a row keeps every row's result independent of the
//     rest of the batch (identical rows give identical spectra: P - Q = 0 exactly).  A wavefront takes 512 / n_fft pairs at a
//     time (one for 1024); per bin k <= n_fft/2: Hermitian split A = (Z_k + conj Z_{n-k}) / 2, B = (Z_k - conj Z_{n-k}) / 2i,
//     the four power values, |P - Q| and |log2(Q + eps) - log2(P + eps)| into the lane's partial sums, G = dloss/dP * 2 * (A, B);
//     the one-sided gradient spectra are extended Hermitian-ly (interior bins halved: the adjoint of an unnormalised rfft),
//     packed as G'_a + i G'_b, and ONE inverse transform per pair gives both frames' gradients.
//   * n_fft = 2048: one real frame per wavefront through a 1024-point complex transform (below).
// The overlap-add / reflect adjoint of the gradient frames is ddsp_stft.hip's gather (deterministic).
// Sums are deterministic: per-wavefront partials in a fixed order, finished in fp64 by ddsp_mss.hip's finish kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_osc_common.h"
#include "ddsp_wave_fft.h"

namespace ddsp_mss {
hipError_t launch_finish(const float *partials, int blocks, float alpha, double inv_n, float *out3, hipStream_t s);   // ddsp_mss.hip
}

namespace {

constexpr int kMaxBlocks = 4096;   // partial sums per scale (the finish kernel's input)

struct MssParams {
    const float *pred, *truth, *window;
    float *grad_frames;            // [B * F, n_fft] or null
    float *partials;               // [grid][2]
    long B, L, F, PR, npairs;      // PR = pairs per batch row = ceil(F / 2)
    int hop;
    float alpha, eps, inv_n;
};

// DDSP_MSS_IEEE (experiments): library log2 and IEEE division instead of v_log_f32 / v_rcp_f32
#ifdef DDSP_MSS_IEEE
__device__ __forceinline__ float fast_log2(float x) { return log2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

__device__ __forceinline__ int reflect_index(int i, int L)
{
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return i;
}

// n_fft = 64 ... 1024: one wavefront owns a unit of PL = max(8, n_fft / 64) points per lane -- 512 / n_fft frame pairs for
// n_fft <= 512, one pair for 1024; radix-R1 / 8 / 8 in registers with two padded LDS exchanges per transform;
// frames are read straight into the transforms' input layout (reflect indexing, window in registers), the spectra are parked in
// natural order in LDS only for the bin loop (a thread owns bins k and n - k), the gradient frames leave through LDS as whole
// 256-byte runs.  Same arithmetic per pair whatever slot it lands in: rows stay independent, identical rows give P - Q = 0.
template <int N>
struct WaveUnit {
    static constexpr int R1 = N / 64;                    // 1, 2, 4, 8, 16
    static constexpr int PL = R1 < 8 ? 8 : R1;           // points per lane
    static constexpr int BT = PL / R1;                   // frame pairs per unit
    static constexpr int STRIDE = N + (R1 < 8 ? 4 * R1 : 0);   // natural-order row of one pair's spectrum (pad: conflict-free stores)
    static constexpr int EXCH = ddsp_wfft::buf_elems<(R1 < 8 ? 8 : R1)>();
    static constexpr int BUF = BT * STRIDE > EXCH ? BT * STRIDE : EXCH;
    static constexpr int BINS = N / 2 + 1;
};

// wavefronts per SIMD the register allocation aims at: up to 256 points four (128 VGPRs; left alone the compiler took 132 and 141
// for 128 and 256 points -- three wavefronts -- which measured 38.5 / 39.2 us against 33.9 / 35.5 us with a handful of spilled
// registers); 512 points three (four would spill 49 registers: no gain), 1024 two (three: 168 VGPRs without a spill but slower)
#ifndef MSS_WAVES
#define MSS_WAVES(n) ((n) <= 256 ? 4 : ((n) <= 512 ? 3 : 2))
#endif
template <int N>
__global__ void __launch_bounds__(64, MSS_WAVES(N)) mss_wave_kernel(MssParams p, long nunits)
{
    using U = WaveUnit<N>;
    using ddsp_wfft::cf;
    constexpr int R1 = U::R1, PL = U::PL, BT = U::BT, STRIDE = U::STRIDE, BINS = U::BINS;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    cf *bufZ = reinterpret_cast<cf *>(smem_f);
    cf *bufW = bufZ + U::BUF;
    const int lane = threadIdx.x;
    const int L = (int)p.L;

    // twiddles and the window values of this lane's input points, for the whole kernel
    ddsp_wfft::Twiddles<(R1 < 8 ? 8 : R1)> tw;           // R1 = 16: the 1024-point transform's; smaller: t2 only
    cf t1s[R1 < 8 ? (R1 > 1 ? R1 : 1) : 1];
    if constexpr (R1 >= 8) {
        ddsp_wfft::make_twiddles<R1>(tw, lane);
    } else {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            float sn, cs;
            sincospif(2.0f * (float)(((lane >> 3) * k2) & 63) / 64.0f, &sn, &cs);
            tw.t2[0][k2] = make_float2(cs, -sn);
        }
#pragma unroll
        for (int k1 = 0; k1 < (R1 > 1 ? R1 : 1); ++k1) {
            float sn, cs;
            sincospif(2.0f * (float)((lane * k1) & (N - 1)) / (float)N, &sn, &cs);
            t1s[k1] = make_float2(cs, -sn);
        }
    }
    float wreg[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) wreg[n1] = p.window[64 * n1 + lane];

    auto forward = [&](cf (&v)[PL], cf *buf) {
        if constexpr (R1 == 16) ddsp_wfft::fft_wave<16, false, false>(v, tw, buf, lane);
        else if constexpr (R1 == 8) ddsp_wfft::fft_wave_batched<8, false>(v, tw.t1, tw.t2[0], buf, lane);
        else ddsp_wfft::fft_wave_batched<R1, false>(v, t1s, tw.t2[0], buf, lane);
    };
    auto inverse = [&](cf (&v)[PL], cf *buf) {
        if constexpr (R1 == 16) ddsp_wfft::fft_wave<16, true, false>(v, tw, buf, lane);
        else if constexpr (R1 == 8) ddsp_wfft::fft_wave_batched<8, true>(v, tw.t1, tw.t2[0], buf, lane);
        else ddsp_wfft::fft_wave_batched<R1, true>(v, t1s, tw.t2[0], buf, lane);
    };
    // result register i of this lane -> natural-order address (pair's row * STRIDE + bin)
    auto natural = [&](int i) {
        if constexpr (R1 == 16) return lane + 64 * (i >> 3) + 128 * (i & 7);          // v[d * 8 + k3] = X[lane + 64 d + 128 k3]
        else {
            const int sq = lane & 7, k2 = lane >> 3;                                   // v[k3] = X_b[k1 + R1 (k2 + 8 k3)], s = b R1 + k1
            return (sq / R1) * STRIDE + (sq % R1) + R1 * (k2 + 8 * i);
        }
    };

    // where a unit's pairs live: lane b computes pair b (the 64-bit divisions happen once), the others read it from there
    struct Slots { long row[BT], frame[BT]; int start[BT], nvalid[BT]; };
    auto get_slots = [&](long unit) {
        long my_row = 0, my_frame = 0;
        int my_start = 0, my_nvalid = 0;
        const long pair = unit * BT + (lane & 7);
        if ((lane & 7) < BT && pair < p.npairs) {
            const long b = pair / p.PR, fa = 2 * (pair - b * p.PR);
            my_row = b * p.L;
            my_frame = b * p.F + fa;
            my_start = (int)(fa * p.hop) - N / 2;
            my_nvalid = (fa + 1 < p.F) ? 2 : 1;
        }
        Slots sl;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            sl.row[b] = ((long)__builtin_amdgcn_readlane((int)(my_row >> 32), b) << 32) | (unsigned)__builtin_amdgcn_readlane((int)my_row, b);
            sl.frame[b] = ((long)__builtin_amdgcn_readlane((int)(my_frame >> 32), b) << 32) | (unsigned)__builtin_amdgcn_readlane((int)my_frame, b);
            sl.start[b] = __builtin_amdgcn_readlane(my_start, b);
            sl.nvalid[b] = __builtin_amdgcn_readlane(my_nvalid, b);
        }
        return sl;
    };
    // frames 2q, 2q + 1 of a row, packed a + i b, straight into the transform's input layout v[b R1 + n1] = z_b[64 n1 + lane]
    // (un-windowed: the window is applied when the values are used, a unit later)
    struct Points { cf v[PL]; };
    auto load = [&](const float *__restrict__ x, const Slots &sl) {
        Points pt;
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const int j = 64 * n1 + lane;
                float re = 0.0f, im = 0.0f;
                if (sl.nvalid[b] > 1 && sl.start[b] >= 0 && sl.start[b] + p.hop + N <= L) {   // wave-uniform: both frames inside the row
                    const float *r = x + sl.row[b] + sl.start[b] + j;
                    re = r[0];
                    im = r[p.hop];
                } else if (sl.nvalid[b] > 0) {                    // the rows' first and last frames: reflect padding
                    const float *r = x + sl.row[b];
                    re = r[reflect_index(sl.start[b] + j, L)];
                    if (sl.nvalid[b] > 1) im = r[reflect_index(sl.start[b] + p.hop + j, L)];
                }
                pt.v[b * R1 + n1] = make_float2(re, im);
            }
        return pt;
    };

    const float inv_ln2 = 1.4426950408889634f;
    float lin = 0.0f, lg = 0.0f;
    // software pipeline over the units: the NEXT unit's frames of both signals are in flight while this unit is transformed
    long unit = blockIdx.x;
    if (unit >= nunits) { if (lane == 0) { p.partials[2 * blockIdx.x] = 0.0f; p.partials[2 * blockIdx.x + 1] = 0.0f; } return; }
    // (1024 points: four sets of sixteen points per lane do not fit the register file beside the transform -- both signals of
    //  the CURRENT unit are read together instead, no look-ahead)
    constexpr bool AHEAD = R1 < 16;
    Slots sl = get_slots(unit);
    Points pp, pq;
    if (AHEAD) { pp = load(p.pred, sl); pq = load(p.truth, sl); }
    for (;;) {
        const long next = unit + gridDim.x;
        Slots sn;
        Points np, nq;
        if (AHEAD) {
            sn = get_slots(next < nunits ? next : unit);
            np = load(p.pred, sn);
            nq = load(p.truth, sn);
        } else {
            pp = load(p.pred, sl);
            pq = load(p.truth, sl);
        }
        const long (&frame)[BT] = sl.frame;
        const int (&nvalid)[BT] = sl.nvalid;
        cf v[PL];
#pragma unroll
        for (int i = 0; i < PL; ++i) v[i] = make_float2(pp.v[i].x * wreg[i % R1], pp.v[i].y * wreg[i % R1]);
        forward(v, bufZ);
#pragma unroll
        for (int i = 0; i < PL; ++i) bufZ[natural(i)] = v[i];
#pragma unroll
        for (int i = 0; i < PL; ++i) v[i] = make_float2(pq.v[i].x * wreg[i % R1], pq.v[i].y * wreg[i % R1]);
        forward(v, bufW);
#pragma unroll
        for (int i = 0; i < PL; ++i) bufW[natural(i)] = v[i];
        DDSP_WAVE_ORDER();

        // split, loss terms, gradient spectrum (in place in bufZ: a lane owns bins k and N - k of its pair)
        for (int t = lane; t < BT * BINS; t += 64) {
            const int b = t / BINS, k = t - b * BINS;
            const int km = (N - k) & (N - 1);
            cf *zrow = bufZ + b * STRIDE;
            const cf *wrow = bufW + b * STRIDE;
            const cf zk = zrow[k], zm = zrow[km], wk = wrow[k], wm = wrow[km];
            // A = (zk + conj zm) / 2, B = -i (zk - conj zm) / 2
            const cf A = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const cf Bq = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const cf C = make_float2(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));
            const cf D = make_float2(0.5f * (wk.y + wm.y), -0.5f * (wk.x - wm.x));
            const float Pa = __fmaf_rn(A.x, A.x, A.y * A.y), Qa = __fmaf_rn(C.x, C.x, C.y * C.y);
            const float Pb = __fmaf_rn(Bq.x, Bq.x, Bq.y * Bq.y), Qb = __fmaf_rn(D.x, D.x, D.y * D.y);
            const float da = Pa - Qa, db = Pb - Qb;
            // (P + eps >= eps > 0 is a normal number: the hardware log2 and reciprocal need no denormal handling; v_rcp_f32 is within
            //  1 ulp, far inside the gradient's tolerance)
            const float ea = fast_log2(Qa + p.eps) - fast_log2(Pa + p.eps);
            const float eb = fast_log2(Qb + p.eps) - fast_log2(Pb + p.eps);
            lin += fabsf(da) + fabsf(db);                         // a frame past the end is all zero: P = Q = 0, both terms vanish
            lg += fabsf(ea) + fabsf(eb);
            if (p.grad_frames) {
                const float ca = 2.0f * p.inv_n * (sgn(da) - p.alpha * sgn(ea) * inv_ln2 * fast_rcp(Pa + p.eps));
                const float cb = 2.0f * p.inv_n * (sgn(db) - p.alpha * sgn(eb) * inv_ln2 * fast_rcp(Pb + p.eps));
                const cf GA = make_float2(ca * A.x, ca * A.y), GB = make_float2(cb * Bq.x, cb * Bq.y);
                if (k == 0 || k == N / 2) {
                    zrow[k] = make_float2(GA.x, GB.x);
                } else {
                    zrow[k] = make_float2(0.5f * (GA.x - GB.y), 0.5f * (GA.y + GB.x));
                    zrow[km] = make_float2(0.5f * (GA.x + GB.y), 0.5f * (GB.x - GA.y));
                }
            }
        }
        DDSP_WAVE_ORDER();
        if (p.grad_frames) {
#pragma unroll
            for (int b = 0; b < BT; ++b)
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) v[b * R1 + n1] = bufZ[b * STRIDE + 64 * n1 + lane];
            DDSP_WAVE_ORDER();
            inverse(v, bufW);
#pragma unroll
            for (int i = 0; i < PL; ++i) bufW[natural(i)] = v[i];
            DDSP_WAVE_ORDER();
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                if (nvalid[b] > 0) {                              // wave-uniform
                    float *dst = p.grad_frames + frame[b] * N;
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const int j = 64 * n1 + lane;
                        const cf y = bufW[b * STRIDE + j];
                        dst[j] = y.x;
                        if (nvalid[b] > 1) dst[N + j] = y.y;
                    }
                }
            }
            DDSP_WAVE_ORDER();
        }
        if (next >= nunits) break;
        unit = next;
        if (AHEAD) { sl = sn; pp = np; pq = nq; }
        else sl = get_slots(unit);
    }
    lin = ddsp_osc::wave_sum(lin);
    lg = ddsp_osc::wave_sum(lg);
    if (lane == 0) {
        p.partials[2 * blockIdx.x] = lin;
        p.partials[2 * blockIdx.x + 1] = lg;
    }
}

// n_fft = 2048: one REAL frame per wavefront through a 1024-point complex transform (2048 complex points of a frame pair
// exceed a wavefront's registers): z[m] = x[2m] w[2m] + i x[2m+1] w[2m+1], Z = FFT_1024(z),
//   Fe = (Z[k] + conj Z[M-k]) / 2,  Fo = -i (Z[k] - conj Z[M-k]) / 2,  T = W_2048^k Fo:   X[k] = Fe + T,  X[M-k] = conj(Fe - T)
// (a lane owns the bins k and M - k, k = 0 .. 512; k = 0 yields the real bins 0 and M = 1024).  Gradient: with the one-sided
// G'[k] = h c_k X[k] (h = 1/2 inside, 1 at bins 0 and M: the adjoint of an unnormalised rfft) the frame's gradient is the
// complex-to-real transform g[2m] + i g[2m+1] = IFFT_1024(Y)[m],
//   S = G'[k] + conj G'[M-k],  D = G'[k] - conj G'[M-k]:   Y[k] = S + i conj(W^k) D,   Y[M-k] = conj(S) + i W^k conj(D).
__global__ void __launch_bounds__(64) mss_wave2048_kernel(MssParams p, long nunits)
{
    using ddsp_wfft::cf;
    constexpr int N = 2048, M = 1024, R1 = 16;
    constexpr int NB = ddsp_wfft::buf_elems<R1>();
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    cf *bufZ = reinterpret_cast<cf *>(smem_f);
    cf *bufW = bufZ + NB;
    const int lane = threadIdx.x;
    const int L = (int)p.L;
    ddsp_wfft::Twiddles<R1> tw;
    ddsp_wfft::make_twiddles<R1>(tw, lane);
    cf wbase;                                             // W_2048^lane
    {
        float sn, cs;
        sincospif(2.0f * (float)lane / (float)N, &sn, &cs);
        wbase = make_float2(cs, -sn);
    }
    float2 wreg[R1];                                      // window at the lane's sample pairs
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) wreg[n1] = reinterpret_cast<const float2 *>(p.window)[64 * n1 + lane];

    auto load = [&](const float *__restrict__ x, long row, int start, cf (&v)[R1]) {
        if (start >= 0 && start + N <= L) {               // wave-uniform: the frame lies inside the row
            const float *r = x + row + start + 2 * lane;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1] = make_float2(r[128 * n1] * wreg[n1].x, r[128 * n1 + 1] * wreg[n1].y);
        } else {
            const float *r = x + row;
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const int i0 = start + 2 * (64 * n1 + lane);
                v[n1] = make_float2(r[reflect_index(i0, L)] * wreg[n1].x, r[reflect_index(i0 + 1, L)] * wreg[n1].y);
            }
        }
    };

    const float inv_ln2 = 1.4426950408889634f;
    float lin = 0.0f, lg = 0.0f;
    for (long unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const long b = unit / p.F, fr = unit - b * p.F;
        const long row = b * p.L;
        const int start = (int)(fr * p.hop) - N / 2;
        cf v[R1];
        load(p.pred, row, start, v);
        ddsp_wfft::fft_wave<R1, false, false>(v, tw, bufZ, lane);
        ddsp_wfft::store_natural<R1>(v, bufZ, lane);
        load(p.truth, row, start, v);
        ddsp_wfft::fft_wave<R1, false, false>(v, tw, bufW, lane);
        ddsp_wfft::store_natural<R1>(v, bufW, lane);
        DDSP_WAVE_ORDER();

        // W_2048^k, k = lane + 64 it, = W_2048^lane * W_32^it: one product with an exact-to-the-ulp constant per trip (advancing a
        // running twiddle by W_32 eight times costs 5e-7 of relative accuracy, which near-empty bins amplify a thousandfold)
        constexpr float c32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                                  0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f};
        constexpr float s32[9] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                                  0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f};
#pragma unroll 1
        for (int it = 0; it < 9; ++it) {
            const int k = lane + 64 * it;
            // (wx + i wy)(c - i s)
            const cf wk = make_float2(__fmaf_rn(wbase.x, c32[it], wbase.y * s32[it]), __fmaf_rn(wbase.y, c32[it], -(wbase.x * s32[it])));
            if (k <= M / 2) {
                const int km = (M - k) & (M - 1);
                const cf zk = bufZ[k], zm = bufZ[km], qk = bufW[k], qm = bufW[km];
                const cf Fe = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)), Fo = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
                const cf Ge = make_float2(0.5f * (qk.x + qm.x), 0.5f * (qk.y - qm.y)), Go = make_float2(0.5f * (qk.y + qm.y), -0.5f * (qk.x - qm.x));
                const cf T = make_float2(__fmaf_rn(wk.x, Fo.x, -(wk.y * Fo.y)), __fmaf_rn(wk.x, Fo.y, wk.y * Fo.x));
                const cf Tq = make_float2(__fmaf_rn(wk.x, Go.x, -(wk.y * Go.y)), __fmaf_rn(wk.x, Go.y, wk.y * Go.x));
                const cf X1 = make_float2(Fe.x + T.x, Fe.y + T.y), X2 = make_float2(Fe.x - T.x, -(Fe.y - T.y));      // bins k, M - k
                const cf Q1 = make_float2(Ge.x + Tq.x, Ge.y + Tq.y), Q2 = make_float2(Ge.x - Tq.x, -(Ge.y - Tq.y));
                const bool twice = k != M / 2;            // k = 512 is its own partner: one bin
                const float P1 = __fmaf_rn(X1.x, X1.x, X1.y * X1.y), P2 = __fmaf_rn(X2.x, X2.x, X2.y * X2.y);
                const float R1q = __fmaf_rn(Q1.x, Q1.x, Q1.y * Q1.y), R2q = __fmaf_rn(Q2.x, Q2.x, Q2.y * Q2.y);
                const float d1 = P1 - R1q, d2 = P2 - R2q;
                const float e1 = fast_log2(R1q + p.eps) - fast_log2(P1 + p.eps);
                const float e2 = fast_log2(R2q + p.eps) - fast_log2(P2 + p.eps);
                lin += fabsf(d1) + (twice ? fabsf(d2) : 0.0f);
                lg += fabsf(e1) + (twice ? fabsf(e2) : 0.0f);
                if (p.grad_frames) {
                    const float h = (k == 0) ? 1.0f : 0.5f;       // bins 0 and M are real and not halved
                    const float c1 = h * 2.0f * p.inv_n * (sgn(d1) - p.alpha * sgn(e1) * inv_ln2 * fast_rcp(P1 + p.eps));
                    const float c2 = h * 2.0f * p.inv_n * (sgn(d2) - p.alpha * sgn(e2) * inv_ln2 * fast_rcp(P2 + p.eps));
                    const cf G1 = make_float2(c1 * X1.x, c1 * X1.y), G2 = make_float2(c2 * X2.x, c2 * X2.y);         // G'[k], G'[M - k]
                    const cf S = make_float2(G1.x + G2.x, G1.y - G2.y), D = make_float2(G1.x - G2.x, G1.y + G2.y);
                    // i conj(w) D = i (wx + i*(-wy))... with w = (wx, wy): conj(w) D = (wx Dx + wy Dy, wx Dy - wy Dx)
                    const cf cD = make_float2(__fmaf_rn(wk.x, D.x, wk.y * D.y), __fmaf_rn(wk.x, D.y, -(wk.y * D.x)));
                    bufZ[k] = make_float2(S.x - cD.y, S.y + cD.x);                                                   // S + i conj(w) D
                    if (k != 0 && twice) {
                        // w conj(D) = (wx Dx + wy Dy, wy Dx - wx Dy)
                        const cf wD = make_float2(__fmaf_rn(wk.x, D.x, wk.y * D.y), __fmaf_rn(wk.y, D.x, -(wk.x * D.y)));
                        bufZ[km] = make_float2(S.x - wD.y, -S.y + wD.x);                                             // conj(S) + i w conj(D)
                    }
                }
            }
        }
        DDSP_WAVE_ORDER();
        if (p.grad_frames) {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1] = bufZ[64 * n1 + lane];
            DDSP_WAVE_ORDER();
            ddsp_wfft::fft_wave<R1, true, false>(v, tw, bufW, lane);
            ddsp_wfft::store_natural<R1>(v, bufW, lane);
            DDSP_WAVE_ORDER();
            float2 *dst = reinterpret_cast<float2 *>(p.grad_frames + unit * N);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) dst[64 * n1 + lane] = bufW[64 * n1 + lane];                            // (g[2m], g[2m+1]) pairs: natural order
            DDSP_WAVE_ORDER();
        }
    }
    lin = ddsp_osc::wave_sum(lin);
    lg = ddsp_osc::wave_sum(lg);
    if (lane == 0) {
        p.partials[2 * blockIdx.x] = lin;
        p.partials[2 * blockIdx.x + 1] = lg;
    }
}

hipError_t launch_wave2048(const MssParams &p0, hipStream_t s, int *blocks_out)
{
    const size_t lds = sizeof(float2) * 2 * ddsp_wfft::buf_elems<16>();
    MssParams p = p0;
    const long nunits = p.B * p.F;
    const int blocks = (int)(nunits < kMaxBlocks ? nunits : kMaxBlocks);
    *blocks_out = blocks;
    hipLaunchKernelGGL(mss_wave2048_kernel, dim3((unsigned)blocks), dim3(64), lds, s, p, nunits);
    return hipGetLastError();
}

template <int N>
hipError_t launch_wave(const MssParams &p0, hipStream_t s, int *blocks_out)
{
    using U = WaveUnit<N>;
    const size_t lds = sizeof(float2) * 2 * U::BUF;
    MssParams p = p0;
    const long nunits = (p.npairs + U::BT - 1) / U::BT;
    const int blocks = (int)(nunits < kMaxBlocks ? nunits : kMaxBlocks);
    *blocks_out = blocks;
    hipLaunchKernelGGL((mss_wave_kernel<N>), dim3((unsigned)blocks), dim3(64), lds, s, p, nunits);
    return hipGetLastError();
}

}  // namespace

extern "C" size_t ddsp_mss_scale_scratch_bytes(void) { return sizeof(float) * 2 * kMaxBlocks; }

extern "C" int ddsp_mss_scale_supported(int n_fft) { return (n_fft >= 64 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0) ? 1 : 0; }

extern "C" int ddsp_mss_scale(const float *x_pred, const float *x_true, const float *window, float *grad_frames, void *scratch, float *out3,
                              long B, long L, int n_fft, int hop, float alpha, float eps, void *stream)
{
    if (!(eps >= 1.1754944e-38f) || !out3) return DDSP_EINVAL;      // (a normal number: the kernels take hardware log2 / reciprocals of P + eps)
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(out3, 0, 3 * sizeof(float), s);   // an empty shard: the mean over no bins is reported as 0
    if (!x_pred || !x_true || !window || !scratch || B < 0 || hop <= 0) return DDSP_EINVAL;
    if (!ddsp_mss_scale_supported(n_fft) || L <= n_fft / 2 || L > (1l << 30) || (L / hop + 2) * (long)hop > (1l << 30)) return DDSP_ERANGE;   // 32-bit positions inside a row
    MssParams p;
    p.pred = x_pred; p.truth = x_true; p.window = window; p.grad_frames = grad_frames; p.partials = (float *)scratch;
    p.B = B; p.L = L; p.F = 1 + L / hop; p.hop = hop;
    const long nframes = B * p.F;
    p.PR = (p.F + 1) / 2;
    p.npairs = B * p.PR;
    const double count = (double)nframes * (double)(n_fft / 2 + 1);
    p.alpha = alpha; p.eps = eps; p.inv_n = (float)(1.0 / count);
    int blocks = 0;
    hipError_t e;
    switch (n_fft) {
    case 64: e = launch_wave<64>(p, s, &blocks); break;
    case 128: e = launch_wave<128>(p, s, &blocks); break;
    case 256: e = launch_wave<256>(p, s, &blocks); break;
    case 512: e = launch_wave<512>(p, s, &blocks); break;
    case 1024: e = launch_wave<1024>(p, s, &blocks); break;
    default: e = launch_wave2048(p, s, &blocks); break;
    }
    if (e != hipSuccess) return (int)e;
    return (int)ddsp_mss::launch_finish(p.partials, blocks, alpha, 1.0 / count, out3, s);
}
