// One scale of the multi-scale spectral loss (loss/mss_loss.py:11-33) as ONE kernel from the two waveforms to the loss
// partials and the gradient frames -- framing, the transforms themselves, the power spectra, both L1 terms, d loss / d spectrum
// and the transform back -- instead of torch.stft's pad / frame / window / library FFT / copies on both signals and, for the
// backward, the zero-filled full-spectrum library transform autograd derives for an rfft.
//
// A workgroup takes PTS = max(n_fft, 1024) complex points at a time = PTS / n_fft frame PAIRS; everything stays in LDS:
//   1. frames 2q and 2q+1 of one batch row of the prediction are packed as one complex sequence z = a w + i b w (reflect
//      padding, window w: torch.stft center=True semantics), likewise the target's frames; pairing a signal with itself
//      keeps both halves of a packed transform at the same magnitude, and pairing inside a row keeps every row's result
//      independent of the rest of the batch (identical rows give identical spectra: P - Q = 0 exactly, sign 0)
//   2. two n_fft-point complex FFTs per pair (Stockham autosort: one radix-2 pass when log2 n_fft is odd, then radix-4
//      passes, natural order in and out, twiddles from an LDS table)
//   3. per bin k <= n_fft/2: Hermitian split A = (Z_k + conj Z_{n-k}) / 2, B = (Z_k - conj Z_{n-k}) / 2i, the four power
//      values, |P - Q| and |log2(Q + eps) - log2(P + eps)| into the lane's partial sums, and G = dloss/dP * 2 * (A, B)
//   4. (gradient wanted) the one-sided gradient spectra are extended Hermitian-ly (interior bins halved: the adjoint of
//      an unnormalised rfft), packed as G'_a + i G'_b, and ONE inverse FFT per pair gives both frames' gradients
// The overlap-add / reflect adjoint of the gradient frames is ddsp_stft.hip's gather (deterministic).
// Sums are deterministic: per-workgroup partials in a fixed order, finished in fp64 by ddsp_mss.hip's finish kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_osc_common.h"

namespace ddsp_mss {
hipError_t launch_finish(const float *partials, int blocks, float alpha, double inv_n, float *out3, hipStream_t s);   // ddsp_mss.hip
}

namespace {

typedef float2 cf;
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;   // partial sums per scale (the finish kernel's input)

struct MssParams {
    const float *pred, *truth, *window;
    float *grad_frames;            // [B * F, n_fft] or null
    float *partials;               // [grid][2]
    long B, L, F, PR, npairs;      // PR = pairs per batch row = ceil(F / 2)
    int hop;
    float alpha, eps, inv_n;
};

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
template <bool INV>
__device__ __forceinline__ cf cmulw(cf a, cf w)   // a * w (forward) / a * conj(w) (inverse); w = e^{-i theta}
{
    if (!INV) return make_float2(__fmaf_rn(a.x, w.x, -(a.y * w.y)), __fmaf_rn(a.x, w.y, a.y * w.x));
    return make_float2(__fmaf_rn(a.x, w.x, a.y * w.y), __fmaf_rn(a.y, w.x, -(a.x * w.y)));
}
template <bool INV>
__device__ __forceinline__ cf rot90(cf a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

// Batched n-point FFTs over the PTS points of `src` (PTS / N independent sequences, each contiguous), through `dst` and back:
// on return `src` points at the result and `dst` at the other buffer.  Ends with a barrier.
template <int N, int PTS, bool INV>
__device__ __forceinline__ void fft_lds(cf *&src, cf *&dst, const cf *__restrict__ tw, int tid)
{
    constexpr int LOG = ilog2(N);
    constexpr int P0 = (LOG & 1) ? 2 : 1;
    if constexpr (LOG & 1) {
        // radix-2, p = 1 (no twiddles): y[2i] = x[i] + x[i + N/2], y[2i + 1] = x[i] - x[i + N/2]
#pragma unroll
        for (int e = 0; e < PTS / 2 / kThreads; ++e) {
            const int i = tid + kThreads * e;
            const int slot = i / (N / 2), il = i & (N / 2 - 1);
            const cf u0 = src[slot * N + il], u1 = src[slot * N + il + N / 2];
            dst[slot * N + 2 * il] = cadd(u0, u1);
            dst[slot * N + 2 * il + 1] = csub(u0, u1);
        }
        __syncthreads();
        cf *t = src; src = dst; dst = t;
    }
#pragma unroll
    for (int q = 0; q < LOG / 2; ++q) {
        const int p = P0 << (2 * q);
        const int stride = N / (4 * p);            // twiddle of (m, k): W_N^(m k stride) = W_{4p}^(m k)
#pragma unroll
        for (int e = 0; e < PTS / 4 / kThreads; ++e) {
            const int i = tid + kThreads * e;
            const int slot = i / (N / 4), il = i & (N / 4 - 1);
            const int k = il & (p - 1), j = ((il - k) << 2) + k;
            const cf *s = src + slot * N + il;
            cf u0 = s[0], u1 = s[N / 4], u2 = s[N / 2], u3 = s[3 * N / 4];
            if (p > 1) {
                u1 = cmulw<INV>(u1, tw[k * stride]);
                u2 = cmulw<INV>(u2, tw[2 * k * stride]);
                u3 = cmulw<INV>(u3, tw[3 * k * stride]);
            }
            const cf s0 = cadd(u0, u2), s1 = cadd(u1, u3), d0 = csub(u0, u2), d1 = rot90<INV>(csub(u1, u3));
            cf *d = dst + slot * N + j;
            d[0] = cadd(s0, s1);
            d[p] = cadd(d0, d1);
            d[2 * p] = csub(s0, s1);
            d[3 * p] = csub(d0, d1);
        }
        __syncthreads();
        cf *t = src; src = dst; dst = t;
    }
}

__device__ __forceinline__ int reflect_index(int i, int L)
{
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return i;
}

// Where a slot's frame pair lives: filled once per workgroup iteration by the first SLOTS threads (the 64-bit divisions happen
// there, not per point).
struct SlotInfo {
    long row;      // b * L: offset of the batch row in x
    long frame;    // b * F + fa: flat index of the pair's first frame (gradient frames)
    int start;     // fa * hop - N/2: position of point 0 of frame a in the row (frame b: + hop)
    int nvalid;    // 0: no pair (past the last row), 1: frame a only (odd frame count), 2: both
};

template <int N, int PTS>
__device__ __forceinline__ void fill_slots(SlotInfo *slots, const MssParams &p, long pair0, int tid)
{
    if (tid < PTS / N) {
        const long pair = pair0 + tid;
        SlotInfo si;
        si.row = 0; si.frame = 0; si.start = 0; si.nvalid = 0;
        if (pair < p.npairs) {
            const long b = pair / p.PR, fa = 2 * (pair - b * p.PR);
            si.row = b * p.L;
            si.frame = b * p.F + fa;
            si.start = (int)(fa * p.hop) - N / 2;
            si.nvalid = (fa + 1 < p.F) ? 2 : 1;
        }
        slots[tid] = si;
    }
}

// z[slot * N + j] = x[b, frame 2q](j) w[j] + i x[b, frame 2q + 1](j) w[j]; a missing frame is zero
template <int N, int PTS>
__device__ __forceinline__ void load_pairs(cf *z, const float *__restrict__ x, const float *__restrict__ win_s, const SlotInfo *slots,
                                           const MssParams &p, int tid)
{
    const int L = (int)p.L;
#pragma unroll
    for (int e = 0; e < PTS / kThreads; ++e) {
        const int pt = tid + kThreads * e;
        const int slot = pt / N, j = pt & (N - 1);
        const SlotInfo si = slots[slot];
        float re = 0.0f, im = 0.0f;
        if (si.nvalid > 0) {
            const float *row = x + si.row;
            re = row[reflect_index(si.start + j, L)];
            if (si.nvalid > 1) im = row[reflect_index(si.start + p.hop + j, L)];
        }
        const float w = win_s[j];
        z[pt] = make_float2(re * w, im * w);
    }
}

template <int N>
__global__ void __launch_bounds__(kThreads) mss_scale_kernel(MssParams p)
{
    constexpr int PTS = N > 1024 ? N : 1024;
    constexpr int SLOTS = PTS / N;
    constexpr int BINS = N / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    cf *X = reinterpret_cast<cf *>(smem_f);
    cf *Y = X + PTS;
    cf *Zb = Y + PTS;
    cf *tw = Zb + PTS;                         // [N] e^{-2 pi i q / N}
    float *win_s = reinterpret_cast<float *>(tw + N);   // [N]
    SlotInfo *slots = reinterpret_cast<SlotInfo *>(win_s + N);
    const int tid = threadIdx.x;
    for (int q = tid; q < N; q += kThreads) {
        float s, c;
        sincospif(2.0f * (float)q / (float)N, &s, &c);
        tw[q] = make_float2(c, -s);
        win_s[q] = p.window[q];
    }
    __syncthreads();

    const float inv_ln2 = 1.4426950408889634f;
    float lin = 0.0f, lg = 0.0f;
    for (long pair0 = (long)blockIdx.x * SLOTS; pair0 < p.npairs; pair0 += (long)gridDim.x * SLOTS) {
        fill_slots<N, PTS>(slots, p, pair0, tid);
        __syncthreads();
        load_pairs<N, PTS>(X, p.pred, win_s, slots, p, tid);
        load_pairs<N, PTS>(Y, p.truth, win_s, slots, p, tid);
        __syncthreads();
        cf *a = X, *b = Zb, *c = Y;
        fft_lds<N, PTS, false>(a, b, tw, tid);      // prediction's packed spectra -> a
        fft_lds<N, PTS, false>(c, b, tw, tid);      // target's -> c

        // split, loss terms, gradient spectrum (in place in a: a thread owns bins k and N - k of its slot)
        for (int t = tid; t < SLOTS * BINS; t += kThreads) {
            const int slot = t / BINS, k = t - slot * BINS;
            const int km = (N - k) & (N - 1);
            const cf zk = a[slot * N + k], zm = a[slot * N + km], wk = c[slot * N + k], wm = c[slot * N + km];
            // A = (zk + conj zm) / 2, B = -i (zk - conj zm) / 2
            const cf A = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            const cf Bq = make_float2(0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x));
            const cf C = make_float2(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));
            const cf D = make_float2(0.5f * (wk.y + wm.y), -0.5f * (wk.x - wm.x));
            const float Pa = __fmaf_rn(A.x, A.x, A.y * A.y), Qa = __fmaf_rn(C.x, C.x, C.y * C.y);
            const float Pb = __fmaf_rn(Bq.x, Bq.x, Bq.y * Bq.y), Qb = __fmaf_rn(D.x, D.x, D.y * D.y);
            const float da = Pa - Qa, db = Pb - Qb;
            const float ea = log2f(Qa + p.eps) - log2f(Pa + p.eps), eb = log2f(Qb + p.eps) - log2f(Pb + p.eps);
            // a frame past the end is all zero: P = Q = 0, both terms vanish exactly
            lin += fabsf(da) + fabsf(db);
            lg += fabsf(ea) + fabsf(eb);
            if (p.grad_frames) {
                const float ca = 2.0f * p.inv_n * (sgn(da) - p.alpha * sgn(ea) * inv_ln2 / (Pa + p.eps));
                const float cb = 2.0f * p.inv_n * (sgn(db) - p.alpha * sgn(eb) * inv_ln2 / (Pb + p.eps));
                const cf GA = make_float2(ca * A.x, ca * A.y), GB = make_float2(cb * Bq.x, cb * Bq.y);
                if (k == 0 || k == N / 2) {
                    a[slot * N + k] = make_float2(GA.x, GB.x);
                } else {
                    a[slot * N + k] = make_float2(0.5f * (GA.x - GB.y), 0.5f * (GA.y + GB.x));
                    a[slot * N + km] = make_float2(0.5f * (GA.x + GB.y), 0.5f * (GB.x - GA.y));
                }
            }
        }
        __syncthreads();
        if (p.grad_frames) {
            fft_lds<N, PTS, true>(a, b, tw, tid);
#pragma unroll
            for (int e = 0; e < PTS / kThreads; ++e) {
                const int pt = tid + kThreads * e;
                const int slot = pt / N, j = pt & (N - 1);
                const SlotInfo si = slots[slot];
                if (si.nvalid > 0) {
                    const cf y = a[pt];
                    float *dst = p.grad_frames + si.frame * N + j;
                    dst[0] = y.x;
                    if (si.nvalid > 1) dst[N] = y.y;
                }
            }
            __syncthreads();
        }
    }
    lin = ddsp_osc::wave_sum(lin);
    lg = ddsp_osc::wave_sum(lg);
    __shared__ float red[2][kThreads / 64];
    if ((tid & 63) == 0) { red[0][tid >> 6] = lin; red[1][tid >> 6] = lg; }
    __syncthreads();
    if (tid == 0) {
        p.partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p.partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <int N>
hipError_t launch(const MssParams &p0, hipStream_t s, int *blocks_out)
{
    constexpr int PTS = N > 1024 ? N : 1024;
    constexpr int SLOTS = PTS / N;
    const size_t lds = sizeof(float2) * (3 * PTS + N) + sizeof(float) * N + sizeof(SlotInfo) * SLOTS;
    if (lds > 64 * 1024) {          // n_fft = 2048: 72 KiB of dynamic LDS needs the opt-in, once per device (the kernel also holds a
                                    // few static words, so the ceiling asked for is what it uses, not the whole 160 KiB)
        static bool raised[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (!raised[dev & 63]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mss_scale_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            raised[dev & 63] = true;
        }
    }
    MssParams p = p0;
    const long groups = (p.npairs + SLOTS - 1) / SLOTS;
    const int blocks = (int)(groups < kMaxBlocks ? groups : kMaxBlocks);
    *blocks_out = blocks;
    hipLaunchKernelGGL((mss_scale_kernel<N>), dim3((unsigned)blocks), dim3(kThreads), lds, s, p);
    return hipGetLastError();
}

}  // namespace

extern "C" size_t ddsp_mss_scale_scratch_bytes(void) { return sizeof(float) * 2 * kMaxBlocks; }

extern "C" int ddsp_mss_scale_supported(int n_fft) { return (n_fft >= 64 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0) ? 1 : 0; }

extern "C" int ddsp_mss_scale(const float *x_pred, const float *x_true, const float *window, float *grad_frames, void *scratch, float *out3,
                              long B, long L, int n_fft, int hop, float alpha, float eps, void *stream)
{
    if (!(eps > 0.0f) || !out3) return DDSP_EINVAL;
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
    case 64: e = launch<64>(p, s, &blocks); break;
    case 128: e = launch<128>(p, s, &blocks); break;
    case 256: e = launch<256>(p, s, &blocks); break;
    case 512: e = launch<512>(p, s, &blocks); break;
    case 1024: e = launch<1024>(p, s, &blocks); break;
    default: e = launch<2048>(p, s, &blocks); break;
    }
    if (e != hipSuccess) return (int)e;
    return (int)ddsp_mss::launch_finish(p.partials, blocks, alpha, 1.0 / count, out3, s);
}
