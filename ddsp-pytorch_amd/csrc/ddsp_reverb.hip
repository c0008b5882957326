// Reverb of the DDSP decoder for MI355X (gfx950) -- SURVEY §8f next row 1, reference model/ddsp/reverb.py:8-49.
//
//   build_impulse (:24-29)   impulse[i] = noise[i] * exp(-softplus(-decay) * t[i] * 500) * sigmoid(wet), tap 0 forced to 1
//   forward       (:31-38)   first len(x) samples of x * impulse; the impulse is zero-padded -- or CROPPED (:34, negative pad)
//                            -- to len(x).  The two 2N-point real transforms stay library FFTs (rocFFT, caller side); this
//                            file provides what surrounds them: the impulse written straight into its padded / cropped
//                            buffer (one launch instead of ~8), the spectral product (and, for the backward, the product with
//                            the conjugate kernel fused with the batch-reduced correlation spectrum), and the impulse's
//                            backward (noise, decay, wet) with a deterministic in-workgroup reduction.
//   live_forward  (:40-49)   the reference slides a one-second history, runs the full 2L-point FFT convolution over it and
//                            keeps the last n samples.  Only those n outputs are computed here, directly:
//                                y[j] = sum_{m <= i} impulse[m] * window[i - m],  i = L - n + j,  window = [history[n:], x]
//                            n x L multiply-adds (90 M at the real-time shape: 2048 samples, L = 44100) split over
//                            (output tile x tap chunk) workgroups -> partial sums -> a fixed-order reduction that also writes
//                            the slid history.  The impulse taps are evaluated on the fly while staging (no impulse buffer).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ddsp_hip.h"

namespace {

// ATen CPU: softplus(x) = x > 20 ? x : log1p(exp(x)) (beta 1, threshold 20); sigmoid(x) = 1 / (1 + exp(-x)).
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// One tap with the reference's rounding order (:25-27): ((-softplus(-decay)) * t) * 500 -> exp -> noise * env -> * sigmoid(wet).
__device__ __forceinline__ float tap_envelope(float neg_sp, float t) { return expf((neg_sp * t) * 500.0f); }

__global__ void __launch_bounds__(256) reverb_impulse_kernel(const float *__restrict__ noise, const float *__restrict__ decay,
                                                             const float *__restrict__ wet, const float *__restrict__ t,
                                                             float *__restrict__ impulse, int length, int n_out)
{
    const float neg_sp = -softplusf_(-decay[0]);
    const float sg = sigmoidf_(wet[0]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_out; i += gridDim.x * 256) {
        float v = 0.0f;
        if (i == 0) v = 1.0f;                                     // :28
        else if (i < length) v = (noise[i] * tap_envelope(neg_sp, t[i])) * sg;
        impulse[i] = v;
    }
}

// Backward of build_impulse for the first n_used (<= length) taps (the others were cropped away or do not exist):
//   d noise[i] = g[i] env[i] sg;  d wet = sg (1 - sg) sum g noise env;  d decay = sigmoid(-decay) * 500 sum g noise env sg t
// One workgroup (<= 48000 taps): fp64 partial sums, fixed-order tree -> deterministic.
__global__ void __launch_bounds__(1024) reverb_impulse_bwd_kernel(const float *__restrict__ grad_impulse, const float *__restrict__ noise,
                                                                  const float *__restrict__ decay, const float *__restrict__ wet,
                                                                  const float *__restrict__ t, float *__restrict__ grad_noise,
                                                                  float *__restrict__ grad_decay, float *__restrict__ grad_wet,
                                                                  int length, int n_used)
{
    __shared__ double red[2][1024];
    const float d = decay[0];
    const float neg_sp = -softplusf_(-d);
    const float sg = sigmoidf_(wet[0]);
    double s_wet = 0.0, s_dec = 0.0;
    for (int i = threadIdx.x; i < length; i += 1024) {
        float gn = 0.0f;
        if (i >= 1 && i < n_used) {
            const float g = grad_impulse[i];
            const float env = tap_envelope(neg_sp, t[i]);
            gn = g * env * sg;
            const float gne = g * noise[i] * env;
            s_wet += (double)gne;
            s_dec += (double)(gne * sg) * (double)(t[i] * 500.0f);
        }
        grad_noise[i] = gn;
    }
    red[0][threadIdx.x] = s_wet;
    red[1][threadIdx.x] = s_dec;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        grad_wet[0] = (float)(red[0][0] * (double)(sg * (1.0f - sg)));
        // a = -softplus(-decay): da/d decay = softplus'(-decay) = sigmoid(-decay) (1 above ATen's threshold)
        const float dsp = (-d > 20.0f) ? 1.0f : sigmoidf_(-d);
        grad_decay[0] = (float)(red[1][0] * (double)dsp);
    }
}

// ---- spectral products around the library FFTs ------------------------------------------------------------------
// Y[r,f] = X[r,f] * K[f]   (fft_convolve's `rfft(signal) * rfft(kernel)`, filtered_noise.py:28-30, kernel shared by the rows)
__global__ void __launch_bounds__(256) spectral_mul_kernel(const float2 *__restrict__ X, const float2 *__restrict__ K,
                                                           float2 *__restrict__ Y, long rows, long bins)
{
    const long total = rows * bins;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float2 x = X[i], k = K[i % bins];
        Y[i] = make_float2(x.x * k.x - x.y * k.y, x.x * k.y + x.y * k.x);
    }
}

// Backward of the causal convolution in the frequency domain, one pass over G = rfft(grad_y):
//   GK[r,f] = G[r,f] conj(K[f])            -> irfft -> grad_x
//   S[f]    = sum_r G[r,f] conj(X[r,f])    -> irfft -> grad_impulse  (rows summed in order: deterministic)
__global__ void __launch_bounds__(256) spectral_bwd_kernel(const float2 *__restrict__ G, const float2 *__restrict__ X,
                                                           const float2 *__restrict__ K, float2 *__restrict__ GK,
                                                           float2 *__restrict__ S, long rows, long bins)
{
    for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < bins; f += (long)gridDim.x * 256) {
        const float2 k = K[f];
        float sr = 0.0f, si = 0.0f;
        for (long r = 0; r < rows; ++r) {
            const float2 g = G[r * bins + f];
            if (GK) GK[r * bins + f] = make_float2(g.x * k.x + g.y * k.y, g.y * k.x - g.x * k.y);
            if (S) {
                const float2 x = X[r * bins + f];
                sr += g.x * x.x + g.y * x.y;
                si += g.y * x.x - g.x * x.y;
            }
        }
        if (S) S[f] = make_float2(sr, si);
    }
}

// ---- live_forward ------------------------------------------------------------------------------------------------
constexpr int kTileJ = 256;   // outputs per workgroup (one per thread)
constexpr int kTileM = 512;   // taps per workgroup

// window[p], p in [0, L): the slid history [history[n:], x] (reverb.py:42-44), read without materialising it
__device__ __forceinline__ float window_at(const float *__restrict__ hist, const float *__restrict__ x, int p, int L, int n)
{
    if (p < 0 || p >= L) return 0.0f;
    return p < L - n ? hist[p + n] : x[p - (L - n)];
}

__global__ void __launch_bounds__(kTileJ) reverb_live_partial_kernel(const float *__restrict__ x, const float *__restrict__ hist,
                                                                     const float *__restrict__ noise, const float *__restrict__ decay,
                                                                     const float *__restrict__ wet, const float *__restrict__ t,
                                                                     float *__restrict__ part, int L, int n)
{
    __shared__ float imp_s[kTileM];
    __shared__ float win_s[kTileJ + kTileM];
    const int j0 = blockIdx.x * kTileJ, m0 = blockIdx.y * kTileM;
    const float neg_sp = -softplusf_(-decay[0]);
    const float sg = sigmoidf_(wet[0]);
    for (int mm = threadIdx.x; mm < kTileM; mm += kTileJ) {
        const int m = m0 + mm;
        float v = 0.0f;
        if (m == 0) v = 1.0f;
        else if (m < L) v = (noise[m] * tap_envelope(neg_sp, t[m])) * sg;
        imp_s[mm] = v;
    }
    // outputs i = L - n + j0 + jl, taps m0 + mm  ->  window index (i - m) = base + jl + (kTileM - 1 - mm)
    const int base = (L - n + j0) - (m0 + kTileM - 1);
    for (int q = threadIdx.x; q < kTileJ + kTileM - 1; q += kTileJ) win_s[q] = window_at(hist, x, base + q, L, n);
    __syncthreads();
    const int jl = threadIdx.x;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};                       // four independent chains over the taps
#pragma unroll 4
    for (int mm = 0; mm < kTileM; mm += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(imp_s[mm + u], win_s[jl + (kTileM - 1) - (mm + u)], acc[u]);
    }
    if (j0 + jl < n) part[(size_t)blockIdx.y * n + j0 + jl] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// y[j] = sum over tap chunks (in order); history_out = the slid window (must not alias history_in: every element moves)
__global__ void __launch_bounds__(256) reverb_live_finish_kernel(const float *__restrict__ part, const float *__restrict__ x,
                                                                 const float *__restrict__ hist_in, float *__restrict__ hist_out,
                                                                 float *__restrict__ y, int L, int n, int chunks)
{
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
        float s = 0.0f;
        for (int c = 0; c < chunks; ++c) s += part[(size_t)c * n + j];
        y[j] = s;
    }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < L; p += stride) hist_out[p] = window_at(hist_in, x, p, L, n);
}

unsigned grid_for(long n, long cap = 4096)
{
    const long want = (n + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want < cap ? want : cap));
}

}  // namespace

extern "C" int ddsp_reverb_impulse(const float *noise, const float *decay, const float *wet, const float *t, float *impulse,
                                   int length, int n_out, void *stream)
{
    if (n_out == 0) return 0;
    if (!noise || !decay || !wet || !t || !impulse || length <= 0 || n_out < 0) return DDSP_EINVAL;
    hipLaunchKernelGGL(reverb_impulse_kernel, dim3(grid_for(n_out)), dim3(256), 0, (hipStream_t)stream, noise, decay, wet, t, impulse,
                       length, n_out);
    return (int)hipGetLastError();
}

extern "C" int ddsp_reverb_impulse_backward(const float *grad_impulse, const float *noise, const float *decay, const float *wet,
                                            const float *t, float *grad_noise, float *grad_decay, float *grad_wet, int length,
                                            int n_used, void *stream)
{
    if (!grad_impulse || !noise || !decay || !wet || !t || !grad_noise || !grad_decay || !grad_wet || length <= 0 || n_used < 0)
        return DDSP_EINVAL;
    if (n_used > length) return DDSP_ERANGE;
    hipLaunchKernelGGL(reverb_impulse_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, grad_impulse, noise, decay, wet, t,
                       grad_noise, grad_decay, grad_wet, length, n_used);
    return (int)hipGetLastError();
}

extern "C" int ddsp_spectral_mul(const float *x_ri, const float *k_ri, float *y_ri, long rows, long bins, void *stream)
{
    if (rows == 0 || bins == 0) return 0;
    if (!x_ri || !k_ri || !y_ri || rows < 0 || bins < 0) return DDSP_EINVAL;
    hipLaunchKernelGGL(spectral_mul_kernel, dim3(grid_for(rows * bins, 8192)), dim3(256), 0, (hipStream_t)stream, (const float2 *)x_ri,
                       (const float2 *)k_ri, (float2 *)y_ri, rows, bins);
    return (int)hipGetLastError();
}

extern "C" int ddsp_spectral_mul_backward(const float *g_ri, const float *x_ri, const float *k_ri, float *gk_ri, float *s_ri,
                                          long rows, long bins, void *stream)
{
    if (bins == 0) return 0;
    if (!g_ri || !k_ri || rows < 0 || bins < 0 || (s_ri && !x_ri) || (!gk_ri && !s_ri)) return DDSP_EINVAL;
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3(grid_for(bins)), dim3(256), 0, (hipStream_t)stream, (const float2 *)g_ri,
                       (const float2 *)x_ri, (const float2 *)k_ri, (float2 *)gk_ri, (float2 *)s_ri, rows, bins);
    return (int)hipGetLastError();
}

extern "C" size_t ddsp_reverb_live_scratch_bytes(int length, int n)
{
    if (length <= 0 || n <= 0) return 0;
    return sizeof(float) * (size_t)((length + kTileM - 1) / kTileM) * (size_t)n;
}

extern "C" int ddsp_reverb_live(const float *x, const float *history_in, float *history_out, const float *noise, const float *decay,
                                const float *wet, const float *t, float *y, void *scratch, int length, int n, void *stream)
{
    if (n == 0) return 0;
    if (!x || !history_in || !history_out || !noise || !decay || !wet || !t || !y || !scratch || length <= 0 || n < 0) return DDSP_EINVAL;
    if (history_in == history_out) return DDSP_EINVAL;
    if (n > length) return DDSP_ERANGE;          // the reference's slice assignment fails for a call longer than the history
    const int chunks = (length + kTileM - 1) / kTileM;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(reverb_live_partial_kernel, dim3((unsigned)((n + kTileJ - 1) / kTileJ), (unsigned)chunks), dim3(kTileJ), 0, s,
                       x, history_in, noise, decay, wet, t, (float *)scratch, length, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(reverb_live_finish_kernel, dim3(grid_for(length > n ? length : n, 256)), dim3(256), 0, s,
                       (const float *)scratch, x, history_in, history_out, y, length, n, chunks);
    return (int)hipGetLastError();
}
