// Framing for the multi-scale spectral loss (loss/mss_loss.py:11-33, restated on torch.stft: center=True, reflect padding,
// periodic Hann window, hop = n_fft / 4).  torch.stft spends a reflection-pad launch, a strided window-multiply launch and
// its layout copies per signal and scale, and in the backward a multiply, an index_add (overlap-add) and the padding's
// adjoint; the transform itself is one batched library FFT over contiguous frames.  These two kernels are everything
// around that FFT:
//   frames[b, f, j] = x[b, reflect(f * hop + j - n_fft/2)] * window[j]                        (one pass, output contiguous)
//   grad_x[b, m]    = sum over (f, j) with reflect(f * hop + j - n_fft/2) == m of grad_frames[b, f, j] * window[j]
// The backward is a GATHER: every output sample adds its (at most twelve) contributions in a fixed order -- no atomics,
// deterministic.  HBM-bound: 4 + 16 bytes per input sample forward (75 % overlap), the reverse backward.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ddsp_hip.h"

namespace {

// torch's 'reflect' padding (no repeated edge sample): -1 -> 1, N -> N - 2
__device__ __forceinline__ long reflect_index(long p, long N)
{
    if (p < 0) p = -p;
    if (p >= N) p = 2 * (N - 1) - p;
    return p;
}

__global__ void __launch_bounds__(256) stft_frames_kernel(const float *__restrict__ x, const float *__restrict__ window,
                                                          float *__restrict__ frames, long B, long N, int n_fft, int hop, long F)
{
    extern __shared__ float win_s[];
    for (int j = threadIdx.x; j < n_fft; j += 256) win_s[j] = window[j];
    __syncthreads();
    const long total = B * F * (long)(n_fft / 4);                 // float4 granules
    const int q4 = n_fft / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long bf = e / q4;
        const int j = (int)(e - bf * q4) * 4;
        const long b = bf / F, f = bf - b * F;
        const long p0 = f * hop + j - n_fft / 2;
        const float *row = x + b * N;
        float4 v;
        if (p0 >= 0 && p0 + 3 < N) {                              // interior: one 16-byte load when aligned, four scalars otherwise
            v = make_float4(row[p0], row[p0 + 1], row[p0 + 2], row[p0 + 3]);
        } else {
            v = make_float4(row[reflect_index(p0, N)], row[reflect_index(p0 + 1, N)], row[reflect_index(p0 + 2, N)],
                            row[reflect_index(p0 + 3, N)]);
        }
        v.x *= win_s[j]; v.y *= win_s[j + 1]; v.z *= win_s[j + 2]; v.w *= win_s[j + 3];
        reinterpret_cast<float4 *>(frames)[e] = v;
    }
}

// contributions of padded position p (0 <= p < N + n_fft) to its source sample: frames f with f * hop <= p < f * hop + n_fft
__device__ __forceinline__ float gather_position(const float *__restrict__ g, const float *win_s, long p, int n_fft, int hop, long F)
{
    long f_hi = p / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    long f_lo = (p - n_fft + hop) / hop;                           // ceil((p - n_fft + 1) / hop) for p - n_fft + 1 > 0
    if (p - n_fft + 1 <= 0) f_lo = 0;
    float s = 0.0f;
    for (long f = f_lo; f <= f_hi; ++f) {
        const int j = (int)(p - f * hop);
        if (j >= 0 && j < n_fft) s += g[f * n_fft + j] * win_s[j];
    }
    return s;
}

__global__ void __launch_bounds__(256) stft_frames_bwd_kernel(const float *__restrict__ grad_frames, const float *__restrict__ window,
                                                              float *__restrict__ grad_x, long B, long N, int n_fft, int hop, long F,
                                                              int accumulate)
{
    extern __shared__ float win_s[];
    for (int j = threadIdx.x; j < n_fft; j += 256) win_s[j] = window[j];
    __syncthreads();
    const int half = n_fft / 2;
    const long total = B * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / N, m = e - b * N;
        const float *g = grad_frames + b * F * n_fft;
        // padded positions whose source is m: the direct one, the mirror about the first sample, the mirror about the last
        float s = gather_position(g, win_s, m + half, n_fft, hop, F);
        if (m >= 1 && m <= half) s += gather_position(g, win_s, half - m, n_fft, hop, F);
        const long pm = half + 2 * (N - 1) - m;                   // > half + N - 1 and < N + n_fft  <=>  N - 1 - half <= m <= N - 2
        if (m <= N - 2 && m >= N - 1 - half) s += gather_position(g, win_s, pm, n_fft, hop, F);
        grad_x[e] = accumulate ? grad_x[e] + s : s;
    }
}

unsigned grid_for(long n)
{
    const long want = (n + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want < 8192 ? want : 8192));
}

}  // namespace

extern "C" int ddsp_stft_frames(const float *x, const float *window, float *frames, long B, long N, int n_fft, int hop, void *stream)
{
    if (B == 0) return 0;
    if (!x || !window || !frames || B < 0 || N <= 0 || n_fft <= 0 || hop <= 0) return DDSP_EINVAL;
    if (n_fft % 4 != 0 || n_fft > 8192 || N <= n_fft / 2) return DDSP_ERANGE;       // reflect padding needs N > n_fft / 2
    const long F = 1 + N / hop;
    hipLaunchKernelGGL(stft_frames_kernel, dim3(grid_for(B * F * (long)(n_fft / 4))), dim3(256), sizeof(float) * (size_t)n_fft,
                       (hipStream_t)stream, x, window, frames, B, N, n_fft, hop, F);
    return (int)hipGetLastError();
}

extern "C" int ddsp_stft_frames_backward(const float *grad_frames, const float *window, float *grad_x, long B, long N, int n_fft, int hop,
                                         int accumulate, void *stream)
{
    if (B == 0) return 0;
    if (!grad_frames || !window || !grad_x || B < 0 || N <= 0 || n_fft <= 0 || hop <= 0) return DDSP_EINVAL;
    if (n_fft % 4 != 0 || n_fft > 8192 || N <= n_fft / 2) return DDSP_ERANGE;
    const long F = 1 + N / hop;
    hipLaunchKernelGGL(stft_frames_bwd_kernel, dim3(grid_for(B * N)), dim3(256), sizeof(float) * (size_t)n_fft, (hipStream_t)stream,
                       grad_frames, window, grad_x, B, N, n_fft, hop, F, accumulate);
    return (int)hipGetLastError();
}
