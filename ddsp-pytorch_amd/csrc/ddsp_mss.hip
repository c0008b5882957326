// One scale of the multi-scale spectral loss (loss/mss_loss.py:11-33) as ONE pass over the two
// complex STFTs instead of ~17 elementwise / reduction launches forward and ~20 backward per scale:
//   P = |S_pred|^2, Q = |S_true|^2,  loss = mean|P - Q| + alpha * mean|log2(Q + eps) - log2(P + eps)|   (:27-31)
// and, in the same pass, d loss / d S_pred (as re/im pairs) for the backward.  HBM-bound: 16 B read + 8 B written per
// bin; the STFTs themselves stay rocFFT (torch.stft).  Sums are deterministic: per-workgroup partials in fixed order,
// finished by one workgroup in fp64.
#include <hip/hip_runtime.h>
#include <math.h>

#include "ddsp_hip.h"
#include "ddsp_osc_common.h"

namespace {

constexpr int kBlocks = 2048;  // partial sums (8 workgroups per CU); grid-stride beyond that

__device__ __forceinline__ float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

__global__ void __launch_bounds__(256) spectral_loss_kernel(const float2 *__restrict__ pred, const float2 *__restrict__ truth,
                                                            float2 *__restrict__ grad, float *__restrict__ partials,
                                                            long n, float alpha, float eps, float inv_n)
{
    float lin = 0.0f, lg = 0.0f;
    const float inv_ln2 = 1.4426950408889634f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float2 a = pred[i], b = truth[i];
        const float P = __fmaf_rn(a.x, a.x, a.y * a.y), Q = __fmaf_rn(b.x, b.x, b.y * b.y);
        const float d = P - Q;
        const float e = log2f(Q + eps) - log2f(P + eps);
        lin += fabsf(d);
        lg += fabsf(e);
        if (grad) {
            // d|P-Q|/dP = sgn(P-Q);  d|log2(Q+eps) - log2(P+eps)|/dP = -sgn(e) / ((P+eps) ln 2);  dP/d(re,im) = 2 (re,im)
            const float c = 2.0f * inv_n * (sgn(d) - alpha * sgn(e) * inv_ln2 / (P + eps));
            grad[i] = make_float2(c * a.x, c * a.y);
        }
    }
    lin = ddsp_osc::wave_sum(lin);
    lg = ddsp_osc::wave_sum(lg);
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lin; red[1][threadIdx.x >> 6] = lg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) spectral_finish_kernel(const float *__restrict__ partials, int blocks, float alpha,
                                                              double inv_n, float *__restrict__ out)
{
    __shared__ double red[2][256];
    double lin = 0.0, lg = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 256) { lin += (double)partials[2 * i]; lg += (double)partials[2 * i + 1]; }
    red[0][threadIdx.x] = lin;
    red[1][threadIdx.x] = lg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l = red[0][0] * inv_n, g = red[1][0] * inv_n;
        out[0] = (float)(l + (double)alpha * g);
        out[1] = (float)l;
        out[2] = (float)g;
    }
}

}  // namespace

namespace ddsp_mss {
// shared with ddsp_mss_fft.hip: the fp64 finish of a scale's per-workgroup partial sums -> out3 = {loss, linear term, log term}
hipError_t launch_finish(const float *partials, int blocks, float alpha, double inv_n, float *out3, hipStream_t s)
{
    hipLaunchKernelGGL(spectral_finish_kernel, dim3(1), dim3(256), 0, s, partials, blocks, alpha, inv_n, out3);
    return hipGetLastError();
}
}  // namespace ddsp_mss

extern "C" size_t ddsp_spectral_loss_scratch_bytes(void) { return sizeof(float) * 2 * kBlocks; }

extern "C" int ddsp_spectral_loss(const float *pred_ri, const float *true_ri, float *grad_ri, void *scratch, float *out3,
                                  long n_bins, float alpha, float eps, void *stream)
{
    if (!(eps > 0.0f)) return DDSP_EINVAL;
    if (n_bins == 0) {   // an empty shard: the mean over no bins is reported as 0 (and there is no gradient to write)
        if (!out3) return DDSP_EINVAL;
        return (int)hipMemsetAsync(out3, 0, 3 * sizeof(float), (hipStream_t)stream);
    }
    if (!pred_ri || !true_ri || !scratch || !out3 || n_bins < 0) return DDSP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    long want = (n_bins + 255) / 256;
    const int blocks = (int)(want < kBlocks ? want : kBlocks);
    hipLaunchKernelGGL(spectral_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float2 *)pred_ri,
                       (const float2 *)true_ri, (float2 *)grad_ri, (float *)scratch, n_bins, alpha, eps, (float)(1.0 / (double)n_bins));
    hipLaunchKernelGGL(spectral_finish_kernel, dim3(1), dim3(256), 0, s, (const float *)scratch, blocks, alpha,
                       1.0 / (double)n_bins, out3);
    return (int)hipGetLastError();
}
