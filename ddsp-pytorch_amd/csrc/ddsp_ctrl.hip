// Output non-linearity of the control network's three heads (model/autoencoder/decoder.py:110-116, `modified_sigmoid`:
// 2 * sigmoid(x)^ln(10) + 1e-7), forward and backward, as one elementwise pass each instead of the four (forward) and
// six (backward) stock launches per head -- the fusion SURVEY §8f row 4 names.  HBM-bound: 8 B / 12 B per element.
#include <hip/hip_runtime.h>
#include <math.h>

#include "ddsp_hip.h"

namespace {

constexpr float kExponent = 2.3026f;   // the reference's literal for ln(10) (decoder.py:115)
constexpr float kFloor = 1e-7f;

__global__ void __launch_bounds__(256) scaled_sigmoid_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float s = 1.0f / (1.0f + expf(-x[i]));
        y[i] = 2.0f * powf(s, kExponent) + kFloor;
    }
}

// d/dx [2 s^p + c] = 2 p s^p (1 - s)
__global__ void __launch_bounds__(256) scaled_sigmoid_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                 float *__restrict__ gx, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float s = 1.0f / (1.0f + expf(-x[i]));
        gx[i] = gy[i] * (2.0f * kExponent * powf(s, kExponent) * (1.0f - s));
    }
}

unsigned grid_for(long n)
{
    const long want = (n + 255) / 256;
    return (unsigned)(want < 4096 ? want : 4096);
}

}  // namespace

extern "C" int ddsp_scaled_sigmoid_forward(const float *x, float *y, long n, void *stream)
{
    if (n == 0) return 0;
    if (!x || !y || n < 0) return DDSP_EINVAL;
    hipLaunchKernelGGL(scaled_sigmoid_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return (int)hipGetLastError();
}

extern "C" int ddsp_scaled_sigmoid_backward(const float *x, const float *grad_y, float *grad_x, long n, void *stream)
{
    if (n == 0) return 0;
    if (!x || !grad_y || !grad_x || n < 0) return DDSP_EINVAL;
    hipLaunchKernelGGL(scaled_sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, grad_y, grad_x, n);
    return (int)hipGetLastError();
}
