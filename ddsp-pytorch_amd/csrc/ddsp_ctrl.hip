// Output non-linearity of the control network's three heads (model/autoencoder/decoder.py:110-116, `modified_sigmoid`:
// 2 * sigmoid(x)^ln(10) + 1e-7), forward and backward, as one elementwise pass each instead of the four (forward) and
// six (backward) stock launches per head -- the fusion SURVEY §8f row 4 names.  HBM-bound: 8 B / 12 B per element.
#include <hip/hip_runtime.h>
#include <math.h>

#include <initializer_list>

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

// ---- the three control heads as one epilogue (decoder.py:96-100, :110-116) ------------------------------------------
// The controller ends in three Linear layers on the same activations (harmonic amplitudes, loudness, filter magnitudes),
// each followed by modified_sigmoid.  Run as ONE GEMM on the concatenated weights, their outputs arrive as one
// [rows, n0+n1+n2] matrix (fp32, or the 16-bit autocast type): this pass applies the sigmoid and writes the three controls
// as separate dense fp32 tensors (what the synthesis kernels take); the backward reads the three upstream gradients and
// writes one gradient matrix in the GEMM's type.
namespace {

template <typename T> __device__ __forceinline__ float head_ld(const T *p, long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void head_st(T *p, long i, float v) { p[i] = (T)v; }

template <typename T>
__global__ void __launch_bounds__(256) heads_fwd_kernel(const T *__restrict__ x, float *__restrict__ o0, float *__restrict__ o1,
                                                        float *__restrict__ o2, long rows, int n0, int n1, int n2)
{
    const int N = n0 + n1 + n2;
    const long total = rows * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / N;
        const int c = (int)(i - r * N);
        const float s = 1.0f / (1.0f + expf(-head_ld(x, i)));
        const float y = 2.0f * powf(s, kExponent) + kFloor;
        if (c < n0) o0[r * n0 + c] = y;
        else if (c < n0 + n1) o1[r * n1 + (c - n0)] = y;
        else o2[r * n2 + (c - n0 - n1)] = y;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) heads_bwd_kernel(const T *__restrict__ x, const float *__restrict__ g0, const float *__restrict__ g1,
                                                        const float *__restrict__ g2, T *__restrict__ gx, long rows, int n0, int n1, int n2)
{
    const int N = n0 + n1 + n2;
    const long total = rows * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / N;
        const int c = (int)(i - r * N);
        const float g = c < n0 ? g0[r * n0 + c] : (c < n0 + n1 ? g1[r * n1 + (c - n0)] : g2[r * n2 + (c - n0 - n1)]);
        const float s = 1.0f / (1.0f + expf(-head_ld(x, i)));
        head_st(gx, i, g * (2.0f * kExponent * powf(s, kExponent) * (1.0f - s)));
    }
}

}  // namespace

extern "C" int ddsp_heads_sigmoid_forward(const void *x, float *out0, float *out1, float *out2, long rows, int n0, int n1, int n2,
                                          int io_type, void *stream)
{
    if (rows == 0) return 0;
    if (!x || !out0 || !out1 || !out2 || rows < 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return DDSP_EINVAL;
    const unsigned grid = grid_for(rows * (n0 + n1 + n2));
    hipStream_t s = (hipStream_t)stream;
    switch (io_type) {
        case 0: hipLaunchKernelGGL(heads_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x, out0, out1, out2, rows, n0, n1, n2); break;
        case 1: hipLaunchKernelGGL(heads_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16 *)x, out0, out1, out2, rows, n0, n1, n2); break;
        case 2: hipLaunchKernelGGL(heads_fwd_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16 *)x, out0, out1, out2, rows, n0, n1, n2); break;
        default: return DDSP_EINVAL;
    }
    return (int)hipGetLastError();
}

extern "C" int ddsp_heads_sigmoid_backward(const void *x, const float *g0, const float *g1, const float *g2, void *grad_x, long rows,
                                           int n0, int n1, int n2, int io_type, void *stream)
{
    if (rows == 0) return 0;
    if (!x || !g0 || !g1 || !g2 || !grad_x || rows < 0 || n0 <= 0 || n1 <= 0 || n2 <= 0) return DDSP_EINVAL;
    const unsigned grid = grid_for(rows * (n0 + n1 + n2));
    hipStream_t s = (hipStream_t)stream;
    switch (io_type) {
        case 0: hipLaunchKernelGGL(heads_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x, g0, g1, g2, (float *)grad_x, rows, n0, n1, n2); break;
        case 1: hipLaunchKernelGGL(heads_bwd_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16 *)x, g0, g1, g2, (__bf16 *)grad_x, rows, n0, n1, n2); break;
        case 2: hipLaunchKernelGGL(heads_bwd_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16 *)x, g0, g1, g2, (_Float16 *)grad_x, rows, n0, n1, n2); break;
        default: return DDSP_EINVAL;
    }
    return (int)hipGetLastError();
}

// ---- LayerNorm + LeakyReLU of the MLP blocks (decoder.py:9-39: Linear -> LayerNorm -> LeakyReLU) ------------------
// One pass forward (y = lrelu(gamma * (x - mean) * rstd + beta); mean and rstd kept per row) and one pass backward
// (dx, plus per-workgroup partial sums of d gamma / d beta finished by a second small kernel: deterministic) instead of
// two launches forward and four backward per block.  One wavefront per row, D/64 elements per lane in registers;
// rows are D = 256 * NV wide (NV = 1..4).  HBM-bound: 8 B per element forward, 16 B backward.
namespace {

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Element type of the activations (x, y and their gradients): fp32, or -- under torch.autocast, where the Linear in front
// hands over bf16 / fp16 and the Linear behind wants it back -- the 16-bit type itself, so that no cast pass runs on either
// side of the fused pass.  Statistics, gamma / beta and all arithmetic stay fp32.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct IoF32 {
    typedef float T;
    static __device__ __forceinline__ float4 ld(const T *p, long i) { return reinterpret_cast<const float4 *>(p)[i]; }
    static __device__ __forceinline__ void st(T *p, long i, float4 v) { reinterpret_cast<float4 *>(p)[i] = v; }
};
template <typename V4, typename E>
struct IoHalf {
    typedef E T;
    static __device__ __forceinline__ float4 ld(const T *p, long i)
    {
        const V4 h = reinterpret_cast<const V4 *>(p)[i];
        const f32x4_t f = __builtin_convertvector(h, f32x4_t);
        return make_float4(f.x, f.y, f.z, f.w);
    }
    static __device__ __forceinline__ void st(T *p, long i, float4 v)
    {
        const f32x4_t f = {v.x, v.y, v.z, v.w};
        reinterpret_cast<V4 *>(p)[i] = __builtin_convertvector(f, V4);     // round to nearest even
    }
};
typedef IoHalf<bf16x4_t, __bf16> IoBf16;
typedef IoHalf<f16x4_t, _Float16> IoF16;

template <int NV, typename IO>
__global__ void __launch_bounds__(256) ln_lrelu_fwd_kernel(const typename IO::T *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, typename IO::T *__restrict__ y,
                                                           float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                           long rows, float eps, float slope)
{
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    float4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        b[j] = reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
    }
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        float4 v[NV];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = IO::ld(x + row * D, lane + 64 * j);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum64(s) * (1.0f / D);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) * (1.0f / D) + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = __fmaf_rn(v[j].x * rstd, g[j].x, b[j].x);
            o.y = __fmaf_rn(v[j].y * rstd, g[j].y, b[j].y);
            o.z = __fmaf_rn(v[j].z * rstd, g[j].z, b[j].z);
            o.w = __fmaf_rn(v[j].w * rstd, g[j].w, b[j].w);
            o.x = o.x > 0.0f ? o.x : o.x * slope;
            o.y = o.y > 0.0f ? o.y : o.y * slope;
            o.z = o.z > 0.0f ? o.z : o.z * slope;
            o.w = o.w > 0.0f ? o.w : o.w * slope;
            IO::st(y + row * D, lane + 64 * j, o);
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

// partials: [gridDim.x][3][D]  (d gamma | d beta | column sums of d x = the bias gradient of the Linear in front), one slab per
// workgroup, summed over its four wavefronts through LDS
template <int NV, typename IO>
__global__ void __launch_bounds__(256) ln_lrelu_bwd_kernel(const typename IO::T *__restrict__ gy, const typename IO::T *__restrict__ x,
                                                           const typename IO::T *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                           typename IO::T *__restrict__ gx, float *__restrict__ partials, long rows, float slope)
{
    constexpr int D = 256 * NV;
    __shared__ float red[4][3][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 g[NV], dg[NV], db[NV], dc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        dg[j] = db[j] = dc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh[NV], d[NV];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 xv = IO::ld(x + row * D, lane + 64 * j);
            const float4 yv = IO::ld(y + row * D, lane + 64 * j);
            float4 gv = IO::ld(gy + row * D, lane + 64 * j);
            gv.x = yv.x > 0.0f ? gv.x : gv.x * slope;      // the activation keeps the sign of its input (slope > 0)
            gv.y = yv.y > 0.0f ? gv.y : gv.y * slope;
            gv.z = yv.z > 0.0f ? gv.z : gv.z * slope;
            gv.w = yv.w > 0.0f ? gv.w : gv.w * slope;
            xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            dg[j].x += gv.x * xh[j].x; dg[j].y += gv.y * xh[j].y; dg[j].z += gv.z * xh[j].z; dg[j].w += gv.w * xh[j].w;
            db[j].x += gv.x; db[j].y += gv.y; db[j].z += gv.z; db[j].w += gv.w;
            d[j] = make_float4(gv.x * g[j].x, gv.y * g[j].y, gv.z * g[j].z, gv.w * g[j].w);
            s1 += (d[j].x + d[j].y) + (d[j].z + d[j].w);
            s2 += (d[j].x * xh[j].x + d[j].y * xh[j].y) + (d[j].z * xh[j].z + d[j].w * xh[j].w);
        }
        const float m1 = wave_sum64(s1) * (1.0f / D), m2 = wave_sum64(s2) * (1.0f / D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = rstd * (d[j].x - m1 - xh[j].x * m2);
            o.y = rstd * (d[j].y - m1 - xh[j].y * m2);
            o.z = rstd * (d[j].z - m1 - xh[j].z * m2);
            o.w = rstd * (d[j].w - m1 - xh[j].w * m2);
            IO::st(gx + row * D, lane + 64 * j, o);
            dc[j].x += o.x; dc[j].y += o.y; dc[j].z += o.z; dc[j].w += o.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        reinterpret_cast<float4 *>(red[wave][0])[lane + 64 * j] = dg[j];
        reinterpret_cast<float4 *>(red[wave][1])[lane + 64 * j] = db[j];
        reinterpret_cast<float4 *>(red[wave][2])[lane + 64 * j] = dc[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const int which = i / D, c = i - which * D;
        partials[((size_t)blockIdx.x * 3 + which) * D + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// columns of the [blocks][3*D] partial slabs -> d gamma | d beta | column sums (nullable); 64 columns x 16 row groups per workgroup, fixed order
__global__ void __launch_bounds__(1024) ln_lrelu_finish_kernel(const float *__restrict__ partials, int blocks, int D,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dxsum)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;   // c over 3 * D columns
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;   // four loads in flight per thread
    int b = grp;
    for (; b + 48 < blocks; b += 64) {
        s0 += partials[(size_t)b * 3 * D + c];
        s1 += partials[(size_t)(b + 16) * 3 * D + c];
        s2 += partials[(size_t)(b + 32) * 3 * D + c];
        s3 += partials[(size_t)(b + 48) * 3 * D + c];
    }
    for (; b < blocks; b += 16) s0 += partials[(size_t)b * 3 * D + c];
    red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x & 63];
        if (c < D) dgamma[c] = t;
        else if (c < 2 * D) dbeta[c - D] = t;
        else if (dxsum) dxsum[c - 2 * D] = t;
    }
}

// ---- the FIRST block of the f0 / loudness stacks (decoder.py:43-44: Linear(1 -> D) -> LayerNorm -> LeakyReLU) --------------------
// Its Linear is an outer product x[row] * w[j] + b[j]: the row is rebuilt from ONE scalar instead of being written by an elementwise
// pass and read back, forward and backward; and since the block's input carries no gradient, the backward does not store d x either:
// d w[j] = sum_rows dx[row][j] * x[row] and d b[j] = sum_rows dx[row][j] are accumulated beside d gamma / d beta (four column
// sums per workgroup, the same finish).  Replaces, per stack and step: addcmul + fp32 LayerNorm pass forward; fp32 LayerNorm backward,
// two stock reductions of [rows, D] (24 us each) and a product pass backward.
template <int NV, typename IO>
__global__ void __launch_bounds__(256) outer_ln_lrelu_fwd_kernel(const float *__restrict__ xs, const float *__restrict__ w,
                                                                 const float *__restrict__ bias, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, typename IO::T *__restrict__ y,
                                                                 float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                                 long rows, float eps, float slope)
{
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63;
    float4 g[NV], b[NV], wv[NV], bv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        b[j] = reinterpret_cast<const float4 *>(beta)[lane + 64 * j];
        wv[j] = reinterpret_cast<const float4 *>(w)[lane + 64 * j];
        bv[j] = reinterpret_cast<const float4 *>(bias)[lane + 64 * j];
    }
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const float xr = xs[row];
        float4 v[NV];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = make_float4(xr * wv[j].x + bv[j].x, xr * wv[j].y + bv[j].y, xr * wv[j].z + bv[j].z, xr * wv[j].w + bv[j].w);   // product, then sum: a K = 1 GEMM
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = wave_sum64(s) * (1.0f / D);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) * (1.0f / D) + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = __fmaf_rn(v[j].x * rstd, g[j].x, b[j].x);
            o.y = __fmaf_rn(v[j].y * rstd, g[j].y, b[j].y);
            o.z = __fmaf_rn(v[j].z * rstd, g[j].z, b[j].z);
            o.w = __fmaf_rn(v[j].w * rstd, g[j].w, b[j].w);
            o.x = o.x > 0.0f ? o.x : o.x * slope;
            o.y = o.y > 0.0f ? o.y : o.y * slope;
            o.z = o.z > 0.0f ? o.z : o.z * slope;
            o.w = o.w > 0.0f ? o.w : o.w * slope;
            IO::st(y + row * D, lane + 64 * j, o);
        }
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

// partials: [gridDim.x][4][D]  (d gamma | d beta | d w | d b)
template <int NV, typename IO>
__global__ void __launch_bounds__(256) outer_ln_lrelu_bwd_kernel(const typename IO::T *__restrict__ gy, const float *__restrict__ xs,
                                                                 const float *__restrict__ w, const float *__restrict__ bias,
                                                                 const typename IO::T *__restrict__ y, const float *__restrict__ gamma,
                                                                 const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                                 float *__restrict__ partials, long rows, float slope)
{
    constexpr int D = 256 * NV;
    __shared__ float red[4][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 g[NV], wv[NV], bv[NV], dg[NV], db[NV], dw[NV], dc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4 *>(gamma)[lane + 64 * j];
        wv[j] = reinterpret_cast<const float4 *>(w)[lane + 64 * j];
        bv[j] = reinterpret_cast<const float4 *>(bias)[lane + 64 * j];
        dg[j] = db[j] = dw[j] = dc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row], xr = xs[row];
        float4 xh[NV], d[NV];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 xv = make_float4(xr * wv[j].x + bv[j].x, xr * wv[j].y + bv[j].y, xr * wv[j].z + bv[j].z, xr * wv[j].w + bv[j].w);
            const float4 yv = IO::ld(y + row * D, lane + 64 * j);
            float4 gv = IO::ld(gy + row * D, lane + 64 * j);
            gv.x = yv.x > 0.0f ? gv.x : gv.x * slope;
            gv.y = yv.y > 0.0f ? gv.y : gv.y * slope;
            gv.z = yv.z > 0.0f ? gv.z : gv.z * slope;
            gv.w = yv.w > 0.0f ? gv.w : gv.w * slope;
            xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            dg[j].x += gv.x * xh[j].x; dg[j].y += gv.y * xh[j].y; dg[j].z += gv.z * xh[j].z; dg[j].w += gv.w * xh[j].w;
            db[j].x += gv.x; db[j].y += gv.y; db[j].z += gv.z; db[j].w += gv.w;
            d[j] = make_float4(gv.x * g[j].x, gv.y * g[j].y, gv.z * g[j].z, gv.w * g[j].w);
            s1 += (d[j].x + d[j].y) + (d[j].z + d[j].w);
            s2 += (d[j].x * xh[j].x + d[j].y * xh[j].y) + (d[j].z * xh[j].z + d[j].w * xh[j].w);
        }
        const float m1 = wave_sum64(s1) * (1.0f / D), m2 = wave_sum64(s2) * (1.0f / D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;                                              // d (pre-activation): never stored
            o.x = rstd * (d[j].x - m1 - xh[j].x * m2);
            o.y = rstd * (d[j].y - m1 - xh[j].y * m2);
            o.z = rstd * (d[j].z - m1 - xh[j].z * m2);
            o.w = rstd * (d[j].w - m1 - xh[j].w * m2);
            dw[j].x += o.x * xr; dw[j].y += o.y * xr; dw[j].z += o.z * xr; dw[j].w += o.w * xr;
            dc[j].x += o.x; dc[j].y += o.y; dc[j].z += o.z; dc[j].w += o.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        reinterpret_cast<float4 *>(red[wave][0])[lane + 64 * j] = dg[j];
        reinterpret_cast<float4 *>(red[wave][1])[lane + 64 * j] = db[j];
        reinterpret_cast<float4 *>(red[wave][2])[lane + 64 * j] = dw[j];
        reinterpret_cast<float4 *>(red[wave][3])[lane + 64 * j] = dc[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * D; i += 256) {
        const int which = i / D, c = i - which * D;
        partials[((size_t)blockIdx.x * 4 + which) * D + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// columns of the [blocks][4*D] partial slabs -> d gamma | d beta | d w | d b (the two-output finish above, four outputs)
__global__ void __launch_bounds__(1024) outer_ln_finish_kernel(const float *__restrict__ partials, int blocks, int D,
                                                               float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2, float *__restrict__ o3)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;   // c over 4 * D columns
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = grp;
    for (; b + 48 < blocks; b += 64) {
        s0 += partials[(size_t)b * 4 * D + c];
        s1 += partials[(size_t)(b + 16) * 4 * D + c];
        s2 += partials[(size_t)(b + 32) * 4 * D + c];
        s3 += partials[(size_t)(b + 48) * 4 * D + c];
    }
    for (; b < blocks; b += 16) s0 += partials[(size_t)b * 4 * D + c];
    red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x & 63];
        const int which = c / D, cc = c - which * D;
        (which == 0 ? o0 : which == 1 ? o1 : which == 2 ? o2 : o3)[cc] = t;
    }
}

constexpr int kLnBlocks = 1024;  // four workgroups per CU: enough rows in flight to cover the HBM latency

}  // namespace

extern "C" size_t ddsp_ln_lrelu_scratch_bytes(int D) { return D > 0 ? sizeof(float) * 3 * (size_t)D * kLnBlocks : 0; }

namespace {

template <typename IO>
int ln_forward(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd, long rows, int D, float eps,
               float slope, hipStream_t s)
{
    if (rows == 0) return 0;
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0) return DDSP_EINVAL;
    if (D <= 0 || D % 256 != 0 || D > 1024) return DDSP_ERANGE;
    typedef typename IO::T T;
    const T *xi = (const T *)x;
    T *yo = (T *)y;
    const long want = (rows + 3) / 4;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096)), blk(256);
    switch (D / 256) {
        case 1: hipLaunchKernelGGL((ln_lrelu_fwd_kernel<1, IO>), grid, blk, 0, s, xi, gamma, beta, yo, mean, rstd, rows, eps, slope); break;
        case 2: hipLaunchKernelGGL((ln_lrelu_fwd_kernel<2, IO>), grid, blk, 0, s, xi, gamma, beta, yo, mean, rstd, rows, eps, slope); break;
        case 3: hipLaunchKernelGGL((ln_lrelu_fwd_kernel<3, IO>), grid, blk, 0, s, xi, gamma, beta, yo, mean, rstd, rows, eps, slope); break;
        default: hipLaunchKernelGGL((ln_lrelu_fwd_kernel<4, IO>), grid, blk, 0, s, xi, gamma, beta, yo, mean, rstd, rows, eps, slope); break;
    }
    return (int)hipGetLastError();
}

template <typename IO>
int ln_backward(const void *grad_y, const void *x, const void *y, const float *gamma, const float *mean, const float *rstd, void *grad_x,
                float *grad_gamma, float *grad_beta, float *grad_xsum, void *scratch, long rows, int D, float slope, hipStream_t s)
{
    if (D <= 0 || D % 256 != 0 || D > 1024) return DDSP_ERANGE;
    if (rows == 0) {   // an empty shard (batch < world size): no rows contribute, the parameter gradients are zero
        if (!grad_gamma || !grad_beta) return DDSP_EINVAL;
        hipError_t e = hipMemsetAsync(grad_gamma, 0, sizeof(float) * (size_t)D, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_beta, 0, sizeof(float) * (size_t)D, s);
        if (e == hipSuccess && grad_xsum) e = hipMemsetAsync(grad_xsum, 0, sizeof(float) * (size_t)D, s);
        return (int)e;
    }
    if (!grad_y || !x || !y || !gamma || !mean || !rstd || !grad_x || !grad_gamma || !grad_beta || !scratch || rows < 0) return DDSP_EINVAL;
    typedef typename IO::T T;
    const T *gy = (const T *)grad_y, *xi = (const T *)x, *yi = (const T *)y;
    T *gx = (T *)grad_x;
    const long want = (rows + 3) / 4;
    const int blocks = (int)(want < kLnBlocks ? want : kLnBlocks);
    const dim3 grid((unsigned)blocks), blk(256);
    float *part = (float *)scratch;
    switch (D / 256) {
        case 1: hipLaunchKernelGGL((ln_lrelu_bwd_kernel<1, IO>), grid, blk, 0, s, gy, xi, yi, gamma, mean, rstd, gx, part, rows, slope); break;
        case 2: hipLaunchKernelGGL((ln_lrelu_bwd_kernel<2, IO>), grid, blk, 0, s, gy, xi, yi, gamma, mean, rstd, gx, part, rows, slope); break;
        case 3: hipLaunchKernelGGL((ln_lrelu_bwd_kernel<3, IO>), grid, blk, 0, s, gy, xi, yi, gamma, mean, rstd, gx, part, rows, slope); break;
        default: hipLaunchKernelGGL((ln_lrelu_bwd_kernel<4, IO>), grid, blk, 0, s, gy, xi, yi, gamma, mean, rstd, gx, part, rows, slope); break;
    }
    hipLaunchKernelGGL(ln_lrelu_finish_kernel, dim3((unsigned)(3 * D / 64)), dim3(1024), 0, s, part, blocks, D, grad_gamma, grad_beta, grad_xsum);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int ddsp_ln_lrelu_forward(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                     long rows, int D, float eps, float slope, void *stream)
{
    return ln_forward<IoF32>(x, gamma, beta, y, mean, rstd, rows, D, eps, slope, (hipStream_t)stream);
}

extern "C" int ddsp_ln_lrelu_backward(const float *grad_y, const float *x, const float *y, const float *gamma, const float *mean,
                                      const float *rstd, float *grad_x, float *grad_gamma, float *grad_beta, float *grad_xsum, void *scratch,
                                      long rows, int D, float slope, void *stream)
{
    return ln_backward<IoF32>(grad_y, x, y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, grad_xsum, scratch, rows, D, slope, (hipStream_t)stream);
}

extern "C" int ddsp_ln_lrelu_forward_16(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                                        long rows, int D, float eps, float slope, int io_type, void *stream)
{
    if (io_type == DDSP_IO_BF16) return ln_forward<IoBf16>(x, gamma, beta, y, mean, rstd, rows, D, eps, slope, (hipStream_t)stream);
    if (io_type == DDSP_IO_F16) return ln_forward<IoF16>(x, gamma, beta, y, mean, rstd, rows, D, eps, slope, (hipStream_t)stream);
    return DDSP_EINVAL;
}

extern "C" int ddsp_ln_lrelu_backward_16(const void *grad_y, const void *x, const void *y, const float *gamma, const float *mean,
                                         const float *rstd, void *grad_x, float *grad_gamma, float *grad_beta, float *grad_xsum, void *scratch,
                                         long rows, int D, float slope, int io_type, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (io_type == DDSP_IO_BF16) return ln_backward<IoBf16>(grad_y, x, y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, grad_xsum, scratch, rows, D, slope, s);
    if (io_type == DDSP_IO_F16) return ln_backward<IoF16>(grad_y, x, y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, grad_xsum, scratch, rows, D, slope, s);
    return DDSP_EINVAL;
}

namespace {

template <typename IO>
int outer_forward(const float *xs, const float *w, const float *bias, const float *gamma, const float *beta, void *y, float *mean,
                  float *rstd, long rows, int D, float eps, float slope, hipStream_t s)
{
    if (rows == 0) return 0;
    if (!xs || !w || !bias || !gamma || !beta || !y || !mean || !rstd || rows < 0) return DDSP_EINVAL;
    if (D != 256 && D != 512) return DDSP_ERANGE;            // (the backward keeps 16 floats of LDS per column and wavefront)
    typedef typename IO::T T;
    T *yo = (T *)y;
    const long want = (rows + 3) / 4;
    const dim3 grid((unsigned)(want < 4096 ? want : 4096)), blk(256);
    if (D == 256) hipLaunchKernelGGL((outer_ln_lrelu_fwd_kernel<1, IO>), grid, blk, 0, s, xs, w, bias, gamma, beta, yo, mean, rstd, rows, eps, slope);
    else hipLaunchKernelGGL((outer_ln_lrelu_fwd_kernel<2, IO>), grid, blk, 0, s, xs, w, bias, gamma, beta, yo, mean, rstd, rows, eps, slope);
    return (int)hipGetLastError();
}

template <typename IO>
int outer_backward(const void *grad_y, const float *xs, const float *w, const float *bias, const void *y, const float *gamma,
                   const float *mean, const float *rstd, float *grad_w, float *grad_b, float *grad_gamma, float *grad_beta,
                   void *scratch, long rows, int D, float slope, hipStream_t s)
{
    if (D != 256 && D != 512) return DDSP_ERANGE;
    if (!grad_w || !grad_b || !grad_gamma || !grad_beta) return DDSP_EINVAL;
    if (rows == 0) {   // an empty shard: the parameter gradients are zero
        hipError_t e = hipSuccess;
        for (float *o : {grad_w, grad_b, grad_gamma, grad_beta})
            if (e == hipSuccess) e = hipMemsetAsync(o, 0, sizeof(float) * (size_t)D, s);
        return (int)e;
    }
    if (!grad_y || !xs || !w || !bias || !y || !gamma || !mean || !rstd || !scratch || rows < 0) return DDSP_EINVAL;
    typedef typename IO::T T;
    const T *gy = (const T *)grad_y, *yi = (const T *)y;
    const long want = (rows + 3) / 4;
    const int blocks = (int)(want < kLnBlocks ? want : kLnBlocks);
    const dim3 grid((unsigned)blocks), blk(256);
    float *part = (float *)scratch;
    if (D == 256) hipLaunchKernelGGL((outer_ln_lrelu_bwd_kernel<1, IO>), grid, blk, 0, s, gy, xs, w, bias, yi, gamma, mean, rstd, part, rows, slope);
    else hipLaunchKernelGGL((outer_ln_lrelu_bwd_kernel<2, IO>), grid, blk, 0, s, gy, xs, w, bias, yi, gamma, mean, rstd, part, rows, slope);
    hipLaunchKernelGGL(outer_ln_finish_kernel, dim3((unsigned)(4 * D / 64)), dim3(1024), 0, s, part, blocks, D, grad_gamma, grad_beta, grad_w, grad_b);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t ddsp_outer_ln_lrelu_scratch_bytes(int D) { return D > 0 ? sizeof(float) * 4 * (size_t)D * kLnBlocks : 0; }

extern "C" int ddsp_outer_ln_lrelu_forward(const float *x, const float *w, const float *bias, const float *gamma, const float *beta, void *y,
                                           float *mean, float *rstd, long rows, int D, float eps, float slope, int io_type, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (io_type == 0) return outer_forward<IoF32>(x, w, bias, gamma, beta, y, mean, rstd, rows, D, eps, slope, s);
    if (io_type == DDSP_IO_BF16) return outer_forward<IoBf16>(x, w, bias, gamma, beta, y, mean, rstd, rows, D, eps, slope, s);
    if (io_type == DDSP_IO_F16) return outer_forward<IoF16>(x, w, bias, gamma, beta, y, mean, rstd, rows, D, eps, slope, s);
    return DDSP_EINVAL;
}

extern "C" int ddsp_outer_ln_lrelu_backward(const void *grad_y, const float *x, const float *w, const float *bias, const void *y,
                                            const float *gamma, const float *mean, const float *rstd, float *grad_w, float *grad_bias,
                                            float *grad_gamma, float *grad_beta, void *scratch, long rows, int D, float slope,
                                            int io_type, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (io_type == 0) return outer_backward<IoF32>(grad_y, x, w, bias, y, gamma, mean, rstd, grad_w, grad_bias, grad_gamma, grad_beta, scratch, rows, D, slope, s);
    if (io_type == DDSP_IO_BF16) return outer_backward<IoBf16>(grad_y, x, w, bias, y, gamma, mean, rstd, grad_w, grad_bias, grad_gamma, grad_beta, scratch, rows, D, slope, s);
    if (io_type == DDSP_IO_F16) return outer_backward<IoF16>(grad_y, x, w, bias, y, gamma, mean, rstd, grad_w, grad_bias, grad_gamma, grad_beta, scratch, rows, D, slope, s);
    return DDSP_EINVAL;
}

// ---- column sums: the bias gradient of a dense layer, sum over the M = batch x frames rows of gy [M, N] ------------------
// (decoder.py:9-39, :60-72: every Linear's bias, the GRU's b_ih / b_hh).  The stock reduction of a tall thin matrix takes
// 12-150 us per layer at the training shape (16 000 rows); this is one streaming pass (64 columns x 4 row lanes per workgroup over a
// chunk of rows; partial sums per chunk) and a small finish over the chunks, both in a fixed order: deterministic.
namespace {

constexpr int kColChunks = 128;

template <typename T> __device__ __forceinline__ float load_as_float(const T *p);
template <> __device__ __forceinline__ float load_as_float<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load_as_float<__bf16>(const __bf16 *p) { return (float)*p; }
template <> __device__ __forceinline__ float load_as_float<_Float16>(const _Float16 *p) { return (float)*p; }

template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T *__restrict__ x, float *__restrict__ partials, long M, int N, long rows_per_chunk)
{
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const long row0 = (long)blockIdx.y * rows_per_chunk;
    long row1 = row0 + rows_per_chunk;
    if (row1 > M) row1 = M;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (col < N) {
        long row = row0 + r;
        for (; row + 12 < row1; row += 16) {
            a0 += load_as_float(x + row * N + col);
            a1 += load_as_float(x + (row + 4) * N + col);
            a2 += load_as_float(x + (row + 8) * N + col);
            a3 += load_as_float(x + (row + 12) * N + col);
        }
        for (; row < row1; row += 4) a0 += load_as_float(x + row * N + col);
    }
    __shared__ float red[4][64];
    red[r][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (r == 0 && col < N) partials[(long)blockIdx.y * N + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// 64 columns x 4 lanes of chunks per workgroup; loads in batches of eight, additions in a fixed order
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float *__restrict__ partials, float *__restrict__ out, int chunks, int N)
{
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int per = (chunks + 3) / 4;
    int ch = r * per;
    const int end = (ch + per < chunks) ? ch + per : chunks;
    float s = 0.0f;
    if (col < N) {
        for (; ch + 8 <= end; ch += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = partials[(long)(ch + e) * N + col];
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
        }
        for (; ch < end; ++ch) s += partials[(long)ch * N + col];
    }
    __shared__ float red[4][64];
    red[r][c] = s;
    __syncthreads();
    if (r == 0 && col < N) out[col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

template <typename T>
int colsum_launch(const void *x, float *out, void *scratch, long M, int N, hipStream_t s)
{
    long chunks = (M + 63) / 64;
    if (chunks > kColChunks) chunks = kColChunks;
    const long rpc = (M + chunks - 1) / chunks;
    chunks = (M + rpc - 1) / rpc;
    hipLaunchKernelGGL((colsum_kernel<T>), dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, s, (const T *)x, (float *)scratch, M, N, rpc);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, (const float *)scratch, out, (int)chunks, N);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t ddsp_colsum_scratch_bytes(int N) { return N > 0 ? sizeof(float) * (size_t)N * kColChunks : 0; }

extern "C" int ddsp_colsum(const void *x, float *out, void *scratch, long M, int N, int io_type, void *stream)
{
    if (N == 0) return 0;
    if (!out || N < 0 || M < 0) return DDSP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return (int)hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);       // the sum over no rows
    if (!x || !scratch) return DDSP_EINVAL;
    if (io_type == 0) return colsum_launch<float>(x, out, scratch, M, N, s);
    if (io_type == DDSP_IO_BF16) return colsum_launch<__bf16>(x, out, scratch, M, N, s);
    if (io_type == DDSP_IO_F16) return colsum_launch<_Float16>(x, out, scratch, M, N, s);
    return DDSP_EINVAL;
}
