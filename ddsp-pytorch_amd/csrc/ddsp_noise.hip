// Filtered-noise generator for MI355X (gfx950) -- replaces the torch-op chain of
// model/ddsp/filtered_noise.py:7-53 (amp_to_impulse_response, fft_convolve, FilteredNoise.forward).
//
// Per frame (b,t), all in LDS:
//   z[n]    = irfft(H + 0j)[n]                       n in [0,S), S = 2(F-1)          (:8-10)
//   kern[j] = roll(pad(roll(z, S/2) * hann_periodic(S), R - S), -S/2)[j]   j in [0,R)  (:14-20; R < S crops)
//   x[m]    = 2*u[m] - 1                                                             (:44-48)
//   y[n]    = sum_{m<=n} x[m] * kern[n-m]            n in [0,R)                      (:25-32)
// fft_convolve zero-pads to 2R, multiplies spectra and keeps the LAST R samples of a buffer whose
// kernel was left-padded by R: that is exactly the first R samples of the linear convolution, the
// tail is discarded and frames are concatenated with no overlap-add (:50-51).
//
// This first version evaluates the inverse real DFT and the truncated convolution directly in fp32
// (F and R/2 multiply-adds per output); see DESIGN.md §5 for the planned in-LDS FFT form.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"

namespace {

struct NoiseParams {
    const float *Hm;
    const float *u;
    float *y;
    int B, T, F, R, S;
    uint64_t seed, offset;
    int accumulate;
};

// Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,0,0), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__global__ void __launch_bounds__(256) noise_frame_kernel(NoiseParams p)
{
    extern __shared__ float smem[];
    float *Hs = smem;             // [F]
    float *ct = Hs + p.F;         // [S]  cos(2*pi*m/S)
    float *kern = ct + p.S;       // [R]
    float *x = kern + p.R;        // [R]
    const long frame = blockIdx.x;
    const int tid = threadIdx.x;
    const int S = p.S, R = p.R, F = p.F, half = S >> 1;

    for (int k = tid; k < F; k += 256) Hs[k] = p.Hm[frame * F + k];
    for (int m = tid; m < S; m += 256) ct[m] = cospif((float)(2 * m) / (float)S);
    for (int j = tid; j < R; j += 256) kern[j] = 0.0f;
    if (p.u) {
        for (int m = tid; m < R; m += 256) x[m] = p.u[frame * R + m] * 2.0f - 1.0f;
    } else {
        const int quads = (R + 3) >> 2;
        for (int q = tid; q < quads; q += 256) {
            const uint64_t ctr = p.offset + (uint64_t)frame * (uint64_t)quads + (uint64_t)q;
            uint32_t r[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < R) x[4 * q + e] = (float)(r[e] >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        }
    }
    __syncthreads();

    // impulse response: inverse real DFT of the (real, zero-phase) magnitudes, windowed and re-wrapped
    const int taps = min(S, R);
    const float invS = 1.0f / (float)S;
    for (int src = tid; src < taps; src += 256) {
        const int n = (src + half) % S;  // roll(+S/2): a1[src] = z[(src - S/2) mod S]
        float v = Hs[0] + ((n & 1) ? -Hs[F - 1] : Hs[F - 1]);
        float acc = 0.0f;
        int idx = 0;
        for (int k = 1; k < F - 1; ++k) {
            idx += n;
            if (idx >= S) idx -= S;
            acc = __fmaf_rn(Hs[k], ct[idx], acc);
        }
        v = __fmaf_rn(2.0f, acc, v) * invS;
        const float win = 0.5f - 0.5f * ct[src];   // torch.hann_window(S) (periodic)
        int j = (src - half) % R;                  // roll(-S/2) on the length-R buffer
        if (j < 0) j += R;
        kern[j] = v * win;
    }
    __syncthreads();

    for (int n = tid; n < R; n += 256) {
        float acc = 0.0f;
        for (int m = 0; m <= n; ++m) acc = __fmaf_rn(x[m], kern[n - m], acc);
        float *dst = p.y + frame * R + n;
        *dst = p.accumulate ? (*dst + acc) : acc;
    }
}

}  // namespace

extern "C" int ddsp_noise_forward(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop,
                                  uint64_t seed, uint64_t offset, int accumulate, void *stream)
{
    if (B == 0) return 0;
    if (!Hmag || !y || B < 0 || T <= 0 || F < 2 || hop <= 0) return DDSP_EINVAL;
    NoiseParams p;
    p.Hm = Hmag; p.u = uniform; p.y = y;
    p.B = B; p.T = T; p.F = F; p.R = hop; p.S = 2 * (F - 1);
    p.seed = seed; p.offset = offset; p.accumulate = accumulate;
    const size_t lds = sizeof(float) * ((size_t)F + p.S + 2 * (size_t)hop);
    if (lds > 64 * 1024 || (long)B * T >= (1L << 31)) return DDSP_ERANGE;
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE, (hipStream_t)stream);
    hipLaunchKernelGGL(noise_frame_kernel, dim3((unsigned)((long)B * T)), dim3(256), lds, (hipStream_t)stream, p);
    ddsp_prof::end(slot, (hipStream_t)stream);
    return (int)hipGetLastError();
}
