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

#include <atomic>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_noise_common.h"

using namespace ddsp_noise;

namespace {

std::atomic<int> g_force_generic{0};  // ddsp_noise_set_generic: tests exercise the one-frame-per-workgroup kernel (read once per launch)

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
            const uint64_t ctr = p.offset + (p.offset_dev ? *p.offset_dev : 0ull) + (uint64_t)frame * (uint64_t)quads + (uint64_t)q;
            uint32_t r[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < R) x[4 * q + e] = philox_to_sample(r[e]);
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


// ---- batched kernel: FB = 64, 32 or 16 frames per workgroup ------------------------------------------
// Used when hop % 8 == 0 and a tile fits in LDS.  512 threads = 8 wavefronts.  FB = 64 (lane = frame)
// while the tile fits in half the CU's LDS (hop <= 128); larger hops take 32 / 16 frames with 2 / 4 lanes per frame.
//   phase 0  H tile -> LDS transposed [k][frame]; cos table
//   phase 1  zero-phase IR by direct inverse real DFT, 4 frames per thread, using
//            cos(2*pi*k*(S/2-n)/S) = (-1)^k cos(2*pi*k*n/S): one pass over k yields z[n] and z[S/2-n];
//            windowed and re-wrapped straight into kern[frame][j]
//   phase 2  noise tile (injected draw or Philox) -> xs[frame][8 + m] (8 leading zeros = causal padding)
//   phase 3  truncated convolution, register blocked: a wavefront owns output chunks c and C-1-c
//            (8 samples each, balanced triangle) of 64 frames; per 8 taps: 6 ds_read_b128, 64 FMAs
constexpr int kNT = 512;       // threads per workgroup

__device__ __forceinline__ int kern_stride(int R) { return R + 4; }
__device__ __forceinline__ int xs_stride(int R) { return R + 12; }

// LPFLOG = log2(lanes per frame) is a template parameter so that the H tile's row stride is a compile-time constant (the
// unrolled cosine sums then address LDS with immediate offsets); SPOW2: S is a power of two (table index by masking).
template <int LPFLOG, bool SPOW2>
__global__ void __launch_bounds__(kNT) noise_batched_kernel(NoiseParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, R = p.R, F = p.F, half = S >> 1;
    const int KS = kern_stride(R), XS = xs_stride(R);
    constexpr int LPF = 1 << LPFLOG;                    // lanes per frame in the convolution phase
    constexpr int FB = 64 >> LPFLOG;                    // frames per workgroup
    constexpr int HS = FB + 4;                          // row stride of the transposed H tile
    float *ct = smem;                                   // [S] (rounded up to a multiple of 4)
    float *kern = ct + ((S + 3) & ~3);                  // [FB][KS]
    float *un = kern + FB * KS;                         // union: Hs [F][HS]  |  xs [FB][XS]
    float *Hs = un, *xs = un;
    const int tid = threadIdx.x;
    const long frame0 = (long)blockIdx.x * FB;
    const long nframes = (long)p.B * p.T;
    const int nf = (int)min((long)FB, nframes - frame0);

    // phase 0: a wavefront per frame row (coalesced global read, transposed padded LDS write; no index division)
    for (int f = tid >> 6; f < FB; f += kNT / 64)
        for (int k = tid & 63; k < F; k += 64) Hs[k * HS + f] = (f < nf) ? p.Hm[(frame0 + f) * F + k] : 0.0f;
    for (int m = tid; m < S; m += kNT) ct[m] = cospif((float)(2 * m) / (float)S);
    if (S < R)                                          // otherwise phase 1 writes every tap
        for (int e = tid; e < FB * KS; e += kNT) kern[e] = 0.0f;
    __syncthreads();

    // phase 1
    const int taps = min(S, R);
    const float invS = 1.0f / (float)S;
    // z[nn] (nn in [0, S/2], = z[S-nn]) of frame f goes, windowed, where roll/pad/roll put samples nn and S-nn
    auto emit_general = [&](int f, int nn, float z) {
#pragma unroll
        for (int wrap = 0; wrap < 2; ++wrap) {
            // roll(z, S/2)[src] holds z[(src + S/2) % S]: z[nn] sits at src = nn + S/2 and src = S/2 - nn
            int src;
            if (!wrap) { if (nn == half) continue; src = nn + half; }
            else       { if (nn == 0) continue;    src = half - nn; }
            if (src >= taps) continue;
            const float win = 0.5f - 0.5f * ct[src];      // torch.hann_window(S), periodic
            // roll(-S/2) on the length-R buffer: jj = (src - S/2) mod R = nn or (-nn) mod R (no division when nn <= R)
            int jj;
            if (!wrap) jj = nn;                           // nn + S/2 < taps <= R  =>  nn < R
            else if (nn < R) jj = R - nn;
            else jj = (R - nn % R) % R;
            kern[f * KS + jj] = z * win;
        }
    };
    // No crop (S <= R, the normal case): both copies of z[nn] -- at j = nn and at j = R - nn -- carry the same window
    // weight, 0.5 - 0.5 cos(2 pi (nn +- S/2) / S) = 0.5 + 0.5 cos(2 pi nn / S): one table read, two stores.
    const bool nocrop = S <= R;
    auto emit = [&](int f, int nn, float z) {
        if (nocrop) {   // wave-uniform
            const float v = z * __fmaf_rn(0.5f, ct[nn], 0.5f);     // nn = S/2: cos = -1, weight 0 (the reference's hann[0])
            if (nn != half) kern[f * KS + nn] = v;
            if (nn != 0) kern[f * KS + R - nn] = v;
        } else {
            emit_general(f, nn, z);
        }
    };
    {
        // n = 0 and n = S/2 need no cosines: plain and alternating sums, 8 lanes per frame
        {
            const int f = min(tid >> 3, FB - 1), part = tid & 7;
            float e = 0.0f, o = 0.0f;
            for (int k = 1 + part; k < half; k += 8) {
                const float h = Hs[k * HS + f];
                if (k & 1) o += h; else e += h;
            }
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) { e += __shfl_xor(e, m); o += __shfl_xor(o, m); }
            if (part == 0 && (tid >> 3) < FB) {
                const float h0 = Hs[f], hn = Hs[half * HS + f];
                emit(f, 0, __fmaf_rn(2.0f, e + o, h0 + hn) * invS);
                if (half > 0) emit(f, half, __fmaf_rn(2.0f, e - o, h0 + ((half & 1) ? -hn : hn)) * invS);
            }
        }
        // n = 1 .. S/4 paired with S/2 - n: cos(2 pi k (S/2 - n) / S) = (-1)^k cos(2 pi k n / S)
        const int nmain = half / 2;
        for (int item = tid; item < nmain * (FB / 4); item += kNT) {
            const int fq = item / nmain, n = 1 + item - fq * nmain;  // n fastest: a wavefront's H reads broadcast
            float ev[4] = {0, 0, 0, 0}, ov[4] = {0, 0, 0, 0};
            int idx = 0;
            const float *hp = &Hs[HS + 4 * fq];                      // bin k = 1
            int k = 1;
            // eight bins per trip: the eight table reads and the eight H reads (immediate offsets) are in flight together
#pragma unroll 1
            for (; k + 7 < half; k += 8, hp += 8 * HS) {
                float c[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    idx += n;
                    if (SPOW2) idx &= S - 1; else if (idx >= S) idx -= S;
                    c[e] = ct[idx];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float4 h = *reinterpret_cast<const float4 *>(hp + e * HS);
                    if ((e & 1) == 0) {   // k odd (the trip starts at an odd bin)
                        ov[0] = __fmaf_rn(h.x, c[e], ov[0]); ov[1] = __fmaf_rn(h.y, c[e], ov[1]); ov[2] = __fmaf_rn(h.z, c[e], ov[2]); ov[3] = __fmaf_rn(h.w, c[e], ov[3]);
                    } else {
                        ev[0] = __fmaf_rn(h.x, c[e], ev[0]); ev[1] = __fmaf_rn(h.y, c[e], ev[1]); ev[2] = __fmaf_rn(h.z, c[e], ev[2]); ev[3] = __fmaf_rn(h.w, c[e], ev[3]);
                    }
                }
            }
            for (; k < half; ++k, hp += HS) {
                idx += n;
                if (SPOW2) idx &= S - 1; else if (idx >= S) idx -= S;
                const float c = ct[idx];
                const float4 h = *reinterpret_cast<const float4 *>(hp);
                if (k & 1) { ov[0] = __fmaf_rn(h.x, c, ov[0]); ov[1] = __fmaf_rn(h.y, c, ov[1]); ov[2] = __fmaf_rn(h.z, c, ov[2]); ov[3] = __fmaf_rn(h.w, c, ov[3]); }
                else       { ev[0] = __fmaf_rn(h.x, c, ev[0]); ev[1] = __fmaf_rn(h.y, c, ev[1]); ev[2] = __fmaf_rn(h.z, c, ev[2]); ev[3] = __fmaf_rn(h.w, c, ev[3]); }
            }
            const float4 h0 = *reinterpret_cast<const float4 *>(&Hs[4 * fq]);
            const float4 hn = *reinterpret_cast<const float4 *>(&Hs[half * HS + 4 * fq]);
            const float h0v[4] = {h0.x, h0.y, h0.z, h0.w}, hnv[4] = {hn.x, hn.y, hn.z, hn.w};
            const int n2 = half - n;
            const float sg1 = (n & 1) ? -1.0f : 1.0f, sg2 = (n2 & 1) ? -1.0f : 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                emit(4 * fq + q, n, __fmaf_rn(2.0f, ev[q] + ov[q], h0v[q] + sg1 * hnv[q]) * invS);
                if (n2 != n) emit(4 * fq + q, n2, __fmaf_rn(2.0f, ev[q] - ov[q], h0v[q] + sg2 * hnv[q]) * invS);
            }
        }
    }
    __syncthreads();

    // phase 2 (overwrites the H tile)
    for (int e = tid; e < FB * 8; e += kNT) xs[(e >> 3) * XS + (e & 7)] = 0.0f;
    if (p.u) {
        for (int f = tid >> 6; f < FB; f += kNT / 64)                     // a wavefront per frame row: no index division
            for (int m = tid & 63; m < R; m += 64) xs[f * XS + 8 + m] = (f < nf) ? p.u[(frame0 + f) * R + m] * 2.0f - 1.0f : 0.0f;
    } else {
        const int quads = R >> 2;                        // R % 8 == 0 here
        const int qshift = (quads & (quads - 1)) == 0 ? __builtin_ctz(quads) : -1;    // power-of-two hops: shift instead of divide
        for (int e = tid; e < FB * quads; e += kNT) {
            const int f = qshift >= 0 ? (e >> qshift) : e / quads, q = e - f * quads;
            const uint64_t ctr = p.offset + (p.offset_dev ? *p.offset_dev : 0ull) + (uint64_t)(frame0 + f) * (uint64_t)quads + (uint64_t)q;
            uint32_t r[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
            float4 v;
            v.x = philox_to_sample(r[0]);
            v.y = philox_to_sample(r[1]);
            v.z = philox_to_sample(r[2]);
            v.w = philox_to_sample(r[3]);
            *reinterpret_cast<float4 *>(&xs[f * XS + 8 + 4 * q]) = v;
        }
    }
    __syncthreads();

    // phase 3: lane = (frame, sub-chunk); a wavefront owns the groups q and Q-1-q of LPF consecutive 8-sample chunks
    const int lane = tid & 63, wv = tid >> 6;
    const int fr = lane >> LPFLOG, sub = lane & (LPF - 1);
    const int C = R >> 3, Q = (C + LPF - 1) >> LPFLOG;
    const float *krow = kern + fr * KS;
    const float *xrow = xs + fr * XS + 8;
    float *yrow = p.y + (frame0 + fr) * R;
    for (int pr = wv; 2 * pr < Q; pr += kNT / 64) {
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const int q = side ? Q - 1 - pr : pr;
            if (side && q == pr) break;
            const int c = q * LPF + sub;
            if (c >= C) continue;
            const int n0 = c << 3;
            float acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
            for (int j = 0; j <= n0; j += 8) {
                const float4 ka = *reinterpret_cast<const float4 *>(krow + j);
                const float4 kb = *reinterpret_cast<const float4 *>(krow + j + 4);
                const float4 xa = *reinterpret_cast<const float4 *>(xrow + n0 - j - 8);
                const float4 xb = *reinterpret_cast<const float4 *>(xrow + n0 - j - 4);
                const float4 xc = *reinterpret_cast<const float4 *>(xrow + n0 - j);
                const float4 xd = *reinterpret_cast<const float4 *>(xrow + n0 - j + 4);
                asm volatile("" ::"v"(xa.x));  // keep the (unused) first lane alive so the window stays 4 x ds_read_b128
                const float kv[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                const float xw[16] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w, xd.x, xd.y, xd.z, xd.w};
#pragma unroll
                for (int v = 0; v < 8; ++v)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = __fmaf_rn(kv[v], xw[8 + u - v], acc[u]);
            }
            if (fr < nf) {
                float4 lo = make_float4(acc[0], acc[1], acc[2], acc[3]), hi = make_float4(acc[4], acc[5], acc[6], acc[7]);
                float4 *dst = reinterpret_cast<float4 *>(yrow + n0);
                if (p.accumulate) {
                    const float4 a = dst[0], b = dst[1];
                    lo.x += a.x; lo.y += a.y; lo.z += a.z; lo.w += a.w;
                    hi.x += b.x; hi.y += b.y; hi.z += b.z; hi.w += b.w;
                }
                dst[0] = lo;
                dst[1] = hi;
            }
        }
    }
}

// =====================================================================================================
// Backward w.r.t. the filter magnitudes H (autograd of filtered_noise.py:40-53; the noise draw is a constant).
//   y[n] = sum_{m<=n} x[m] kern[n-m]                 =>  d/dkern[j] = sum_{d=0}^{R-1-j} x[d] g[j+d]
//   kern[j] = z[(src+S/2)%S] * hann[src], src <-> j  =>  d/dz folded onto n in [0,S/2]:  Gs[n]
//   z[n] = (H0 + (-1)^n H_{S/2} + 2 sum_k H_k cos(2 pi k n / S)) / S
//                                                    =>  d/dH_k = c_k/S (Gs[0] + (-1)^k Gs[S/2] + sum_{n=1}^{S/2-1} Gs[n] cos(2 pi k n / S)),
//                                                        c_k = 1 for k in {0, S/2}, 2 otherwise
// =====================================================================================================
struct NoiseBwdParams {
    const float *g;      // [B,T*R] upstream gradient
    const float *u;      // [B,T,R] the forward's uniform draw (nullable -> Philox from seed/offset, as the forward)
    float *gH;           // [B,T,F]
    int B, T, F, R, S;
    int lpf_log;         // batched kernel: log2(lanes per frame)
    uint64_t seed, offset;
    const uint64_t *offset_dev;   // nullable: the draw started at offset + *offset_dev (the forward's device counter)
};

__device__ __forceinline__ float noise_sample(const float *u, long frame, int m, int R, uint64_t seed, uint64_t offset)
{
    if (u) return u[frame * R + m] * 2.0f - 1.0f;
    const int quads = (R + 3) >> 2;
    const uint64_t ctr = offset + (uint64_t)frame * (uint64_t)quads + (uint64_t)(m >> 2);
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return philox_to_sample(r[m & 3]);
}

// Generic backward: one frame per workgroup, any hop / F.
__global__ void __launch_bounds__(256) noise_bwd_frame_kernel(NoiseBwdParams p)
{
    extern __shared__ float smem[];
    const int S = p.S, R = p.R, F = p.F, half = S >> 1;
    float *ct = smem;          // [S]
    float *x = ct + S;         // [R]
    float *g = x + R;          // [R]
    float *Gs = g + R;         // [half + 1]
    const long frame = blockIdx.x;
    const int tid = threadIdx.x;
    for (int m = tid; m < S; m += 256) ct[m] = cospif((float)(2 * m) / (float)S);
    for (int m = tid; m < R; m += 256) {
        x[m] = noise_sample(p.u, frame, m, R, p.seed, p.offset + (p.offset_dev ? *p.offset_dev : 0ull));
        g[m] = p.g[frame * R + m];
    }
    for (int n = tid; n <= half; n += 256) Gs[n] = 0.0f;
    __syncthreads();
    const int taps = min(S, R);
    for (int src = tid; src < taps; src += 256) {
        int j = (src - half) % R;
        if (j < 0) j += R;
        float acc = 0.0f;
        for (int d = 0; d + j < R; ++d) acc = __fmaf_rn(x[d], g[j + d], acc);
        const int nf = (src + half) % S;
        const int nn = nf <= half ? nf : S - nf;
        atomicAdd(&Gs[nn], acc * (0.5f - 0.5f * ct[src]));   // at most two addends per bin: order-independent
    }
    __syncthreads();
    const float invS = 1.0f / (float)S;
    for (int k = tid; k < F; k += 256) {
        float acc = 0.0f;
        int idx = 0;
        for (int n = 1; n < half; ++n) {
            idx += k;
            if (idx >= S) idx -= S;
            acc = __fmaf_rn(Gs[n], ct[idx], acc);
        }
        const float edge = Gs[0] + ((k & 1) ? -Gs[half] : Gs[half]);
        const float ck = (k == 0 || k == half) ? 1.0f : 2.0f;
        p.gH[frame * F + k] = ck * invS * (edge + acc);
    }
}

// Batched backward: 64 frames per workgroup, lane = frame (mirror of noise_batched_kernel).
__global__ void __launch_bounds__(kNT) noise_bwd_batched_kernel(NoiseBwdParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = p.S, R = p.R, F = p.F, half = S >> 1;
    const int XS = R + 4, GS = R + 12;
    const int LPF = 1 << p.lpf_log, FB = 64 >> p.lpf_log, HS = FB + 4;   // lanes per frame, frames per workgroup (as the forward)
    float *ct = smem;                                   // [S] rounded up to a multiple of 4
    float *xs = ct + ((S + 3) & ~3);                    // [FB][XS]
    float *gs = xs + FB * XS;                           // [FB][GS]  (7+ zeros after the R samples)
    float *GsT = gs + FB * GS;                          // [(half+1)][HS]  folded d/dz, transposed
    const int tid = threadIdx.x;
    const long frame0 = (long)blockIdx.x * FB;
    const long nframes = (long)p.B * p.T;
    const int nf = (int)min((long)FB, nframes - frame0);

    for (int m = tid; m < S; m += kNT) ct[m] = cospif((float)(2 * m) / (float)S);
    for (int e = tid; e < (half + 1) * HS; e += kNT) GsT[e] = 0.0f;
    for (int e = tid; e < FB * GS; e += kNT) {
        const int f = e / GS, m = e - f * GS;
        gs[e] = (f < nf && m < R) ? p.g[(frame0 + f) * R + m] : 0.0f;
    }
    if (p.u) {
        for (int e = tid; e < FB * R; e += kNT) {
            const int f = e / R, m = e - f * R;
            xs[f * XS + m] = (f < nf) ? p.u[(frame0 + f) * R + m] * 2.0f - 1.0f : 0.0f;
        }
    } else {
        const int quads = R >> 2;
        for (int e = tid; e < FB * quads; e += kNT) {
            const int f = e / quads, q = e - f * quads;
            const uint64_t ctr = p.offset + (p.offset_dev ? *p.offset_dev : 0ull) + (uint64_t)(frame0 + f) * (uint64_t)quads + (uint64_t)q;
            uint32_t r[4];
            philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
            float4 v;
            v.x = philox_to_sample(r[0]);
            v.y = philox_to_sample(r[1]);
            v.z = philox_to_sample(r[2]);
            v.w = philox_to_sample(r[3]);
            *reinterpret_cast<float4 *>(&xs[f * XS + 4 * q]) = v;
        }
    }
    __syncthreads();

    // correlation d/dkern[j] = sum_d x[d] g[j+d], 8 taps x 8 lags per step, chunk c paired with C-1-c
    const int lane = tid & 63, wv = tid >> 6;
    const int fr = lane >> p.lpf_log, sub = lane & (LPF - 1);
    const int C = R >> 3, Q = (C + LPF - 1) >> p.lpf_log;
    const int taps = min(S, R);
    const float *xrow = xs + fr * XS;
    const float *grow = gs + fr * GS;
    for (int pr = wv; 2 * pr < Q; pr += kNT / 64) {
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const int qg = side ? Q - 1 - pr : pr;
            if (side && qg == pr) break;
            const int c = qg * LPF + sub;
            if (c >= C) continue;
            const int j0 = c << 3;
            float acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
            for (int d = 0; d + j0 < R; d += 8) {
                const float4 xa = *reinterpret_cast<const float4 *>(xrow + d);
                const float4 xb = *reinterpret_cast<const float4 *>(xrow + d + 4);
                const float4 ga = *reinterpret_cast<const float4 *>(grow + j0 + d);
                const float4 gb = *reinterpret_cast<const float4 *>(grow + j0 + d + 4);
                const float4 gc = *reinterpret_cast<const float4 *>(grow + j0 + d + 8);
                const float4 gd = *reinterpret_cast<const float4 *>(grow + j0 + d + 12);
                asm volatile("" ::"v"(gd.w));  // keep the window as 4 x ds_read_b128
                const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                const float gw[16] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w, gc.x, gc.y, gc.z, gc.w, gd.x, gd.y, gd.z, gd.w};
#pragma unroll
                for (int v = 0; v < 8; ++v)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = __fmaf_rn(xv[v], gw[u + v], acc[u]);
            }
            // window, fold onto n in [0, S/2] (two taps can meet in one bin: a 2-term sum is order-independent)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                // j = (src - half) mod R with src in [0, taps)  <=>  src = (j + half) mod R when that is < taps
                int src = (j + half) % R;
                if (src < taps && ((src - half) % R + R) % R == j) {
                    const int nfull = (src + half) % S;
                    const int nn = nfull <= half ? nfull : S - nfull;
                    atomicAdd(&GsT[nn * HS + fr], acc[u] * (0.5f - 0.5f * ct[src]));
                }
            }
        }
    }
    __syncthreads();

    // d/dH_k: cosine sums over n, k = 1..S/4 paired with S/2-k, 4 frames per thread
    const float invS = 1.0f / (float)S;
    {   // k = 0 and k = S/2: plain and alternating sums, 8 lanes per frame
        const int f = min(tid >> 3, FB - 1), part = tid & 7;
        float e = 0.0f, o = 0.0f;
        for (int n = 1 + part; n < half; n += 8) {
            const float v = GsT[n * HS + f];
            if (n & 1) o += v; else e += v;
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { e += __shfl_xor(e, m); o += __shfl_xor(o, m); }
        if (part == 0 && (tid >> 3) < FB && f < nf) {
            const float v0 = GsT[f], vh = GsT[half * HS + f];
            p.gH[(frame0 + f) * F + 0] = invS * (v0 + vh + e + o);
            if (half > 0) p.gH[(frame0 + f) * F + half] = invS * (v0 + ((half & 1) ? -vh : vh) + e - o);
        }
    }
    const int nmain = half / 2;
    for (int item = tid; item < nmain * (FB / 4); item += kNT) {
        const int fq = item / nmain, k = 1 + item - fq * nmain;
        float ev[4] = {0, 0, 0, 0}, ov[4] = {0, 0, 0, 0};
        int idx = 0;
        for (int n = 1; n < half; ++n) {
            idx += k;
            if (idx >= S) idx -= S;
            const float cc = ct[idx];
            const float4 v = *reinterpret_cast<const float4 *>(&GsT[n * HS + 4 * fq]);
            if (n & 1) { ov[0] = __fmaf_rn(v.x, cc, ov[0]); ov[1] = __fmaf_rn(v.y, cc, ov[1]); ov[2] = __fmaf_rn(v.z, cc, ov[2]); ov[3] = __fmaf_rn(v.w, cc, ov[3]); }
            else       { ev[0] = __fmaf_rn(v.x, cc, ev[0]); ev[1] = __fmaf_rn(v.y, cc, ev[1]); ev[2] = __fmaf_rn(v.z, cc, ev[2]); ev[3] = __fmaf_rn(v.w, cc, ev[3]); }
        }
        const float4 v0 = *reinterpret_cast<const float4 *>(&GsT[4 * fq]);
        const float4 vh = *reinterpret_cast<const float4 *>(&GsT[half * HS + 4 * fq]);
        const float v0v[4] = {v0.x, v0.y, v0.z, v0.w}, vhv[4] = {vh.x, vh.y, vh.z, vh.w};
        const int k2 = half - k;
        const float sg1 = (k & 1) ? -1.0f : 1.0f, sg2 = (k2 & 1) ? -1.0f : 1.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 4 * fq + q;
            if (f < nf) {
                p.gH[(frame0 + f) * F + k] = 2.0f * invS * (v0v[q] + sg1 * vhv[q] + ev[q] + ov[q]);
                if (k2 != k) p.gH[(frame0 + f) * F + k2] = 2.0f * invS * (v0v[q] + sg2 * vhv[q] + ev[q] - ov[q]);
            }
        }
    }
}

size_t bwd_batched_lds_bytes(int F, int R, int lpf_log)
{
    const int S = 2 * (F - 1), FB = 64 >> lpf_log;
    return sizeof(float) * (((S + 3) & ~3) + (size_t)FB * (R + 4) + (size_t)FB * (R + 12) + (size_t)(S / 2 + 1) * (FB + 4));
}

int pick_bwd_lpf_log(int F, int R)
{
    for (int limit : {48 * 1024, 80 * 1024, 160 * 1024})
        for (int l = 0; l <= 3; ++l)
            if (bwd_batched_lds_bytes(F, R, l) <= (size_t)limit) return l;
    return -1;
}

size_t batched_lds_bytes(int F, int R, int lpf_log)
{
    const int S = 2 * (F - 1), FB = 64 >> lpf_log;
    const size_t ua = (size_t)(FB + 4) * F, ub = (size_t)FB * (R + 12);
    const size_t un = ua > ub ? ua : ub;
    return sizeof(float) * (((S + 3) & ~3) + (size_t)FB * (R + 4) + un);
}

// Lanes per frame (log2) of the batched forward kernel: 64 frames per workgroup when the tile fits in ~half the
// CU's LDS (two workgroups per CU), else 32 / 16 frames; -1 when even 16 frames do not fit (generic kernel then).
int pick_lpf_log(int F, int R, int mode)
{
    if (mode >> 8) return (mode >> 8) - 1;  // tuning: ddsp_noise_set_generic((l + 1) << 8)
    // measured (hop 128, F 65): 32 frames / 35 KB per workgroup (4 workgroups per CU) beats 64 frames / 70 KB by 14 %
    for (int l = 0; l <= 3; ++l)
        if (batched_lds_bytes(F, R, l) <= 40 * 1024) return l;
    for (int l = 0; l <= 3; ++l)
        if (batched_lds_bytes(F, R, l) <= 80 * 1024) return l;
    for (int l = 0; l <= 3; ++l)
        if (batched_lds_bytes(F, R, l) <= 160 * 1024) return l;
    return -1;
}

}  // namespace

namespace {
int noise_forward_impl(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop, uint64_t seed,
                       uint64_t offset, const uint64_t *offset_dev, int accumulate, void *workspace, size_t workspace_bytes,
                       void *stream);
// where the whole-batch matrix product pays for its extra launches (cosine operand + product; measured crossovers at 195 bands,
// hop 512: forward between 2 752 and 5 504 frames, backward below 688): the real-time callback's 4 frames and the reference's
// own training batch (16 x 172 frames) keep the cosine sums in the forward
constexpr long kIrProductMinFramesFwd = 4096, kIrProductMinFramesBwd = 512;
}

extern "C" int ddsp_noise_forward(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop,
                                  uint64_t seed, uint64_t offset, int accumulate, void *stream)
{
    return noise_forward_impl(Hmag, uniform, y, B, T, F, hop, seed, offset, nullptr, accumulate, nullptr, 0, stream);
}

extern "C" int ddsp_noise_forward_counter(const float *Hmag, float *y, int B, int T, int F, int hop, uint64_t seed,
                                          const uint64_t *counter_dev, int accumulate, void *stream)
{
    if (!counter_dev) return DDSP_EINVAL;
    return noise_forward_impl(Hmag, nullptr, y, B, T, F, hop, seed, 0, counter_dev, accumulate, nullptr, 0, stream);
}

extern "C" size_t ddsp_noise_workspace_bytes(int B, int T, int F, int hop)
{
    if (B <= 0 || T <= 0 || F < 2 || hop <= 0) return 0;
    const long frames = (long)B * T;
    return (ir_product_shape(F, hop) && frames >= kIrProductMinFramesBwd) ? ir_workspace_bytes(frames, F) : 0;
}

extern "C" int ddsp_noise_forward_ws(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop, uint64_t seed,
                                     uint64_t offset, const uint64_t *counter_dev, int accumulate, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    if (uniform && counter_dev) return DDSP_EINVAL;
    return noise_forward_impl(Hmag, uniform, y, B, T, F, hop, seed, offset, counter_dev, accumulate, workspace, workspace_bytes, stream);
}

namespace {
int noise_forward_impl(const float *Hmag, const float *uniform, float *y, int B, int T, int F, int hop, uint64_t seed,
                       uint64_t offset, const uint64_t *offset_dev, int accumulate, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (B == 0) return 0;
    if (!Hmag || !y || B < 0 || T <= 0 || F < 2 || hop <= 0) return DDSP_EINVAL;
    NoiseParams p;
    p.Hm = Hmag; p.u = uniform; p.y = y;
    p.B = B; p.T = T; p.F = F; p.R = hop; p.S = 2 * (F - 1);
    p.seed = seed; p.offset = offset; p.offset_dev = offset_dev; p.accumulate = accumulate; p.lpf_log = 0;
    p.zrows = nullptr; p.zs = 0;
    if ((long)B * T >= (1L << 31)) return DDSP_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    const int mode = g_force_generic.load(std::memory_order_relaxed);
    // 195 bands at hop 512 (the reference's default shape) with a workspace: the impulse responses of the whole batch as one
    // matrix product (ddsp_noise_ir.hip), which the FFT form below then reads instead of summing cosines; mode bit 4 (tests, A/B) keeps the sums
    if (workspace && !(mode & (3 | 16)) && ir_product_shape(F, hop) && (long)B * T >= kIrProductMinFramesFwd &&
        workspace_bytes >= ir_workspace_bytes((long)B * T, F) && ((uintptr_t)workspace % 16) == 0) {
        hipError_t ie = hipSuccess;
        p.zrows = launch_noise_ir(Hmag, (long)B * T, F, workspace, s, &ie);
        if (!p.zrows) return (int)ie;
        p.zs = ir_row_stride(F);
    }
    // hop 512: the in-LDS FFT form (ddsp_noise_fft.hip); mode bit 1 (tests, A/B) keeps the direct forms, bit 2 takes the
    // FFT form for hop 256 as well (correct there too, just not faster)
    if (!(mode & 3)) {
        hipError_t fe = hipSuccess;
        if (launch_noise_fft(p, s, (mode & 4) != 0, &fe)) return (int)fe;
    }
    // hop 128 / 65 bands (the 16 kHz configurations): the wavefront-private form (ddsp_noise_wave.hip); mode bit 3 keeps the batched kernel
    if (!(mode & (1 | 8))) {
        hipError_t we = hipSuccess;
        const long done = launch_noise_wave(p, s, &we);
        if (done < 0) return (int)we;
        if (done == (long)B * T) return 0;
        if (done > 0) {                                       // a remainder of fewer than 16 frames: the kernels below, same counters
            p.Hm += done * F;
            if (p.u) p.u += done * hop;
            p.y += done * hop;
            p.offset += (uint64_t)done * (uint64_t)((hop + 3) / 4);
            B = 1;
            T = (int)((long)p.B * p.T - done);
            p.B = B;
            p.T = T;
        }
    }
    const int lpf_log = pick_lpf_log(F, hop, mode);
    // (the batched kernel stores whole float4s: an output buffer that is not 16-byte aligned takes the generic kernel)
    if (!(mode & 1) && hop % 8 == 0 && lpf_log >= 0 && ((uintptr_t)y % 16) == 0) {
        const size_t blds = batched_lds_bytes(F, hop, lpf_log);
        p.lpf_log = lpf_log;
        const int fb = 64 >> lpf_log;
        const long blocks = ((long)B * T + fb - 1) / fb;
        const bool spow2 = (p.S & (p.S - 1)) == 0;
        hipError_t le = hipSuccess;
#define DDSP_NOISE_LAUNCH(L, P2)                                                                                       \
        do {                                                                                                           \
            static bool attr_set[64] = {};                                                                             \
            le = ddsp_allow_big_lds((const void *)noise_batched_kernel<L, P2>, attr_set);                              \
            if (le != hipSuccess) return (int)le;                                                                      \
            const int slot = ddsp_prof::begin(ddsp_prof::NOISE, s);                                                    \
            hipLaunchKernelGGL((noise_batched_kernel<L, P2>), dim3((unsigned)blocks), dim3(kNT), blds, s, p);          \
            ddsp_prof::end(slot, s);                                                                                   \
        } while (0)
        switch (lpf_log * 2 + (spow2 ? 1 : 0)) {
            case 0: DDSP_NOISE_LAUNCH(0, false); break;
            case 1: DDSP_NOISE_LAUNCH(0, true); break;
            case 2: DDSP_NOISE_LAUNCH(1, false); break;
            case 3: DDSP_NOISE_LAUNCH(1, true); break;
            case 4: DDSP_NOISE_LAUNCH(2, false); break;
            case 5: DDSP_NOISE_LAUNCH(2, true); break;
            case 6: DDSP_NOISE_LAUNCH(3, false); break;
            default: DDSP_NOISE_LAUNCH(3, true); break;
        }
#undef DDSP_NOISE_LAUNCH
        return (int)hipGetLastError();
    }
    const size_t lds = sizeof(float) * ((size_t)F + p.S + 2 * (size_t)hop);
    if (lds > 160 * 1024) return DDSP_ERANGE;
    {
        static bool attr_set[64] = {};
        const hipError_t ae = ddsp_allow_big_lds((const void *)noise_frame_kernel, attr_set);
        if (ae != hipSuccess) return (int)ae;
    }
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE, s);
    hipLaunchKernelGGL(noise_frame_kernel, dim3((unsigned)((long)B * T)), dim3(256), lds, s, p);
    ddsp_prof::end(slot, s);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int ddsp_noise_set_generic(int on)
{
    if (on != 0 && !ddsp_hooks_on()) return DDSP_EPERM;
    g_force_generic.store(on, std::memory_order_relaxed);
    return 0;
}

static int noise_backward_impl(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop,
                               uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *workspace, size_t workspace_bytes,
                               void *stream);

extern "C" int ddsp_noise_backward(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop,
                                   uint64_t seed, uint64_t offset, void *stream)
{
    return noise_backward_impl(grad_y, uniform, grad_H, B, T, F, hop, seed, offset, nullptr, nullptr, 0, stream);
}

extern "C" int ddsp_noise_backward_counter(const float *grad_y, float *grad_H, int B, int T, int F, int hop, uint64_t seed,
                                           const uint64_t *counter_dev, void *stream)
{
    if (!counter_dev) return DDSP_EINVAL;
    return noise_backward_impl(grad_y, nullptr, grad_H, B, T, F, hop, seed, 0, counter_dev, nullptr, 0, stream);
}

extern "C" int ddsp_noise_backward_ws(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop, uint64_t seed,
                                      uint64_t offset, const uint64_t *counter_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (uniform && counter_dev) return DDSP_EINVAL;
    return noise_backward_impl(grad_y, uniform, grad_H, B, T, F, hop, seed, offset, counter_dev, workspace, workspace_bytes, stream);
}

static int noise_backward_impl(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop,
                               uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *workspace, size_t workspace_bytes,
                               void *stream)
{
    if (B == 0) return 0;
    if (!grad_y || !grad_H || B < 0 || T <= 0 || F < 2 || hop <= 0) return DDSP_EINVAL;
    if ((long)B * T >= (1L << 31)) return DDSP_ERANGE;
    NoiseBwdParams p;
    p.g = grad_y; p.u = uniform; p.gH = grad_H;
    p.B = B; p.T = T; p.F = F; p.R = hop; p.S = 2 * (F - 1);
    p.seed = seed; p.offset = offset; p.offset_dev = offset_dev;
    hipStream_t s = (hipStream_t)stream;
    if (!(g_force_generic.load(std::memory_order_relaxed) & 3)) {    // hop 512: correlation in the in-LDS FFT form (mode bits 0 / 1 keep the direct forms)
        hipError_t fe = hipSuccess;
        // (a workspace is used only by the shapes of ddsp_noise_workspace_bytes; mode bit 4 keeps the direct kernels there)
        const bool ws_ok = workspace && !(g_force_generic.load(std::memory_order_relaxed) & 16) && ir_product_shape(F, hop) &&
                           (long)B * T >= kIrProductMinFramesBwd && workspace_bytes >= ir_workspace_bytes((long)B * T, F) &&
                           ((uintptr_t)workspace % 16) == 0;
        if (launch_noise_fft_backward(grad_y, uniform, grad_H, B, T, F, hop, seed, offset, offset_dev, ws_ok ? workspace : nullptr, s, &fe))
            return (int)fe;
    }
    const int lpf_log = pick_bwd_lpf_log(F, hop);
    p.lpf_log = lpf_log < 0 ? 0 : lpf_log;
    if (!(g_force_generic.load(std::memory_order_relaxed) & 1) && hop % 8 == 0 && lpf_log >= 0) {
        const size_t blds = bwd_batched_lds_bytes(F, hop, lpf_log);
        static bool attr_set[64] = {};
        const hipError_t ae = ddsp_allow_big_lds((const void *)noise_bwd_batched_kernel, attr_set);
        if (ae != hipSuccess) return (int)ae;
        const int fb = 64 >> lpf_log;
        const long blocks = ((long)B * T + fb - 1) / fb;
        hipLaunchKernelGGL(noise_bwd_batched_kernel, dim3((unsigned)blocks), dim3(kNT), blds, s, p);
        return (int)hipGetLastError();
    }
    const size_t lds = sizeof(float) * ((size_t)p.S + 2 * (size_t)hop + p.S / 2 + 1);
    if (lds > 160 * 1024) return DDSP_ERANGE;
    {
        static bool attr_set[64] = {};
        const hipError_t ae = ddsp_allow_big_lds((const void *)noise_bwd_frame_kernel, attr_set);
        if (ae != hipSuccess) return (int)ae;
    }
    hipLaunchKernelGGL(noise_bwd_frame_kernel, dim3((unsigned)((long)B * T)), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}
