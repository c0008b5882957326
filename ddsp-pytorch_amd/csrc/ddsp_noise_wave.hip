// Filtered noise for hop 128 / 65 bands (the 16 kHz configurations of BASELINE.json: cfg2, cfg4, cfg5), round-3 form.
// Same arithmetic contract as ddsp_noise.hip (model/ddsp/filtered_noise.py:7-53); different organisation (noise section):
//
//   * ONE WAVEFRONT owns a group of 16 frames at a time and shares nothing with other wavefronts: no workgroup barrier anywhere
//     (64-thread workgroups, 17.4 KB of LDS each, eight per CU = two per SIMD; LDS operations of one wavefront execute in order,
//     which is all the hand-offs between the stages need).  Persistent: a wavefront walks groups g, g + grid, ...
//   * the NEXT group's filter magnitudes (16 x 65 floats, contiguous) are fetched into registers while this group is convolved.
//   * impulse responses (:8-20): z = irfft(H) is a product with ONE cosine matrix shared by every frame,
//         E'[f][n] = sum_{k even} w_k H[f][k] cos(2 pi k n / 128)   (w = 1/2 for k in {0, 64}),   O[f][n] = sum_{k odd} H[f][k] cos(2 pi k n / 128),
//         z[n] = (E' + O) / 64,  z[64 - n] = (E' - O) / 64,  n = 0..31   (cos(2 pi k (64 - n) / 128) = (-1)^k cos(2 pi k n / 128)),
//     i.e. a [16 frames x 33] x [33 x 32] and a [16 x 32] x [32 x 32] fp32 product: v_mfma_f32_16x16x4_f32 with the cosine
//     operand resident in registers for the whole kernel (exact fp32 fused multiply-adds in k order: the matrix cores are used
//     because this step IS a dense contraction with a shared operand; they run beside the vector pipe of the SIMD's other
//     wavefront).  The result layout puts a fixed tap index n on every lane: the PRODUCING lane windows z (periodic Hann,
//     :15) and stores it at both of its wrapped positions (:14,:19-20) -- no scatter pass, no table.  n = 32 is one signed sum.
//   * noise (:44-48): the injected draw or Philox4x32-10, counter layout of ddsp_noise_common.h (streams identical to every other form).
//   * truncated convolution (:25-32), register blocked 8 outputs x 8 taps (6 ds_read_b128 per 64 multiply-adds).  Lanes are
//     (frame, quarter): the lanes that share a ds_read_b128 service group are 16 DIFFERENT frames of the same quarter, and the
//     row strides are 4 x odd, so every read is bank-conflict-free whatever the quarters are working on.  A quarter owns the
//     output chunks {15-s, s} then {8+s, 7-s}: 17 blocks per pass for every lane (uniform trip count), the switch between a
//     pass's two chunks is the only divergent instruction group (a register move).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_noise_common.h"

using namespace ddsp_noise;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int R = 128;            // hop = samples per frame
constexpr int F = 65;             // bands; S = 2 (F - 1) = 128 = R: the impulse response fills the frame exactly
constexpr int FG = 16;            // frames per group
constexpr int KS = 132;           // kern row stride (4 x 33)
constexpr int XS = 140;           // noise row stride: 8 leading zeros + 128 samples + 4 (4 x 35)
constexpr int kLdsFloats = FG * KS + FG * XS;

#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// even-bin / odd-bin index a lane (quarter q = lane >> 4) feeds into K-step s of the products; chosen so that the two
// quarters that share a ds_read_b32 service group read bins 16 apart (row stride 65: different banks for all 32 lanes)
__device__ __forceinline__ int bin_index(int s, int q) { return 16 * (s >> 2) + 2 * (s & 3) + (q >> 1) + 8 * (q & 1); }

__global__ void __launch_bounds__(64) noise_wave_kernel(NoiseParams p, long ngroups)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *kern = smem;                       // [FG][KS]; its first FG*F floats double as the H tile between two groups
    float *xs = smem + FG * KS;               // [FG][XS]
    float *Hs = kern;
    const int lane = threadIdx.x;
    const long nframes = (long)p.B * p.T;

    // ---- per-lane constants ------------------------------------------------------------------------------------
    // matrix-core operand lanes: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[4 (lane >> 4) + r][lane & 15]
    const int mi = lane & 15, mq = lane >> 4;
    float Be[2][9], Bo[2][8];                 // cosine operand, tile t: n = 16 t + mi
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + mi;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int e = (s < 8) ? bin_index(s, mq) : (mq == 0 ? 32 : -1);
            const float w = (e < 0) ? 0.0f : ((e == 0 || e == 32) ? 0.5f : 1.0f);
            Be[t][s] = w * cospif((float)((2 * (e < 0 ? 0 : e) * n) & 127) * (1.0f / 64.0f));
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int o = bin_index(s, mq);
            Bo[t][s] = cospif((float)(((2 * o + 1) * n) & 127) * (1.0f / 64.0f));
        }
    }
    // window weights of the taps this lane produces: z[n] * win(n), z[64 - n] * win(64 - n), win(m) = 0.5 + 0.5 cos(2 pi m / 128),
    // with the 1/64 of the transform folded in
    float w1[2], w2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + mi;
        w1[t] = __fmaf_rn(0.5f, cospif((float)n * (1.0f / 64.0f)), 0.5f) * (1.0f / 64.0f);
        w2[t] = __fmaf_rn(0.5f, cospif((float)(64 - n) * (1.0f / 64.0f)), 0.5f) * (1.0f / 64.0f);
    }
    // convolution lanes: (quarter cs, frame cf) with the 16 lanes of a ds_read_b128 service group = 16 frames of one quarter
    const int l5 = lane & 31, seg = l5 >> 2;
    const int cf = ((seg >> 1) << 2) | (l5 & 3);
    const int cs = 2 * (lane >> 5) + ((0x96 >> seg) & 1);
    const float *krow = kern + cf * KS;
    const float *xrow = xs + cf * XS + 8;

    for (int e = lane; e < FG * 8; e += 64) xs[(e >> 3) * XS + (e & 7)] = 0.0f;      // causal padding, written once
    const uint64_t base_off = p.offset + (p.offset_dev ? *p.offset_dev : 0ull);

    // the H tile of a group: FG * F = 1040 contiguous floats = 260 float4 (group bases are 16-byte aligned: 4160 bytes apart)
    auto load_tile = [&](long g, float4 (&h)[5]) {
        const long first = g * FG;
        const long avail = (nframes - first) * F;                 // floats of the tile that exist (ragged last group)
        const float *src = p.Hm + first * F;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q4 = lane + 64 * j;                         // float4 index inside the tile
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (q4 < 260) {
                if (4L * q4 + 3 < avail) v = *reinterpret_cast<const float4 *>(src + 4 * q4);
                else if (4L * q4 < avail) {
                    float tmp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    for (int c = 0; c < 4; ++c) if (4L * q4 + c < avail) tmp[c] = src[4 * q4 + c];
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            h[j] = v;
        }
    };

    float4 hreg[5];
    long g = blockIdx.x;
    if (g < ngroups) load_tile(g, hreg);

    for (; g < ngroups; g += gridDim.x) {
        const long frame0 = g * FG;
        const int nf = (int)min((long)FG, nframes - frame0);

        // ---- 0. H tile -> LDS (over the previous group's impulse responses: this wavefront is done with them) -------
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q4 = lane + 64 * j;
            if (q4 < 260) *reinterpret_cast<float4 *>(Hs + 4 * q4) = hreg[j];
        }
        DDSP_WAVE_ORDER();
        if (g + gridDim.x < ngroups) load_tile(g + gridDim.x, hreg);          // in flight during everything below

        // ---- 1. impulse responses on the matrix cores ---------------------------------------------------------------
        float Ae[9], Ao[8];
        {
            const float *hrow = Hs + mi * F;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int b = bin_index(s, mq);
                Ae[s] = hrow[2 * b];
                Ao[s] = hrow[2 * b + 1];
            }
            Ae[8] = (mq == 0) ? hrow[64] : 0.0f;
        }
        DDSP_WAVE_ORDER();
        v4f accE[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}}, accO[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            accE[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ae[s], Be[0][s], accE[0], 0, 0, 0);
            accE[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ae[s], Be[1][s], accE[1], 0, 0, 0);
            if (s < 8) {
                accO[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ao[s], Bo[0][s], accO[0], 0, 0, 0);
                accO[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ao[s], Bo[1][s], accO[1], 0, 0, 0);
            }
        }
        // n = 32: cos(2 pi k 32 / 128) = cos(pi k / 2): (-1)^(k/2) for even k, 0 for odd k; the sign of this lane's even bins is
        // that of its quarter ((mq >> 1) odd <=> bin_index odd)
        float z32;
        {
            float t = 0.5f * Ae[0];
            if (mq != 0) t = Ae[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) t += Ae[s];
            t += 0.5f * Ae[8];
            if (mq & 2) t = -t;
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            z32 = t * (0.5f / 64.0f);                              // win(32) = 0.5
        }
        // taps: lane (n = 16 t + mi) holds frames 4 mq + r.  z[n] -> kern[n], kern[128 - n]; z[64 - n] -> kern[64 - n], kern[64 + n]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 16 * t + mi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *kr = kern + (4 * mq + r) * KS;
                const float v1 = (accE[t][r] + accO[t][r]) * w1[t];
                const float v2 = (accE[t][r] - accO[t][r]) * w2[t];
                kr[n] = v1;
                if (n != 0) { kr[128 - n] = v1; kr[64 - n] = v2; }
                kr[64 + n] = v2;                                  // n = 0: win(64) = 0, the reference's hann[0]
            }
        }
        if (mq == 0) { kern[mi * KS + 32] = z32; kern[mi * KS + 96] = z32; }
        DDSP_WAVE_ORDER();

        // ---- 2. noise -> xs[frame][8 + m] -------------------------------------------------------------------------------
        if (p.u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qi = lane + 64 * j, fr = qi >> 5, q = qi & 31;
                float4 v = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
                if (fr < nf) v = *reinterpret_cast<const float4 *>(p.u + (frame0 + fr) * R + 4 * q);
                v.x = v.x * 2.0f - 1.0f; v.y = v.y * 2.0f - 1.0f; v.z = v.z * 2.0f - 1.0f; v.w = v.w * 2.0f - 1.0f;
                *reinterpret_cast<float4 *>(xs + fr * XS + 8 + 4 * q) = v;
            }
        } else {
            const uint64_t c0 = base_off + (uint64_t)frame0 * 32u + (uint64_t)lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qi = lane + 64 * j, fr = qi >> 5, q = qi & 31;
                const uint64_t ctr = c0 + 64u * j;
                uint32_t rr[4];
                philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rr);
                float4 v;
                v.x = philox_to_sample(rr[0]); v.y = philox_to_sample(rr[1]); v.z = philox_to_sample(rr[2]); v.w = philox_to_sample(rr[3]);
                *reinterpret_cast<float4 *>(xs + fr * XS + 8 + 4 * q) = v;
            }
        }
        DDSP_WAVE_ORDER();

        // ---- 3. truncated convolution -----------------------------------------------------------------------------------
        float *yrow = p.y + (frame0 + cf) * R;
        const bool live = cf < nf;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int cA = pass ? 8 + cs : 15 - cs, cB = pass ? 7 - cs : cs;      // cA + cB = 15: (cA + 1) + (cB + 1) = 17 blocks
            float4 ya0, ya1, yb0, yb1;
            if (p.accumulate && live) {                                          // read early, used at the end of the pass
                ya0 = *reinterpret_cast<const float4 *>(yrow + 8 * cA); ya1 = *reinterpret_cast<const float4 *>(yrow + 8 * cA + 4);
                yb0 = *reinterpret_cast<const float4 *>(yrow + 8 * cB); yb1 = *reinterpret_cast<const float4 *>(yrow + 8 * cB + 4);
            }
            float acc[8], first[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[u] = 0.0f; first[u] = 0.0f; }
            const float *kp = krow;
            const float *xp = xrow + 8 * cA - 8;                  // window [n0 - j - 8, n0 - j + 8)
            const int tsw = cA + 1;
#pragma unroll 1
            for (int t = 0; t < 17; ++t) {
                if (t == tsw) {                                   // chunk A done for this lane: park it, start chunk B
#pragma unroll
                    for (int u = 0; u < 8; ++u) { first[u] = acc[u]; acc[u] = 0.0f; }
                    kp = krow;
                    xp = xrow + 8 * cB - 8;
                }
                const float4 ka = *reinterpret_cast<const float4 *>(kp);
                const float4 kb = *reinterpret_cast<const float4 *>(kp + 4);
                const float4 xa = *reinterpret_cast<const float4 *>(xp);
                const float4 xb = *reinterpret_cast<const float4 *>(xp + 4);
                const float4 xc = *reinterpret_cast<const float4 *>(xp + 8);
                const float4 xd = *reinterpret_cast<const float4 *>(xp + 12);
                asm volatile("" ::"v"(xa.x));                     // keep the window as 4 x ds_read_b128
                const float kv[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
                const float xw[16] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w, xd.x, xd.y, xd.z, xd.w};
#pragma unroll
                for (int v = 0; v < 8; ++v)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = __fmaf_rn(kv[v], xw[8 + u - v], acc[u]);
                kp += 8;
                xp -= 8;
            }
            if (live) {
                float4 a0 = make_float4(first[0], first[1], first[2], first[3]), a1 = make_float4(first[4], first[5], first[6], first[7]);
                float4 b0 = make_float4(acc[0], acc[1], acc[2], acc[3]), b1 = make_float4(acc[4], acc[5], acc[6], acc[7]);
                if (p.accumulate) {
                    a0.x += ya0.x; a0.y += ya0.y; a0.z += ya0.z; a0.w += ya0.w; a1.x += ya1.x; a1.y += ya1.y; a1.z += ya1.z; a1.w += ya1.w;
                    b0.x += yb0.x; b0.y += yb0.y; b0.z += yb0.z; b0.w += yb0.w; b1.x += yb1.x; b1.y += yb1.y; b1.z += yb1.z; b1.w += yb1.w;
                }
                *reinterpret_cast<float4 *>(yrow + 8 * cA) = a0; *reinterpret_cast<float4 *>(yrow + 8 * cA + 4) = a1;
                *reinterpret_cast<float4 *>(yrow + 8 * cB) = b0; *reinterpret_cast<float4 *>(yrow + 8 * cB + 4) = b1;
            }
        }
        DDSP_WAVE_ORDER();
    }
}

}  // namespace

namespace ddsp_noise {

bool launch_noise_wave(const NoiseParams &p, hipStream_t s, hipError_t *err)
{
    if (p.R != R || p.F != F || p.S != R) return false;
    if (((uintptr_t)p.y % 16) != 0 || ((uintptr_t)p.Hm % 16) != 0 || (p.u && ((uintptr_t)p.u % 16) != 0)) return false;
    const long nframes = (long)p.B * p.T;
    const long ngroups = (nframes + FG - 1) / FG;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { *err = e; return true; }
    static int cached[64] = {};
    if (!cached[dev & 63]) {
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) { *err = e; return true; }
        cached[dev & 63] = cus;
    }
    cus = cached[dev & 63];
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    const long resident = (long)cus * 8;                          // 17.4 KB each: eight wavefronts per CU, two per SIMD
    const long grid = ngroups < resident ? ngroups : resident;
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE, s);
    hipLaunchKernelGGL(noise_wave_kernel, dim3((unsigned)grid), dim3(64), lds, s, p, ngroups);
    ddsp_prof::end(slot, s);
    *err = hipGetLastError();
    return true;
}

}  // namespace ddsp_noise
