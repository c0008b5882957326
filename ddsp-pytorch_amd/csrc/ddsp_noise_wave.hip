// Filtered noise for hop 128 / 65 bands (the 16 kHz configurations of BASELINE.json: cfg2, cfg4, cfg5), round-3 form.
// Same arithmetic contract as ddsp_noise.hip (model/ddsp/filtered_noise.py:7-53); different organisation (noise section):
//
//   * ONE WAVEFRONT owns a group of 16 frames at a time and shares nothing with other wavefronts: no workgroup barrier anywhere
//     (64-thread workgroups, 17.4 KB of LDS each, four per CU by default -- launch_noise_wave says why not eight; LDS operations of one wavefront execute in order,
//     which is all the hand-offs between the stages need).  Persistent: a wavefront walks groups g, g + grid, ...
//   * software pipeline over the groups, written so that every global access sits in straight-line code (the compiler's
//     `s_waitcnt vmcnt` are then exact counts, not drains): the NEXT group's filter magnitudes (16 x 65 contiguous floats) are
//     fetched into registers before this group's noise draw and convolution; when accumulating, this group's 8 KB of y is read
//     (whole lines) between the two convolution passes and added + stored inside the next group's impulse-response step.
//   * impulse responses (:8-20): z = irfft(H) is a product with ONE cosine matrix shared by every frame,
//         E'[f][n] = sum_{k even} w_k H[f][k] cos(2 pi k n / 128)   (w = 1/2 for k in {0, 64}),   O[f][n] = sum_{k odd} H[f][k] cos(2 pi k n / 128),
//         z[n] = (E' + O) / 64,  z[64 - n] = (E' - O) / 64,  n = 0..31   (cos(2 pi k (64 - n) / 128) = (-1)^k cos(2 pi k n / 128)),
//     i.e. a [16 frames x 32] x [32 x 32] product for each parity (bin 64 is a rank-one term added by the tap stage).  The
//     contraction has a shared operand, so it runs on the matrix cores -- as SPLIT bf16: every fp32 value is hi + mid + lo, three
//     bf16 terms that represent it exactly (8 + 8 + 8 significand bits), and a product is the six cross terms of weight >= 2^-16
//     (hi hi, hi mid, mid hi, hi lo, mid mid, lo hi) of v_mfma_f32_16x16x32_bf16, fp32 accumulate: 24 instructions per group and
//     the dropped terms are below 2^-26 |a||b|, a quarter of an fp32 rounding.  (The fp32 matrix instruction,
//     v_mfma_f32_16x16x4_f32 x 34, was exact in k order but does NOT run beside the vector pipe -- measured,
//     tools/microbench/int_mfma_rates.hip -- and cost 1088 cycles per group; the split form holds the issue port a quarter of
//     that and its splitting arithmetic is ~90 vector instructions.  Measured same box: 0.1035 -> 0.0994 ms.)  The cosine
//     operand's split is resident in registers for the whole kernel.  The result layout puts a
//     fixed tap index n on every lane: the PRODUCING lane windows z (periodic Hann, :15) and stores it at both of its wrapped
//     positions (:14,:19-20) -- no scatter pass, no table.  n = 32 is one signed sum.
//   * noise (:44-48): the injected draw or Philox4x32-10, counter layout of ddsp_noise_common.h (streams identical to every other form).
//   * truncated convolution (:25-32), register blocked 8 outputs x 8 taps (6 ds_read_b128 per 64 multiply-adds), operands of the
//     next block read while this one is multiplied.  Lanes are (frame, quarter): the lanes that share a ds_read_b128 service
//     group are 16 DIFFERENT frames of the same quarter, and the row strides are 4 x odd, so every read is bank-conflict-free
//     whatever the quarters are working on.  A quarter owns the output chunks {15-s, s} then {8+s, 7-s}: 17 blocks per pass for
//     every lane (uniform trip count); the switch between a pass's two chunks is the only divergent instruction group.
//   * output: the group's 16 x 128 results are contiguous in y; staged through LDS (over the noise tile) and stored as whole lines.
//
// Only whole groups: launch_noise_wave() hands a remainder of fewer than 16 frames to the batched kernel (same Philox counters).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_noise_common.h"

using namespace ddsp_noise;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// register-resident aggregates handed between the steps BY VALUE (arrays passed by reference to the step lambdas ended up in
// scratch memory, whose loads share the vector-memory counter with the prefetches)
// (v4f, the compiler's own vector type, not HIP's float4 class: aggregates of the latter are not split into registers)
struct Tile { v4f v[5]; };           // a group's 16 x 65 filter magnitudes, 260 x 16 bytes over 64 lanes
struct Lines { v4f v[8]; };          // a group's 16 x 128 outputs, 512 x 16 bytes over 64 lanes
struct Pass { float v[16]; };        // a lane's two output chunks of one convolution pass
struct Spectra { v4f e[2], o[2], h64; float z32; };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split { bf16x8 p[3]; };       // eight fp32 values as three bf16 terms each, hi + mid + lo == the value exactly
// x = hi + mid + lo with every term a bf16: both residuals are exact in fp32 (8 + 8 + 8 significand bits)
__device__ __forceinline__ void split3(float x, Split &d, int j)
{
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    d.p[0][j] = hi; d.p[1][j] = mid; d.p[2][j] = (__bf16)r2;
}

constexpr int R = 128;            // hop = samples per frame
constexpr int F = 65;             // bands; S = 2 (F - 1) = 128 = R: the impulse response fills the frame exactly
constexpr int FG = 16;            // frames per group
constexpr int KS = 132;           // kern / staging row stride (4 x 33)
constexpr int XS = 140;           // noise row stride: 8 leading zeros + 128 samples + 4 (4 x 35)
constexpr int kLdsFloats = FG * KS + FG * XS;

// DDSP_NOISE_ABL (tools/build_variant.sh): timing-only ablation builds (results wrong): 1 = one convolution block per pass
// instead of 17, 2 = no Philox rounds, 3 = no matrix-core products, 4 = no read / write of y
#ifndef DDSP_NOISE_ABL
#define DDSP_NOISE_ABL 0
#endif

// DDSP_NOISE_STAMPS (tools/build_variant.sh ... -DDDSP_NOISE_STAMPS): per-phase cycle counts of every wavefront, summed over its
// groups, written to the buffer passed as the (then unused) injected draw: [grid][8] uint64 = setup, first convolution pass,
// second pass, staging, H->LDS + operand reads + products, pending output, taps, noise.  tools/microbench/noise_stamps.py reads them.
#ifdef DDSP_NOISE_STAMPS
#define DDSP_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                           stamp_sum[i] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define DDSP_STAMP(i) do { } while (0)
#endif

#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

template <bool ACC>   // ACC: add to the output buffer's contents (harmonics + noise, decoder.py:132) instead of overwriting them
__global__ void __launch_bounds__(64, 2) noise_wave_kernel(NoiseParams p, long ngroups)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *kern = smem;                       // [FG][KS]; its first FG*F floats double as the H tile between two groups
    float *xs = smem + FG * KS;               // [FG][XS]; doubles as the output staging tile [FG][KS] between two groups
    float *Hs = kern, *ys = xs;
    const int lane = threadIdx.x;
#ifdef DDSP_NOISE_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
    unsigned long long *stamp_out = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.u)) + 8 * blockIdx.x;
    p.u = nullptr;
#endif

    // ---- per-lane constants, resident for the whole kernel -------------------------------------------------------------
    // matrix-core operand lanes: A[i = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][n = lane & 15], D[4 (lane >> 4) + r][lane & 15]
    const int mi0 = lane & 15, mq0 = lane >> 4;
    // the 128 distinct values cos(2 pi m / 128), two per lane, through LDS once (38 library cosines per lane cost a tenth of the
    // kernel's time at eight groups per wavefront)
    float *ctab = xs;
    ctab[lane] = cospif((float)lane * (1.0f / 64.0f));
    ctab[lane + 64] = cospif((float)(lane + 64) * (1.0f / 64.0f));
    DDSP_WAVE_ORDER();
    // cosine operand of v_mfma_f32_16x16x32_bf16, tile t (n = 16 t + mi): B[k = 8 mq + j][n], k = the even bin 2k / the odd bin 2k + 1
    Split Be[2], Bo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + mi0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * mq0 + j;
            split3((k == 0 ? 0.5f : 1.0f) * ctab[(2 * k * n) & 127], Be[t], j);
            split3(ctab[((2 * k + 1) * n) & 127], Bo[t], j);
        }
    }
    // window weights of the taps this lane produces: z[n] * win(n), z[64 - n] * win(64 - n), win(m) = 0.5 + 0.5 cos(2 pi m / 128),
    // with the 1/64 of the transform folded in
    float w1[2], w2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + mi0;
        w1[t] = __fmaf_rn(0.5f, ctab[n], 0.5f) * (1.0f / 64.0f);
        w2[t] = __fmaf_rn(0.5f, ctab[64 - n], 0.5f) * (1.0f / 64.0f);
    }
    DDSP_WAVE_ORDER();                        // (the table's floats are the noise tile's from here on)
    const uint64_t base_off = p.offset + (p.offset_dev ? *p.offset_dev : 0ull);

    // the H tile of a group: FG * F = 1040 contiguous floats = 260 float4 (group bases are 16-byte aligned: 4160 bytes apart)
    auto load_tile = [&](long g) {
        const v4f *src = reinterpret_cast<const v4f *>(p.Hm + g * FG * F);
        Tile h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h.v[j] = src[lane + 64 * j];
        h.v[4] = src[256 + (lane & 3)];                           // float4 256..259 (lanes 0..3 keep theirs)
        return h;
    };

    // ---- the steps of one group -------------------------------------------------------------------------------------------
    // (lane-derived indices and LDS addresses are recomputed inside each step from an opaque copy of the lane id: hoisted out of
    //  the loop they would all stay live across it, some forty registers, and push the convolution's operands into scratch)
    auto opaque_lane = [&]() { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };

    // H tile (registers) -> LDS, operand reads and splits, the 24 products.  Results stay in accE / accO.
    auto products = [&](const Tile h) {
        const int ln = opaque_lane(), mi = ln & 15, mq = ln >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<v4f *>(Hs + 4 * (ln + 64 * j)) = h.v[j];
        if (ln < 4) *reinterpret_cast<v4f *>(Hs + 4 * (ln + 256)) = h.v[4];
        DDSP_WAVE_ORDER();
        // lane (frame mi, quarter mq) reads H[mi][16 mq .. 16 mq + 16): even positions are its 8 even bins, odd its 8 odd bins
        // (row stride 65: the 32 lanes of a service group hit 32 different banks)
        const float *hrow = Hs + mi * F + 16 * mq;
        float hv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) hv[j] = hrow[j];
        const float h64 = Hs[mi * F + 64];
        Spectra sp;
#pragma unroll
        for (int r = 0; r < 4; ++r) sp.h64[r] = 0.5f * Hs[(4 * mq + r) * F + 64];
        DDSP_WAVE_ORDER();
        Split Ae, Ao;
#pragma unroll
        for (int j = 0; j < 8; ++j) { split3(hv[2 * j], Ae, j); split3(hv[2 * j + 1], Ao, j); }
        v4f (&accE)[2] = sp.e, (&accO)[2] = sp.o;
        accE[0] = accE[1] = accO[0] = accO[1] = (v4f){0, 0, 0, 0};
        // six of the nine cross terms, smallest first: the three dropped ones are below 2^-26 of |a||b|
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int i = 0; i < (DDSP_NOISE_ABL == 3 ? 1 : 6); ++i) {
            accE[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ae.p[PA[i]], Be[0].p[PB[i]], accE[0], 0, 0, 0);
            accE[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ae.p[PA[i]], Be[1].p[PB[i]], accE[1], 0, 0, 0);
            accO[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ao.p[PA[i]], Bo[0].p[PB[i]], accO[0], 0, 0, 0);
            accO[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ao.p[PA[i]], Bo[1].p[PB[i]], accO[1], 0, 0, 0);
        }
        // n = 32: cos(pi k / 2) = (-1)^(k/2) for even k, 0 for odd k; even bin 2 (8 mq + j) has the sign (-1)^j
        float t = (mq == 0 ? 0.5f : 1.0f) * hv[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += (j & 1) ? -hv[2 * j] : hv[2 * j];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        sp.z32 = (t + 0.5f * h64) * (0.5f / 64.0f);                 // bin 64: (-1)^32; win(32) = 0.5
        return sp;
    };

    // taps: lane (n = 16 t + mi) holds frames 4 mq + r.  z[n] -> kern[n], kern[128 - n]; z[64 - n] -> kern[64 - n], kern[64 + n]
    auto taps = [&](const Spectra sp) {
        const int ln = opaque_lane(), mi = ln & 15, mq = ln >> 4;
        const v4f (&accE)[2] = sp.e, (&accO)[2] = sp.o;
        const float z32 = sp.z32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 16 * t + mi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *kr = kern + (4 * mq + r) * KS;
                const float ev = accE[t][r] + ((mi & 1) ? -sp.h64[r] : sp.h64[r]);      // bin 64: cos(pi n) / 2
                const float v1 = (ev + accO[t][r]) * w1[t];
                const float v2 = (ev - accO[t][r]) * w2[t];
                kr[n] = v1;
                if (n != 0) { kr[128 - n] = v1; kr[64 - n] = v2; }
                kr[64 + n] = v2;                                  // n = 0: win(64) = 0, the reference's hann[0]
            }
        }
        if (mq == 0) { kern[mi * KS + 32] = z32; kern[mi * KS + 96] = z32; }
        DDSP_WAVE_ORDER();
    };

    // noise -> xs[frame][8 + m], and the causal padding xs[frame][0..8) (the staging tile of the previous group overwrote it)
    auto draw = [&](long frame0) {
        const int ln = opaque_lane();
        for (int e = ln; e < FG * 8; e += 64) xs[(e >> 3) * XS + (e & 7)] = 0.0f;
        if (p.u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qi = ln + 64 * j, fr = qi >> 5, q = qi & 31;
                float4 v = *reinterpret_cast<const float4 *>(p.u + (frame0 + fr) * R + 4 * q);
                v.x = v.x * 2.0f - 1.0f; v.y = v.y * 2.0f - 1.0f; v.z = v.z * 2.0f - 1.0f; v.w = v.w * 2.0f - 1.0f;
                *reinterpret_cast<float4 *>(xs + fr * XS + 8 + 4 * q) = v;
            }
        } else {
            const uint64_t c0 = base_off + (uint64_t)frame0 * 32u + (uint64_t)ln;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int qi = ln + 64 * j, fr = qi >> 5, q = qi & 31;
                const uint64_t ctr = c0 + 64u * j;
                uint32_t rr[4];
                if (DDSP_NOISE_ABL == 2) { rr[0] = (uint32_t)ctr; rr[1] = rr[0] * 3u; rr[2] = rr[0] ^ 77u; rr[3] = rr[0] + 5u; }
                else philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rr);
                float4 v;
                v.x = philox_to_sample(rr[0]); v.y = philox_to_sample(rr[1]); v.z = philox_to_sample(rr[2]); v.w = philox_to_sample(rr[3]);
                *reinterpret_cast<float4 *>(xs + fr * XS + 8 + 4 * q) = v;
            }
        }
        DDSP_WAVE_ORDER();
    };

    // One convolution pass: 17 blocks; the lane's output chunk cA (cA + 1 blocks) then cB = 15 - cA (cB + 1 blocks).  Fully
    // unrolled: the operands of block t + 1 are read while block t is multiplied; block t reads k at kq + 8 t and the noise
    // window at xq + 8 (16 - t), where the per-lane bases switch from chunk A's to chunk B's at t = tsw (9..16).
    auto conv_pass = [&](int pass) {
        const int ln = opaque_lane();
        // convolution lanes: (quarter cs, frame cf) with the 16 lanes of a ds_read_b128 service group = 16 frames of one quarter
        const int l5 = ln & 31, seg = l5 >> 2;
        const int cf = ((seg >> 1) << 2) | (l5 & 3);
        const int cs = 2 * (ln >> 5) + ((0x96 >> seg) & 1);
        const float *krow = kern + cf * KS;
        const float *xrow = xs + cf * XS + 8;
        const int cA = pass ? 8 + cs : 15 - cs, cB = 15 - cA;
        const int tsw = cA + 1;
        const float *kq_a = krow, *kq_b = krow - 8 * tsw;
        const float *xq_a = xrow + 8 * cA - 8 - 128, *xq_b = xrow + 8 * cB - 8 + 8 * tsw - 128;
        float acc[8], first[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc[u] = 0.0f; first[u] = 0.0f; }
        float4 cur[6], nxt[6];
        auto fetch = [&](float4 (&d)[6], const float *kq, const float *xq, int t) {
            d[0] = *reinterpret_cast<const float4 *>(kq + 8 * t);
            d[1] = *reinterpret_cast<const float4 *>(kq + 8 * t + 4);
            d[2] = *reinterpret_cast<const float4 *>(xq + 8 * (16 - t));
            d[3] = *reinterpret_cast<const float4 *>(xq + 8 * (16 - t) + 4);
            d[4] = *reinterpret_cast<const float4 *>(xq + 8 * (16 - t) + 8);
            d[5] = *reinterpret_cast<const float4 *>(xq + 8 * (16 - t) + 12);
        };
        fetch(cur, kq_a, xq_a, 0);
#pragma unroll
        for (int t = 0; t < (DDSP_NOISE_ABL == 1 ? 1 : 17); ++t) {
            if (t < 16) {
                if (t + 1 < 9) fetch(nxt, kq_a, xq_a, t + 1);                    // every lane is still in its first chunk
                else { const bool b = t + 1 >= tsw; fetch(nxt, b ? kq_b : kq_a, b ? xq_b : xq_a, t + 1); }
            }
            if (t >= 9 && t == tsw) {                             // chunk A done for this lane: park it, chunk B starts at zero
                asm volatile("" ::: "memory");                    // (a real branch: taken by a quarter of the lanes at one t each)
#pragma unroll
                for (int u = 0; u < 8; ++u) { first[u] = acc[u]; acc[u] = 0.0f; }
            }
            asm volatile("" ::"v"(cur[2].x));                     // keep the window as 4 x ds_read_b128
            const float kv[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
            const float xw[16] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w,
                                  cur[4].x, cur[4].y, cur[4].z, cur[4].w, cur[5].x, cur[5].y, cur[5].z, cur[5].w};
#pragma unroll
            for (int v = 0; v < 8; ++v)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] = __fmaf_rn(kv[v], xw[8 + u - v], acc[u]);
#pragma unroll
            for (int e = 0; e < 6; ++e) cur[e] = nxt[e];
            __builtin_amdgcn_sched_barrier(0);                    // one block ahead, not seventeen
        }
        Pass out;
#pragma unroll
        for (int u = 0; u < 8; ++u) { out.v[u] = first[u]; out.v[8 + u] = acc[u]; }
        return out;
    };

    // both passes' results -> the staging tile (over the noise tile: the convolution is done with it)
    auto stage = [&](const Pass r0, const Pass r1) {
        const int ln = opaque_lane();
        const int l5 = ln & 31, seg = l5 >> 2;
        const int cf = ((seg >> 1) << 2) | (l5 & 3);
        const int cs = 2 * (ln >> 5) + ((0x96 >> seg) & 1);
        DDSP_WAVE_ORDER();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float (&r)[16] = pass ? r1.v : r0.v;
            const int cA = pass ? 8 + cs : 15 - cs, cB = 15 - cA;
            float *ya = ys + cf * KS + 8 * cA, *yb = ys + cf * KS + 8 * cB;
            *reinterpret_cast<float4 *>(ya) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4 *>(ya + 4) = make_float4(r[4], r[5], r[6], r[7]);
            *reinterpret_cast<float4 *>(yb) = make_float4(r[8], r[9], r[10], r[11]);
            *reinterpret_cast<float4 *>(yb + 4) = make_float4(r[12], r[13], r[14], r[15]);
        }
        DDSP_WAVE_ORDER();
    };

    // the group's 16 x 128 outputs are contiguous in y: 512 float4, eight per lane, whole lines
    auto read_y = [&](long frame0) {
        const v4f *src = reinterpret_cast<const v4f *>(p.y) + frame0 * (R / 4);
        Lines yv;
#pragma unroll
        for (int j = 0; j < 8; ++j) yv.v[j] = src[lane + 64 * j];
        return yv;
    };
    auto store_y = [&](long frame0, const Lines yv) {
        const int ln = opaque_lane();
        v4f *dst = reinterpret_cast<v4f *>(p.y) + frame0 * (R / 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i4 = ln + 64 * j;
            v4f o = *reinterpret_cast<const v4f *>(ys + (i4 >> 5) * KS + 4 * (i4 & 31));
            if (ACC && DDSP_NOISE_ABL != 4) o += yv.v[j];
            if (DDSP_NOISE_ABL != 4 || o[0] == 123.0f) dst[i4] = o;
        }
        DDSP_WAVE_ORDER();
    };

    // ---- the pipeline -------------------------------------------------------------------------------------------------------
    // Global reads are issued between a group's two convolution passes -- the next group's H tile, then (ACC) this group's y lines --
    // and consumed after the second pass: some 8000 cycles later, which is what an HBM read takes under this kernel's own load
    // (measured with the DDSP_NOISE_STAMPS build: reads issued 3000 cycles ahead still stalled the wavefront for 3000 more).
    long g = blockIdx.x;
    if (g >= ngroups) return;
    Tile hreg = load_tile(g);
    Lines yv;
#pragma unroll
    for (int j = 0; j < 8; ++j) yv.v[j] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    taps(products(hreg));
    draw(g * FG);
    DDSP_STAMP(0);
    for (;;) {
        {   // The two wavefronts of a SIMD take turns at the higher priority, 41 us each (ddsp_osc_chunk.hip: take_turn; the
            // arbiter otherwise favours the older one for the whole launch and the younger one finishes alone): 0.1034 ->
            // 0.0995 ms at the bench shape, same box, three interleaved rounds (epochs of 5 / 10 / 20 us: no gain).
            const unsigned slot = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));
            const unsigned epoch = (unsigned)(__builtin_amdgcn_s_memrealtime() >> 12);
            if ((slot + epoch) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        const long frame0 = g * FG;
        const long gn = g + gridDim.x;
        const Pass r0 = conv_pass(0);
        DDSP_STAMP(1);
        hreg = load_tile(gn < ngroups ? gn : g);                  // (the last group re-reads its own tile: no branch around the loads)
        if (ACC && DDSP_NOISE_ABL != 4) yv = read_y(frame0);
        const Pass r1 = conv_pass(1);
        DDSP_STAMP(2);
        stage(r0, r1);
        DDSP_STAMP(3);
        if (gn >= ngroups) break;
        const Spectra sp = products(hreg);                        // next group's impulse responses
        DDSP_STAMP(4);
        store_y(frame0, yv);                                      // this group's add + store, behind the products
        DDSP_STAMP(5);
        taps(sp);
        DDSP_STAMP(6);
        draw(gn * FG);
        DDSP_STAMP(7);
        g = gn;
    }
    store_y(g * FG, yv);
#ifdef DDSP_NOISE_STAMPS
    DDSP_STAMP(5);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) stamp_out[i] = stamp_sum[i];
#endif
}

}  // namespace

namespace {
constexpr int kDefaultWavesPerCu = 4;
std::atomic<int> g_residency{0};      // ddsp_noise_set_residency: wavefronts per CU of noise_wave_kernel (0: the default)
}  // namespace

// Production tuning knob (not a test hook): wavefronts per CU, 1..8, of the hop-128 noise kernel's persistent grid; 0 restores the
// default.  Results do not depend on it (a group of 16 frames is computed by one wavefront either way).
extern "C" int ddsp_noise_set_residency(int waves_per_cu)
{
    if (waves_per_cu < 0 || waves_per_cu > 8) return DDSP_EINVAL;
    g_residency.store(waves_per_cu, std::memory_order_relaxed);
    return 0;
}
extern "C" int ddsp_noise_get_residency(void)
{
    const int v = g_residency.load(std::memory_order_relaxed);
    return v >= 1 && v <= 8 ? v : kDefaultWavesPerCu;
}

namespace ddsp_noise {

long launch_noise_wave(const NoiseParams &p, hipStream_t s, hipError_t *err)
{
    if (p.R != R || p.F != F || p.S != R) return 0;
    if (((uintptr_t)p.y % 16) != 0 || ((uintptr_t)p.Hm % 16) != 0 || (p.u && ((uintptr_t)p.u % 16) != 0)) return 0;
    const long nframes = (long)p.B * p.T;
    const long ngroups = nframes / FG;                            // whole groups only
    if (ngroups == 0) return 0;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { *err = e; return -1; }
    static int cached[64] = {};
    if (!cached[dev & 63]) {
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) { *err = e; return -1; }
        cached[dev & 63] = cus;
    }
    cus = cached[dev & 63];
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    // FOUR wavefronts per CU by default (17.4 KB of LDS each), not the eight that fit two per SIMD: at eight, this kernel's FMA + LDS +
    // HBM activity makes the chip's power management drop the shader clock (2.41 -> ~2.1 GHz; it takes ~25 ms of load to come back), so
    // the oscillator's kernels of the NEXT step pay 0.15 ms for the 0.04 ms saved here.  Where the clock gives way differs between
    // boxes of the same model (below 8, below 6, below 5 wavefronts per CU on the three kinds measured); one wavefront per SIMD held it
    // on all of them.  ddsp_noise_set_residency lets a caller that has measured ITS box take more (ddsp_pytorch_amd.
    // calibrate_noise_residency does the measuring; profiles/r04_clock_ramp.txt, tools/microbench/noise_residency_sweep.sh).
    const int asked = g_residency.load(std::memory_order_relaxed);
    const long per_cu = asked >= 1 && asked <= 8 ? asked : kDefaultWavesPerCu;
    // (a launch of at most one group per wavefront at eight per CU -- cfg2, the training step: ~20 us -- is over before the trip)
    long resident = ngroups <= (long)cus * 8 ? (long)cus * 8 : (long)cus * per_cu;
    // tuning experiments (DDSP_TEST_HOOKS=1 processes only; read once): wavefronts per CU
    static const long env_waves = [] { const char *ev = getenv("DDSP_NOISE_WAVES"); return (ev && ddsp_hooks_on()) ? atol(ev) : 0L; }();
    if (env_waves > 0 && env_waves <= 8) resident = (long)cus * env_waves;
    const long grid = ngroups < resident ? ngroups : resident;
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE, s);
    if (p.accumulate) hipLaunchKernelGGL(noise_wave_kernel<true>, dim3((unsigned)grid), dim3(64), lds, s, p, ngroups);
    else hipLaunchKernelGGL(noise_wave_kernel<false>, dim3((unsigned)grid), dim3(64), lds, s, p, ngroups);
    ddsp_prof::end(slot, s);
    *err = hipGetLastError();
    return *err == hipSuccess ? ngroups * FG : -1;
}

}  // namespace ddsp_noise
