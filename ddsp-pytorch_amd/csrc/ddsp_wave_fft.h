// Wavefront-private complex FFTs in registers + LDS exchanges (gfx950), shared by the filtered-noise FFT form
// (ddsp_noise_fft.hip) and the spectral-loss scales (ddsp_mss_fft.hip).  One wavefront owns its transforms and shares nothing
// with other wavefronts: LDS operations of one wavefront execute in order, so the stage hand-offs need no barrier.
//
// FFT of N = 64 * R1 points on one wavefront: n = 64 n1 + 8 n2 + n3, k = k1 + R1 k2 + 8 R1 k3;
//   radix-R1 over n1 in registers (lane = 8 n2 + n3) -> twiddle W_N^(lane k1) -> LDS exchange -> radix-8 over n2
//   (lane = k1 + R1 n3') -> twiddle W_64^(n3 k2) -> LDS exchange -> radix-8 over n3 (lane = k1 + R1 k2).
// Twiddles live in registers; the exchange addresses are linear (lane base + immediate offsets) and padded against bank conflicts.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace ddsp_wfft {

typedef float2 cf;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(__fmaf_rn(a.x, b.x, -(a.y * b.y)), __fmaf_rn(a.x, b.y, a.y * b.x)); }
// a * w for the forward transform, a * conj(w) for the inverse (w always holds the FORWARD twiddle e^{-i theta})
template <bool INV>
__device__ __forceinline__ cf cmulw(cf a, cf w)
{
    if (!INV) return make_float2(__fmaf_rn(a.x, w.x, -(a.y * w.y)), __fmaf_rn(a.x, w.y, a.y * w.x));
    return make_float2(__fmaf_rn(a.x, w.x, a.y * w.y), __fmaf_rn(a.y, w.x, -(a.x * w.y)));
}
// times -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ cf rot90(cf a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
// times W8^1 = e^{-i pi/4} and W8^3 = e^{-3 i pi/4} (conjugated for the inverse)
template <bool INV>
__device__ __forceinline__ cf mul_w8_1(cf a)
{
    constexpr float r = 0.70710678118654752f;
    return INV ? make_float2((a.x - a.y) * r, (a.x + a.y) * r) : make_float2((a.x + a.y) * r, (a.y - a.x) * r);
}
template <bool INV>
__device__ __forceinline__ cf mul_w8_3(cf a)
{
    constexpr float r = 0.70710678118654752f;
    return INV ? make_float2(-(a.x + a.y) * r, (a.x - a.y) * r) : make_float2((a.y - a.x) * r, -(a.x + a.y) * r);
}

// natural order in, natural order out
template <bool INV>
__device__ __forceinline__ void dft4(cf &p0, cf &p1, cf &p2, cf &p3)
{
    const cf s0 = cadd(p0, p2), s1 = cadd(p1, p3), d0 = csub(p0, p2), d1 = rot90<INV>(csub(p1, p3));
    p0 = cadd(s0, s1); p1 = cadd(d0, d1); p2 = csub(s0, s1); p3 = csub(d0, d1);
}

template <bool INV>
__device__ __forceinline__ void dft8(cf (&v)[8])
{
    cf a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    cf b0 = csub(v[0], v[4]), b1 = mul_w8_1<INV>(csub(v[1], v[5])), b2 = rot90<INV>(csub(v[2], v[6])), b3 = mul_w8_3<INV>(csub(v[3], v[7]));
    dft4<INV>(a0, a1, a2, a3);   // X[0], X[2], X[4], X[6]
    dft4<INV>(b0, b1, b2, b3);   // X[1], X[3], X[5], X[7]
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// HALFZERO: v[8..15] are zero (a length-N/2 signal zero-padded to N): the first radix-2 layer costs only its twiddles
template <bool INV, bool HALFZERO>
__device__ __forceinline__ void dft16(cf (&v)[16])
{
    constexpr float C = 0.92387953251128674f, S = 0.38268343236508977f;   // cos, sin of pi/8
    const cf w1 = make_float2(C, -S), w3 = make_float2(S, -C), w5 = make_float2(-S, -C), w7 = make_float2(-C, -S);
    cf a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = HALFZERO ? v[j] : cadd(v[j], v[j + 8]);
        b[j] = HALFZERO ? v[j] : csub(v[j], v[j + 8]);
    }
    b[1] = cmulw<INV>(b[1], w1);
    b[2] = mul_w8_1<INV>(b[2]);
    b[3] = cmulw<INV>(b[3], w3);
    b[4] = rot90<INV>(b[4]);
    b[5] = cmulw<INV>(b[5], w5);
    b[6] = mul_w8_3<INV>(b[6]);
    b[7] = cmulw<INV>(b[7], w7);
    dft8<INV>(a);   // X[2m]
    dft8<INV>(b);   // X[2m+1]
#pragma unroll
    for (int m = 0; m < 8; ++m) { v[2 * m] = a[m]; v[2 * m + 1] = b[m]; }
}

template <int R1, bool INV, bool HALFZERO>
__device__ __forceinline__ void dft_r1(cf (&v)[R1])
{
    if constexpr (R1 == 16) dft16<INV, HALFZERO>(v);
    else {
        if constexpr (HALFZERO) {
            // v[4..7] zero: a_j = v_j, b_j = v_j W8^j
            cf a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
            cf b0 = v[0], b1 = mul_w8_1<INV>(v[1]), b2 = rot90<INV>(v[2]), b3 = mul_w8_3<INV>(v[3]);
            dft4<INV>(a0, a1, a2, a3);
            dft4<INV>(b0, b1, b2, b3);
            v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
            v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
        } else {
            dft8<INV>(v);
        }
    }
}

// LDS operations of one wavefront execute in order: between a stage's stores and the next stage's loads only the
// compiler has to be kept from reordering.
#define DDSP_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

// Per-wavefront twiddles of the 64*R1-point transform (forward values; the inverse conjugates on the fly).
template <int R1>
struct Twiddles {
    cf t1[R1];                 // W_N^(lane * k1)
    cf t2[R1 / 8][8];          // W_64^(n3 * k2) for the lane's n3 of step 2 (one per DFT the lane does there)
};

template <int R1>
__device__ __forceinline__ void make_twiddles(Twiddles<R1> &tw, int lane)
{
    constexpr int N = 64 * R1;
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        float s, c;
        sincospif(2.0f * (float)((lane * k1) & (N - 1)) / (float)N, &s, &c);
        tw.t1[k1] = make_float2(c, -s);
    }
#pragma unroll
    for (int d = 0; d < R1 / 8; ++d) {
        const int n3 = (lane / R1) + (64 / R1) * d;      // step-2 lane = k1 + R1 * (n3 mod (64/R1)), DFT d = n3 div (64/R1)
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            float s, c;
            sincospif(2.0f * (float)((n3 * k2) & 63) / 64.0f, &s, &c);
            tw.t2[d][k2] = make_float2(c, -s);
        }
    }
}

// Exchange addresses (in float2 units inside one buffer of buf_elems<R1>() elements).  Both are SEPARABLE in their arguments --
// lane-dependent base + compile-time offset, so the unrolled loops address LDS with immediates -- and laid out so that b64
// accesses do not collide on banks (a b64 access of 32 lanes covers the 64 banks once):
// step 1 -> 2, R1 = 16: element (k1, n2, n3) at 136 n2 + 17 n3 + k1.  A write (lanes (n2, n3), one k1) touches
//   8 n2 + 17 n3 mod 32: 32 different residues; the step-2 reads (lanes k1 + 16 m) are consecutive apart from one pad per 16 lanes.
// step 1 -> 2, R1 = 8: element (k1, n2, n3) at 68 k1 + 8 n2 + n3.  A write (lanes 8 n2 + n3, one k1) is 64 CONSECUTIVE elements;
//   a step-2 read (lanes k1 + 8 n3, one n2) touches 4 k1 + n3 mod 32: 32 different residues per half wavefront.  (Rounds 2-3 used
//   81 n2 + 9 n3 + k1 here: both its writes and its reads were two-way conflicted -- a quarter to a third of the LDS cycles of the
//   spectral-loss scales below 1024 points, profiles/r03_mss_pmc.json.)
template <int R1>
__device__ __forceinline__ int addr1(int k1, int n2, int n3) { return (R1 == 16) ? 136 * n2 + 17 * n3 + k1 : 68 * k1 + 8 * n2 + n3; }
// step 2 -> 3: element (c = k1 + R1 k2, n3) at ROW2 n3 + c (R1 = 8: 72 = 8 mod 32 puts the four n3 of a half-wavefront write on
// different banks; R1 = 16: 128, the writes of a half wavefront are 2 x 16 consecutive elements 128 apart ... see fft_wave)
template <int R1>
__device__ __forceinline__ int addr2(int c, int n3) { return ((R1 == 16) ? 128 : 72) * n3 + c; }
template <int R1>
constexpr int buf_elems() { return (R1 == 16) ? 8 * 136 : 576; }

// 64*R1-point complex FFT of the wavefront's data.  In: v[n1] = x[64 n1 + lane].  Out: v[d * 8 + k3] = X[c + 8 R1 k3] with
// c = lane + 64 d (d < R1/8).  `buf`: N float2 of LDS, free to clobber.
template <int R1, bool INV, bool HALFZERO>
__device__ __forceinline__ void fft_wave(cf (&v)[R1], const Twiddles<R1> &tw, cf *buf, int lane)
{
    constexpr int ND = R1 / 8;                     // radix-8 DFTs per lane in steps 2 and 3
    // step 1: radix-R1 over n1, twiddle, scatter
    dft_r1<R1, INV, HALFZERO>(v);
    {
        const int n2 = lane >> 3, n3 = lane & 7;
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            const cf a = (k1 == 0) ? v[0] : cmulw<INV>(v[k1], tw.t1[k1]);
            buf[addr1<R1>(k1, n2, n3)] = a;
        }
    }
    DDSP_WAVE_ORDER();
    // step 2: radix-8 over n2, twiddle, scatter
    cf u[ND][8];
    {
        const int k1 = lane & (R1 - 1), m3 = lane / R1;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int n3 = m3 + (64 / R1) * d;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) u[d][n2] = buf[addr1<R1>(k1, n2, n3)];
        }
        DDSP_WAVE_ORDER();
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int n3 = m3 + (64 / R1) * d;
            dft8<INV>(u[d]);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const cf a = (k2 == 0) ? u[d][0] : cmulw<INV>(u[d][k2], tw.t2[d][k2]);
                buf[addr2<R1>(k1 + R1 * k2, n3)] = a;
            }
        }
    }
    DDSP_WAVE_ORDER();
    // step 3: radix-8 over n3
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int c = lane + 64 * d;
        cf t[8];
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) t[n3] = buf[addr2<R1>(c, n3)];
        dft8<INV>(t);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) v[d * 8 + k3] = t[k3];
    }
    DDSP_WAVE_ORDER();
}

// natural-order store of the result layout of fft_wave
template <int R1>
__device__ __forceinline__ void store_natural(const cf (&v)[R1], cf *dst, int lane)
{
#pragma unroll
    for (int d = 0; d < R1 / 8; ++d)
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) dst[lane + 64 * d + 8 * R1 * k3] = v[d * 8 + k3];
}


// ---- 512 points per wavefront as 8 / R1 independent transforms of N = 64 R1 points (R1 = 1, 2, 4, 8) -----------------------
// Eight points per lane whatever N is: step 1 is a radix-R1 butterfly per transform (nothing for R1 = 1) and its twiddle
// W_N^(lane k1); what is left are eight 64-point sequences (s = b R1 + k1) across the lanes -- exactly steps 2 and 3 of the
// 512-point transform above, with s in the role of k1.
//   In:  v[b R1 + n1] = x_b[64 n1 + lane].   Out: v[k3] = X_b[k1 + R1 (k2 + 8 k3)] on lane s + 8 k2, s = b R1 + k1.
// t1[k1] = W_N^(lane k1) (k1 < R1), t2[k2] = W_64^((lane >> 3) k2); `buf`: buf_elems<8>() float2 of LDS, free to clobber.
template <int R1, bool INV>
__device__ __forceinline__ void fft_wave_batched(cf (&v)[8], const cf *t1, const cf (&t2)[8], cf *buf, int lane)
{
    static_assert(R1 == 1 || R1 == 2 || R1 == 4 || R1 == 8, "transform sizes 64, 128, 256, 512");
    // step 1: radix-R1 over n1 inside each transform, twiddle, scatter
#pragma unroll
    for (int b = 0; b < 8 / R1; ++b) {
        if constexpr (R1 == 2) {
            const cf a0 = v[2 * b], a1 = v[2 * b + 1];
            v[2 * b] = cadd(a0, a1);
            v[2 * b + 1] = csub(a0, a1);
        } else if constexpr (R1 == 4) {
            dft4<INV>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
        } else if constexpr (R1 == 8) {
            dft8<INV>(v);
        }
    }
    {
        const int n2 = lane >> 3, n3 = lane & 7;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k1 = s % R1;
            const cf a = (k1 == 0) ? v[s] : cmulw<INV>(v[s], t1[k1]);
            buf[addr1<8>(s, n2, n3)] = a;
        }
    }
    DDSP_WAVE_ORDER();
    // step 2: radix-8 over n2, twiddle, scatter
    cf u[8];
    {
        const int s = lane & 7, n3 = lane >> 3;
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) u[n2] = buf[addr1<8>(s, n2, n3)];
        DDSP_WAVE_ORDER();
        dft8<INV>(u);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const cf a = (k2 == 0) ? u[0] : cmulw<INV>(u[k2], t2[k2]);
            buf[addr2<8>(s + 8 * k2, n3)] = a;
        }
    }
    DDSP_WAVE_ORDER();
    // step 3: radix-8 over n3
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) u[n3] = buf[addr2<8>(lane, n3)];
    dft8<INV>(u);
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) v[k3] = u[k3];
    DDSP_WAVE_ORDER();
}

}  // namespace ddsp_wfft
