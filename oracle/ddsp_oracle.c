/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the DDSP synthesis hot path.
 *
 * A scalar C restatement of what the reference's torch-CPU code computes for
 *   model/ddsp/harmonic_oscillator.py:24-75  (OscillatorBank.forward / .live)
 *   model/ddsp/filtered_noise.py:7-53        (amp_to_impulse_response, fft_convolve, FilteredNoise.forward)
 * written from the bit-level spec in SURVEY.md Appendix A (not from the reference text).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (ddsp-pytorch_amd/) never does.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this file against the
 * fixtures in tests/golden/ that tools/make_goldens.py captured by importing the
 * reference in the build container (phase path bit-exact, outputs <= 1e-6).
 *
 * Build: see oracle/Makefile.  Must be compiled with -ffp-contract=off: every fused
 * multiply-add below is explicit (fmaf) because the rounding points are the spec.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TWO_PI32 6.2831854820251465f /* fl32(2*pi), harmonic_oscillator.py:34,42 */

/* F.interpolate(mode='linear', align_corners=False, scale_factor=hop) source index and
 * weights for output sample i (harmonic_oscillator.py:52-55). */
static inline void upsample_index(int64_t i, float scale, int T, int *i0, int *i1, float *w0, float *w1)
{
    float src = fmaf(scale, (float)i + 0.5f, -0.5f);
    if (src < 0.0f) src = 0.0f;
    int64_t k = (int64_t)floorf(src);
    if (k > T - 1) k = T - 1;
    float lam = src - (float)k;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    *i0 = (int)k;
    *i1 = (int)k + (k < T - 1 ? 1 : 0);
    if (scale == 1.0f) *i1 = *i0; /* scale_factor 1: ATen copies the input (no neighbour term, so no NaN bleed) */
    *w1 = lam;
    *w0 = 1.0f - lam;
}

static inline float lerp_spec(float w0, float x0, float w1, float x1)
{
    return fmaf(w0, x0, w1 * x1); /* fl32(fma(w0, x[i0], fl32(w1*x[i1]))) */
}

/* Frame-rate preparation (harmonic_oscillator.py:24-35): per (t,k) the phase increment in
 * rad/sample and the masked, re-normalised harmonic amplitude. */
static void prepare_frames(const float *f0, const float *c, int T, int H, int sample_rate,
                           float *w /*[T,H]*/, float *amp /*[T,H]*/)
{
    const float nyq = (float)(sample_rate / 2); /* integer floor, then compared as float */
    const float sr = (float)sample_rate;
    for (int t = 0; t < T; ++t) {
        float s = 0.0f;
        for (int k = 0; k < H; ++k) {
            float hz = (float)(k + 1) * f0[t];
            float a0 = (hz > nyq) ? 0.0f : c[t * H + k];
            amp[t * H + k] = a0;
            s += a0;
            float rad = hz * TWO_PI32;
            w[t * H + k] = rad / sr;
        }
        for (int k = 0; k < H; ++k) amp[t * H + k] = amp[t * H + k] / s;
    }
}

/*
 * Oscillator bank, one batch row.  dbg_* (nullable) receive [N,H] intermediates:
 * inc (upsampled increments), cum (fl32 of the double running sum, before the modulo), phi.
 * live_phase (nullable, [H]): added to the first increment row and overwritten with the
 * last phase row (harmonic_oscillator.py:70,72) -- caller passes it for batch row 0 only.
 */
static void osc_row(const float *f0, const float *c, const float *a, float *y,
                    float *dbg_inc, float *dbg_cum, float *dbg_phi, float *live_phase,
                    int T, int H, int hop, int sample_rate)
{
    const int64_t N = (int64_t)T * hop;
    const float scale = (float)(1.0 / (double)hop);
    float *w = (float *)malloc(sizeof(float) * T * H);
    float *amp = (float *)malloc(sizeof(float) * T * H);
    double *acc = (double *)calloc(H, sizeof(double));
    prepare_frames(f0, c, T, H, sample_rate, w, amp);
    for (int64_t i = 0; i < N; ++i) {
        int i0, i1;
        float w0, w1;
        upsample_index(i, scale, T, &i0, &i1, &w0, &w1);
        const float L = lerp_spec(w0, a[i0], w1, a[i1]);
        float sum = 0.0f;
        for (int k = 0; k < H; ++k) {
            float inc = lerp_spec(w0, w[i0 * H + k], w1, w[i1 * H + k]);
            if (i == 0 && live_phase) inc = inc + live_phase[k];
            acc[k] += (double)inc;              /* cumsum: double accumulator ... */
            const float P = (float)acc[k];      /* ... rounded to fp32 per output (App. A item 5) */
            const float phi = fmodf(P, TWO_PI32);
            const float A = lerp_spec(w0, amp[i0 * H + k], w1, amp[i1 * H + k]);
            sum += (L * A) * sinf(phi);
            if (dbg_inc) dbg_inc[i * H + k] = inc;
            if (dbg_cum) dbg_cum[i * H + k] = P;
            if (dbg_phi) dbg_phi[i * H + k] = phi;
            if (i == N - 1 && live_phase) live_phase[k] = phi;
        }
        y[i] = sum;
    }
    free(w);
    free(amp);
    free(acc);
}

int ddsp_oracle_osc(const float *f0, const float *c, const float *a, float *y,
                    float *dbg_inc, float *dbg_cum, float *dbg_phi, float *live_phase,
                    int B, int T, int H, int hop, int sample_rate)
{
    if (B < 0 || T <= 0 || H <= 0 || hop <= 0 || sample_rate <= 0) return 1;
    const int64_t N = (int64_t)T * hop;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        osc_row(f0 + (int64_t)b * T, c + (int64_t)b * T * H, a + (int64_t)b * T, y + b * N,
                dbg_inc ? dbg_inc + b * N * H : NULL, dbg_cum ? dbg_cum + b * N * H : NULL,
                dbg_phi ? dbg_phi + b * N * H : NULL, (b == 0) ? live_phase : NULL, T, H, hop, sample_rate);
    }
    return 0;
}

/* Frame-rate quantities only (for checking the GPU prep kernel): w, amp as [B,T,H]. */
int ddsp_oracle_osc_frames(const float *f0, const float *c, float *w, float *amp,
                           int B, int T, int H, int sample_rate)
{
    for (int b = 0; b < B; ++b)
        prepare_frames(f0 + (int64_t)b * T, c + (int64_t)b * T * H, T, H, sample_rate,
                       w + (int64_t)b * T * H, amp + (int64_t)b * T * H);
    return 0;
}

/*
 * amp_to_impulse_response (filtered_noise.py:7-22) for one frame, in double:
 *   ir = irfft(H + 0j) (length S = 2(F-1)); roll(+S/2); * periodic hann(S); pad/crop to R; roll(-S/2).
 * The reference evaluates this with fp32 FFTs; the fp64 direct form agrees to <= 2e-7.
 */
static void frame_impulse(const float *Hm, int F, int R, double *kern /*[R]*/)
{
    const int S = 2 * (F - 1);
    const int half = S / 2;
    double *z = (double *)malloc(sizeof(double) * (S > 0 ? S : 1));
    for (int n = 0; n < S; ++n) {
        double v = (double)Hm[0] + ((n & 1) ? -1.0 : 1.0) * (double)Hm[F - 1];
        for (int k = 1; k < F - 1; ++k) v += 2.0 * (double)Hm[k] * cos(2.0 * M_PI * (double)k * (double)n / (double)S);
        z[n] = v / (double)S;
    }
    /* a2[j] = z[(j - half) mod S] * hann[j],   j in [0,S) ; then padded/cropped to R ; then rolled by -half */
    for (int j = 0; j < R; ++j) kern[j] = 0.0;
    for (int j = 0; j < R; ++j) {
        int src = (j + half) % R; /* roll(-S//2) on the length-R array: out[j] = in[(j + half) mod R] */
        if (src < S) {
            /* hann evaluated in fp32 as torch.hann_window(S, dtype=float32) does */
            float win = 0.5f - 0.5f * (float)cos(2.0 * M_PI * (double)src / (double)S);
            kern[j] = z[((src - half) % S + S) % S] * (double)win;
        }
    }
    free(z);
}

/*
 * FilteredNoise.forward (filtered_noise.py:40-53) with the uniform draw injected:
 *   u [B,T,R] in [0,1)  ->  x = u*2-1 ;  y[n] = sum_{m<=n} x[m] * kern[n-m]  (fft_convolve keeps the
 *   first R samples of the linear convolution, :25-32) ; frames concatenated, no overlap-add.
 * dbg_ir (nullable): [B,T,R] impulse responses.
 */
int ddsp_oracle_noise(const float *Hm, const float *u, float *y, float *dbg_ir, int B, int T, int F, int R)
{
    if (B < 0 || T <= 0 || F < 2 || R <= 0) return 1;
    const int64_t frames = (int64_t)B * T;
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < frames; ++f) {
        double *kern = (double *)malloc(sizeof(double) * R);
        frame_impulse(Hm + f * F, F, R, kern);
        const float *uu = u + f * R;
        for (int n = 0; n < R; ++n) {
            double s = 0.0;
            for (int m = 0; m <= n; ++m) s += (double)(uu[m] * 2.0f - 1.0f) * kern[n - m];
            y[f * R + n] = (float)s;
        }
        if (dbg_ir)
            for (int n = 0; n < R; ++n) dbg_ir[f * R + n] = (float)kern[n];
        free(kern);
    }
    return 0;
}

/*
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
 * library's philox4x32 with its default 10 rounds).  Restated from the paper: multipliers 0xD2511F53 / 0xCD9E8D57,
 * Weyl key increments 0x9E3779B9 / 0xBB67AE85 applied between rounds.  Pinned by the library's published
 * known-answer vectors (tests/test_oracle_golden.py::test_philox_known_answers).
 * This is the generator behind FilteredNoise(rng='device') -- the throughput replacement of the reference's
 * torch.rand draw (filtered_noise.py:44-48); the reference itself has no counterpart, so the KAT vectors are the pin.
 */
void ddsp_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/*
 * The uniform draw of the in-kernel stream, laid out like torch.rand(B,T,R) (include/ddsp_hip.h, ddsp_noise_forward):
 * frame f (= b*T + t), sample m: word (m & 3) of Philox(counter = offset + f*ceil(R/4) + (m >> 2), key = seed),
 * counter and key as 64-bit values split low word first, upper counter words 0; u = (word >> 8) * 2^-24 in [0,1).
 * The kernels then form x = 2u - 1 exactly as the reference does with its own draw (:45).
 */
int ddsp_oracle_philox_uniform(uint64_t seed, uint64_t offset, int64_t frames, int R, float *u)
{
    if (frames < 0 || R <= 0) return 1;
    const int64_t quads = (R + 3) / 4;
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < frames; ++f)
        for (int64_t q = 0; q < quads; ++q) {
            const uint64_t c = offset + (uint64_t)f * (uint64_t)quads + (uint64_t)q;
            const uint32_t ctr[4] = {(uint32_t)c, (uint32_t)(c >> 32), 0u, 0u};
            uint32_t w[4];
            ddsp_oracle_philox4x32_10(ctr, key, w);
            for (int e = 0; e < 4 && 4 * q + e < R; ++e) u[f * R + 4 * q + e] = (float)(w[e] >> 8) * (1.0f / 16777216.0f);
        }
    return 0;
}

int ddsp_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
