// Shared device helpers and host-side parameter block of the oscillator kernels
// (ddsp_osc.hip: forward, ddsp_osc_bwd.hip: backward).  See ddsp_osc.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace ddsp_osc {


constexpr float kTwoPi32 = 6.2831854820251465f;     // fl32(2*pi): the modulus the reference uses (:34,:42)
constexpr float kInvTwoPi32 = 0.15915493667125702f; // fl32(1/fl32(2*pi))
constexpr float kRevPerRad = 0.15915494309189535f;  // 1/(2*pi) for v_sin_f32 (argument in revolutions)
constexpr float kFastPhaseLimit = 1.0e7f;           // fast modulo is exact while P/2pi32 < 2^21
constexpr float kRoundMagic = 12582912.0f;          // 1.5*2^23: (x + magic) - magic = rint(x) for |x| < 2^22
constexpr int kFrameScratchTag = 0x46524d45;        // flag word [2] of a scratch buffer the FRAME kernels filled (what the backward re-walks)

struct OscParams {
    const float *f0, *c, *a;
    float *y;
    float *w, *amp;   // scratch [B,T,H]: written by the totals kernel (row t), read by the synth kernels
    double *loc;      // scratch [B,T,H]: exclusive prefix of the frame totals inside the frame's superblock
    double *sup;      // scratch [B,NSB,H]: superblock totals, then (in place) their exclusive scan along t
    int *redo_flag;   // scratch: set by the FAST synth kernel when a wavefront needs the EXACT one; 64 bytes further on:
                      // four 64-bit words {shader clock, 100 MHz clock} at the start and at the end of one synth wavefront (ddsp_osc_clock)
    const float *live_in;
    float *live_out;
    float *dbg_phi;
    const float *grad_y;  // backward: [B,N]
    float *part_c;        // backward scratch [B,T,3,H]: partial d/d(amp) aimed at rows t-1, t, t+1
    float *part_a;        // backward scratch [B,T,3]:   partial d/d(a)
    float *grad_c, *grad_a;
    int B, T, H, R;
    int K;            // harmonics per lane (template instance to launch)
    int logG;         // lanes per frame group
    int NSB;          // superblocks per batch row = ceil(T / (256 >> logG))
    int force_exact;
    int stage_out;    // synth: stage 32 output samples per frame in LDS, store whole 128-byte lines (pow2 hop >= 64, G in 4..16)
    int pow2;         // hop is a power of two and the clip has <= 2^23 samples: incremental weights, uniform loops
    // chunked form (ddsp_osc_chunk.hip): power-of-two hop >= 64, 4..16 lanes per row group
    double *ctot;     // scratch [B,NC,H]: chunk totals, then (in place) their exclusive scan along the row
    int *rlive;       // scratch [B,NC]: 1 + highest harmonic slot with a non-zero amplitude anywhere in the row's chunk
    int *perm;        // scratch [NC, RB*64/G]: per chunk index, the batch rows grouped by how many slots they walk (-1 = none)
    int *redo;        // scratch [RB*NC]: wave tasks the fast synth kernel declined
    int *frame_flag;  // the frame layout's flag words inside this scratch buffer (chunked launches clear the tag there)
    int Lc, NC, RB;   // chunk length in samples, chunks per row, row blocks (64/G rows each)
    int lgR;          // log2(R)
    int nres;         // wavefronts per SIMD the chunks were sized for (turn-taking modulus, <= 3)
    int turn_shift;   // log2 of the turn-taking epoch in 100 MHz ticks
    float inv2R;      // 1/(2R)
    float scale;      // fl32(1/R): F.interpolate's source-index scale
    float nyquist;    // float(sample_rate // 2)
    float sr;         // float(sample_rate)
};

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx b and b + 8 share an L2, observed
// behaviour: speed only).  Consecutive *logical* blocks cover consecutive frames of one batch row and share halo rows of
// the frame-rate scratch, so logical blocks [i * n/8, (i + 1) * n/8) are handed to the hardware blocks of one XCD.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nblocks)
{
    const unsigned per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;          // the ragged tail keeps its position
    return (bid & 7u) * per + (bid >> 3);
}

// Shader-clock probe: one wavefront of the synth kernel stamps {s_memtime, s_memrealtime} when it starts and when it ends;
// (delta shader ticks) / (delta 100 MHz ticks) * 0.1 = the GHz the kernel actually ran at (ddsp_osc_clock reads it back).
__device__ __forceinline__ void clock_stamp(int *redo_flag, int which)
{
    unsigned long long *w = reinterpret_cast<unsigned long long *>(redo_flag + 16) + 2 * which;
    w[0] = __builtin_amdgcn_s_memtime();
    w[1] = __builtin_amdgcn_s_memrealtime();
}

// ---- cross-lane helpers -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the G = 2^logG lanes of a group (groups are G-aligned); every lane of the group gets the total
// (each step is a symmetric exchange).
__device__ __forceinline__ float group_sum(float v, int logG)
{
    if (logG >= 1) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if (logG >= 2) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if (logG >= 3) v += dpp_mov<0x141>(v);  // row_half_mirror
    if (logG >= 4) v += dpp_mov<0x140>(v);  // row_mirror
    if (logG >= 5) v += __shfl_xor(v, 16);
    if (logG >= 6) v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// torch CPU `%=` on floats (aten remainder): fmod, then shifted into the divisor's sign.
__device__ __forceinline__ float remainder_two_pi(float p)
{
    float r = fmodf(p, kTwoPi32);
    if (r != 0.0f && r < 0.0f) r += kTwoPi32;
    return r;
}

enum { MODE_TOTALS = 0, MODE_SYNTH = 1 };
// FAST: production path.  EXACT: bit-exact modulo (libm fmodf), live state and debug outputs; it also
// repairs the wavefronts the FAST synth kernel declined (phases outside the fast modulo's exact range).
enum { VAR_FAST = 0, VAR_EXACT = 1 };

template <int K>
struct FrameState {
    double acc[K];
    float x0[K], x1[K];  // frame-rate increments at the bracketing frames i0, i1
    float a0[K], da[K];  // amplitude at i0 and (amp[i1] - amp[i0])   (synth only)
};

// First sample index n in [0,R) of frame t whose interpolation source index is >= t, i.e. where
// F.interpolate switches from frames (t-1,t) to (t,t+1).  Evaluated with the reference's own fp32
// expression so that it is right for every hop (R/2 for even hops in exact arithmetic).
__device__ __forceinline__ int split_index(int t, int R, float scale)
{
    int m = R >> 1;
    if (t == 0) return m;  // both halves clamp to frame 0: any split gives identical results
    const float tf = (float)t;
    const int base = t * R;
    while (m > 0 && __fmaf_rn(scale, (float)(base + m - 1) + 0.5f, -0.5f) >= tf) --m;
    while (m < R && __fmaf_rn(scale, (float)(base + m) + 0.5f, -0.5f) < tf) ++m;
    return m;
}

// F.interpolate(linear, align_corners=False) weights of output sample i against source frame i0: App. A item 4
__device__ __forceinline__ void upsample_weights(float scale, int i, float i0f, float &w0, float &w1)
{
    float src = __fmaf_rn(scale, (float)i + 0.5f, -0.5f);
    src = fmaxf(src, 0.0f);
    w1 = fminf(fmaxf(src - i0f, 0.0f), 1.0f);
    w0 = 1.0f - w1;
}

// rad/sample of harmonic h (0-based) at fundamental f: two roundings and a true division (:26-35, App. A item 3)
__device__ __forceinline__ float frame_increment(int h, float f, float sr)
{
    const float hz = (float)(h + 1) * f;
    const float rad = hz * kTwoPi32;
    return rad / sr;
}

template <int K>
__device__ __forceinline__ void load_synth_segment(const OscParams &p, FrameState<K> &st, int b, int j, int i0, int i1,
                                                   float &L0, float &L1)
{
    const int G = 1 << p.logG;
    const long rowbase = (long)b * p.T;
    const float *w0row = p.w + (rowbase + i0) * p.H;
    const float *w1row = p.w + (rowbase + i1) * p.H;
    const float *a0row = p.amp + (rowbase + i0) * p.H;
    const float *a1row = p.amp + (rowbase + i1) * p.H;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const int h = j + m * G;
        const bool ok = h < p.H;
        st.x0[m] = ok ? w0row[h] : 0.0f;
        st.x1[m] = ok ? w1row[h] : 0.0f;
        const float u0 = ok ? a0row[h] : 0.0f;
        const float u1 = ok ? a1row[h] : 0.0f;
        st.a0[m] = u0;
        st.da[m] = u1 - u0;
    }
    L0 = p.a[rowbase + i0];
    L1 = p.a[rowbase + i1];
}


// ---- host side (defined in ddsp_osc.hip) ---------------------------------------------------------------
struct Tiling { int K, logG; };
bool pick_tiling(int H, long frames, Tiling *out);
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
// scratch layout: w | amp | loc | sup | flag; sup is sized for the smallest superblock (G = 64: 4 frames)
inline size_t sup_elems(int B, int T, int H) { return (size_t)B * ((size_t)(T + 3) / 4) * H; }
// Fills the shape-derived fields and carves the scratch buffer; returns false if no tiling exists for H.
bool setup_params(OscParams &p, void *scratch, int B, int T, int H, int hop, int sample_rate);
// chunked form (ddsp_osc_chunk.hip)
bool chunked_eligible(const OscParams &p);
extern std::atomic<int> g_chunk_any_batch;
size_t chunk_scratch_bytes(int B, int T, int H);
size_t frame_scratch_bytes(int B, int T, int H);
void pick_chunks(int T, int R, int RB, int cus, int wg_per_cu, int *Lc_out, int *NC_out);
hipError_t launch_chunked_k(const OscParams &p, void *scratch, hipStream_t s);
hipError_t chunk_geometry_k(OscParams &p, int *cus, int *wg_per_cu);
const int *chunk_flag_words(OscParams p);

}  // namespace ddsp_osc
