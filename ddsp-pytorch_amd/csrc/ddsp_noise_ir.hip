// Impulse responses of the filtered-noise path when 2 (F - 1) is NOT a power of two (the reference's default: 195 bands at hop 512,
// config/default.py:15,19; S = 388 = 4 x 97 has no radix-2 transform), as ONE matrix product for the whole batch:
//
//     z[frame][n] = sum_k c_k H[frame][k] cos(2 pi k n / S),   n = 0 .. S/2,   c_0 = c_{S/2} = 1, c_k = 2 otherwise
//
// = irfft of the zero-phase magnitudes (model/ddsp/filtered_noise.py:8-10) up to the 1/S the consumer folds into its window.  The
// cosine matrix is shared by every frame, so the contraction runs on the matrix cores -- as SPLIT bf16 (ddsp_noise_wave.hip): every
// fp32 value is hi + mid + lo, three bf16 terms that represent it exactly, and a product is the six cross terms of weight >= 2^-16
// of v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the dropped terms are below 2^-26 |a||b|.  noise_fft_kernel (ddsp_noise_fft.hip)
// then reads z instead of building it from F x S/4 cosine sums per frame pair on the vector pipe, which was 60 % of its time.
// The same product, transposed, is the backward's last step (dH = dz C^T).
//
// Folded by the parity of the bin: cos(2 pi k (S/2 - n) / S) = (-1)^k cos(2 pi k n / S), so with E / O = the sums over the even / odd
// bins for n = 0 .. S/4 only,  z[n] = E[n] + O[n],  z[S/2 - n] = E[n] - O[n]:  two [frames x 98] x [98 x 98] products instead of one
// [frames x 195] x [195 x 195] (-38 % matrix-core work and operand traffic).  Backward the same way round: the taps are folded
// (P[n] = dz[n] + dz[S/2 - n], M[n] = dz[n] - dz[S/2 - n]) and the even / odd bins come from P / M.
//
// Layout: a workgroup of four wavefronts takes 64 frames; each wavefront keeps ITS 16 frames' rows as the A operands (even / odd bins,
// or P / M) in registers for the whole tile; the cosine operand (1 KB fragments in the exact register layout, built once per call by
// noise_ir_table_kernel into the caller's workspace: 168 KB, L2-resident) streams through a double-buffered LDS stage one output tile
// (16 columns of both parities: 2 x 4 x 3 fragments = 24 KB) at a time, shared by the four wavefronts.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ddsp_hip.h"
#include "ddsp_internal.h"
#include "ddsp_noise_common.h"

using namespace ddsp_noise;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte load from a 4-byte-aligned address
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split { bf16x8 p[3]; };
// x = hi + mid + lo with every term a bf16: both residuals are exact in fp32 (8 + 8 + 8 significand bits)
__device__ __forceinline__ void split3(float x, Split &d, int j)
{
    const __bf16 hi = (__bf16)x;
    const float r1 = x - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    d.p[0][j] = hi; d.p[1][j] = mid; d.p[2][j] = (__bf16)r2;
}

constexpr int KT = 4;             // contraction steps of 32 per parity: S/4 + 1 <= 128
constexpr int NT = 7;             // output tiles of 16 per parity:      S/4 + 1 in (96, 112]  <=>  F in [194, 225]
constexpr int FR = 2 * KT * 3;    // fragments of one output tile: (parity, step, term)

// Fragment (nt, parity, kt, term) of the cosine operand, B[i = 32 kt + 8 (lane >> 4) + j][o = 16 nt + (lane & 15)], j = 0..7 in one
// 16-byte word per lane; i = contraction index, o = output index.  Forward (transpose == 0): i = k' (bin 2 k' + parity), o = tap n;
// backward: i = tap n, o = k'.  Value c_bin cos(2 pi bin n / S) for bin <= S/2 and n <= S/4, else 0.  cos(2 pi m / S) =
// cospif(2 m / S): the values of the direct kernels' table (ddsp_noise_fft.hip: ctab).
__global__ void __launch_bounds__(64) noise_ir_table_kernel(bf16x8 *table, int F, int transpose)
{
    const int nt = blockIdx.x / (2 * KT), par = (blockIdx.x / KT) & 1, kt = blockIdx.x % KT, lane = threadIdx.x;
    const int half = F - 1, S = 2 * half, NQ = half / 2 + 1;
    const int o = 16 * nt + (lane & 15);
    Split b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 32 * kt + 8 * (lane >> 4) + j;
        const int bin = 2 * (transpose ? o : i) + par, n = transpose ? i : o;
        float v = 0.0f;
        if (bin <= half && n < NQ) {
            const int m = (int)(((long)bin * (long)n) % (long)S);
            v = ((bin == 0 || bin == half) ? 1.0f : 2.0f) * cospif((float)(2 * m) / (float)S);
        }
        split3(v, b, j);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) table[((long)nt * FR + (par * KT + kt) * 3 + t) * 64 + lane] = b.p[t];
}

struct IrParams {
    const float *in;      // [frames][in_stride], columns [0, F) used
    const bf16x8 *table;  // NT x FR fragments of 64 lanes
    float *out;           // [frames][out_stride], columns [0, F) written
    float *maxabs;        // nullable: maxabs[frame * out_stride] = max_i |in[frame][i]| (a spare column of `out`)
    long frames;
    int F, in_stride, out_stride;
};

// TRANSPOSE == false: in = H (bins), out = z S (taps).  true: in = dz (taps), out = dH (bins).
template <bool TRANSPOSE>
__global__ void __launch_bounds__(256, 2) noise_ir_kernel(IrParams p)
{
    extern __shared__ __attribute__((aligned(16))) bf16x8 stage[];      // [2][FR][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4;
    const int half = p.F - 1, NQ = half / 2 + 1;
    // the fragments of a chunk this thread moves: wave, wave + 4, ... (FR = 24: six each)
    auto fetch = [&](int nt, bf16x8 (&pre)[FR / 4]) {
        const bf16x8 *src = p.table + (long)nt * FR * 64 + lane;
#pragma unroll
        for (int e = 0; e < FR / 4; ++e) pre[e] = src[(long)(wave + 4 * e) * 64];
    };
    auto park = [&](int buf, const bf16x8 (&pre)[FR / 4]) {
#pragma unroll
        for (int e = 0; e < FR / 4; ++e) stage[(buf * FR + wave + 4 * e) * 64 + lane] = pre[e];
    };
    const long tiles = (p.frames + 63) / 64;
    bf16x8 pre[FR / 4];
    fetch(0, pre);
    park(0, pre);
    __syncthreads();
    int buf = 0;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long f0 = tile * 64 + wave * 16;
        // ---- this wavefront's 16 rows as the two A operands: A[m = lane & 15][i = 32 kt + 8 (lane >> 4) + j] ------------------
        const long fr = f0 + mi;
        const bool ok = fr < p.frames;
        const float *row = p.in + (ok ? fr : 0) * (long)p.in_stride;
        Split Ae[KT], Ao[KT];
        float mx = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int i0 = 32 * kt + 8 * mq;
            float ve[8], vo[8];
            // Steps kt < KT - 1 lie inside every row this kernel is built for (2 * 96 + 16 <= 194 <= F; taps 96 <= S/4 and their mirrors
            // >= 0): unconditional 16-byte loads from 4-byte-aligned addresses.  The last step reads clamped indices and selects zeros
            // (no lane-dependent branches: with them the compiler spilled 30-80 registers).
            if (!TRANSPOSE) {
                // even / odd bins 2 (i0 + j), 2 (i0 + j) + 1: sixteen consecutive floats (rows start on any 4-byte boundary: F is odd)
                float v[16];
                if (kt < KT - 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f_u t = *reinterpret_cast<const v4f_u *>(row + 2 * i0 + 4 * q);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[4 * q + c] = t[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float t = row[min(2 * i0 + c, half)];
                        v[c] = 2 * i0 + c <= half ? t : 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ve[j] = ok ? v[2 * j] : 0.0f;
                    vo[j] = ok ? v[2 * j + 1] : 0.0f;
                    mx = fmaxf(mx, fmaxf(fabsf(ve[j]), fabsf(vo[j])));
                }
            } else {
                // folded taps: P[n] = dz[n] + dz[S/2 - n], M[n] = dz[n] - dz[S/2 - n] (the middle tap of an even S/2 counts once)
                float a[8], b[8];
                if (kt < KT - 1) {
                    const v4f_u a0 = *reinterpret_cast<const v4f_u *>(row + i0), a1 = *reinterpret_cast<const v4f_u *>(row + i0 + 4);
                    const v4f_u b1 = *reinterpret_cast<const v4f_u *>(row + half - i0 - 7), b0 = *reinterpret_cast<const v4f_u *>(row + half - i0 - 3);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { a[c] = a0[c]; a[4 + c] = a1[c]; b[c] = b0[3 - c]; b[4 + c] = b1[3 - c]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { a[j] = row[min(i0 + j, half)]; b[j] = row[max(half - i0 - j, 0)]; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = i0 + j, n2 = half - n;
                    const bool in = ok && n < NQ, pair = in && n2 != n;
                    ve[j] = (in ? a[j] : 0.0f) + (pair ? b[j] : 0.0f);
                    vo[j] = pair ? a[j] - b[j] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { split3(ve[j], Ae[kt], j); split3(vo[j], Ao[kt], j); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!TRANSPOSE && p.maxabs) {
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (mq == 0 && ok) p.maxabs[fr * (long)p.out_stride] = mx;
        }
        // ---- one output tile (16 columns of both parities) per chunk of the cosine operand -----------------------------------
#pragma unroll 1                                                      // (unrolled, the seven chunks' 64-bit addresses are spilled)
        for (int nt = 0; nt < NT; ++nt) {
            fetch(nt + 1 < NT ? nt + 1 : 0, pre);                    // the next chunk (the next tile's first after the last)
            v4f e0 = {0.0f, 0.0f, 0.0f, 0.0f}, e1 = e0, o0 = e0, o1 = e0;     // two chains per parity: a product does not wait for the one before it
            const bf16x8 *bsrc = stage + (long)buf * FR * 64 + lane;
            // six of the nine cross terms, smallest first (ddsp_noise_wave.hip)
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                bf16x8 be[3], bo[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { be[t] = bsrc[(kt * 3 + t) * 64]; bo[t] = bsrc[((KT + kt) * 3 + t) * 64]; }
#pragma unroll
                for (int i = 0; i < 6; i += 2) {
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ae[kt].p[PA[i]], be[PB[i]], e0, 0, 0, 0);
                    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ao[kt].p[PA[i]], bo[PB[i]], o0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ae[kt].p[PA[i + 1]], be[PB[i + 1]], e1, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ao[kt].p[PA[i + 1]], bo[PB[i + 1]], o1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);                   // (operand reads of later steps hoisted above: 100 more registers)
            }
            const v4f E = e0 + e1, O = o0 + o1;
            // the next chunk goes to LDS BEFORE this tile's stores are issued: the wait for its loads then covers only stores that are
            // a whole chunk old (vector-memory operations retire in order)
            park(buf ^ 1, pre);
            // D[m = 4 (lane >> 4) + r][o = lane & 15]
            const int o = 16 * nt + mi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long f = f0 + 4 * mq + r;
                float *dst = p.out + f * (long)p.out_stride;
                if (f < p.frames) {
                    if (!TRANSPOSE) {          // tap o and its mirror S/2 - o
                        if (o < NQ) {
                            dst[o] = E[r] + O[r];
                            if (half - o != o) dst[half - o] = E[r] - O[r];
                        }
                    } else {                   // bins 2 o, 2 o + 1
                        if (2 * o <= half) dst[2 * o] = E[r];
                        if (2 * o + 1 <= half) dst[2 * o + 1] = O[r];
                    }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
    }
}

int device_cus(hipError_t *err)
{
    int dev = 0, cus = 0;
    *err = hipGetDevice(&dev);
    if (*err != hipSuccess) return 0;
    static int cached[64] = {};
    if (!cached[dev & 63]) {
        *err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (*err != hipSuccess) return 0;
        cached[dev & 63] = cus;
    }
    return cached[dev & 63];
}

}  // namespace

namespace ddsp_noise {

bool ir_product_shape(int F, int hop)
{
    const int S = 2 * (F - 1), NQ = (F - 1) / 2 + 1;
    return hop == 512 && S < hop && NQ > 16 * (NT - 1) && NQ <= 16 * NT;
}

int ir_row_stride(int F) { return 16 * ((F + 15) / 16) + 4; }           // z row: >= F columns + a 16-byte tail (max |H| in its first float)

size_t ir_table_bytes(int) { return (size_t)NT * FR * 64 * sizeof(bf16x8); }

size_t ir_workspace_bytes(long frames, int F) { return ir_table_bytes(F) + (size_t)frames * ir_row_stride(F) * sizeof(float); }

float *ir_rows(void *workspace, int F) { return reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ir_table_bytes(F)); }

// The cosine operand into the head of the workspace (transpose: the backward's, bins on the output side).
hipError_t launch_ir_table(void *workspace, int F, int transpose, hipStream_t s)
{
    hipLaunchKernelGGL(noise_ir_table_kernel, dim3((unsigned)(NT * 2 * KT)), dim3(64), 0, s, reinterpret_cast<bf16x8 *>(workspace), F, transpose);
    return hipGetLastError();
}

// transpose == 0: out[frame][n] = sum_k c_k in[frame][k] cos(2 pi k n / S) (max |in| of a frame to maxabs[frame * out_stride] if given);
// transpose != 0: out[frame][k] = c_k sum_n in[frame][n] cos(2 pi k n / S); k, n < F; with the operand launch_ir_table left in the workspace.
hipError_t launch_ir_product(const float *in, int in_stride, float *out, int out_stride, float *maxabs, long frames, int F, int transpose,
                             const void *workspace, hipStream_t s)
{
    hipError_t err = hipSuccess;
    const int cus = device_cus(&err);
    if (err != hipSuccess) return err;
    IrParams p;
    p.in = in; p.table = reinterpret_cast<const bf16x8 *>(workspace); p.out = out; p.maxabs = maxabs; p.frames = frames;
    p.F = F; p.in_stride = in_stride; p.out_stride = out_stride;
#ifndef DDSP_IR_WG_PER_CU
#define DDSP_IR_WG_PER_CU 2
#endif
    const long tiles = (frames + 63) / 64, resident = (long)cus * DDSP_IR_WG_PER_CU;
    const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
    const size_t lds = (size_t)2 * FR * 64 * sizeof(bf16x8);
    if (transpose) hipLaunchKernelGGL(noise_ir_kernel<true>, dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(noise_ir_kernel<false>, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Forward: z rows (x S) of every frame into the workspace, | table | z [frames][ir_row_stride] |, max |H| of a frame in column
// ir_row_stride - 4 of its row.  Returns the z rows (nullptr on a launch error).
const float *launch_noise_ir(const float *Hmag, long frames, int F, void *workspace, hipStream_t s, hipError_t *err)
{
    const int zs = ir_row_stride(F);
    float *z = ir_rows(workspace, F);
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE_IR, s);
    *err = launch_ir_table(workspace, F, 0, s);
    if (*err == hipSuccess) *err = launch_ir_product(Hmag, F, z, zs, z + zs - 4, frames, F, 0, workspace, s);
    ddsp_prof::end(slot, s);
    return *err == hipSuccess ? z : nullptr;
}

}  // namespace ddsp_noise
