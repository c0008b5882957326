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
// Layout: a workgroup of four wavefronts takes 64 frames; each wavefront keeps ITS 16 frames' rows as the A operand in registers
// (KT x 3 split fragments) for the whole tile; the cosine operand (KT x NT tiles x 3 terms, 1 KB fragments in the exact register
// layout, built once per call by noise_ir_table_kernel into the caller's workspace: 273 KB at F = 195, L2-resident) streams through
// a double-buffered LDS stage one output tile (16 taps: KT x 3 fragments = 21 KB) at a time, shared by the four wavefronts -- so
// the L1 moves 273 KB per 64 frames and the LDS serves 4 x that, both under the matrix cores' 7 x 13 x 6 x 16 cycles.
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

constexpr int KT = 7;             // contraction steps of 32: F in (192, 224]

// Fragment (nt, kt, term) of the cosine operand, B[i = 32 kt + 8 (lane >> 4) + j][o = 16 nt + (lane & 15)], j = 0..7 in one 16-byte
// word per lane.  i = contraction index, o = output index; the bin (whose weight c applies) is i in the forward (transpose == 0:
// i = bin k, o = tap n) and o in the backward (i = tap n, o = bin k).  cos(2 pi m / S) = cospif(2 m / S): the values of the direct
// kernels' table (ddsp_noise_fft.hip: ctab).
__global__ void __launch_bounds__(64) noise_ir_table_kernel(bf16x8 *table, int F, int S, int NT, int transpose)
{
    const int nt = blockIdx.x / KT, kt = blockIdx.x % KT, lane = threadIdx.x;
    const int o = 16 * nt + (lane & 15);
    Split b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 32 * kt + 8 * (lane >> 4) + j;
        float v = 0.0f;
        if (i < F && o < F) {
            const int bin = transpose ? o : i;
            const int m = (int)(((long)i * (long)o) % (long)S);
            v = ((bin == 0 || bin == F - 1) ? 1.0f : 2.0f) * cospif((float)(2 * m) / (float)S);
        }
        split3(v, b, j);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) table[((long)(nt * KT + kt) * 3 + t) * 64 + lane] = b.p[t];
}

struct IrParams {
    const float *in;      // [frames][in_stride], columns [0, F) used
    const bf16x8 *table;  // NT x KT x 3 fragments of 64 lanes
    float *out;           // [frames][out_stride], columns [0, out_cols) written
    float *maxabs;        // nullable: maxabs[frame * out_stride] = max_i |in[frame][i]| (a spare column of `out`)
    long frames;
    int F, NT, in_stride, out_stride, out_cols;
};

__global__ void __launch_bounds__(256, 2) noise_ir_kernel(IrParams p)
{
    extern __shared__ __attribute__((aligned(16))) bf16x8 stage[];      // [2][KT * 3][64]
    constexpr int FR = KT * 3;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, mi = lane & 15, mq = lane >> 4;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // the fragments of a chunk this thread moves: wave, wave + 4, ... < FR
    auto fetch = [&](int nt, bf16x8 (&pre)[(FR + 3) / 4]) {
        const bf16x8 *src = p.table + (long)nt * FR * 64 + lane;
#pragma unroll
        for (int e = 0; e < (FR + 3) / 4; ++e) {
            const int f = wave + 4 * e;
            pre[e] = f < FR ? src[(long)f * 64] : zero8;
        }
    };
    auto park = [&](int buf, const bf16x8 (&pre)[(FR + 3) / 4]) {
#pragma unroll
        for (int e = 0; e < (FR + 3) / 4; ++e) {
            const int f = wave + 4 * e;
            if (f < FR) stage[(buf * FR + f) * 64 + lane] = pre[e];
        }
    };
    const long tiles = (p.frames + 63) / 64;
    bf16x8 pre[(FR + 3) / 4];
    fetch(0, pre);
    park(0, pre);
    __syncthreads();
    int buf = 0;
    for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long f0 = tile * 64 + wave * 16;
        // ---- this wavefront's 16 rows as the A operand: A[m = lane & 15][i = 32 kt + 8 (lane >> 4) + j] --------------------
        const long fr = f0 + mi;
        const bool ok = fr < p.frames;
        const float *row = p.in + (ok ? fr : 0) * (long)p.in_stride;
        Split A[KT];
        float mx = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float v[8];
            const int i0 = 32 * kt + 8 * mq;
            if (ok && i0 + 8 <= p.F) {                                // two 16-byte loads (rows start on any 4-byte boundary: F is odd)
                const v4f_u lo = *reinterpret_cast<const v4f_u *>(row + i0), hi = *reinterpret_cast<const v4f_u *>(row + i0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ok && i0 + j < p.F) ? row[i0 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { mx = fmaxf(mx, fabsf(v[j])); split3(v[j], A[kt], j); }
        }
        if (p.maxabs) {
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (mq == 0 && ok) p.maxabs[fr * (long)p.out_stride] = mx;
        }
        // ---- one output tile (16 columns) per chunk of the cosine operand -----------------------------------------------------
        for (int nt = 0; nt < p.NT; ++nt) {
            fetch(nt + 1 < p.NT ? nt + 1 : 0, pre);                  // the next chunk (the next tile's first after the last)
            v4f acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;         // two chains: a product does not wait for the one before it
            const bf16x8 *bsrc = stage + (long)buf * FR * 64 + lane;
            // six of the nine cross terms, smallest first (ddsp_noise_wave.hip)
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                bf16x8 b[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) b[t] = bsrc[(kt * 3 + t) * 64];
#pragma unroll
                for (int i = 0; i < 6; i += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kt].p[PA[i]], b[PB[i]], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kt].p[PA[i + 1]], b[PB[i + 1]], acc1, 0, 0, 0);
                }
            }
            const v4f acc = acc0 + acc1;
            // the next chunk goes to LDS BEFORE this tile's stores are issued: the wait for its loads then covers only stores that are
            // a whole chunk old (vector-memory operations retire in order: behind fresh stores it was a full HBM write latency per chunk)
            park(buf ^ 1, pre);
            // D[m = 4 (lane >> 4) + r][o = lane & 15]
            const int o = 16 * nt + mi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long f = f0 + 4 * mq + r;
                if (f < p.frames && o < p.out_cols) p.out[f * (long)p.out_stride + o] = acc[r];
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
    const int S = 2 * (F - 1);
    return hop == 512 && S < hop && F > 32 * (KT - 1) && F <= 32 * KT;
}

int ir_row_stride(int F) { return 16 * ((F + 15) / 16) + 4; }           // z row: 16 NT columns + a 16-byte tail (max |H| in its first float)

size_t ir_table_bytes(int F) { return (size_t)((F + 15) / 16) * KT * 3 * 64 * sizeof(bf16x8); }

size_t ir_workspace_bytes(long frames, int F) { return ir_table_bytes(F) + (size_t)frames * ir_row_stride(F) * sizeof(float); }

float *ir_rows(void *workspace, int F) { return reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ir_table_bytes(F)); }

// The cosine operand into the head of the workspace (transpose: the backward's, bins on the output side).
hipError_t launch_ir_table(void *workspace, int F, int transpose, hipStream_t s)
{
    const int NT = (F + 15) / 16, S = 2 * (F - 1);
    hipLaunchKernelGGL(noise_ir_table_kernel, dim3((unsigned)(NT * KT)), dim3(64), 0, s, reinterpret_cast<bf16x8 *>(workspace), F, S, NT, transpose);
    return hipGetLastError();
}

// out[frame][o] = sum_i in[frame][i] C[i][o], i, o < F, with the operand launch_ir_table left in the workspace.
hipError_t launch_ir_product(const float *in, int in_stride, float *out, int out_stride, int out_cols, float *maxabs, long frames, int F,
                             const void *workspace, hipStream_t s)
{
    hipError_t err = hipSuccess;
    const int cus = device_cus(&err);
    if (err != hipSuccess) return err;
    IrParams p;
    p.in = in; p.table = reinterpret_cast<const bf16x8 *>(workspace); p.out = out; p.maxabs = maxabs; p.frames = frames;
    p.F = F; p.NT = (F + 15) / 16; p.in_stride = in_stride; p.out_stride = out_stride; p.out_cols = out_cols;
#ifndef DDSP_IR_WG_PER_CU
#define DDSP_IR_WG_PER_CU 3
#endif
    const long tiles = (frames + 63) / 64, resident = (long)cus * DDSP_IR_WG_PER_CU;
    const size_t lds = (size_t)2 * KT * 3 * 64 * sizeof(bf16x8);
    hipLaunchKernelGGL(noise_ir_kernel, dim3((unsigned)(tiles < resident ? tiles : resident)), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Forward: z rows (x S) of every frame into the workspace, | table | z [frames][ir_row_stride] |, max |H| of a frame in column
// 16 NT of its row.  Returns the z rows (nullptr on a launch error).
const float *launch_noise_ir(const float *Hmag, long frames, int F, void *workspace, hipStream_t s, hipError_t *err)
{
    const int NT = (F + 15) / 16, zs = ir_row_stride(F);
    float *z = ir_rows(workspace, F);
    const int slot = ddsp_prof::begin(ddsp_prof::NOISE_IR, s);
    *err = launch_ir_table(workspace, F, 0, s);
    if (*err == hipSuccess) *err = launch_ir_product(Hmag, F, z, zs, 16 * NT, z + 16 * NT, frames, F, workspace, s);
    ddsp_prof::end(slot, s);
    return *err == hipSuccess ? z : nullptr;
}

}  // namespace ddsp_noise
