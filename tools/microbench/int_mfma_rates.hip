// Microbenchmarks that size the round-3 filtered-noise kernel (noise section):
//   (1) issue cost of the 32-bit integer multiplies Philox4x32 is made of (v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32) against the
//       24-bit forms and v_xor3_b32;
//   (2) fp32 matrix-core instructions (the inverse real DFT of the filter magnitudes as a product with a shared cosine matrix)
//       alone and interleaved with independent v_fma_f32 of the SAME wave and of ANOTHER wave on the SIMD (does the matrix pipe
//       run beside the vector pipe?).
// Build: hipcc -O3 --offload-arch=gfx950 int_mfma_rates.hip -o int_mfma_rates.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 1 << 13;
constexpr int UNROLL = 16;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

enum Op { MUL_LO, MUL_HI, MAD64, MUL24, XOR3, FMA32, MFMA16, MFMA32, MFMA16_FMA_SAME, MFMA16_ROLE_SPLIT, PHILOX, MFMA16_XOR_SAME, MFMA16_MAD64_SAME, MFMA16_DSREAD_SAME, CVT_U32_F32, ADD_U32, TOTALS_F64, TOTALS_INT };

__device__ __forceinline__ void philox(unsigned c0, unsigned c1, unsigned k0, unsigned k1, unsigned (&out)[4])
{
    unsigned c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
#ifdef PHILOX_XOR2
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
#else
        const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c[1], k0, 0x96), n1 = (unsigned)p1;
        const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c[3], k1, 0x96), n3 = (unsigned)p0;
#endif
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float *out, unsigned seed, unsigned long long *stamps)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned x[UNROLL];
    unsigned long long w[UNROLL];
    float f[UNROLL];
    const unsigned m = 0xD2511F53u + seed;
    const float a = 1.0001f + seed, b = 0.5f;
    v4f acc4[4];
    v16f acc16[2];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { x[u] = seed + threadIdx.x * 7u + u; w[u] = x[u]; f[u] = (float)u + threadIdx.x; }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc4[q] = (v4f){0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) acc16[q] = (v16f){0};
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < ITERS; ++it) {
        if (OP == PHILOX) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned r[4];
                philox(x[u] + it, x[u + 4], seed, seed + 1, r);
                x[u] ^= r[0] ^ r[1]; x[u + 4] ^= r[2] ^ r[3];
            }
            continue;
        }
        if (OP == MFMA16_ROLE_SPLIT) {
            // waves 0,1 (SIMDs 0,2 or so) only MFMA, waves 2,3 only FMA -- with 2+ workgroups per CU both kinds share SIMDs
            if ((wave & 1) == 0) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                }
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[u]) : "v"(x[u]), "v"(m) : "vcc");
            if (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            if (OP == XOR3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
            if (OP == MFMA16) acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
            if (OP == MFMA32) acc16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[u & 1], 0, 0, 0);
            if (OP == CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(x[u]) : "v"(f[u]));
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            if (OP == TOTALS_F64) {   // the oscillator's frame-totals chain: mul, fma, cvt f64, add f64
                float t; double dd;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(b), "v"(f[u]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(f[u]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd) : "v"(t));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[u]) : "v"(dd));
            }
            if (OP == TOTALS_INT) {   // the same with the increments pre-scaled to integers: mul, fma, cvt u32, add u32
                float t; unsigned ti;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(b), "v"(f[u]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(f[u]));
                asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(ti) : "v"(t));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[u]) : "v"(ti));
            }
            if (OP == MFMA16_XOR_SAME) {
                acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[u]) : "v"(m));
            }
            if (OP == MFMA16_MAD64_SAME) {
                acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[u]) : "v"(x[u]), "v"(m) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[(u + 1) & 15]) : "v"(x[u]), "v"(m) : "vcc");
            }
            if (OP == MFMA16_FMA_SAME) {
                acc4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[u & 3], 0, 0, 0);
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(b), "v"(a));
            }
        }
        if (OP == MAD64) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) x[u] = (unsigned)(w[u] >> 32) ^ (unsigned)w[u];   // (counted separately: 1 v_xor per mad)
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
    float s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += f[u] + (float)x[u];
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc4[q][0] + acc4[q][3];
    s += acc16[0][0] + acc16[1][7];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char *name, double insts_per_iter, float *dout, const char *unit = "wave-instr")
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *dst;
    CK(hipMalloc(&dst, 16));
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, dout, 1u, dst);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, dout, 1u, dst);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2];
        CK(hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)st[0] / (double)st[1] * 0.1;
        const double winst = (double)grid * 4 * ITERS * insts_per_iter;
        const double per_simd_ns = ms * 1e6 / (winst / (256.0 * 4));
        printf("%-28s waves/SIMD=%d  %8.3f ms  %.3f ns/%s/SIMD  clock %.2f GHz -> %.2f cyc\n", name, wps, ms, per_simd_ns, unit, ghz, per_simd_ns * ghz);
    }
    CK(hipFree(dst));
}

int main()
{
    float *dout;
    CK(hipMalloc(&dout, 1024));
    run<FMA32>("v_fma_f32", UNROLL, dout);
    run<MUL_LO>("v_mul_lo_u32", UNROLL, dout);
    run<MUL_HI>("v_mul_hi_u32", UNROLL, dout);
    run<MAD64>("v_mad_u64_u32 (+1 v_xor)", UNROLL, dout);
    run<MUL24>("v_mul_u32_u24", UNROLL, dout);
    run<XOR3>("v_xor_b32", UNROLL, dout);
    run<PHILOX>("philox4x32-10 block", 4, dout, "block");
    run<MFMA16>("v_mfma_f32_16x16x4_f32", UNROLL, dout);
    run<MFMA32>("v_mfma_f32_32x32x2_f32", UNROLL, dout);
    run<MFMA16_FMA_SAME>("mfma16x16x4 + 4 fma (same wave)", UNROLL, dout, "group");
    run<CVT_U32_F32>("v_cvt_u32_f32", UNROLL, dout);
    run<ADD_U32>("v_add_u32", UNROLL, dout);
    run<TOTALS_F64>("totals chain f64 (4 instr)", UNROLL, dout, "chain");
    run<TOTALS_INT>("totals chain int (4 instr)", UNROLL, dout, "chain");
    run<MFMA16_XOR_SAME>("mfma16x16x4 + 4 xor (same wave)", UNROLL, dout, "group");
    run<MFMA16_MAD64_SAME>("mfma16x16x4 + 2 mad64 (same wave)", UNROLL, dout, "group");
    return 0;
}
