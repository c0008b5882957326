// Shared by the filtered-noise kernels (ddsp_noise.hip: direct forms, ddsp_noise_fft.hip: in-LDS FFT form).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddsp_noise {

struct NoiseParams {
    const float *Hm;
    const float *u;
    float *y;
    int B, T, F, R, S;
    uint64_t seed, offset;
    const uint64_t *offset_dev;  // nullable: the draw starts at offset + *offset_dev (a device counter: hipGraph replays)
    int accumulate;
    int lpf_log; // batched kernel: log2(lanes per frame) -> 64 >> lpf_log frames per workgroup
    const float *zrows;          // nullable (FFT form only): impulse responses already built by ddsp_noise_ir.hip, [B*T][zs]:
    int zs;                      //   z[n] * S at columns [0, S/2], max |H| of the frame at column zs - 4
};

// Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,0,0), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        // (three-input xor as ONE instruction, v_bitop3_b32 with truth table 0x96: 20 instead of 40 xors per block)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c[1], k0, 0x96);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c[3], k1, 0x96);
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}


// One U[0,1) draw of the in-kernel stream as x = 2u - 1, u = (r >> 8) 2^-24 (:44-45).  m = r >> 8 < 2^24 converts exactly, m 2^-23 is
// exact and so is the subtraction (a multiple of 2^-23 in [-1, 1)): ONE fused multiply-add gives the very bits of (u * 2) - 1.
__device__ __forceinline__ float philox_to_sample(uint32_t r) { return __fmaf_rn((float)(r >> 8), 1.0f / 8388608.0f, -1.0f); }

// Launches the FFT form when the shape is one it is built for and faster at (hop 512, S <= hop; with force_fft also hop 256);
// returns false (and launches nothing) otherwise.  *err receives the launch status.
bool launch_noise_fft(const NoiseParams &p, hipStream_t s, bool force_fft, hipError_t *err);

// Backward of the noise path in the in-LDS FFT form (hop 512; 257 bands, or -- given a workspace of ir_workspace_bytes -- the
// shapes of ir_product_shape, whose dH step is then one matrix product); returns false (and launches nothing) for other shapes.
// *err receives the launch status.
bool launch_noise_fft_backward(const float *grad_y, const float *uniform, float *grad_H, int B, int T, int F, int hop, uint64_t seed,
                               uint64_t offset, const uint64_t *offset_dev, void *workspace, hipStream_t s, hipError_t *err);

// Impulse responses as one split-bf16 matrix product for the whole batch (ddsp_noise_ir.hip): the shapes it is built for, the
// workspace it needs (| cosine operand | z rows |), and the launch -> the z rows inside the workspace (nullptr: launch error in *err).
bool ir_product_shape(int F, int hop);
int ir_row_stride(int F);
size_t ir_workspace_bytes(long frames, int F);
const float *launch_noise_ir(const float *Hmag, long frames, int F, void *workspace, hipStream_t s, hipError_t *err);
float *ir_rows(void *workspace, int F);
hipError_t launch_ir_table(void *workspace, int F, int transpose, hipStream_t s);
hipError_t launch_ir_product(const float *in, int in_stride, float *out, int out_stride, float *maxabs, long frames, int F, int transpose,
                             const void *workspace, hipStream_t s);

// Launches the wavefront-private hop-128 / 65-band form (ddsp_noise_wave.hip) on the leading whole groups of 16 frames when the
// shape is the one it is built for.  Returns the number of frames it took (0: not its shape, nothing launched; the caller runs
// the remaining frames -- fewer than 16 -- through another kernel, with the Philox offset advanced), -1 on a launch error (*err).
long launch_noise_wave(const NoiseParams &p, hipStream_t s, hipError_t *err);

}  // namespace ddsp_noise
