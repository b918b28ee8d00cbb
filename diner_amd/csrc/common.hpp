// Shared device helpers of the gfx950 DINER kernels.
//
// Arithmetic contract: every helper performs the reference's fp32 operations in the reference's
// order, one rounding per op (the library is compiled with -ffp-contract=off; fused multiply-adds
// appear only where written as __builtin_fmaf, at the places where ATen's own kernels contract --
// pinned bit-exact against the reference by tests/test_oracle_golden.py on the CPU restatement).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diner_hip.h"

#define DINER_WAVE 64
#define DINER_SIGMA_PAD 100  // exponential padding of the sigma map (image_encoder.py:173)

namespace diner {

void set_error(const char *fmt, ...);
int check_launch(const char *what);

// Launch state that belongs to the DEVICE, not the process (a process may render on cuda:1 after cuda:0): the CU count that sizes the
// persistent grids, and the raised dynamic-LDS limit of a kernel (hipFuncSetAttribute acts on the current device's copy of the function).
// Both are cached per device ordinal (api.hip); thread-safe (atomics; a race only repeats an idempotent call).
int device_cus();                                                    // CUs of the current device (256 on MI355X)
enum LdsSlot { LDS_SLOT_F16_LINZ = 0, LDS_SLOT_F16_LATENT = 2, LDS_SLOT_F16_TRACE = 4, LDS_SLOT_TRAIN_CORE0 = 6, LDS_SLOT_DW512 = 10, LDS_SLOT_COUNT = 12 };   // (the f16 slots come in pairs: view-sequential / views-in-tile)
int ensure_dynamic_lds(const void *kernel, int bytes, int slot);    // DINER_OK, or DINER_E_LAUNCH with the error set

// --------------------------------------------------------------------------------------------
// camera of one source view, held in registers (SGPRs once the compiler sees it is uniform)
// --------------------------------------------------------------------------------------------
struct View {
    float r[9];  // rotation rows
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ View load_view(const DinerScene &s, int sb, int v)
{
    View o;
    const float *P = s.poses + ((int64_t)sb * s.NV + v) * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.r[i * 3 + 0] = P[i * 4 + 0];
        o.r[i * 3 + 1] = P[i * 4 + 1];
        o.r[i * 3 + 2] = P[i * 4 + 2];
        o.t[i] = P[i * 4 + 3];
    }
    const float *f = s.focal + ((int64_t)sb * s.NV + v) * 2, *c = s.c + ((int64_t)sb * s.NV + v) * 2;
    o.fx = f[0]; o.fy = f[1]; o.cx = c[0]; o.cy = c[1];
    return o;
}

// R*x as torch.matmul evaluates it (BLAS: k-ordered FMA chain) -- nerf_renderer.py:100,103
__device__ __forceinline__ void rotate(const View &v, float x, float y, float z, float &ox, float &oy, float &oz)
{
    ox = __builtin_fmaf(v.r[2], z, __builtin_fmaf(v.r[1], y, v.r[0] * x));
    oy = __builtin_fmaf(v.r[5], z, __builtin_fmaf(v.r[4], y, v.r[3] * x));
    oz = __builtin_fmaf(v.r[8], z, __builtin_fmaf(v.r[7], y, v.r[6] * x));
}

// world point -> camera point + NDC uv -- nerf_renderer.py:99-110 == pixelnerf.py:91-108
__device__ __forceinline__ void project(const View &v, float iw, float ih, float x, float y, float z,
                                        float &px, float &py, float &pz, float &u, float &w)
{
    rotate(v, x, y, z, px, py, pz);
    px = px + v.t[0]; py = py + v.t[1]; pz = pz + v.t[2];
    u = px / pz; w = py / pz;
    u = u * v.fx; w = w * v.fy;
    u = u + v.cx; w = w + v.cy;
    u = u / iw * 2.0f - 1.0f;
    w = w / ih * 2.0f - 1.0f;
}

// sin() of the positional encodings.  The argument is the reference's fp32 value (one fma, positional_encoding.py:45-49); its sine comes
// from the hardware's v_sin_f32 (sin(2 pi x) of an argument in revolutions) behind a compensated reduction: r1 = RN(a * C_HI),
// r2 = the part of a / 2pi that r1 lost (exact by fma) + a * C_LO, sin(2 pi (fract(r1) + r2)) -- accurate for every finite a.  Measured
// over the encodings' range (tools/sin_probe.hip, 1.7e7 arguments up to 705 rad): max |error| 4.2e-7 against 6.8e-8 for sinf -- below
// the 2^-22 x 16 granularity of the fp16 hi/lo operand split the value goes through next -- for 6 instructions instead of the ~90 of
// the inlined sinf (with its Payne-Hanek path), 48 times per point and view.  Used by the default (f16x3) inference kernel
// (points_mlp_f16.hip) and by the training path's point_inputs_kernel (train.hip); the exact-fp32 inference kernel (points_mlp.hip)
// keeps sinf.  -DDINER_PE_LIBM=1: sinf everywhere.
#ifndef DINER_PE_LIBM
#define DINER_PE_LIBM 0
#endif
__device__ __forceinline__ float pe_sin(float a)
{
    if (DINER_PE_LIBM) return sinf(a);
    const float C_HI = 0.15915494f;                                             // fp32(1 / 2pi)
    const float C_LO = (float)(0.15915494309189535 - (double)0.15915494f);
    const float r1 = a * C_HI;
    const float r2 = __builtin_fmaf(a, C_HI, -r1) + a * C_LO;
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r1) + r2);
}

// grid_sample(align_corners=False) un-normalisation, contracted as ATen compiles it
__device__ __forceinline__ float unnorm(float u, float half_size) { return __builtin_fmaf(u + 1.0f, half_size, -0.5f); }

// clamp_max(size-1, clamp_min(0, x)); NaN -> 0
__device__ __forceinline__ float clipf(float x, float hi)
{
    float y = (x > 0.0f) ? x : 0.0f;
    return (y < hi) ? y : hi;
}

// float pixel index -> int in [0,size-1] (callers have already range-checked finite values; the
// clamp makes every address safe for NaN/inf inputs as well)
__device__ __forceinline__ int safe_idx(float f, int size)
{
    int i = (int)f;  // v_cvt_i32_f32 saturates, NaN -> 0
    return i < 0 ? 0 : (i > size - 1 ? size - 1 : i);
}

// --------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (perf mode: statistically equivalent to torch.rand/randn, not
// bit-matching any torch generator -- parity tests inject explicit noise instead)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// --------------------------------------------------------------------------------------------
// wave-level primitives (64 lanes)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scans over the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_up(v, o, 64);
        if (lane >= o) v = n * v;
    }
    return v;
}
__device__ __forceinline__ int wave_scan_add_i(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}

}  // namespace diner
