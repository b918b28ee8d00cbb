// Alpha compositing of one ray per wavefront -- replaces the tail of
// NeRFRendererDGS.composite (reference src/models/nerf_renderer.py:299-301, 341-360).
//
// HBM-bound: per ray it reads K*(16+4) B (rgbsigma, z) + 32 B (ray) and writes 16 B (+4K if the
// weights are requested).  Lanes stride over the K samples (float4 loads of rgbsigma are
// coalesced: 1 KiB per wave instruction); the transmittance prod_{i<k}(1 - alpha_i + 1e-10) is a
// multiplicative wave scan (6 shuffle steps) with a carry between 64-sample chunks; the
// weighted sums are butterfly reductions.
#include "common.hpp"

namespace diner {

constexpr int COMPOSITE_WAVES = 4;  // rays per 256-thread workgroup

__global__ __launch_bounds__(COMPOSITE_WAVES * 64) void composite_kernel(
    const float *__restrict__ rays, const float *__restrict__ z, const float4 *__restrict__ rgbsigma,
    int64_t N, int K, int white_bkgd, float *__restrict__ rgb_out, float *__restrict__ depth_out,
    float *__restrict__ weights_out, unsigned int *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * COMPOSITE_WAVES + (threadIdx.x >> 6);
    if (ray >= N) return;  // whole wave exits together
    const float far = rays[ray * 8 + 7];
    const float *zr = z + ray * K;
    const float4 *cr = rgbsigma + ray * K;
    float carry = 1.0f, acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
    bool bad = false;  // a non-finite rgb-sigma sample (an MLP activation left the range of the arithmetic in use)
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const bool on = k < K;
        float zk = 0.f, zn = 0.f;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
            zk = zr[k];
            zn = (k + 1 < K) ? zr[k + 1] : far;       // :299-301 last delta reaches to `far`
            c = cr[k];
        }
        const float delta = zn - zk;
        bad |= !(__builtin_isfinite(c.x) && __builtin_isfinite(c.y) && __builtin_isfinite(c.z) && __builtin_isfinite(c.w));
        const float sg = c.w < 0.0f ? 0.0f : c.w;      // relu applied again (:344); NaN stays NaN like torch.relu
        float alpha = 1.0f - expf(-delta * sg);
        float keep = 1.0f - alpha + 1e-10f;            // :347-349
        if (!on) { alpha = 0.0f; keep = 1.0f; }
        const float incl = wave_scan_mul(keep, lane);  // prod_{i<=k} within the chunk
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = carry * excl;                  // cumprod([1, keep...])[k]  (:350)
        const float w = alpha * T;                     // :351
        carry = carry * __shfl(incl, 63, 64);
        if (on && weights_out) weights_out[ray * K + k] = w;
        acc_r += w * c.x; acc_g += w * c.y; acc_b += w * c.z;   // :355
        acc_d += w * zk;                                          // :356
        acc_w += w;
    }
    acc_r = wave_sum(acc_r); acc_g = wave_sum(acc_g); acc_b = wave_sum(acc_b);
    acc_d = wave_sum(acc_d); acc_w = wave_sum(acc_w);
    if (status && __any(bad) && lane == 0) atomicOr(status, DINER_STATUS_NONFINITE);  // rare: only when something is wrong
    if (lane == 0) {
        if (white_bkgd) {                               // :357-360
            acc_r = acc_r + 1.0f - acc_w; acc_g = acc_g + 1.0f - acc_w; acc_b = acc_b + 1.0f - acc_w;
        }
        rgb_out[ray * 3 + 0] = acc_r; rgb_out[ray * 3 + 1] = acc_g; rgb_out[ray * 3 + 2] = acc_b;
        depth_out[ray] = acc_d;
    }
}

int launch_composite(const float *rays, const float *z, const float *rgbsigma, int64_t N, int K,
                     int white_bkgd, float *rgb, float *depth, float *weights, unsigned int *status, hipStream_t st)
{
    if (N == 0) return DINER_OK;
    const int64_t blocks = (N + COMPOSITE_WAVES - 1) / COMPOSITE_WAVES;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(COMPOSITE_WAVES * 64), 0, st, rays, z,
                       (const float4 *)rgbsigma, N, K, white_bkgd, rgb, depth, weights, status);
    return check_launch("composite_kernel");
}

}  // namespace diner
