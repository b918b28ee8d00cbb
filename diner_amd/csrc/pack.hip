// Once-per-encode() re-packing of the model's maps into the layouts the kernels read.
// Pure data movement (HBM-bound, coalesced on the write side; the strided reads of the NCHW
// latent are staged through LDS so both sides move full lines).
#include "common.hpp"

namespace diner {

// depths/depths_std [N,1,H,W], normals [N,3,H,W] -> maps [N,H,W,8] = nx ny nz depth | sigma 0 0 0
// One texel is 32 B so the three nearest look-ups of the sampler (image_encoder.py:129-204) hit one
// sector when they agree on the texel (always, inside the image).
__global__ void pack_maps_kernel(const float *__restrict__ d, const float *__restrict__ s,
                                 const float *__restrict__ n, int64_t N, int64_t HW, float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * HW) return;
    const int64_t img = i / HW, p = i - img * HW;
    const float *nb = n + img * 3 * HW + p;
    out[i * 2 + 0] = make_float4(nb[0], nb[HW], nb[2 * HW], d[i]);
    out[i * 2 + 1] = make_float4(s[i], 0.f, 0.f, 0.f);
}

// latent [N,C,h,w] -> [N,h,w,C] (a texel = 2 KB contiguous).  Tile: 32 pixels x C channels via LDS.
template <int C>
__global__ __launch_bounds__(256) void pack_latent_kernel(const float *__restrict__ in, int64_t hw,
                                                          float *__restrict__ out)
{
    __shared__ float tile[32][C + 1];
    const int64_t img = blockIdx.y, p0 = (int64_t)blockIdx.x * 32;
    const float *src = in + img * C * hw;
    const int px = threadIdx.x & 31, c0 = threadIdx.x >> 5;
    for (int c = c0; c < C; c += 8)
        tile[px][c] = (p0 + px < hw) ? src[(int64_t)c * hw + p0 + px] : 0.f;
    __syncthreads();
    float *dst = out + (img * hw + p0) * C;
    for (int i = threadIdx.x; i < 32 * C; i += 256) {
        const int p = i / C, k = i - p * C;
        if (p0 + p < hw) dst[(int64_t)p * C + k] = tile[p][k];
    }
}

int launch_pack_maps(const float *d, const float *s, const float *n, int64_t N, int H, int W, float *out,
                     hipStream_t st)
{
    const int64_t total = N * H * W;
    if (total == 0) return DINER_OK;
    hipLaunchKernelGGL(pack_maps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, s, n, N,
                       (int64_t)H * W, (float4 *)out);
    return check_launch("pack_maps_kernel");
}

int launch_pack_latent(const float *in, int64_t N, int C, int h, int w, float *out, hipStream_t st)
{
    if (C != DINER_D_LATENT) { set_error("pack_latent: C=%d unsupported (need %d)", C, DINER_D_LATENT); return DINER_E_UNSUPPORTED; }
    const int64_t hw = (int64_t)h * w;
    if (N * hw == 0) return DINER_OK;
    hipLaunchKernelGGL(pack_latent_kernel<DINER_D_LATENT>, dim3((unsigned)((hw + 31) / 32), (unsigned)N), dim3(256), 0, st,
                       in, hw, out);
    return check_launch("pack_latent_kernel");
}

}  // namespace diner
