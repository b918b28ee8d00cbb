// The two small producers on either side of the render path, per image,
// not per ray -- HBM-bound elementwise kernels, one thread per pixel:
//   gen_rays      reference src/util/cam_geometry.py:36-79   (producer of the `rays` tensor)
//   depth2normal  reference src/util/depth2normal.py:7-87    (normal maps inside PixelNeRF.encode)
//   decode_depth  reference src/data/dtu.py:90-124,220-223, src/data/facescape.py:54-56,68-106,266
//                 (TransMVSNet's uint16 depth / confidence planes -> depth [m], depth std)
#include "common.hpp"

namespace diner {

// rays [B,H,W,8] = origin(3), unit direction(3), near, far; pixel centres, OpenCV convention.
// extr [B,4,4] world->cam, intr [B,3,3], near/far [B].
__global__ void gen_rays_kernel(const float *__restrict__ extr, const float *__restrict__ intr,
                                const float *__restrict__ z_near, const float *__restrict__ z_far, int B, int H, int W,
                                float *__restrict__ rays)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int b = (int)(i / ((int64_t)H * W));
    const int p = (int)(i - (int64_t)b * H * W), y = p / W, x = p - y * W;
    const float *E = extr + b * 16, *Kk = intr + b * 9;
    const float fx = Kk[0], fy = Kk[4], cx = Kk[2], cy = Kk[5];
    float dx = (((float)x + 0.5f) - cx) / fx, dy = (((float)y + 0.5f) - cy) / fy, dz = 1.0f;  // :62-63
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);                                        // :64 pow(2).sum().sqrt()
    dx = dx / n; dy = dy / n; dz = dz / n;
    // world direction = R^T d (bmm: k-ordered FMA chain), origin = -R^T t (:67-72)
    float *o = rays + i * 8;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[3 + r] = __builtin_fmaf(E[2 * 4 + r], dz, __builtin_fmaf(E[1 * 4 + r], dy, E[0 * 4 + r] * dx));
        o[r] = __builtin_fmaf(-1.0f * E[2 * 4 + r], E[2 * 4 + 3], __builtin_fmaf(-1.0f * E[1 * 4 + r], E[1 * 4 + 3], (-1.0f * E[0 * 4 + r]) * E[0 * 4 + 3]));
    }
    o[6] = z_near[b];
    o[7] = z_far[b];
}

// One thread per OUTPUT pixel.  out(y, x) = in(y*stride, x*stride): torchvision's NEAREST resize for a downsample
// of 1/stride (dtu.py:113-117).  depth = ((u16 * mul0) / div) * mul1 with one rounding per operation, the order of
// dtu.py:104-105,119 (Facescape: div = mul1 = 1, facescape.py:80-91); std = a * conf + b with conf decoded the
// same way (dtu.py:220-223, facescape.py:54-56,266).  `mesh` (optional, Facescape depth_type "merge",
// facescape.py:96-104): mesh-rendered depth that wins wherever it is non-zero, with confidence 0.8.
__global__ void decode_depth_kernel(const unsigned short *__restrict__ depth, const unsigned short *__restrict__ conf,
                                    const unsigned short *__restrict__ mesh, int64_t N, int Hin, int Win, int stride, float mul0, float div,
                                    float mul1, float std_a, float std_b, float *__restrict__ depth_out, float *__restrict__ std_out,
                                    float *__restrict__ mask_out)
{
    const int Ho = Hin / stride, Wo = Win / stride;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Ho * Wo) return;
    const int64_t n = i / ((int64_t)Ho * Wo);
    const int p = (int)(i - n * Ho * Wo), y = p / Wo, x = p - y * Wo;
    const int64_t src = (n * Hin + (int64_t)y * stride) * Win + (int64_t)x * stride;
    float d = (float)depth[src] * mul0, c = (float)conf[src] * mul0;
    if (mesh) {
        const float m = (float)mesh[src] * mul0, mc = m == 0.0f ? 0.0f : 0.8f;
        d = (m == 0.0f && d != 0.0f) ? d : m;
        c = (mc == 0.0f && c != 0.0f) ? c : mc;
    }
    d = d / div;
    c = c / div;
    if (mask_out) mask_out[i] = d > 0.0f ? 1.0f : 0.0f;   // dtu.py:118 (before the scene scale)
    d = d * mul1;
    c = c * mul1;
    depth_out[i] = d;
    std_out[i] = std_a * c + std_b;
}

// camera-space point of pixel (x,y) with replicate padding (depth2normal.py:22-33)
__device__ __forceinline__ void d2n_point(const float *__restrict__ d, int H, int W, float fx, float fy, float cx, float cy,
                                          int x, int y, float &px, float &py, float &pz)
{
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    const float dep = d[(int64_t)y * W + x];
    px = ((((float)x + 0.5f) - cx) / fx) * dep;
    py = ((((float)y + 0.5f) - cy) / fy) * dep;
    pz = 1.0f * dep;
}

// un-cleaned normal at (x,y): normalize(cross(down - up, right - left)) (:47-55); also reports which of the
// four neighbours have x == 0 (the reference's "hole" test, :61-72)
__device__ __forceinline__ void d2n_raw(const float *__restrict__ d, int H, int W, float fx, float fy, float cx, float cy,
                                        int x, int y, float &nx, float &ny, float &nz, int &offy, int &offx)
{
    float dnx, dny, dnz, upx, upy, upz, rx, ry, rz, lx, ly, lz;
    d2n_point(d, H, W, fx, fy, cx, cy, x, y + 1, dnx, dny, dnz);
    d2n_point(d, H, W, fx, fy, cx, cy, x, y - 1, upx, upy, upz);
    d2n_point(d, H, W, fx, fy, cx, cy, x + 1, y, rx, ry, rz);
    d2n_point(d, H, W, fx, fy, cx, cy, x - 1, y, lx, ly, lz);
    const float vx = dnx - upx, vy = dny - upy, vz = dnz - upz, hx = rx - lx, hy = ry - ly, hz = rz - lz;
    const float cxp = vy * hz - vz * hy, cyp = vz * hx - vx * hz, czp = vx * hy - vy * hx;
    const float n = sqrtf(cxp * cxp + cyp * cyp + czp * czp);
    nx = cxp / n; ny = cyp / n; nz = czp / n;
    offy = (dnx == 0.0f ? -1 : 0) + (upx == 0.0f ? 1 : 0);
    offx = (rx == 0.0f ? -1 : 0) + (lx == 0.0f ? 1 : 0);
}

// dmap [N,1,H,W], intr [N,3,3] -> normals [N,3,H,W]
__global__ void depth2normal_kernel(const float *__restrict__ dmap, const float *__restrict__ intr, int N, int H, int W,
                                    float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int n = (int)(i / ((int64_t)H * W));
    const int p = (int)(i - (int64_t)n * H * W), y = p / W, x = p - y * W;
    const float *d = dmap + (int64_t)n * H * W, *Kk = intr + n * 9;
    const float fx = Kk[0], fy = Kk[4], cx = Kk[2], cy = Kk[5];
    float nx, ny, nz;
    int offy, offx;
    d2n_raw(d, H, W, fx, fy, cx, cy, x, y, nx, ny, nz, offy, offx);
    if (offy != 0 || offx != 0) {  // next to a hole: take the (un-cleaned) normal one pixel further inside (:74-78)
        int qy = y + offy, qx = x + offx, t0, t1;
        qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
        qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);
        d2n_raw(d, H, W, fx, fy, cx, cy, qx, qy, nx, ny, nz, t0, t1);
    }
    if (d[p] == 0.0f) { nx = 0.0f; ny = 0.0f; nz = 0.0f; }                        // :79
    float *o = out + (int64_t)n * 3 * H * W + p;
    o[0] = nx; o[(int64_t)H * W] = ny; o[2 * (int64_t)H * W] = nz;
}

// depth2normal fused into the map packing: depths/depths_std [N,1,H,W] + K [N,3,3]
// -> maps [N,H,W,8] = nx ny nz depth | sigma 0 0 0 without materialising the NCHW normal tensor.
__global__ void pack_maps_from_depth_kernel(const float *__restrict__ dmap, const float *__restrict__ dstd,
                                            const float *__restrict__ intr, int N, int H, int W, float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * H * W) return;
    const int n = (int)(i / ((int64_t)H * W));
    const int p = (int)(i - (int64_t)n * H * W), y = p / W, x = p - y * W;
    const float *d = dmap + (int64_t)n * H * W, *Kk = intr + n * 9;
    const float fx = Kk[0], fy = Kk[4], cx = Kk[2], cy = Kk[5];
    float nx, ny, nz;
    int offy, offx;
    d2n_raw(d, H, W, fx, fy, cx, cy, x, y, nx, ny, nz, offy, offx);
    if (offy != 0 || offx != 0) {
        int qy = y + offy, qx = x + offx, t0, t1;
        qy = qy < 0 ? 0 : (qy > H - 1 ? H - 1 : qy);
        qx = qx < 0 ? 0 : (qx > W - 1 ? W - 1 : qx);
        d2n_raw(d, H, W, fx, fy, cx, cy, qx, qy, nx, ny, nz, t0, t1);
    }
    if (d[p] == 0.0f) { nx = 0.0f; ny = 0.0f; nz = 0.0f; }
    out[i * 2 + 0] = make_float4(nx, ny, nz, d[p]);
    out[i * 2 + 1] = make_float4(dstd[i], 0.f, 0.f, 0.f);
}

int launch_pack_maps_from_depth(const float *dmap, const float *dstd, const float *intr, int N, int H, int W, float *out,
                                hipStream_t st)
{
    const int64_t total = (int64_t)N * H * W;
    if (total == 0) return DINER_OK;
    hipLaunchKernelGGL(pack_maps_from_depth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dmap, dstd, intr, N,
                       H, W, (float4 *)out);
    return check_launch("pack_maps_from_depth_kernel");
}

int launch_gen_rays(const float *extr, const float *intr, const float *zn, const float *zf, int B, int H, int W, float *rays,
                    hipStream_t st)
{
    const int64_t total = (int64_t)B * H * W;
    if (total == 0) return DINER_OK;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, extr, intr, zn, zf, B, H, W, rays);
    return check_launch("gen_rays_kernel");
}

int launch_depth2normal(const float *dmap, const float *intr, int N, int H, int W, float *out, hipStream_t st)
{
    const int64_t total = (int64_t)N * H * W;
    if (total == 0) return DINER_OK;
    hipLaunchKernelGGL(depth2normal_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dmap, intr, N, H, W, out);
    return check_launch("depth2normal_kernel");
}

int launch_decode_depth(const unsigned short *depth, const unsigned short *conf, const unsigned short *mesh, int64_t N, int Hin, int Win,
                        int stride, float mul0, float div, float mul1, float std_a, float std_b, float *depth_out, float *std_out,
                        float *mask_out, hipStream_t st)
{
    const int64_t n = N * (Hin / stride) * (Win / stride);
    if (n == 0) return DINER_OK;
    hipLaunchKernelGGL(decode_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, depth, conf, mesh, N, Hin, Win, stride, mul0,
                       div, mul1, std_a, std_b, depth_out, std_out, mask_out);
    return check_launch("decode_depth_kernel");
}

}  // namespace diner
