// Depth-guided ray sampler, one ray per wavefront -- replaces
//   NeRFRendererDGS.sample_coarse            (reference src/models/nerf_renderer.py:39-63)
//   NeRFRendererDGS.sample_depthguided       (:65-284)
//   NeRFRendererDGS.fill_up_uniform_samples  (:367-397)
// and the map look-ups they call (src/models/image_encoder.py:129-204,
// src/util/torch_helpers.py:100-160,294-302).
//
// Layout: the NC candidates of a ray are striped over the 64 lanes (candidate j = 64*c + lane), so
// neighbouring lanes project to neighbouring texels of the packed [H,W,8] map (one 32-B texel holds
// normal, depth and sigma) and a wave instruction touches a handful of cache lines; everything
// per candidate stays in registers (no [NV,NR*NC] temporaries, the reference's ~40 of them are
// what makes its sampler bandwidth-bound).  Cross-lane work -- the occlusion cumprod, the top-(K-G)
// selection, the weighted mean/std -- uses wave ballots, shuffles and scans; the final K samples
// are bitonic-sorted in LDS.
//
// Selection: the reference sorts all NC likelihoods (argsort, :172) but only the SET of the K-G
// largest matters afterwards (everything is re-sorted by z in fill_up, :396).  The kernel finds the
// (K-G)-th largest likelihood by a 30-step bisection on the float bit pattern (likelihoods are in
// [0,1], so the bit pattern is monotonic) counting with ballots, then keeps everything above it and
// the lowest-index ties -- no sort of the candidates at all.
#include "common.hpp"

namespace diner {

struct MapGeo {
    int H, W;
    float halfW, halfH, Wm1, Hm1;          // depth / normal maps
    float sfx, sfy, halfWp, halfHp, Wpm1, Hpm1;  // sigma map extended by DINER_SIGMA_PAD
};

// Likelihood of one candidate under one source view -- nerf_renderer.py:107-128.
// tex: the view's packed map, 2 float4 per texel = {nx,ny,nz,depth},{sigma,0,0,0}.
__device__ __forceinline__ float view_likelihood(const float4 *__restrict__ tex, const View &vw, const MapGeo &g,
                                                 float iw, float ih, float dcx, float dcy, float dcz,
                                                 float x, float y, float z, float half_step, float ddmax)
{
    float px, py, pz, u, w;
    project(vw, iw, ih, x, y, z, px, py, pz, u, w);
    // depth: nearest, border (image_encoder.py:129-151)
    const float ux = unnorm(u, g.halfW), uy = unnorm(w, g.halfH);
    const int dx = safe_idx(__builtin_rintf(clipf(ux, g.Wm1)), g.W), dy = safe_idx(__builtin_rintf(clipf(uy, g.Hm1)), g.H);
    const float depth = tex[((int64_t)dy * g.W + dx) * 2].w;
    if (!(fabsf(depth - pz) < ddmax)) return 0.0f;                      // depth_dist_mask (:122)
    // sigma: nearest in the exponentially padded map, zeros outside (image_encoder.py:153-180,
    // torch_helpers.py:100-160): ring at Chebyshev distance d>=1 = border * exp((d-1)/12 * ln 2)
    const float jx = __builtin_rintf(unnorm(u * g.sfx, g.halfWp)), jy = __builtin_rintf(unnorm(w * g.sfy, g.halfHp));
    if (!(jx >= 0.0f && jx <= g.Wpm1 && jy >= 0.0f && jy <= g.Hpm1)) return 0.0f;
    const int sx = (int)jx - DINER_SIGMA_PAD, sy = (int)jy - DINER_SIGMA_PAD;
    const int ox = sx < 0 ? -sx : (sx > g.W - 1 ? sx - (g.W - 1) : 0);
    const int oy = sy < 0 ? -sy : (sy > g.H - 1 ? sy - (g.H - 1) : 0);
    const int cheb = ox > oy ? ox : oy;
    const int cx = sx < 0 ? 0 : (sx > g.W - 1 ? g.W - 1 : sx), cy = sy < 0 ? 0 : (sy > g.H - 1 ? g.H - 1 : sy);
    float sigma = tex[((int64_t)cy * g.W + cx) * 2 + 1].x;
    if (cheb > 1) sigma = sigma * expf((float)(cheb - 1) / 12.0f * 0.6931471805599453f);
    if (sigma == 0.0f) return 0.0f;                                      // bg_mask (:123)
    // normal: nearest, zeros (image_encoder.py:182-204)
    const float nxf = __builtin_rintf(ux), nyf = __builtin_rintf(uy);
    float cosd = 0.0f;
    if (nxf >= 0.0f && nxf <= g.Wm1 && nyf >= 0.0f && nyf <= g.Hm1) {
        const float4 n = tex[((int64_t)(int)nyf * g.W + (int)nxf) * 2];
        cosd = dcx * n.x + dcy * n.y + dcz * n.z;                        // :119
    }
    if (!(cosd <= 0.0f)) return 0.0f;                                    // :121
    const float den = sigma * 1.4142135623730951f;
    const float a = erff((pz + half_step - depth) / den), b = erff((pz - half_step - depth) / den);
    return fabsf(0.5f * (a - b));                                        // :125-128
}

// fill_up_uniform_samples (:376-396) on the K values of one ray held in LDS (one wave):
// after the reference's first sort the m zeros sit in columns n_neg .. n_neg+m-1 (n_neg = number of
// negative samples, normally 0), so the i-th empty slot becomes near + (n_neg+i)*step + u_i*step;
// then one ascending bitonic sort (:396) over n2 = pow2 >= K with +inf padding.
__device__ __forceinline__ void fill_up_and_sort(float *zs, int K, float near, float far, const float *u_fill_row,
                                                 int64_t gray, uint2 key, int lane)
{
    int m = 0, n_neg = 0;
    for (int i0 = 0; i0 < K; i0 += 64) {
        const int i = i0 + lane;
        const float v = i < K ? zs[i] : 1.0f;
        m += __popcll(__ballot(v == 0.0f));
        n_neg += __popcll(__ballot(v < 0.0f));
    }
    if (m > 0) {
        const float step = (far - near) / (float)m;                   // :388
        int seen = 0;
        for (int i0 = 0; i0 < K; i0 += 64) {
            const int i = i0 + lane;
            const bool zero = i < K && zs[i] == 0.0f;
            const unsigned long long zm = __ballot(zero);
            if (zero) {
                const int rank = seen + __popcll(zm & ((1ull << lane) - 1ull));
                float u;
                if (u_fill_row) u = u_fill_row[rank];
                else u = u01(philox4x32(make_uint4((uint32_t)rank, 2u, (uint32_t)gray, (uint32_t)(gray >> 32)), key).x);
                const float zmiss = near + (float)(n_neg + rank) * step;  // :389
                zs[i] = zmiss + u * step;                                  // :390
            }
            seen += __popcll(zm);
        }
    }
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int i = K + lane; i < n2; i += 64) zs[i] = __builtin_inff();
    __syncthreads();
    for (int kk = 2; kk <= n2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const float a = zs[i], b = zs[l];
                const bool asc = (i & kk) == 0;
                if ((a > b) == asc) { zs[i] = b; zs[l] = a; }
            }
            __syncthreads();
        }
    }
}

// stratified candidate j of a ray (sample_coarse, :53-60): t_j = linspace(0, 1-1/NC, NC)[j] + u/NC,
// z = near*(1-t) + far*t; torch.linspace in fp32 evaluates symmetrically from both ends.
struct CandGeo { float end, step, lstep; };
__device__ __forceinline__ CandGeo cand_geo(int NC)
{
    const double stepd = 1.0 / (double)NC;
    CandGeo g;
    g.end = (float)(1.0 - stepd); g.step = (float)stepd; g.lstep = g.end / (float)(NC - 1);
    return g;
}
__device__ __forceinline__ float candidate_from_u(int j, int NC, float near, float far, float u, const CandGeo &g)
{
    float t = (NC == 1) ? 0.0f : (j < NC / 2 ? g.lstep * (float)j : g.end - g.lstep * (float)(NC - j - 1));
    t = t + u * g.step;
    return near * (1.0f - t) + far * t;
}
// in-kernel noise of candidate j: one Philox call yields the 4 candidates {lane + 64*(4g .. 4g+3)} of a lane
__device__ __forceinline__ float candidate_u_philox(int j, int64_t gray, uint2 key)
{
    const int comp = (j >> 6) & 3;
    const uint4 r = philox4x32(make_uint4((uint32_t)((j & 63) | ((j >> 8) << 6)), 0u, (uint32_t)gray, (uint32_t)(gray >> 32)), key);
    return u01(comp == 0 ? r.x : comp == 1 ? r.y : comp == 2 ? r.z : r.w);
}

template <int CPL>  // candidates per lane: NC <= 64*CPL
__global__ __launch_bounds__(64) void sampler_kernel(
    DinerScene s, const float *__restrict__ rays, DinerTargetCam cam, float *__restrict__ rays_gen, int64_t NR, DinerSamplerCfg cfg,
    const float *__restrict__ u_coarse, const float *__restrict__ n_gauss, const float *__restrict__ u_fill,
    const float *__restrict__ z_cand, uint64_t seed, float *__restrict__ z_out, float *__restrict__ z_dg_out,
    float *__restrict__ lik_out)
{
    extern __shared__ float zs[];  // n2 = pow2 >= K floats
    const int lane = threadIdx.x;
    const int sb = blockIdx.y;
    const int64_t ray = blockIdx.x, gray = (int64_t)sb * NR + ray;
    const int NC = cfg.n_candidates, K = cfg.n_samples, G = cfg.n_gaussian, keep = K - G;
    float ox, oy, oz, dx, dy, dz, near, far;
    if (cam.extrinsics) {
        // gen_rays fused (src/util/cam_geometry.py:36-79, same arithmetic as encode_glue.hip's gen_rays_kernel): ray `ray` of scene
        // sb is pixel (ray / W, ray % W) of the target camera; stored once for the point kernel and the compositing
        const float *E = cam.extrinsics + sb * 16, *Kk = cam.intrinsics + sb * 9;
        const int py = (int)(ray / cam.W), px = (int)(ray - (int64_t)py * cam.W);
        float cx_ = (((float)px + 0.5f) - Kk[2]) / Kk[0], cy_ = (((float)py + 0.5f) - Kk[5]) / Kk[4], cz_ = 1.0f;   // :62-63
        const float n = sqrtf(cx_ * cx_ + cy_ * cy_ + cz_ * cz_);                                                   // :64
        cx_ = cx_ / n; cy_ = cy_ / n; cz_ = cz_ / n;
        float o3[3], d3[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {   // world direction = R^T d, origin = -R^T t (:67-72)
            d3[r] = __builtin_fmaf(E[2 * 4 + r], cz_, __builtin_fmaf(E[1 * 4 + r], cy_, E[0 * 4 + r] * cx_));
            o3[r] = __builtin_fmaf(-1.0f * E[2 * 4 + r], E[2 * 4 + 3], __builtin_fmaf(-1.0f * E[1 * 4 + r], E[1 * 4 + 3], (-1.0f * E[0 * 4 + r]) * E[0 * 4 + 3]));
        }
        ox = o3[0]; oy = o3[1]; oz = o3[2]; dx = d3[0]; dy = d3[1]; dz = d3[2];
        near = cam.z_near[sb]; far = cam.z_far[sb];
        if (lane < 8) rays_gen[gray * 8 + lane] = lane == 0 ? ox : lane == 1 ? oy : lane == 2 ? oz : lane == 3 ? dx : lane == 4 ? dy : lane == 5 ? dz : lane == 6 ? near : far;
    } else {
        const float *rp = rays + gray * 8;
        ox = rp[0]; oy = rp[1]; oz = rp[2]; dx = rp[3]; dy = rp[4]; dz = rp[5]; near = rp[6]; far = rp[7];
    }
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));

    // ---- candidates (sample_coarse, :53-60) ---------------------------------------------------
    float z[CPL], L[CPL];
    {
        const CandGeo cg = cand_geo(NC);
#pragma unroll
        for (int g4 = 0; g4 < CPL / 4; ++g4) {  // 4 chunks (candidates lane + 64*(4*g4 + 0..3)) share one Philox call
            uint4 rnd = make_uint4(0u, 0u, 0u, 0u);
            if (!z_cand && !u_coarse) rnd = philox4x32(make_uint4((uint32_t)(lane | (g4 << 6)), 0u, (uint32_t)gray, (uint32_t)(gray >> 32)), key);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = g4 * 4 + cc, j = c * 64 + lane;
                L[c] = 0.0f;
                z[c] = 0.0f;
                if (j >= NC) continue;
                if (z_cand) { z[c] = z_cand[gray * NC + j]; continue; }
                const float u = u_coarse ? u_coarse[gray * NC + j] : u01(cc == 0 ? rnd.x : cc == 1 ? rnd.y : cc == 2 ? rnd.z : rnd.w);
                z[c] = candidate_from_u(j, NC, near, far, u, cg);
            }
        }
    }

    // ---- per-view surface likelihood, max over views (:94-129) ----------------------------------
    MapGeo g;
    g.H = s.H; g.W = s.W;
    g.halfW = (float)s.W / 2.0f; g.halfH = (float)s.H / 2.0f; g.Wm1 = (float)(s.W - 1); g.Hm1 = (float)(s.H - 1);
    const int Wp = s.W + 2 * DINER_SIGMA_PAD, Hp = s.H + 2 * DINER_SIGMA_PAD;
    g.sfx = (float)s.W / (float)Wp; g.sfy = (float)s.H / (float)Hp;
    g.halfWp = (float)Wp / 2.0f; g.halfHp = (float)Hp / 2.0f; g.Wpm1 = (float)(Wp - 1); g.Hpm1 = (float)(Hp - 1);
    const float half_step = ((far - near) / (float)NC) / 2.0f;  // :95, :126
    for (int v = 0; v < s.NV; ++v) {
        const View vw = load_view(s, sb, v);
        float dcx, dcy, dcz;
        rotate(vw, dx, dy, dz, dcx, dcy, dcz);  // raydirs_cam (:103)
        const float4 *tex = (const float4 *)s.maps + ((int64_t)sb * s.NV + v) * s.H * s.W * 2;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = c * 64 + lane;
            if (j < NC) {
                const float x = ox + z[c] * dx, y = oy + z[c] * dy, zz = oz + z[c] * dz;  // :96
                const float l = view_likelihood(tex, vw, g, s.image_w, s.image_h, dcx, dcy, dcz, x, y, zz,
                                                half_step, cfg.depth_diff_max);
                L[c] = l > L[c] ? l : L[c];
            }
        }
    }
    if (lik_out) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (c * 64 + lane < NC) lik_out[gray * NC + c * 64 + lane] = L[c];
    }

    // ---- occlusion-aware likelihood O_j = L_j * prod_{i<j}(1-L_i) and its moments (:131-132,181-185)
    float wsum = 0.0f;
    float O[CPL];
    {
        float carry = 1.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const float incl = wave_scan_mul(1.0f - L[c], lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            O[c] = L[c] * (carry * excl);
            carry = carry * __shfl(incl, 63, 64);
            wsum += O[c];
        }
    }
    bool hit = false;
#pragma unroll
    for (int c = 0; c < CPL; ++c) hit |= (O[c] != 0.0f);
    hit = __any(hit);
    float gmean = 0.0f, gstd = 0.0f;
    if (G > 0 && hit) {  // weighted_mean_n_std, torch_helpers.py:294-302
        wsum = wave_sum(wsum);
        float m = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) m += z[c] * (O[c] / wsum);
        gmean = wave_sum(m);
        float var = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const float d = z[c] - gmean; var += d * d * (O[c] / wsum); }
        gstd = sqrtf(wave_sum(var));
    }

    // ---- top-(K-G) selection by likelihood, ties to the lower index (:172-178) ------------------
    int n_kept = 0;
    if (keep > 0) {
        uint32_t T = 0;
        int n_nz = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) n_nz += __popcll(__ballot(L[c] > 0.0f));
        // fewer non-zero likelihoods than slots: everything non-zero is kept, no threshold to search for
        for (int b = (n_nz <= keep) ? -1 : 29; b >= 0; --b) {
            const uint32_t cand = T | (1u << b);
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) cnt += __popcll(__ballot(__float_as_uint(L[c]) >= cand));
            if (cnt >= keep) T = cand;
        }
        // T = bit pattern of the keep-th largest likelihood, or 0 if fewer than `keep` are non-zero
        int n_gt = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) n_gt += __popcll(__ballot(__float_as_uint(L[c]) > T));
        const int n_eq = (T == 0) ? 0 : keep - n_gt;  // ties at the cut that are still kept
        int seen_eq = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const uint32_t k = __float_as_uint(L[c]);
            const unsigned long long eqm = __ballot(k == T && T != 0);
            const int my_eq_rank = seen_eq + __popcll(eqm & ((1ull << lane) - 1ull));
            const bool kept = (k > T) || (k == T && T != 0 && my_eq_rank < n_eq);
            const unsigned long long km = __ballot(kept);
            if (kept) zs[n_kept + __popcll(km & ((1ull << lane) - 1ull))] = z[c];  // z=0 where L==0: not kept
            n_kept += __popcll(km);
            seen_eq += __popcll(eqm);
        }
    }
    for (int i = n_kept + lane; i < keep; i += 64) zs[i] = 0.0f;     // empty slots (:176-178)
    for (int i = lane; i < G; i += 64) {                              // gaussian slots (:186-190)
        float val = 0.0f;
        if (hit) {
            float n;
            if (n_gauss) n = n_gauss[gray * G + i];
            else {
                const uint4 r = philox4x32(make_uint4((uint32_t)i, 1u, (uint32_t)gray, (uint32_t)(gray >> 32)), key);
                n = sqrtf(-2.0f * logf(1.0f - u01(r.x))) * cosf(6.283185307179586f * u01(r.y));
            }
            val = n * gstd + gmean;
        }
        zs[keep + i] = val;
    }
    __syncthreads();
    if (z_dg_out)
        for (int i = lane; i < K; i += 64) z_dg_out[gray * K + i] = zs[i];

    fill_up_and_sort(zs, K, near, far, u_fill ? u_fill + gray * K : nullptr, gray, key, lane);
    for (int i = lane; i < K; i += 64) z_out[gray * K + i] = zs[i];
}

// standalone stages (reference stage boundaries) ------------------------------------------------
__global__ __launch_bounds__(256) void sample_coarse_kernel(const float *__restrict__ rays, int64_t N, int NC,
                                                            const float *__restrict__ u_coarse, uint64_t seed,
                                                            float *__restrict__ z_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * NC) return;
    const int64_t ray = i / NC;
    const int j = (int)(i - ray * NC);
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    const float u = u_coarse ? u_coarse[ray * NC + j] : candidate_u_philox(j, ray, key);
    z_out[i] = candidate_from_u(j, NC, rays[ray * 8 + 6], rays[ray * 8 + 7], u, cand_geo(NC));
}

__global__ __launch_bounds__(64) void fill_up_kernel(const float *__restrict__ rays, const float *__restrict__ z_in,
                                                     int64_t N, int K, const float *__restrict__ u_fill, uint64_t seed,
                                                     float *__restrict__ z_out)
{
    extern __shared__ float zs[];
    const int lane = threadIdx.x;
    const int64_t ray = blockIdx.x;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = lane; i < K; i += 64) zs[i] = z_in[ray * K + i];
    __syncthreads();
    fill_up_and_sort(zs, K, rays[ray * 8 + 6], rays[ray * 8 + 7], u_fill ? u_fill + ray * K : nullptr, ray, key, lane);
    for (int i = lane; i < K; i += 64) z_out[ray * K + i] = zs[i];
}

static size_t sort_lds_bytes(int K)
{
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    return sizeof(float) * (size_t)n2;
}

int launch_sample_coarse(const float *rays, int64_t N, int NC, const float *u_coarse, uint64_t seed, float *z_out,
                         hipStream_t st)
{
    if (N * NC == 0) return DINER_OK;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((N * NC + 255) / 256)), dim3(256), 0, st, rays, N, NC,
                       u_coarse, seed, z_out);
    return check_launch("sample_coarse_kernel");
}

int launch_fill_up(const float *rays, const float *z_in, int64_t N, int K, const float *u_fill, uint64_t seed,
                   float *z_out, hipStream_t st)
{
    if (N == 0) return DINER_OK;
    hipLaunchKernelGGL(fill_up_kernel, dim3((unsigned)N), dim3(64), sort_lds_bytes(K), st, rays, z_in, N, K, u_fill,
                       seed, z_out);
    return check_launch("fill_up_kernel");
}

int launch_sampler(const DinerScene &s, const float *rays, const DinerTargetCam *cam, float *rays_gen, int64_t NR, const DinerSamplerCfg &cfg,
                   const float *u_coarse, const float *n_gauss, const float *u_fill, const float *z_cand,
                   uint64_t seed, float *z_out, float *z_dg_out, float *lik_out, hipStream_t st)
{
    if (NR == 0 || s.SB == 0) return DINER_OK;
    const size_t lds = sort_lds_bytes(cfg.n_samples);
    const dim3 grid((unsigned)NR, (unsigned)s.SB), block(64);
    DinerTargetCam tc = {nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (cam) tc = *cam;
#define DINER_LAUNCH_SAMPLER(CPL)                                                                              \
    hipLaunchKernelGGL(sampler_kernel<CPL>, grid, block, lds, st, s, rays, tc, rays_gen, NR, cfg, u_coarse, n_gauss, u_fill, \
                       z_cand, seed, z_out, z_dg_out, lik_out)
    const int NC = cfg.n_candidates;
    if (NC <= 64 * 4) DINER_LAUNCH_SAMPLER(4);
    else if (NC <= 64 * 16) DINER_LAUNCH_SAMPLER(16);
    else if (NC <= 64 * 32) DINER_LAUNCH_SAMPLER(32);
    else if (NC <= 64 * 64) DINER_LAUNCH_SAMPLER(64);   // (4096 candidates: 64 per lane stay in registers; beyond that they would spill)
    else { set_error("sampler: n_candidates=%d > 4096 unsupported", NC); return DINER_E_UNSUPPORTED; }
#undef DINER_LAUNCH_SAMPLER
    return check_launch("sampler_kernel");
}

}  // namespace diner
