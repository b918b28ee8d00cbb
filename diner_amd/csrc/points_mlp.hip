// Fused per-point evaluation: multi-view projection + positional encoding + bilinear latent gather
// + the 15-layer fusion MLP + sigmoid/relu head, one 64-point tile per 512-thread workgroup.
// Replaces, for points = o + z*d (reference src/models/nerf_renderer.py:304-307):
//   PixelNeRF.forward            src/models/pixelnerf.py:55-145
//   PositionalEncoding.forward   src/models/positional_encoding.py:33-53
//   SpatialEncoder.index / index_depth   src/models/image_encoder.py:97-151
//   ResnetFC.forward / ResnetBlockFC.forward / combine   src/models/resnetfc.py:129-159, 61-69, 9-14
//
// This is the MFMA-bound part of the path: 2*(NV*2,387,456 + 1,050,624) FLOP per point
// (21.2 MFLOP at NV=4), >99.9% of the path's arithmetic.  fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32, bit-identical to a k-ordered fmaf chain) because the parity bar is
// 1e-4 abs on RGB/sigma.
//
// Structure (8 waves = 2 per SIMD, <= 256 registers each):
//   * the hidden state x [64 points x 512] never leaves the register file: wave w owns columns
//     64w..64w+63 as a 2x2 grid of 32x32 accumulator tiles (64 registers); `net` (the block's
//     inner activation) and the running view-sum are two more such grids;
//   * the A operand of every layer (raw inputs, the gathered latent, relu(x), relu(net)) is staged
//     in ONE 128-KiB LDS image laid out [k/8][k%2][row][(k/2)%4], so that an MFMA A-fragment for
//     four consecutive k-pairs is a single conflict-free ds_read_b128;
//   * weights are pre-packed (diner_pack_mlp) in exactly the B-fragment order, so each wave streams
//     its 64 output columns with 1-KiB coalesced global_load_dwordx4 straight from L2 into
//     registers, software-pipelined one k-block (16 MFMAs = 1024 cycles) ahead; the second wave of
//     each SIMD covers what latency is left;
//   * views are processed one after another (rows = points, not point x view), so the two
//     post-mean blocks run on full 64-row tiles instead of NV-times-padded ones.
#include "common.hpp"

namespace diner {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_P = 64;           // points per workgroup
constexpr int NWAVES = 8;            // waves per workgroup
constexpr int CT = 16 / NWAVES;      // 32-column tiles per wave
constexpr int HID = DINER_D_HIDDEN;  // 512
constexpr int NJB_FULL = HID / 8;    // 64 k-blocks of 8
constexpr int NJB_IN = 7;            // lin_in: 55 inputs padded to 56
constexpr int A_F4 = NJB_FULL * 2 * TILE_P;  // float4 entries of the A image (8192 = 128 KiB)

// ---- packed weight image (shared with pack_mlp_kernel) -----------------------------------------
// layer order: 0 lin_in | 1..3 lin_z[b] | 4..8 fc_0[b] | 9..13 fc_1[b] | 14 lin_out
// layer block: [col_tile][jb][lane][4]: lane = h*32+c holds W[n = 32*col_tile + c][k = 8*jb + 2*ji + h], ji = 0..3
constexpr int64_t W_FULL = 16LL * NJB_FULL * 256;  // floats of one 512x512 layer
constexpr int64_t W_IN = 16LL * NJB_IN * 256;
constexpr int64_t W_OUT = 1LL * NJB_FULL * 256;
constexpr int64_t OFF_LIN_IN = 0;
constexpr int64_t OFF_LIN_Z = OFF_LIN_IN + W_IN;
constexpr int64_t OFF_FC0 = OFF_LIN_Z + 3 * W_FULL;
constexpr int64_t OFF_FC1 = OFF_FC0 + 5 * W_FULL;
constexpr int64_t OFF_LIN_OUT = OFF_FC1 + 5 * W_FULL;
constexpr int64_t OFF_BIAS = OFF_LIN_OUT + W_OUT;   // biases: 14 x 512, then lin_out padded to 32
constexpr int64_t PACKED_FLOATS = OFF_BIAS + 14 * 512 + 32;
__host__ __device__ constexpr int bias_slot_lin_in() { return 0; }
__host__ __device__ constexpr int bias_slot_lin_z(int b) { return 1 + b; }
__host__ __device__ constexpr int bias_slot_fc0(int b) { return 4 + b; }
__host__ __device__ constexpr int bias_slot_fc1(int b) { return 9 + b; }

int64_t mlp_packed_floats() { return PACKED_FLOATS; }

// one thread per packed weight float
__global__ void pack_mlp_kernel(DinerMlpRaw raw, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PACKED_FLOATS) return;
    if (i >= OFF_BIAS) {  // biases
        const int64_t b = i - OFF_BIAS;
        if (b >= 14 * 512) { const int c = (int)(b - 14 * 512); out[i] = c < 4 ? raw.lin_out_b[c] : 0.0f; return; }
        const int slot = (int)(b / 512), c = (int)(b % 512);
        const float *src = slot == 0 ? raw.lin_in_b : slot < 4 ? raw.lin_z_b[slot - 1] : slot < 9 ? raw.fc0_b[slot - 4] : raw.fc1_b[slot - 9];
        out[i] = src[c];
        return;
    }
    const float *w;
    int64_t rel;
    int njb, in_dim, out_dim;
    if (i < OFF_LIN_Z) { w = raw.lin_in_w; rel = i; njb = NJB_IN; in_dim = DINER_D_IN; out_dim = HID; }
    else if (i < OFF_FC0) { const int b = (int)((i - OFF_LIN_Z) / W_FULL); w = raw.lin_z_w[b]; rel = (i - OFF_LIN_Z) % W_FULL; njb = NJB_FULL; in_dim = HID; out_dim = HID; }
    else if (i < OFF_FC1) { const int b = (int)((i - OFF_FC0) / W_FULL); w = raw.fc0_w[b]; rel = (i - OFF_FC0) % W_FULL; njb = NJB_FULL; in_dim = HID; out_dim = HID; }
    else if (i < OFF_LIN_OUT) { const int b = (int)((i - OFF_FC1) / W_FULL); w = raw.fc1_w[b]; rel = (i - OFF_FC1) % W_FULL; njb = NJB_FULL; in_dim = HID; out_dim = HID; }
    else { w = raw.lin_out_w; rel = i - OFF_LIN_OUT; njb = NJB_FULL; in_dim = HID; out_dim = 4; }
    const int ji = (int)(rel & 3), lane = (int)((rel >> 2) & 63);
    const int64_t blk = rel >> 8;
    const int jb = (int)(blk % njb), tile = (int)(blk / njb);
    const int n = tile * 32 + (lane & 31), k = jb * 8 + ji * 2 + (lane >> 5);
    out[i] = (n < out_dim && k < in_dim) ? w[(int64_t)n * in_dim + k] : 0.0f;
}

int launch_pack_mlp(const DinerMlpRaw &raw, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(pack_mlp_kernel, dim3((unsigned)((PACKED_FLOATS + 255) / 256)), dim3(256), 0, st, raw, out);
    return check_launch("pack_mlp_kernel");
}

// ---- LDS A image ---------------------------------------------------------------------------------
// float offset of element (row, k): [k/8][k%2][row][(k/2)%4]
__device__ __forceinline__ int a_off(int row, int k) { return ((((k >> 3) * 2 + (k & 1)) * TILE_P + row) << 2) + ((k >> 1) & 3); }

// acc[tm][tn] += A[64 x 8*NJB] * W^T for this wave's 4 column tiles.
// A4: LDS image; Wl: packed layer block; wave w reads col tiles CT*w .. CT*w+CT-1.
template <int NJB>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[2][CT], const f32x4 *A4, const f32x4 *__restrict__ Wl, int wave, int lane)
{
    const f32x4 *ap = A4 + (lane >> 5) * TILE_P + (lane & 31);
    const f32x4 *bp = Wl + (int64_t)wave * CT * NJB * 64 + lane;
    f32x4 b_cur[CT], b_nxt[CT];
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) b_cur[tn] = bp[(int64_t)tn * NJB * 64];
#pragma unroll 2
    for (int jb = 0; jb < NJB; ++jb) {
        const int jn = jb + 1 < NJB ? jb + 1 : jb;  // last iteration re-loads (harmless, keeps the loop branch-free)
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) b_nxt[tn] = bp[((int64_t)tn * NJB + jn) * 64];
        const f32x4 a0 = ap[jb * 2 * TILE_P], a1 = ap[jb * 2 * TILE_P + 32];
#pragma unroll
        for (int ji = 0; ji < 4; ++ji) {
#pragma unroll
            for (int tn = 0; tn < CT; ++tn) {
                acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[ji], b_cur[tn][ji], acc[0][tn], 0, 0, 0);
                acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ji], b_cur[tn][ji], acc[1][tn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) b_cur[tn] = b_nxt[tn];
    }
}

__device__ __forceinline__ void acc_set_bias(f32x16 (&acc)[2][CT], const float *__restrict__ bias, int wave, int lane)
{
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) {
        const float b = bias[wave * (32 * CT) + tn * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][tn][i] = b; acc[1][tn][i] = b; }
    }
}
__device__ __forceinline__ void acc_add_bias(f32x16 (&acc)[2][CT], const float *__restrict__ bias, int wave, int lane)
{
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) {
        const float b = bias[wave * (32 * CT) + tn * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][tn][i] += b; acc[1][tn][i] += b; }
    }
}

// relu(acc) -> LDS A image (this wave's 64 columns become k = 64w .. 64w+63 of the next layer)
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[2][CT], float *A, int wave, int lane)
{
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) {
        const int k = wave * (32 * CT) + tn * 32 + c;
        float *col = A + ((((k >> 3) * 2 + (k & 1)) * TILE_P) << 2) + ((k >> 1) & 3);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = tm * 32 + 8 * (i >> 2) + 4 * h + (i & 3);  // C/D layout of the 32x32 MFMA
                const float v = acc[tm][tn][i];
                col[row << 2] = v < 0.0f ? 0.0f : v;  // keeps NaN, like torch.relu
            }
    }
}

struct Tap {        // bilinear footprint of one (point, view) in the latent map
    int o00, o01, o10, o11;  // float4 offsets of the 4 texels (clamped, always readable)
    float nw, ne, sw, se;    // weights; a tap outside the map has its weight forced to 0
};

__global__ __launch_bounds__(NWAVES * 64) void points_mlp_kernel(DinerScene s, const float *__restrict__ Wp,
                                                         const float *__restrict__ rays, const float *__restrict__ zsamp,
                                                         int64_t NR, int K, float *__restrict__ rgbsigma)
{
    __shared__ f32x4 lds[A_F4 + TILE_P * 2];  // A image + one Tap per row (all LDS in ONE array)
    f32x4 *A4 = lds;
    float *A = (float *)lds;
    Tap *taps = (Tap *)(lds + A_F4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sb = blockIdx.y;
    const int64_t P = NR * (int64_t)K;
    // XCD-aware tile order: consecutive tiles (neighbouring samples of a ray, then neighbouring rays ->
    // neighbouring latent texels) stay on one XCD's L2.  Bijective for any grid size.
    int64_t tile;
    {
        const int64_t nwg = gridDim.x, b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const float *bias = Wp + OFF_BIAS;

    // this thread's point for the geometry phase (row = tid & 63; the 8 waves split the 56 inputs)
    const int row = tid & 63;
    int64_t p = tile * TILE_P + row;
    if (p > P - 1) p = P - 1;  // tail tile: duplicate the last point, masked at the store
    const int64_t ray = p / K;
    const float *rp = rays + ((int64_t)sb * NR + ray) * 8;
    const float zz = zsamp[(int64_t)sb * P + p];
    const float dwx = rp[3], dwy = rp[4], dwz = rp[5];
    const float wx = rp[0] + zz * dwx, wy = rp[1] + zz * dwy, wz = rp[2] + zz * dwz;  // :304

    f32x16 x[2][CT], net[2][CT], xsum[2][CT];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < CT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) xsum[tm][tn][i] = 0.0f;

    const float sxl = ((float)s.w - s.feature_padding * 2.0f) / (float)s.w;  // image_encoder.py:113-114
    const float syl = ((float)s.h - s.feature_padding * 2.0f) / (float)s.h;

    for (int v = 0; v < s.NV; ++v) {
        // ---- geometry + positional encodings -> A[:, 0:56]; bilinear footprint -> taps ----------
        {
            const View vw = load_view(s, sb, v);
            float px, py, pz, u, w;
            project(vw, s.image_w, s.image_h, wx, wy, wz, px, py, pz, u, w);   // pixelnerf.py:91-93,105-108
            float dcx, dcy, dcz;
            rotate(vw, dwx, dwy, dwz, dcx, dcy, dcz);                            // :99-101
            const float4 *tex = (const float4 *)s.maps + ((int64_t)sb * s.NV + v) * s.H * s.W * 2;
            const int ddx = safe_idx(__builtin_rintf(clipf(unnorm(u, (float)s.W / 2.0f), (float)(s.W - 1))), s.W);
            const int ddy = safe_idx(__builtin_rintf(clipf(unnorm(w, (float)s.H / 2.0f), (float)(s.H - 1))), s.H);
            const float delta = tex[((int64_t)ddy * s.W + ddx) * 2].w - pz;     // :114-115
            const float half_pi = 1.5707963267948966f;
            for (int e = wave * 7; e < wave * 7 + 7; ++e) {                   // input layout :128
                float val;
                if (e < 3) val = e == 0 ? px : e == 1 ? py : pz;
                else if (e < 39) { const int j = (e - 3) / 3, i = (e - 3) % 3;    // positional_encoding.py:45-49
                    val = sinf(__builtin_fmaf(i == 0 ? px : i == 1 ? py : pz, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
                else if (e < 42) val = e == 39 ? dcx : e == 40 ? dcy : dcz;
                else if (e == 42) val = delta;
                else if (e < 55) { const int j = e - 43;
                    val = sinf(__builtin_fmaf(delta, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
                else val = 0.0f;
                A[a_off(row, e)] = val;
            }
            if (wave == 0) {  // bilinear / border footprint in the latent map (image_encoder.py:97-127)
                const float ix = clipf(unnorm(u * sxl, (float)s.w / 2.0f), (float)(s.w - 1));
                const float iy = clipf(unnorm(w * syl, (float)s.h / 2.0f), (float)(s.h - 1));
                const float x0f = floorf(ix), y0f = floorf(iy);
                const float fx = ix - x0f, ex = 1.0f - fx, fy = iy - y0f, ey = 1.0f - fy;
                const int x0 = safe_idx(x0f, s.w), y0 = safe_idx(y0f, s.h);
                const bool x1ok = x0 + 1 <= s.w - 1, y1ok = y0 + 1 <= s.h - 1;
                const int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
                Tap t;
                const int f4 = HID / 4;
                t.o00 = (y0 * s.w + x0) * f4; t.o01 = (y0 * s.w + x1) * f4;
                t.o10 = (y1 * s.w + x0) * f4; t.o11 = (y1 * s.w + x1) * f4;
                t.nw = ey * ex; t.ne = x1ok ? ey * fx : 0.0f;
                t.sw = y1ok ? fy * ex : 0.0f; t.se = (x1ok && y1ok) ? fy * fx : 0.0f;
                taps[row] = t;
            }
        }
        __syncthreads();
        acc_set_bias(x, bias + 512 * bias_slot_lin_in(), wave, lane);
        gemm_tile<NJB_IN>(x, A4, (const f32x4 *)(Wp + OFF_LIN_IN), wave, lane);   // resnetfc.py:139
        __syncthreads();

        const f32x4 *lat = (const f32x4 *)s.latent + ((int64_t)sb * s.NV + v) * s.h * s.w * (HID / 4);
        for (int b = 0; b < DINER_COMBINE_LAYER; ++b) {
            // ---- z = bilinear latent of the 64 points -> A (each wave gathers 8 rows) -----------
#pragma unroll 2
            for (int rr = 0; rr < TILE_P / NWAVES; ++rr) {
                const int r = wave * (TILE_P / NWAVES) + rr;
                const Tap t = taps[r];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int q = lane + 64 * half;  // float4 index inside the texel: channels 4q..4q+3
                    const f32x4 a = lat[t.o00 + q], bb = lat[t.o01 + q], c = lat[t.o10 + q], d = lat[t.o11 + q];
#pragma unroll
                    for (int i = 0; i < 4; ++i)  // ATen's accumulation order nw,ne,sw,se with contracted FMAs
                        A[a_off(r, 4 * q + i)] =
                            __builtin_fmaf(d[i], t.se, __builtin_fmaf(c[i], t.sw, __builtin_fmaf(bb[i], t.ne, a[i] * t.nw)));
                }
            }
            __syncthreads();
            acc_add_bias(x, bias + 512 * bias_slot_lin_z(b), wave, lane);             // :152-153 x = x + lin_z(z)
            gemm_tile<NJB_FULL>(x, A4, (const f32x4 *)(Wp + OFF_LIN_Z + b * W_FULL), wave, lane);
            __syncthreads();
            store_relu(x, A, wave, lane);                                              // :62 fc_0(relu(x))
            __syncthreads();
            acc_set_bias(net, bias + 512 * bias_slot_fc0(b), wave, lane);
            gemm_tile<NJB_FULL>(net, A4, (const f32x4 *)(Wp + OFF_FC0 + b * W_FULL), wave, lane);
            __syncthreads();
            store_relu(net, A, wave, lane);                                            // :63 fc_1(relu(net))
            __syncthreads();
            acc_add_bias(x, bias + 512 * bias_slot_fc1(b), wave, lane);               // :69 x + dx
            gemm_tile<NJB_FULL>(x, A4, (const f32x4 *)(Wp + OFF_FC1 + b * W_FULL), wave, lane);
            __syncthreads();
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < CT; ++tn) xsum[tm][tn] += x[tm][tn];                  // :146-149
    }
    {
        const float nv = (float)s.NV;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < CT; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) xsum[tm][tn][i] = xsum[tm][tn][i] / nv;  // combine(): mean over views
    }
    for (int b = DINER_COMBINE_LAYER; b < DINER_N_BLOCKS; ++b) {
        store_relu(xsum, A, wave, lane);
        __syncthreads();
        acc_set_bias(net, bias + 512 * bias_slot_fc0(b), wave, lane);
        gemm_tile<NJB_FULL>(net, A4, (const f32x4 *)(Wp + OFF_FC0 + b * W_FULL), wave, lane);
        __syncthreads();
        store_relu(net, A, wave, lane);
        __syncthreads();
        acc_add_bias(xsum, bias + 512 * bias_slot_fc1(b), wave, lane);
        gemm_tile<NJB_FULL>(xsum, A4, (const f32x4 *)(Wp + OFF_FC1 + b * W_FULL), wave, lane);
        __syncthreads();
    }
    store_relu(xsum, A, wave, lane);                                                   // :158 lin_out(relu(x))
    __syncthreads();
    if (wave < 2) {  // lin_out: one 32-column tile (4 real outputs), wave w = rows 32w..32w+31
        f32x16 o;
        const float bo = bias[14 * 512 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = bo;
        const f32x4 *ap = A4 + (lane >> 5) * TILE_P + wave * 32 + (lane & 31);
        const f32x4 *bp = (const f32x4 *)(Wp + OFF_LIN_OUT) + lane;
#pragma unroll 4
        for (int jb = 0; jb < NJB_FULL; ++jb) {
            const f32x4 a = ap[jb * 2 * TILE_P], bq = bp[jb * 64];
#pragma unroll
            for (int ji = 0; ji < 4; ++ji) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ji], bq[ji], o, 0, 0, 0);
        }
        const int c = lane & 31, h = lane >> 5;
        if (c < 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = wave * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                const int64_t pp = tile * TILE_P + r;
                if (pp < P) {
                    const float val = o[i];                                            // pixelnerf.py:139-143
                    rgbsigma[((int64_t)sb * P + pp) * 4 + c] = c < 3 ? 1.0f / (1.0f + expf(-val)) : (val < 0.0f ? 0.0f : val);
                }
            }
        }
    }
}

// ---- once per encode(): G_b = lin_z[b](latent) as three extra feature maps ------------------------
// lin_z is linear and SpatialEncoder.index is a convex combination of 4 texels, so
//   lin_z[b](bilerp(F)(uv)) = bilerp(lin_z[b](F))(uv)            (resnetfc.py:152, image_encoder.py:97-127)
// up to fp32 rounding.  Evaluating lin_z on the latent MAP once per encode (NV*h*w rows instead of
// NV * points rows per frame: 0.64 TFLOP instead of 211 TFLOP at the headline config) removes three of
// the nine per-view GEMMs from the per-point kernel, which then adds bilerp(G_b) straight into its
// accumulators.  Exact fp32 MFMA here (bit-identical to an fmaf chain), bias included -- and with it the bias the residual stream
// receives at the same place (the bilinear weights of a point sum to 1, so a constant of the map arrives as that constant):
//   map 0 also carries lin_in's bias      (x = lin_in(in) + b_in;  x += lin_z[0](z)        resnetfc.py:139,152),
//   map b >= 1 also carries fc_1[b-1]'s   (x = x + fc_1(...) + b1; x += lin_z[b](z)        resnetfc.py:69,152),
// so the per-point kernel adds three bias vectors per view fewer (points_mlp_f16.hip, LINZ path).
// in: latent [N,h,w,512] NHWC rows; out: [3][N,h,w,512].
__global__ __launch_bounds__(NWAVES * 64) void linz_maps_kernel(const float *__restrict__ latent, int64_t rows,
                                                                const float *__restrict__ Wp, float *__restrict__ out)
{
    __shared__ f32x4 lds[A_F4];
    f32x4 *A4 = lds;
    float *A = (float *)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * TILE_P;
    // stage 64 latent rows (2 KiB each) into the A image
    for (int rr = 0; rr < TILE_P / NWAVES; ++rr) {
        const int r = wave * (TILE_P / NWAVES) + rr;
        const int64_t gr = row0 + r < rows ? row0 + r : rows - 1;
        const f32x4 *src = (const f32x4 *)(latent + gr * HID);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q = lane + 64 * half;
            const f32x4 v = src[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) A[a_off(r, 4 * q + i)] = v[i];
        }
    }
    __syncthreads();
    const float *bias = Wp + OFF_BIAS;
    for (int b = 0; b < DINER_COMBINE_LAYER; ++b) {
        f32x16 acc[2][CT];
        acc_set_bias(acc, bias + 512 * bias_slot_lin_z(b), wave, lane);
        acc_add_bias(acc, bias + 512 * (b == 0 ? bias_slot_lin_in() : bias_slot_fc1(b - 1)), wave, lane);
        gemm_tile<NJB_FULL>(acc, A4, (const f32x4 *)(Wp + OFF_LIN_Z + b * W_FULL), wave, lane);
        float *dst = out + (int64_t)b * rows * HID;
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int tn = 0; tn < CT; ++tn)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t gr = row0 + tm * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                    if (gr < rows) dst[gr * HID + wave * (32 * CT) + tn * 32 + c] = acc[tm][tn][i];
                }
    }
}

int launch_linz_maps(const float *latent_nhwc, int64_t rows, const float *mlp_packed, float *out, hipStream_t st)
{
    if (rows == 0) return DINER_OK;
    hipLaunchKernelGGL(linz_maps_kernel, dim3((unsigned)((rows + TILE_P - 1) / TILE_P)), dim3(NWAVES * 64), 0, st, latent_nhwc,
                       rows, mlp_packed, out);
    return check_launch("linz_maps_kernel");
}

int launch_points_mlp(const DinerScene &s, const float *mlp_packed, const float *rays, const float *z, int64_t NR,
                      int K, float *rgbsigma, hipStream_t st)
{
    const int64_t P = NR * (int64_t)K;
    if (P == 0 || s.SB == 0) return DINER_OK;
    if (s.C != DINER_D_LATENT) { set_error("render_points: latent channels C=%d unsupported (need %d)", s.C, DINER_D_LATENT); return DINER_E_UNSUPPORTED; }
    if (s.num_freqs != 6) { set_error("render_points: num_freqs=%d unsupported (need 6)", s.num_freqs); return DINER_E_UNSUPPORTED; }
    const int64_t tiles = (P + TILE_P - 1) / TILE_P;
    if (tiles > 0x7fffffffLL) { set_error("render_points: too many points (%lld)", (long long)P); return DINER_E_INVALID; }
    hipLaunchKernelGGL(points_mlp_kernel, dim3((unsigned)tiles, (unsigned)s.SB), dim3(NWAVES * 64), 0, st, s, mlp_packed, rays, z,
                       NR, K, rgbsigma);
    return check_launch("points_mlp_kernel");
}

}  // namespace diner
