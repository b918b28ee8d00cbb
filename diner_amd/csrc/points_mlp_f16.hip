// Fused per-point evaluation, split-fp16 ("f16x3") variant of points_mlp.hip: same structure, same
// reference functions replaced (PixelNeRF.forward src/models/pixelnerf.py:55-145,
// PositionalEncoding.forward src/models/positional_encoding.py:33-53, SpatialEncoder.index/
// index_depth src/models/image_encoder.py:97-151, ResnetFC.forward src/models/resnetfc.py:129-159),
// but every fp32 GEMM operand is split into two fp16 numbers and the product is evaluated as three
// fp16 MFMAs with fp32 accumulation:
//
//     a = a_hi + a_lo,  w = w_hi + w_lo           (hi = fp16(v), lo = fp16(v - hi): 22+ significant bits)
//     a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi    (dropped a_lo*w_lo <= 2^-22 |a w|)
//
// v_mfma_f32_32x32x16_f16 keeps fp16 subnormal inputs (probed on gfx950: tools/mfma_probe.hip), so
// the representation error of an operand is max(2^-23 |v|, 2^-25): fp32-grade.  The whole hidden
// state is carried scaled by 2^-4 (inputs and biases are pre-scaled, the head multiplies by 16: all
// exact), which moves the fp16 overflow point of an activation to 1.0e6 at no cost in the hot loops.
// An activation beyond the fp16 range becomes inf and the outputs NaN -- loud, never silently wrong.
// Parity with the reference stays inside the 1e-4 bar (tests/test_gpu_parity.py, both precisions).
//
// Why: fp32-input MFMA runs at the vector rate (157 TFLOP/s); fp16 MFMA at ~16x that per
// instruction, so three of them are ~5x faster per product.  What then bounds the kernel is the
// weight stream (1 MiB per layer per 64-point tile from L2) and the activation re-staging between
// layers, see DESIGN.md §4.
#include "common.hpp"

namespace diner {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace f16x3 {

constexpr int TILE_P = 64;
constexpr int NWAVES = 8;
constexpr int CT = 16 / NWAVES;  // 32-column tiles per wave
constexpr int HID = DINER_D_HIDDEN;
constexpr int NKB_FULL = HID / 16;  // 32 k-blocks of 16
constexpr int NKB_IN = 4;           // lin_in: 55 inputs padded to 64
constexpr int UNITS = 64 * TILE_P;  // 16-byte units per A image (hi or lo): [k/8 (64)][row (64)], 64 KiB
constexpr float ACT_SCALE = 0.0625f;   // the hidden state, its inputs and the biases are carried * 2^-4

// ---- packed weight image (halfs) --------------------------------------------------------------
// layer block: [col_tile][kb][part hi=0/lo=1][lane][8]: lane = h*32+c holds
//   W[n = 32*col_tile + c][k = 16*kb + 8*h + j], j = 0..7, split into hi / lo
constexpr int64_t W_FULL = 16LL * NKB_FULL * 2 * 64 * 8;  // halfs of one 512x512 layer (= 512*512*2)
constexpr int64_t W_IN = 16LL * NKB_IN * 2 * 64 * 8;
constexpr int64_t W_OUT = 1LL * NKB_FULL * 2 * 64 * 8;
constexpr int64_t OFF_LIN_IN = 0;
constexpr int64_t OFF_LIN_Z = OFF_LIN_IN + W_IN;
constexpr int64_t OFF_FC0 = OFF_LIN_Z + 3 * W_FULL;
constexpr int64_t OFF_FC1 = OFF_FC0 + 5 * W_FULL;
constexpr int64_t OFF_LIN_OUT = OFF_FC1 + 5 * W_FULL;
constexpr int64_t W_HALFS = OFF_LIN_OUT + W_OUT;
constexpr int64_t BIAS_FLOATS = 14 * 512 + 32;   // fp32 biases appended after the halfs
constexpr int64_t PACKED_FLOATS = W_HALFS / 2 + BIAS_FLOATS;

__global__ void pack_kernel(DinerMlpRaw raw, _Float16 *__restrict__ outw, float *__restrict__ outb)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < BIAS_FLOATS) {
        if (i >= 14 * 512) { const int c = (int)(i - 14 * 512); outb[i] = c < 4 ? raw.lin_out_b[c] * ACT_SCALE : 0.0f; }
        else {
            const int slot = (int)(i / 512), c = (int)(i % 512);
            const float *src = slot == 0 ? raw.lin_in_b : slot < 4 ? raw.lin_z_b[slot - 1] : slot < 9 ? raw.fc0_b[slot - 4] : raw.fc1_b[slot - 9];
            outb[i] = src[c] * ACT_SCALE;
        }
    }
    if (i >= W_HALFS) return;
    const float *w;
    int64_t rel;
    int nkb, in_dim, out_dim;
    if (i < OFF_LIN_Z) { w = raw.lin_in_w; rel = i; nkb = NKB_IN; in_dim = DINER_D_IN; out_dim = HID; }
    else if (i < OFF_FC0) { const int b = (int)((i - OFF_LIN_Z) / W_FULL); w = raw.lin_z_w[b]; rel = (i - OFF_LIN_Z) % W_FULL; nkb = NKB_FULL; in_dim = HID; out_dim = HID; }
    else if (i < OFF_FC1) { const int b = (int)((i - OFF_FC0) / W_FULL); w = raw.fc0_w[b]; rel = (i - OFF_FC0) % W_FULL; nkb = NKB_FULL; in_dim = HID; out_dim = HID; }
    else if (i < OFF_LIN_OUT) { const int b = (int)((i - OFF_FC1) / W_FULL); w = raw.fc1_w[b]; rel = (i - OFF_FC1) % W_FULL; nkb = NKB_FULL; in_dim = HID; out_dim = HID; }
    else { w = raw.lin_out_w; rel = i - OFF_LIN_OUT; nkb = NKB_FULL; in_dim = HID; out_dim = 4; }
    const int j = (int)(rel & 7), lane = (int)((rel >> 3) & 63), part = (int)((rel >> 9) & 1);
    const int64_t blk = rel >> 10;
    const int kb = (int)(blk % nkb), tile = (int)(blk / nkb);
    const int n = tile * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
    const float v = (n < out_dim && k < in_dim) ? w[(int64_t)n * in_dim + k] : 0.0f;
    const _Float16 hi = (_Float16)v;
    outw[i] = part == 0 ? hi : (_Float16)(v - (float)hi);
}

// ---- LDS A images -------------------------------------------------------------------------------
// 16-byte unit (u = k/8, row) of an image sits at u*64 + (row ^ (u & 63)): the XOR makes the
// row-per-lane reads of the MFMA, the k-per-lane writes of the gather and the column-per-lane
// writes of the accumulator store all bank-conflict-free.
__device__ __forceinline__ int unit(int u, int row) { return u * TILE_P + (row ^ (u & 63)); }

__device__ __forceinline__ void split(float s, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
}

// acc[tm][tn] += A[64 x 16*NKB] * W^T for this wave's CT column tiles, three fp16 MFMAs per product.
template <int NKB>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[2][CT], const h8 *Ahi, const h8 *Alo, const h8 *__restrict__ Wl,
                                          int wave, int lane)
{
    const int r = lane & 31, hh = lane >> 5;
    const h8 *bp = Wl + (int64_t)wave * CT * NKB * 2 * 64 + lane;
    h8 b_cur[CT][2], b_nxt[CT][2];
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) { b_cur[tn][0] = bp[(int64_t)tn * NKB * 128]; b_cur[tn][1] = bp[(int64_t)tn * NKB * 128 + 64]; }
#pragma unroll 2
    for (int kb = 0; kb < NKB; ++kb) {
        const int kn = kb + 1 < NKB ? kb + 1 : kb;
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) {
            b_nxt[tn][0] = bp[((int64_t)tn * NKB + kn) * 128];
            b_nxt[tn][1] = bp[((int64_t)tn * NKB + kn) * 128 + 64];
        }
        const int u = kb * 2 + hh;
        const int o0 = unit(u, r), o1 = unit(u, 32 + r);
        const h8 ah0 = Ahi[o0], ah1 = Ahi[o1], al0 = Alo[o0], al1 = Alo[o1];
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) {  // small terms first
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, b_cur[tn][0], acc[0][tn], 0, 0, 0);
            acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, b_cur[tn][0], acc[1][tn], 0, 0, 0);
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, b_cur[tn][1], acc[0][tn], 0, 0, 0);
            acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, b_cur[tn][1], acc[1][tn], 0, 0, 0);
            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, b_cur[tn][0], acc[0][tn], 0, 0, 0);
            acc[1][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, b_cur[tn][0], acc[1][tn], 0, 0, 0);
        }
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) { b_cur[tn][0] = b_nxt[tn][0]; b_cur[tn][1] = b_nxt[tn][1]; }
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

// relu(acc) -> split -> LDS A images: this wave's 64 columns become k = 64w .. 64w+63 of the next layer.
// Lane c holds column k = base + c; neighbouring lanes hold neighbouring k, so an even/odd lane pair
// swaps one value per register pair (DPP quad_perm, no LDS) and each lane writes one packed (k, k+1)
// dword per image.
__device__ __forceinline__ float swap_xor1(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
}
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[2][CT], _Float16 *Ahi, _Float16 *Alo, int wave, int lane)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int c = lane & 31, h = lane >> 5;
    const bool odd = lane & 1;
#pragma unroll
    for (int tn = 0; tn < CT; ++tn) {
        const int k = wave * (32 * CT) + tn * 32 + c, u = k >> 3, j0 = (k & 7) & ~1;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                // even lane writes (k, k+1) of register i's row; odd lane writes (k-1, k) of register i+1's row
                float a0 = acc[tm][tn][i], a1 = acc[tm][tn][i + 1];
                asm volatile("" : "+v"(a0), "+v"(a1));  // keep the two extracts static (no dynamic vector index)
                a0 = a0 > 0.0f ? a0 : 0.0f;
                a1 = a1 > 0.0f ? a1 : 0.0f;
                const float keep = odd ? a1 : a0, give = odd ? a0 : a1;
                const float got = swap_xor1(give);
                const float v0 = odd ? got : keep, v1 = odd ? keep : got;           // values at k even, k odd
                const int row = tm * 32 + 8 * (i >> 2) + 4 * h + (i & 3) + (odd ? 1 : 0);  // C/D layout of the 32x32 MFMA
                _Float16 h0, l0, h1, l1;
                split(v0, h0, l0);
                split(v1, h1, l1);
                const int o = unit(u, row) * 8 + j0;
                *(h2 *)(Ahi + o) = h2{h0, h1};
                *(h2 *)(Alo + o) = h2{l0, l1};
            }
    }
}

struct Tap {
    int o00, o01, o10, o11;  // float4 offsets of the 4 texels (clamped, always readable)
    float nw, ne, sw, se;    // weights; a tap outside the map has its weight forced to 0
};

__global__ __launch_bounds__(NWAVES * 64) void points_mlp_f16_kernel(DinerScene s, const float *__restrict__ Wp,
                                                                     const float *__restrict__ rays,
                                                                     const float *__restrict__ zsamp, int64_t NR, int K,
                                                                     float *__restrict__ rgbsigma)
{
    __shared__ h8 lds[2 * UNITS + TILE_P * 2];  // A_hi | A_lo | one Tap per row (all LDS in ONE array)
    h8 *Ahi8 = lds, *Alo8 = lds + UNITS;
    _Float16 *Ahi = (_Float16 *)Ahi8, *Alo = (_Float16 *)Alo8;
    Tap *taps = (Tap *)(lds + 2 * UNITS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sb = blockIdx.y;
    const int64_t P = NR * (int64_t)K;
    int64_t tile;
    {   // XCD-aware tile order (bijective for any grid size)
        const int64_t nwg = gridDim.x, b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    const _Float16 *Wh = (const _Float16 *)Wp;
    const float *bias = Wp + W_HALFS / 2;

    const int row = tid & 63;
    int64_t p = tile * TILE_P + row;
    if (p > P - 1) p = P - 1;
    const int64_t ray = p / K;
    const float *rp = rays + ((int64_t)sb * NR + ray) * 8;
    const float zz = zsamp[(int64_t)sb * P + p];
    const float dwx = rp[3], dwy = rp[4], dwz = rp[5];
    const float wx = rp[0] + zz * dwx, wy = rp[1] + zz * dwy, wz = rp[2] + zz * dwz;  // nerf_renderer.py:304

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
        // ---- geometry + positional encodings -> A[:, 0:64] (55 real inputs); footprint -> taps -----
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
            for (int e = wave * 8; e < wave * 8 + 8; ++e) {                      // input layout :128
                float val;
                if (e < 3) val = e == 0 ? px : e == 1 ? py : pz;
                else if (e < 39) { const int j = (e - 3) / 3, i = (e - 3) % 3;    // positional_encoding.py:45-49
                    val = sinf(__builtin_fmaf(i == 0 ? px : i == 1 ? py : pz, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
                else if (e < 42) val = e == 39 ? dcx : e == 40 ? dcy : dcz;
                else if (e == 42) val = delta;
                else if (e < 55) { const int j = e - 43;
                    val = sinf(__builtin_fmaf(delta, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
                else val = 0.0f;
                _Float16 hi, lo;
                split(val * ACT_SCALE, hi, lo);
                const int o = unit(e >> 3, row) * 8 + (e & 7);
                Ahi[o] = hi;
                Alo[o] = lo;
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
        acc_set_bias(x, bias, wave, lane);
        gemm_tile<NKB_IN>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_LIN_IN), wave, lane);   // resnetfc.py:139
        __syncthreads();

        const f32x4 *lat = (const f32x4 *)s.latent + ((int64_t)sb * s.NV + v) * s.h * s.w * (HID / 4);
        for (int b = 0; b < DINER_COMBINE_LAYER; ++b) {
            // ---- z = bilinear latent of the 64 points -> A images (each wave gathers 8 rows);
            //      lane l takes channels 8l..8l+7 = one 16-byte unit of each image
#pragma unroll 2
            for (int rr = 0; rr < TILE_P / NWAVES; ++rr) {
                const int r = wave * (TILE_P / NWAVES) + rr;
                const Tap t = taps[r];
                h8 vh, vl;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int q = 2 * lane + half;
                    const f32x4 a = lat[t.o00 + q], bb = lat[t.o01 + q], c = lat[t.o10 + q], d = lat[t.o11 + q];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {  // ATen's accumulation order nw,ne,sw,se with contracted FMAs
                        const float o = __builtin_fmaf(d[i], t.se, __builtin_fmaf(c[i], t.sw, __builtin_fmaf(bb[i], t.ne, a[i] * t.nw)));
                        _Float16 hi, lo;
                        split(o * ACT_SCALE, hi, lo);
                        vh[half * 4 + i] = hi;
                        vl[half * 4 + i] = lo;
                    }
                }
                const int o = unit(lane, r);
                Ahi8[o] = vh;
                Alo8[o] = vl;
            }
            __syncthreads();
            acc_add_bias(x, bias + 512 * (1 + b), wave, lane);                          // :152-153 x = x + lin_z(z)
            gemm_tile<NKB_FULL>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_LIN_Z + b * W_FULL), wave, lane);
            __syncthreads();
            store_relu(x, Ahi, Alo, wave, lane);                                        // :62 fc_0(relu(x))
            __syncthreads();
            acc_set_bias(net, bias + 512 * (4 + b), wave, lane);
            gemm_tile<NKB_FULL>(net, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC0 + b * W_FULL), wave, lane);
            __syncthreads();
            store_relu(net, Ahi, Alo, wave, lane);                                      // :63 fc_1(relu(net))
            __syncthreads();
            acc_add_bias(x, bias + 512 * (9 + b), wave, lane);                          // :69 x + dx
            gemm_tile<NKB_FULL>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC1 + b * W_FULL), wave, lane);
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
        store_relu(xsum, Ahi, Alo, wave, lane);
        __syncthreads();
        acc_set_bias(net, bias + 512 * (4 + b), wave, lane);
        gemm_tile<NKB_FULL>(net, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC0 + b * W_FULL), wave, lane);
        __syncthreads();
        store_relu(net, Ahi, Alo, wave, lane);
        __syncthreads();
        acc_add_bias(xsum, bias + 512 * (9 + b), wave, lane);
        gemm_tile<NKB_FULL>(xsum, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC1 + b * W_FULL), wave, lane);
        __syncthreads();
    }
    store_relu(xsum, Ahi, Alo, wave, lane);                                             // :158 lin_out(relu(x))
    __syncthreads();
    if (wave < 2) {  // lin_out: one 32-column tile (4 real outputs), wave w = rows 32w..32w+31
        f32x16 o;
        const float bo = bias[14 * 512 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = bo;
        const int r = lane & 31, hh = lane >> 5;
        const h8 *bp = (const h8 *)(Wh + OFF_LIN_OUT) + lane;
#pragma unroll 4
        for (int kb = 0; kb < NKB_FULL; ++kb) {
            const int oa = unit(kb * 2 + hh, wave * 32 + r);
            const h8 ah = Ahi8[oa], al = Alo8[oa], bh = bp[kb * 128], bl = bp[kb * 128 + 64];
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, o, 0, 0, 0);
        }
        const int c = lane & 31, h = lane >> 5;
        if (c < 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = wave * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                const int64_t pp = tile * TILE_P + rr;
                if (pp < P) {
                    const float val = o[i] * (1.0f / ACT_SCALE);                       // pixelnerf.py:139-143
                    rgbsigma[((int64_t)sb * P + pp) * 4 + c] = c < 3 ? 1.0f / (1.0f + expf(-val)) : (val > 0.0f ? val : 0.0f);
                }
            }
        }
    }
}

}  // namespace f16x3

int64_t mlp_f16_packed_floats() { return f16x3::PACKED_FLOATS; }

int launch_pack_mlp_f16(const DinerMlpRaw &raw, float *out, hipStream_t st)
{
    using namespace f16x3;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((W_HALFS + 255) / 256)), dim3(256), 0, st, raw, (_Float16 *)out,
                       out + W_HALFS / 2);
    return check_launch("f16x3::pack_kernel");
}

int launch_points_mlp_f16(const DinerScene &s, const float *mlp_packed, const float *rays, const float *z, int64_t NR,
                          int K, float *rgbsigma, hipStream_t st)
{
    using namespace f16x3;
    const int64_t P = NR * (int64_t)K;
    if (P == 0 || s.SB == 0) return DINER_OK;
    if (s.C != DINER_D_LATENT) { set_error("render_points: latent channels C=%d unsupported (need %d)", s.C, DINER_D_LATENT); return DINER_E_UNSUPPORTED; }
    if (s.num_freqs != 6) { set_error("render_points: num_freqs=%d unsupported (need 6)", s.num_freqs); return DINER_E_UNSUPPORTED; }
    const int64_t tiles = (P + TILE_P - 1) / TILE_P;
    if (tiles > 0x7fffffffLL) { set_error("render_points: too many points (%lld)", (long long)P); return DINER_E_INVALID; }
    hipLaunchKernelGGL(points_mlp_f16_kernel, dim3((unsigned)tiles, (unsigned)s.SB), dim3(NWAVES * 64), 0, st, s, mlp_packed,
                       rays, z, NR, K, rgbsigma);
    return check_launch("points_mlp_f16_kernel");
}

}  // namespace diner
