// Training path: forward with saved activations + backward of
//   NeRFRendererDGS.composite   reference src/models/nerf_renderer.py:286-365
//   PixelNeRF.forward           reference src/models/pixelnerf.py:55-145
//   ResnetFC.forward            reference src/models/resnetfc.py:129-159
// as used by DINER.calc_losses (src/models/diner.py:217-290): gradients w.r.t. the fusion-MLP parameters and
// the encoder's latent maps (the sampler is @torch.no_grad in the reference, points/viewdirs carry no grad).
//
// Unlike the fused inference kernels this path is layer by layer: every layer's input must be kept for the
// weight gradients anyway, so activations live in HBM as row-major [rows, 512] fp32 matrices (rows = point x
// view, ~24 GB for a 4096-ray x 40-sample x 4-view step -- 288 GB of HBM make that a non-issue).  GEMMs:
//   gemm_kernel        exact fp32 MFMA (v_mfma_f32_32x32x2_f32), stride-described operands:
//                        C[m][n] (+)= sum_k opA(A[m*sam + k*sak]) * opB(B[k*sbk + n*sbn])  (+ bias[n]) (* [S[m][n] > 0])
//                        forward      Y  = relu?(X) W^T + b (+ Y)          A = X (k contiguous), B = W (k contiguous)
//                        backward dX  dX = (dY W) * [S > 0] (+ dX)         A = dY (k contiguous), B = W (n contiguous)
//                        backward dW  dW += dY^T relu?(X), split over rows A = dY (m contiguous), B = X (n contiguous), atomic C
//   gemm_f16x3_kernel  the same contract in fp32-grade fp16 arithmetic (hi/lo split, 3 MFMAs per product), used for dW
//   gemm_panel_kernel  forward / dX in f16x3 arithmetic with the weight pre-split into fp16 planes
// plus small per-point kernels (inputs/bilinear gather, view mean, head, compositing backward, bilinear
// scatter-add, bias-gradient + max|.| reduction).  Measured rates: DESIGN.md section 4.4.
#include "common.hpp"

namespace diner {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace train {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;  // block tile; LDS tile row stride (floats, 16-byte aligned rows)

struct GemmArgs {
    const float *A, *B, *bias, *S;
    float *C;
    int64_t M;
    int N, K;
    int64_t sam, sak, sbk, sbn;  // element strides of the logical A[m][k], B[k][n]
    int64_t ldc, lds_;           // row strides of C and of the mask S
    int relu_a, relu_b, accumulate, atomic;
    int64_t k_chunk;             // split-K: blockIdx.z handles k in [z*k_chunk, (z+1)*k_chunk)
    // f16x3 kernel only: operands are multiplied by a power of two before the fp16 hi/lo split (C is divided by
    // the product): 2^exp_a / 2^exp_b, or -- when amax_a / amax_b point at a device word holding the bit pattern
    // of max|operand| (diner_train_amax) -- the power of two that maps that maximum into [2^13, 2^14)
    const unsigned int *amax_a, *amax_b;
    int exp_a, exp_b;
};

// One operand tile (128 x 16, as [k][m]) = 512 float4, two per thread.  KC: the operand is contiguous along the
// contraction index (float4 along k, transposed into the tile), else along the tile's long index.
// Loads are unconditional (out-of-range pieces read a clamped in-range address and are zeroed by `ok` when the
// tile is stored): a load under a branch makes hipcc wait for each one separately.
template <bool KC>
__device__ __forceinline__ unsigned tile_load(f32x4 (&v)[2], const float *__restrict__ base, int64_t s_long, int64_t s_k, int64_t l0,
                                              int64_t l_end, int64_t k0, int64_t k_end, int tid)
{
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if (KC) {
            const int64_t l = l0 + (idx >> 2), k = k0 + (idx & 3) * 4;
            const bool in = l < l_end && k < k_end;
            ok |= (unsigned)in << i;
            v[i] = *(const f32x4 *)(base + (l < l_end ? l : l_end - 1) * s_long + (k < k_end ? k : k_end - 4));
        } else {
            const int64_t k = k0 + (idx >> 5), l = l0 + (idx & 31) * 4;
            const bool in = k < k_end && l < l_end;
            ok |= (unsigned)in << i;
            v[i] = *(const f32x4 *)(base + (k < k_end ? k : k_end - 1) * s_k + (l < l_end ? l : l_end - 4));
        }
    }
    return ok;
}
template <bool KC>
__device__ __forceinline__ void tile_store(float (*T)[LDT], const f32x4 (&v)[2], unsigned ok, int relu, int tid)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        f32x4 x = v[i];
        const float lo = relu ? 0.f : -__builtin_inff();
        for (int j = 0; j < 4; ++j) x[j] = ((ok >> i) & 1u) ? (x[j] < lo ? lo : x[j]) : 0.f;  // NaN-keeping floor (fmaxf would launder NaN into 0)
        if (KC) {
            const int l = idx >> 2, kq = (idx & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) T[kq + j][l] = x[j];
        } else {
            const int k = idx >> 5, lq = (idx & 31) * 4;
            *(f32x4 *)&T[k][lq] = x;
        }
    }
}

// Block -> output tile.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2: the
// column blocks of one 128-row tile (they all read the same A tile, the big streamed operand) are given to
// consecutive workgroups of ONE XCD, so A leaves HBM once instead of once per column block.
__device__ __forceinline__ void tile_of(const GemmArgs &g, int64_t &m0, int &n0, int64_t lin = -1)
{
    const int64_t gm = (g.M + BM - 1) / BM;
    if (lin < 0) lin = blockIdx.x;
    const int gn = (g.N + BN - 1) / BN;
    const int64_t full = gm / 8 * 8;
    int64_t mt, nb;
    if (lin < full * gn) { const int64_t j = lin / 8; nb = j % gn; mt = j / gn * 8 + lin % 8; }
    else { const int64_t r = lin - full * gn; mt = full + r / gn; nb = r % gn; }
    m0 = mt * BM;
    n0 = (int)nb * BN;
}

// LDS slot of tile row l for an operand staged by the f16x3 kernel's transposing store (f16g::tile_store<false>): the 4 x 4 index
// transpose inside every 16-row block (an involution), which makes that store conflict-free; the accumulator rows / columns
// come out in slot order and are mapped back here.
__device__ __forceinline__ int slot16(int x) { return (x & ~15) | ((x & 3) << 2) | ((x >> 2) & 3); }

// C layout of the 32x32 MFMA accumulators: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
// PA / PB: the A / B operand tile was staged in slot order (rows / columns of the tile permuted by slot16)
template <bool PA = false, bool PB = false>
__device__ __forceinline__ void epilogue(const GemmArgs &g, const f32x16 (&acc)[2][2], int64_t m0, int n0, int wm, int wn, int lane,
                                         float unscale)
{
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int nc = wn + tb * 32 + (lane & 31), n = n0 + (PB ? slot16(nc) : nc);
        if (n >= g.N) continue;
        const float bias = (g.bias && blockIdx.z == 0) ? g.bias[n] : 0.0f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            // all 16 reads of the tile (old C, mask) are issued before the first dependent store: one memory
            // round trip per tile instead of one per element
            const int mbl = wm + ta * 32 + 4 * (lane >> 5);
            auto row_of = [&](int i) -> int64_t { const int r = mbl + (i & 3) + 8 * (i >> 2); return m0 + (PA ? slot16(r) : r); };
            float old[16], msk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { old[i] = 0.0f; msk[i] = 1.0f; }
            if (g.accumulate && !g.atomic) {  // uniform branches, unconditional loads from clamped rows
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t m = row_of(i);
                    old[i] = g.C[(m < g.M ? m : g.M - 1) * g.ldc + n];
                }
            }
            if (g.S) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t m = row_of(i);
                    msk[i] = g.S[(m < g.M ? m : g.M - 1) * g.lds_ + n];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = row_of(i);
                if (m >= g.M) continue;
                float v = acc[ta][tb][i] * unscale + bias;
                v = msk[i] > 0.0f ? v : 0.0f;
                float *c = g.C + m * g.ldc + n;
                if (g.atomic) atomicAdd(c, v);
                else *c = old[i] + v;
            }
        }
    }
}

// AK: A contiguous along k (sak == 1) else along m (sam == 1).  BNC: B contiguous along n (sbn == 1) else along k.
// 4 waves as 2 x 2, each a 64 x 64 output (2 x 2 MFMA tiles); global loads of tile t+1 are in flight while tile t
// is multiplied out of LDS (double-buffered, one barrier per k-step).
template <bool AK, bool BNC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g)
{
    __shared__ float As[2][BK][LDT], Bs[2][BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t m0;
    int n0;
    tile_of(g, m0, n0);
    const int64_t kbeg = (int64_t)blockIdx.z * g.k_chunk;
    const int64_t kend = kbeg + g.k_chunk < g.K ? kbeg + g.k_chunk : g.K;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    f32x4 ra[2], rb[2];
    unsigned oka = tile_load<AK>(ra, g.A, g.sam, g.sak, m0, g.M, kbeg, kend, tid);
    unsigned okb = tile_load<!BNC>(rb, g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid);
    tile_store<AK>(As[0], ra, oka, g.relu_a, tid);
    tile_store<!BNC>(Bs[0], rb, okb, g.relu_b, tid);
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {
            oka = tile_load<AK>(ra, g.A, g.sam, g.sak, m0, g.M, k0 + BK, kend, tid);
            okb = tile_load<!BNC>(rb, g.B, g.sbn, g.sbk, n0, g.N, k0 + BK, kend, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float a0 = As[buf][kr][wm + c], a1 = As[buf][kr][wm + 32 + c];
            const float b0 = Bs[buf][kr][wn + c], b1 = Bs[buf][kr][wn + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            tile_store<AK>(As[buf ^ 1], ra, oka, g.relu_a, tid);
            tile_store<!BNC>(Bs[buf ^ 1], rb, okb, g.relu_b, tid);
        }
        __syncthreads();
        buf ^= 1;
    }
    epilogue(g, acc, m0, n0, wm, wn, lane, 1.0f);
}

// ---- the same GEMM with fp32-grade fp16 arithmetic (precision "f16x3", the renderer's default) --------------------
// Every operand element is scaled by a power of two, split into fp16 hi + lo while it is staged into LDS, and each
// product is three v_mfma_f32_32x32x16_f16 (lo*hi, hi*lo, hi*hi) with fp32 accumulation -- the arithmetic of
// points_mlp_f16.hip.  128 x 128 x 32 block tiles, 4 waves as 2 x 2 (64 x 64 each), LDS images double-buffered.
// An image holds 16-byte units (u = k/8, row) at u*128 + (row ^ 4u): an MFMA fragment (8 consecutive k of one
// row) is one conflict-free ds_read_b128.
namespace f16g {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int BKH = 32, UNITS_T = (BKH / 8) * 128;

__device__ __forceinline__ int unit(int u, int row) { return u * 128 + (row ^ (4 * u)); }

__device__ __forceinline__ void scale_of(const unsigned int *amax, int static_exp, float &s, float &inv)
{
    int e = static_exp;
    if (amax) {
        const unsigned int b = *amax;
        const int ex = (int)((b >> 23) & 0xffu) - 127;
        e = (b == 0u) ? 0 : 13 - ex;
    }
    e = e < -100 ? -100 : e > 100 ? 100 : e;
    s = __uint_as_float((unsigned int)(127 + e) << 23);
    inv = __uint_as_float((unsigned int)(127 - e) << 23);
}

// One operand tile = 128 (long index l) x 32 (k) fp32 = 1024 float4, four per thread.
// KC (contiguous along k): float4 along k.  else: a 4(k) x 4(l) micro-tile per thread, float4 along l.
template <bool KC>
__device__ __forceinline__ unsigned tile_load(f32x4 (&v)[4], const float *__restrict__ base, int64_t s_long, int64_t s_k, int64_t l0,
                                              int64_t l_end, int64_t k0, int64_t k_end, int tid)
{
    unsigned ok = 0;  // unconditional loads from clamped addresses + a validity bit per piece (see train::tile_load)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KC) {
            const int idx = tid + 256 * i;
            const int64_t l = l0 + (idx >> 3), k = k0 + (idx & 7) * 4;
            ok |= (unsigned)(l < l_end && k < k_end) << i;
            v[i] = *(const f32x4 *)(base + (l < l_end ? l : l_end - 1) * s_long + (k < k_end ? k : k_end - 4));
        } else {
            // thread = (k-quad kq4 of 8, l-quad lq4 of 32); a 16-lane group = 4 k-quads x 4 consecutive l-quads (see tile_store)
            const int kq4 = (tid & 3) | ((tid >> 4) & 4), lq4 = ((tid >> 2) & 15) | ((tid >> 3) & 16);
            const int64_t k = k0 + kq4 * 4 + i, l = l0 + lq4 * 4;
            ok |= (unsigned)(k < k_end && l < l_end) << i;
            v[i] = *(const f32x4 *)(base + (k < k_end ? k : k_end - 1) * s_k + (l < l_end ? l : l_end - 4));
        }
    }
    return ok;
}

// (x * sc, floored) -> fp16 hi / lo pairs: hi = cvt_pk(t), lo = fma_mix(hi * -1 + t) rounded once to fp16 = (f16)(t - (float)hi) (the
// difference is exact in fp32).  2.5 VALU slots per value (4.5 with the relu) where the C++ form costs hipcc about 8: scalar converts
// both ways, v_pack; in the dW kernel that split, not the MFMAs, was the longest phase of a k-step.
// RELU keeps NaN like torch.relu: v_cmp_ngt + v_cndmask, the 4 compares ahead of the 4 selects (gfx950: 2 wait states between a VALU
// write of an SGPR and its VALU read).
template <bool RELU>
__device__ __forceinline__ void split4_pk(float x0, float x1, float x2, float x3, float sc, unsigned &h01, unsigned &h23, unsigned &l01, unsigned &l23)
{
    float t0, t1, t2, t3;
    if constexpr (RELU) {
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_mul_f32 %4, %12, %16\n\tv_mul_f32 %5, %13, %16\n\tv_mul_f32 %6, %14, %16\n\tv_mul_f32 %7, %15, %16\n\t"
                     "v_cmp_ngt_f32_e64 %8, 0, %4\n\tv_cmp_ngt_f32_e64 %9, 0, %5\n\tv_cmp_ngt_f32_e64 %10, 0, %6\n\tv_cmp_ngt_f32_e64 %11, 0, %7\n\t"
                     "v_cndmask_b32_e64 %4, 0, %4, %8\n\tv_cndmask_b32_e64 %5, 0, %5, %9\n\tv_cndmask_b32_e64 %6, 0, %6, %10\n\tv_cndmask_b32_e64 %7, 0, %7, %11\n\t"
                     "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
                     "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(sc));
    } else {
        asm volatile("v_mul_f32 %4, %8, %12\n\tv_mul_f32 %5, %9, %12\n\tv_mul_f32 %6, %10, %12\n\tv_mul_f32 %7, %11, %12\n\t"
                     "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
                     "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(sc));
    }
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void put4(h8 *Thi, h8 *Tlo, int l, int kq, float x0, float x1, float x2, float x3, float floor_, float sc)
{
    u32x2 hi, lo;   // floor_ = 0 applies the relu, -inf does nothing (wave-uniform); NaN stays NaN
    unsigned a, b, c, d;
    if (floor_ == 0.0f) split4_pk<true>(x0, x1, x2, x3, sc, a, b, c, d);
    else split4_pk<false>(x0, x1, x2, x3, sc, a, b, c, d);
    hi.x = a; hi.y = b; lo.x = c; lo.y = d;
    const int o = unit(kq >> 3, l) * 8 + (kq & 4);
    *(u32x2 *)((_Float16 *)Thi + o) = hi;
    *(u32x2 *)((_Float16 *)Tlo + o) = lo;
}

template <bool KC>
__device__ __forceinline__ void tile_store(h8 *Thi, h8 *Tlo, const f32x4 (&v)[4], unsigned ok, float floor_, float sc, int tid)
{
    f32x4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = v[i];
    if (ok != 0xFu) {   // ragged edge of the operand only
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = ((ok >> i) & 1u) ? v[i][j] : 0.0f;
    }
    if (KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            put4(Thi, Tlo, idx >> 3, (idx & 7) * 4, x[i][0], x[i][1], x[i][2], x[i][3], floor_, sc);
        }
    } else {
        // The thread holds 4 (k) x 4 (l); row l + c goes to LDS slot slot16(l + c) = (l & ~15) | 4c | (lq4 & 3).  ds_write_b64 is
        // serviced per 16 contiguous lanes on 32 banks: the group's lanes are 4 k-quads (both halves of 2 units) x 4 l-quads, and
        // with the slots' low bits = the l-quad their 16 half-cells fall on 16 different bank pairs (before: 4-way conflicts,
        // 60 % of the kernel's LDS cycles).  The MFMA rows then come out in slot order: epilogue<PA, PB>.
        const int kq4 = (tid & 3) | ((tid >> 4) & 4), lq4 = ((tid >> 2) & 15) | ((tid >> 3) & 16);
        const int kq = kq4 * 4, sl = ((lq4 * 4) & ~15) | (lq4 & 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) put4(Thi, Tlo, sl + 4 * c, kq, x[0][c], x[1][c], x[2][c], x[3][c], floor_, sc);
    }
}

// The k-loop is a 3-stage pipeline: while tile t is multiplied out of LDS buffer t&1, tile t+1 sits in registers
// (its loads were issued one step earlier and are split/stored into the other LDS buffer after the MFMAs) and the
// loads of tile t+2 are issued.  One step of lead is not enough: the A operand streams from HBM (~2 us) and a
// step is ~0.5 us of MFMA -- with a single stage in flight the kernel ran at 1.1 TB/s, latency-bound.
template <bool AK, bool BNC>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_kernel(GemmArgs g)
{
    __shared__ h8 T[2][4][UNITS_T];  // [buffer][A hi, A lo, B hi, B lo][unit]  (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t m0;
    int n0;
    // Split-K launches (the dW GEMMs: 16 output tiles x 160 row chunks): the 16 workgroups of a chunk read the same 4 + 4 operand
    // slices.  Workgroups go to the 8 XCDs round-robin by launch order, so in launch order each XCD's L2 would see one or two
    // tiles of EVERY chunk and fetch every slice itself (4x the operand bytes from the Infinity Cache / HBM).  Re-deal the
    // (tile, chunk) pairs so that XCD x works through whole chunks x, x + 8, ...: one fetch per slice, 3 of 4 reads hit L2.
    int64_t bx = blockIdx.x, bz = blockIdx.z;
    if (gridDim.z % 8 == 0 && !g.bias) {   // (the epilogue adds the bias in the workgroups with blockIdx.z == 0)
        const int64_t id = (int64_t)blockIdx.z * gridDim.x + blockIdx.x, xcd = id % 8, j = id / 8;
        bz = (j / gridDim.x) * 8 + xcd;
        bx = j % gridDim.x;
    }
    tile_of(g, m0, n0, bx);
    const int64_t kbeg = bz * g.k_chunk;
    const int64_t kend = kbeg + g.k_chunk < g.K ? kbeg + g.k_chunk : g.K;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    float sa, ia, sb, ib;
    scale_of(g.amax_a, g.exp_a, sa, ia);
    scale_of(g.amax_b, g.exp_b, sb, ib);
    const float fa = g.relu_a ? 0.0f : -__builtin_inff(), fb = g.relu_b ? 0.0f : -__builtin_inff();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    f32x4 ra[2][4], rb[2][4];
    const int r = lane & 31, h = lane >> 5;
    const int64_t steps = (kend - kbeg + BKH - 1) / BKH;
    unsigned oka[2], okb[2];
    oka[0] = tile_load<AK>(ra[0], g.A, g.sam, g.sak, m0, g.M, kbeg, kend, tid);
    okb[0] = tile_load<!BNC>(rb[0], g.B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid);
    oka[1] = tile_load<AK>(ra[1], g.A, g.sam, g.sak, m0, g.M, kbeg + BKH, kend, tid);      // all-invalid past kend
    okb[1] = tile_load<!BNC>(rb[1], g.B, g.sbn, g.sbk, n0, g.N, kbeg + BKH, kend, tid);
    tile_store<AK>(T[0][0], T[0][1], ra[0], oka[0], fa, sa, tid);
    tile_store<!BNC>(T[0][2], T[0][3], rb[0], okb[0], fb, sb, tid);
    __syncthreads();
// DINER_DW_INTERLEAVE (default on): the step is ONE basic block -- loads of tile t+2 unconditional (past the end they re-read clamped
// addresses with all pieces invalid), the MFMAs of tile t, then split + store of tile t+1 into the other LDS buffer (past the end: zeros
// into a buffer nobody reads) -- and the scheduler is told to deal the step's 8 global loads and the split's VALU work BETWEEN the 24
// MFMAs (sched_group_barrier, as in the panel kernel): before, the three phases ran one after the other in every wave (in-kernel stamps
// of a 4.3 k-cycle step: 1.0 k issuing the loads, 1.05 k MFMA, 2.0 k waiting for tile t+1 + split + store).
#ifndef DINER_DW_INTERLEAVE
#define DINER_DW_INTERLEAVE 1
#endif
#if DINER_DW_INTERLEAVE
#define DINER_GEMM_STEP(SL)                                                                                      \
    {                                                                                                            \
        const int64_t k2 = kbeg + (t + 2) * BKH;                                                                 \
        oka[SL] = tile_load<AK>(ra[SL], g.A, g.sam, g.sak, m0, g.M, k2, kend, tid);                              \
        okb[SL] = tile_load<!BNC>(rb[SL], g.B, g.sbn, g.sbk, n0, g.N, k2, kend, tid);                            \
        _Pragma("unroll") for (int ks = 0; ks < BKH / 16; ++ks) {                                                \
            const int u = ks * 2 + h;                                                                            \
            h8 ah[2], al[2], bh[2], bl[2];                                                                       \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
                const int oa = unit(u, wm + 32 * q + r), ob = unit(u, wn + 32 * q + r);                          \
                ah[q] = T[SL][0][oa]; al[q] = T[SL][1][oa];                                                      \
                bh[q] = T[SL][2][ob]; bl[q] = T[SL][3][ob];                                                      \
            }                                                                                                    \
            _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                                     \
                _Pragma("unroll") for (int tb = 0; tb < 2; ++tb) {                                               \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bl[tb], acc[ta][tb], 0, 0, 0);  \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                }                                                                                                \
        }                                                                                                        \
        tile_store<AK>(T[1 - SL][0], T[1 - SL][1], ra[1 - SL], oka[1 - SL], fa, sa, tid);                        \
        tile_store<!BNC>(T[1 - SL][2], T[1 - SL][3], rb[1 - SL], okb[1 - SL], fb, sb, tid);                      \
        /* sched_group_barrier masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write, 0x002 VALU */  \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  /* fragments of the first k-step */                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            if (i_ < 16 && (i_ & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     \
            if (i_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                                   \
            if (i_ >= 8 && (i_ & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                     \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __syncthreads();                                                                                         \
    }
#else
#define DINER_GEMM_STEP(SL)                                                                                      \
    {                                                                                                            \
        const int64_t k2 = kbeg + (t + 2) * BKH;                                                                 \
        if (t + 2 < steps) {                                                                                     \
            oka[SL] = tile_load<AK>(ra[SL], g.A, g.sam, g.sak, m0, g.M, k2, kend, tid);                          \
            okb[SL] = tile_load<!BNC>(rb[SL], g.B, g.sbn, g.sbk, n0, g.N, k2, kend, tid);                        \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < BKH / 16; ++ks) {                                                \
            const int u = ks * 2 + h;                                                                            \
            h8 ah[2], al[2], bh[2], bl[2];                                                                       \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
                const int oa = unit(u, wm + 32 * q + r), ob = unit(u, wn + 32 * q + r);                          \
                ah[q] = T[SL][0][oa]; al[q] = T[SL][1][oa];                                                      \
                bh[q] = T[SL][2][ob]; bl[q] = T[SL][3][ob];                                                      \
            }                                                                                                    \
            _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                                     \
                _Pragma("unroll") for (int tb = 0; tb < 2; ++tb) {                                               \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bl[tb], acc[ta][tb], 0, 0, 0);  \
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                }                                                                                                \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (t + 1 < steps) {                                                                                     \
            tile_store<AK>(T[1 - SL][0], T[1 - SL][1], ra[1 - SL], oka[1 - SL], fa, sa, tid);                    \
            tile_store<!BNC>(T[1 - SL][2], T[1 - SL][3], rb[1 - SL], okb[1 - SL], fb, sb, tid);                  \
        }                                                                                                        \
        __syncthreads();                                                                                         \
    }
#endif
    for (int64_t t = 0; t < steps; ++t) {
        DINER_GEMM_STEP(0)
        if (++t >= steps) break;
        DINER_GEMM_STEP(1)
    }
#undef DINER_GEMM_STEP
    epilogue<!AK, BNC>(g, acc, m0, n0, wm, wn, lane, ia * ib);
}

// ---- the weight-gradient GEMM of a 512 x 512 layer: dW[i][j] += sum_m dY[m][i] * relu?(X[m][j])  (reference: autograd of nn.Linear in
//      src/models/resnetfc.py:61-69 as trained by DINER.calc_losses src/models/diner.py:217-290) -- round 3.
// Both operands stream from HBM along the CONTRACTION index (rows m: 655,360 per training step) and are split into fp16 hi / lo in
// the kernel, so the split's VALU work per loaded element is what the generic 128 x 128 kernel above spent its time on (2.0 k of a
// 4.3 k-cycle step).  Here ONE workgroup of 8 waves owns a 256 x 256 output tile (the whole 512 x 512 result is 4 tiles; split-K over
// 64 row chunks fills the 256 CUs): every loaded element feeds twice the MFMAs, a wave owns 128 x 64 (8 accumulator tiles, 128
// registers), the LDS images of a 32-row step are 64 KiB, double-buffered in the CU's 160 KiB.  Staging is the generic kernel's: a
// 4 (k) x 4 (l) micro-tile per thread, transposed in registers, stored in slot16 order (conflict-free ds_write_b64), fragments are
// single conflict-free ds_read_b128; the accumulator rows / columns come out in slot order (undone in the atomic epilogue).
// The 4 tiles of a row chunk run on ONE XCD (they read the same two operand slices: one HBM fetch, three L2 hits).
#ifndef DINER_DW512
#define DINER_DW512 1      // 0: the generic split-K kernel for the weight gradients too (A/B knob)
#endif
namespace dw512 {
using f16g::h8;
using f16g::put4;
using f16g::scale_of;
constexpr int TM = 256, KS = 32, NT = 512;
constexpr int PLANE = (KS / 8) * TM;                   // 16-byte units of one fp16 plane of one operand: 1024 = 16 KiB
constexpr int LDS_BYTES = 2 * 4 * PLANE * 16;          // [2 buffers][A hi, A lo, B hi, B lo] = 128 KiB

struct Args {
    const float *dY, *X;
    float *dW;
    int64_t M, ldy, ldx, ldw;      // contraction length (rows), row strides
    int relu_x;
    const unsigned int *amax_y;    // power-of-two scaling of dY from its measured maximum (or exp_y), X by exp_x: see GemmArgs
    int exp_y, exp_x;
    int64_t k_chunk;               // rows per workgroup (multiple of 32)
    int nchunks;                   // multiple of 8
};

__device__ __forceinline__ int unit(int u, int row) { return u * TM + (row ^ (4 * u)); }

// one operand tile = 32 (k) x 256 (l) fp32 = 2048 float4, four per thread: thread = (k-quad kq4 of 8, l-quad lq4 of 64), float4 along l
__device__ __forceinline__ void thread_of(int tid, int &kq4, int &lq4)
{
    kq4 = (tid & 3) | ((tid >> 4) & 4);
    lq4 = ((tid >> 2) & 15) | ((tid >> 3) & 16) | ((tid >> 3) & 32);
}
// generic form (the chunk's first tile and its ragged end): rows clamped, a validity bit per piece
__device__ __forceinline__ unsigned load_tile(f32x4 (&v)[4], const float *__restrict__ base, int64_t ld, int l0, int64_t k0, int64_t k_end, int tid)
{
    int kq4, lq4;
    thread_of(tid, kq4, lq4);
    unsigned ok = 0;   // unconditional loads from clamped rows (a load under a branch makes hipcc wait per load)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + kq4 * 4 + i;
        ok |= (unsigned)(k < k_end) << i;
        v[i] = *(const f32x4 *)(base + (k < k_end ? k : k_end - 1) * ld + l0 + lq4 * 4);
    }
    return ok;
}
// MASK: pieces whose bit in `ok` is clear are zeros (ragged end only); RELU: the floor at 0 that keeps NaN (split4_pk<true>)
template <bool MASK, bool RELU>
__device__ __forceinline__ void store_tile(h8 *Thi, h8 *Tlo, const f32x4 (&v)[4], unsigned ok, float sc, int tid)
{
    int kq4, lq4;
    thread_of(tid, kq4, lq4);
    f32x4 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = (!MASK || ((ok >> i) & 1u)) ? v[i][j] : 0.0f;
    // row l + c of the tile goes to LDS slot slot16(l + c) = (l & ~15) | 4c | (lq4 & 3): see f16g::tile_store<false>
    const int kq = kq4 * 4, sl = ((lq4 * 4) & ~15) | (lq4 & 3);
    _Float16 *ph = (_Float16 *)Thi + unit(kq >> 3, sl) * 8 + (kq & 4), *pl = (_Float16 *)Tlo + unit(kq >> 3, sl) * 8 + (kq & 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // slots sl + 4c: sl's bits 2, 3 are clear, and the unit's swizzle (row ^ 4u) only touches those: + 4c units... see below
        f16g::u32x2 hi, lo;
        unsigned a, b, cc, d;
        f16g::split4_pk<RELU>(x[0][c], x[1][c], x[2][c], x[3][c], sc, a, b, cc, d);
        hi.x = a; hi.y = b; lo.x = cc; lo.y = d;
        const int o = unit(kq >> 3, sl + 4 * c) * 8 + (kq & 4);
        *(f16g::u32x2 *)((_Float16 *)Thi + o) = hi;
        *(f16g::u32x2 *)((_Float16 *)Tlo + o) = lo;
    }
    (void)ph; (void)pl;
}

template <bool RELU_X>
__global__ __launch_bounds__(512) void dw512_kernel(Args g)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    h8 *T = (h8 *)lds_raw;                                           // T[(buf * 4 + plane) * PLANE + unit]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the 4 output tiles of a chunk on one XCD (workgroups are dealt to the XCDs round-robin by launch order)
    const int lin = blockIdx.x, xcd = lin % 8, jj = lin / 8;
    const int chunk = (jj / 4) * 8 + xcd, tile = jj % 4;
    const int i0 = (tile >> 1) * TM, j0 = (tile & 1) * TM;
    const int64_t kbeg = (int64_t)chunk * g.k_chunk;
    const int64_t kend = kbeg + g.k_chunk < g.M ? kbeg + g.k_chunk : g.M;
    if (kbeg >= kend) return;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;          // the wave's 128 x 64 of the tile
    float sa, ia, sb, ib;
    scale_of(g.amax_y, g.exp_y, sa, ia);
    scale_of(nullptr, g.exp_x, sb, ib);
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    f32x4 ra[4], rb[4];
    const int r = lane & 31, h = lane >> 5;
    const int64_t steps = (kend - kbeg + KS - 1) / KS, full = (kend - kbeg) / KS;   // tiles; tiles without a ragged end
    int kq4, lq4;
    thread_of(tid, kq4, lq4);
    // this thread's 4 x 4 micro-tile of the NEXT tile to load: row kbeg + KS (t + 1) + 4 kq4, columns 4 lq4 .. + 3 of the operand's slice
    const float *pa = g.dY + (kbeg + kq4 * 4) * g.ldy + i0 + lq4 * 4, *pb = g.X + (kbeg + kq4 * 4) * g.ldx + j0 + lq4 * 4;
    const int64_t sta = KS * g.ldy, stb = KS * g.ldx;

    unsigned oka = load_tile(ra, g.dY, g.ldy, i0, kbeg, kend, tid);
    unsigned okb = load_tile(rb, g.X, g.ldx, j0, kbeg, kend, tid);
    store_tile<true, false>(T + 0 * PLANE, T + 1 * PLANE, ra, oka, sa, tid);
    store_tile<true, RELU_X>(T + 2 * PLANE, T + 3 * PLANE, rb, okb, sb, tid);
    __syncthreads();

    // the 48 MFMAs of a tile out of LDS buffer `sl`: the wave's 128 x 64, two 16-deep halves
    auto multiply = [&](int sl) {
        const h8 *Ahi = T + (sl * 4 + 0) * PLANE, *Alo = T + (sl * 4 + 1) * PLANE, *Bhi = T + (sl * 4 + 2) * PLANE, *Blo = T + (sl * 4 + 3) * PLANE;
#pragma unroll
        for (int ks = 0; ks < KS / 16; ++ks) {
            const int u = ks * 2 + h;
            h8 bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { const int ob = unit(u, wn + 32 * q + r); bh[q] = Bhi[ob]; bl[q] = Blo[ob]; }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                const int oa = unit(u, wm + 32 * ta + r);
                const h8 ah = Ahi[oa], al = Alo[oa];
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[tb], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[tb], acc[ta][tb], 0, 0, 0);
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[tb], acc[ta][tb], 0, 0, 0);
                }
            }
        }
    };

    int64_t t = 0;
    // ---- hot loop: steps whose NEXT tile is a full one.  ONE basic block: the 8 loads of tile t + 1 (plain pointer increments, no
    //      clamps, no masks), the 48 MFMAs of tile t out of buffer t & 1, split + store of tile t + 1 into the other buffer; the
    //      scheduler deals fragment reads, the split's VALU work and the stores between the MFMAs (sched_group_barrier)
    for (; t + 1 < full; ++t) {
        const int sl = (int)(t & 1);
        pa += sta; pb += stb;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = *(const f32x4 *)(pa + i * g.ldy); rb[i] = *(const f32x4 *)(pb + i * g.ldx); }
        multiply(sl);
        h8 *N = T + ((1 - sl) * 4) * PLANE;
        store_tile<false, false>(N, N + PLANE, ra, 0xFu, sa, tid);
        store_tile<false, RELU_X>(N + 2 * PLANE, N + 3 * PLANE, rb, 0xFu, sb, tid);
        /* sched_group_barrier masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write, 0x002 VALU */
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);   // the step's 8 global loads first: they are consumed at its end
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // B fragments + the first A pair
#pragma unroll
        for (int i_ = 0; i_ < 48; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i_ % 6 == 0 && i_ < 42) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (i_ >= 12) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            if (i_ >= 12 && (i_ & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // ---- the chunk's last steps (at most two): the next tile is ragged or does not exist
    for (; t < steps; ++t) {
        const int sl = (int)(t & 1);
        const int64_t k1 = kbeg + (t + 1) * KS;
        oka = load_tile(ra, g.dY, g.ldy, i0, k1, kend, tid);
        okb = load_tile(rb, g.X, g.ldx, j0, k1, kend, tid);
        multiply(sl);
        h8 *N = T + ((1 - sl) * 4) * PLANE;
        store_tile<true, false>(N, N + PLANE, ra, oka, sa, tid);
        store_tile<true, RELU_X>(N + 2 * PLANE, N + 3 * PLANE, rb, okb, sb, tid);
        __syncthreads();
    }
    // epilogue: C layout of the 32x32 accumulators: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5); rows and columns are
    // in slot order (both operands were staged by the transposing store)
    const float unscale = ia * ib;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int j = j0 + slot16(wn + tb * 32 + (lane & 31));
#pragma unroll
        for (int ta = 0; ta < 4; ++ta) {
            const int rb0 = wm + ta * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + slot16(rb0 + (i & 3) + 8 * (i >> 2));
                atomicAdd(g.dW + (int64_t)row * g.ldw + j, acc[ta][tb][i] * unscale);
            }
        }
    }
}
}  // namespace dw512

// max |x| over n floats as a bit pattern (non-negative floats order like unsigned integers); *out zeroed by the launcher
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, int64_t n, unsigned int *__restrict__ out)
{
    float m = 0.f;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = ((const f32x4 *)x)[i];
        m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
// One pass over a gradient matrix for both of its reductions: db[n] += sum_m dY[m][n] (bias gradient) and
// *amax = max |dY| (scale of the f16x3 GEMMs that consume dY).  A thread owns one float4 of columns and every
// (256 / (N/4))-th row, so a wave reads whole contiguous rows.  Requires N % 4 == 0 and (N/4) | 256.
__global__ __launch_bounds__(256) void colsum_amax_kernel(const float *__restrict__ dY, int64_t M, int N, int64_t ld, float *__restrict__ db,
                                                          unsigned int *__restrict__ amax)
{
    __shared__ f32x4 part[256];
    const int groups = N >> 2, cg = threadIdx.x % groups, rl = threadIdx.x / groups, lanes = 256 / groups;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    float mx = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * lanes + rl; m < M; m += (int64_t)gridDim.x * lanes) {
        const f32x4 v = *(const f32x4 *)(dY + m * ld + 4 * cg);
        sum += v;
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
    }
    if (amax) {
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(amax, __float_as_uint(mx));
    }
    if (db) {
        part[threadIdx.x] = sum;
        __syncthreads();
        if (rl == 0) {
            for (int i = 1; i < lanes; ++i) sum += part[i * groups + cg];
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(db + 4 * cg + j, sum[j]);
        }
    }
}
}  // namespace f16g

// ---- weight-panel GEMM (f16x3): C[128 rows x 256 cols] per workgroup, B = a 512-column weight matrix given as
// pre-split fp16 hi/lo planes (diner_train_split_panel) ---------------------------------------------------------
// The forward and the dX GEMMs of the training path multiply a huge streamed activation / gradient matrix by a
// 512 x K weight.  In gemm_f16x3_kernel every A tile is split into fp16 hi/lo by four workgroups and every weight
// tile by thousands, and that fp32->fp16x2 conversion (VALU) costs more than the MFMAs (timing-only ablation:
// 1.70 ms with, 0.95 ms without it).  Here the weights arrive already split (no VALU: 16-byte loads become
// 16-byte LDS units), a workgroup owns 256 columns so A is split twice instead of four times, and the two
// workgroups of a row panel sit on one XCD (one HBM read of A).  8 waves as 2 x 4, 64 x 64 outputs each, k-steps
// of 32, LDS double-buffered; A is staged two steps ahead (HBM), the weights one step ahead (L2).
namespace panel {
using f16g::h4;
using f16g::h8;
constexpr int PM = 128, PNW = 256, PN = 512, PK = 32;
constexpr int A_UNITS = (PK / 8) * PM, B_UNITS = (PK / 8) * PNW;  // 16-byte units (u = k/8, row) per image and stage
constexpr int STAGE_UNITS = 2 * A_UNITS + 2 * B_UNITS;            // 48 KiB

struct PanelArgs {
    const float *A;
    int64_t sam;                 // A[m*sam + k], k contiguous
    const _Float16 *Bhi, *Blo;   // panel layout [k/32][512][32], zero-padded to a multiple of 32 in k, scaled by 2^exp_b
    const float *bias, *S, *addend;
    float *C;
    int64_t M;
    int K;
    int64_t ldc, lds_, ldadd;
    int relu_a;
    const unsigned int *amax_a;
    int exp_a, exp_b;
};

__device__ __forceinline__ int unit(int u, int row, int rows) { return u * rows + (row ^ (4 * u)); }

struct StageA {  // 128 x 32 fp32 of A: two float4 per thread
    f32x4 a[2];
    unsigned ok;
};
struct StageB {  // 256 x 32 halfs x {hi, lo} of the weights: two 16-byte units per plane and thread
    h8 bh[2], bl[2];
};

__device__ __forceinline__ void load_a(StageA &st, const PanelArgs &g, int64_t m0, int k0, int tid)
{
    st.ok = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i;
        const int64_t m = m0 + (idx >> 3);
        const int k = k0 + (idx & 7) * 4;
        st.ok |= (unsigned)(m < g.M && k < g.K) << i;
        st.a[i] = *(const f32x4 *)(g.A + (m < g.M ? m : g.M - 1) * g.sam + (k < g.K ? k : g.K - 4));
    }
}
__device__ __forceinline__ void load_b(StageB &st, const PanelArgs &g, int n0, int k0, int tid)
{
    const int64_t off = ((int64_t)(k0 / PK) * PN + n0) * PK;  // this workgroup's 256 rows of the k-step: 16 KiB contiguous
    const h8 *ph = (const h8 *)(g.Bhi + off), *pl = (const h8 *)(g.Blo + off);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        st.bh[i] = ph[tid + 512 * i];
        st.bl[i] = pl[tid + 512 * i];
    }
}
__device__ __forceinline__ void store_a(h8 *T, const StageA &st, float floor_, float sc, int tid)
{
    h8 *Ahi = T, *Alo = T + A_UNITS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i, row = idx >> 3, kq = (idx & 7) * 4;
        f16g::u32x2 hi, lo;
        unsigned a, b, c, d;
        if (floor_ == 0.0f) f16g::split4_pk<true>(st.a[i][0], st.a[i][1], st.a[i][2], st.a[i][3], sc, a, b, c, d);
        else f16g::split4_pk<false>(st.a[i][0], st.a[i][1], st.a[i][2], st.a[i][3], sc, a, b, c, d);
        if (!((st.ok >> i) & 1u)) a = b = c = d = 0u;       // (a piece past the edge was loaded from a clamped, valid address)
        hi.x = a; hi.y = b; lo.x = c; lo.y = d;
        const int o = unit(kq >> 3, row, PM) * 8 + (kq & 4);
        *(f16g::u32x2 *)((_Float16 *)Ahi + o) = hi;
        *(f16g::u32x2 *)((_Float16 *)Alo + o) = lo;
    }
}
__device__ __forceinline__ void store_b(h8 *T, const StageB &st, int tid)
{
    h8 *Bhi = T + 2 * A_UNITS, *Blo = T + 2 * A_UNITS + B_UNITS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 512 * i, n = q >> 2, u = q & 3;  // plane order: [n][32] = four units per row
        Bhi[unit(u, n, PNW)] = st.bh[i];
        Blo[unit(u, n, PNW)] = st.bl[i];
    }
}

__global__ __launch_bounds__(512) void gemm_panel_kernel(PanelArgs g)
{
    __shared__ h8 T[2][STAGE_UNITS];  // 2 x 48 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the two column halves of a row panel are consecutive workgroups of ONE XCD (ids congruent mod 8)
    int64_t m0;
    int n0;
    {
        const int64_t gm = (g.M + PM - 1) / PM, lin = blockIdx.x, full = gm / 8 * 8;
        int64_t mt, nb;
        if (lin < full * 2) { const int64_t j = lin / 8; nb = j & 1; mt = (j >> 1) * 8 + lin % 8; }
        else { const int64_t q = lin - full * 2; mt = full + (q >> 1); nb = q & 1; }
        m0 = mt * PM;
        n0 = (int)nb * PNW;
    }
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
    float sa, ia, sb, ib;
    f16g::scale_of(g.amax_a, g.exp_a, sa, ia);
    f16g::scale_of(nullptr, g.exp_b, sb, ib);
    const float fa = g.relu_a ? 0.0f : -__builtin_inff();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    const int steps = (g.K + PK - 1) / PK;
    const int r = lane & 31, h = lane >> 5;
    StageA sta[2];
    StageB stb;
    load_a(sta[0], g, m0, 0, tid);
    load_b(stb, g, n0, 0, tid);
    load_a(sta[1], g, m0, PK, tid);  // all-invalid (zeros) when there is no second step
    store_a(T[0], sta[0], fa, sa, tid);
    store_b(T[0], stb, tid);
    __syncthreads();
#define DINER_PANEL_STEP(SL)                                                                                     \
    {   /* ONE basic block per step (loads unconditional: past the end they re-read the last tile / load zeros),     \
           so that the scheduler can be told to spread the step's 6 global loads and the split of tile t+1 (VALU)     \
           BETWEEN the MFMAs: a wave issues in order, and after the barrier all 8 waves otherwise queue their loads   \
           at the texture-address unit at once (48 KiB per step = 768 clocks at 64 B/clk) before the first MFMA.      \
           Same-box A/B: 1.370 -> 1.288 ms per forward GEMM */                                                        \
        const int tb_ = t + 1 < steps ? t + 1 : steps - 1, ta_ = t + 2;                                          \
        load_b(stb, g, n0, tb_ * PK, tid);                                                                       \
        load_a(sta[SL], g, m0, ta_ * PK, tid);                                                                   \
        {                                                                                                        \
            const h8 *Ahi = T[SL], *Alo = T[SL] + A_UNITS, *Bhi = T[SL] + 2 * A_UNITS, *Blo = Bhi + B_UNITS;     \
            _Pragma("unroll") for (int ks = 0; ks < PK / 16; ++ks) {                                             \
                const int u = ks * 2 + h;                                                                        \
                h8 ah[2], al[2], bh[2], bl[2];                                                                   \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                  \
                    const int oa = unit(u, wm + 32 * q + r, PM), ob = unit(u, wn + 32 * q + r, PNW);             \
                    ah[q] = Ahi[oa]; al[q] = Alo[oa];                                                            \
                    bh[q] = Bhi[ob]; bl[q] = Blo[ob];                                                            \
                }                                                                                                \
                _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                                 \
                    _Pragma("unroll") for (int tb = 0; tb < 2; ++tb) {                                           \
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bl[tb], acc[ta][tb], 0, 0, 0);  \
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ta], bh[tb], acc[ta][tb], 0, 0, 0);  \
                    }                                                                                            \
            }                                                                                                    \
        }                                                                                                        \
        store_a(T[1 - SL], sta[1 - SL], fa, sa, tid);                                                            \
        /* sched_group_barrier masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x002 VALU */                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  /* fragments of the first k-step */                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
            if (i_ < 12 && (i_ & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     \
            if (i_ < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                   \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        store_b(T[1 - SL], stb, tid);                                                                            \
        __syncthreads();                                                                                         \
    }
    for (int t = 0; t < steps; ++t) {
        DINER_PANEL_STEP(0)
        if (++t >= steps) break;
        DINER_PANEL_STEP(1)
    }
#undef DINER_PANEL_STEP
    const float unscale = ia * ib;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int n = n0 + wn + tb * 32 + (lane & 31);
        const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
            const int64_t mb = m0 + wm + ta * 32 + 4 * h;
            float old[16], msk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { old[i] = 0.0f; msk[i] = 1.0f; }
            if (g.addend) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t m = mb + (i & 3) + 8 * (i >> 2);
                    old[i] = g.addend[(m < g.M ? m : g.M - 1) * g.ldadd + n];
                }
            }
            if (g.S) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t m = mb + (i & 3) + 8 * (i >> 2);
                    msk[i] = g.S[(m < g.M ? m : g.M - 1) * g.lds_ + n];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = mb + (i & 3) + 8 * (i >> 2);
                if (m >= g.M) continue;
                float v = acc[ta][tb][i] * unscale + bias;
                v = msk[i] > 0.0f ? v : 0.0f;
                g.C[m * g.ldc + n] = old[i] + v;
            }
        }
    }
}

// B[n][k] = (transpose ? W[k*ld + n] : W[n*ld + k]) * 2^exp -> fp16 hi / lo planes in panel layout [k/32][512][32]
__global__ void split_panel_kernel(const float *__restrict__ W, int K, int64_t ld, int transpose, int exp_, _Float16 *__restrict__ hi,
                                   _Float16 *__restrict__ lo)
{
    const int kpad = (K + PK - 1) / PK * PK;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)PN * kpad) return;
    const int kk = (int)(i % PK), n = (int)((i / PK) % PN), ks = (int)(i / (PK * PN)), k = ks * PK + kk;
    float sc, inv;
    f16g::scale_of(nullptr, exp_, sc, inv);
    const float v = k < K ? (transpose ? W[(int64_t)k * ld + n] : W[(int64_t)n * ld + k]) * sc : 0.0f;
    const _Float16 hv = (_Float16)v;
    hi[i] = hv;
    lo[i] = (_Float16)(v - (float)hv);
}
}  // namespace panel

int launch_gemm(const GemmArgs &g, int precision, hipStream_t st)
{
    if (g.M == 0 || g.N == 0) return DINER_OK;
    const int64_t kc = g.k_chunk > 0 ? g.k_chunk : g.K;
    GemmArgs a = g;
    a.k_chunk = kc;
    const dim3 grid((unsigned)(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN)), 1, (unsigned)((g.K + kc - 1) / kc));
    const bool ak = g.sak == 1, bnc = g.sbn == 1;
    if (!ak && g.sam != 1) { set_error("gemm: A must be contiguous along m or k"); return DINER_E_INVALID; }
    if (!bnc && g.sbk != 1) { set_error("gemm: B must be contiguous along k or n"); return DINER_E_INVALID; }
    if (precision == DINER_PRECISION_F16X3) {
        if (g.k_chunk > 0 && kc % f16g::BKH) { set_error("gemm: k_chunk must be a multiple of 32 in f16x3 mode"); return DINER_E_INVALID; }
        // the weight gradient of a 512 x 512 layer (dW += dY^T relu?(X), atomic split-K, both operands row-major along the contraction):
        // the dedicated 256 x 256-tile kernel
        if (DINER_DW512 && !ak && bnc && g.atomic && g.M == f16g::dw512::NT && g.N == f16g::dw512::NT && !g.bias && !g.S && !g.relu_a && !g.amax_b &&
            g.K >= 64 * 32 * 4 && g.sak % 4 == 0 && g.sbk % 4 == 0 && g.ldc % 1 == 0) {
            f16g::dw512::Args d{g.A, g.B, g.C, g.K, g.sak, g.sbk, g.ldc, g.relu_b, g.amax_a, g.exp_a, g.exp_b, 0, 64};
            d.k_chunk = ((g.K + d.nchunks - 1) / d.nchunks + 31) / 32 * 32;
            const void *fn = g.relu_b ? (const void *)f16g::dw512::dw512_kernel<true> : (const void *)f16g::dw512::dw512_kernel<false>;
            if (const int rc = ensure_dynamic_lds(fn, f16g::dw512::LDS_BYTES, LDS_SLOT_DW512 + (g.relu_b ? 1 : 0))) return rc;
            void *kargs[] = {(void *)&d};
            if (hipLaunchKernel(fn, dim3(4 * d.nchunks), dim3(512), kargs, f16g::dw512::LDS_BYTES, st) != hipSuccess)
                return check_launch("train::dw512_kernel(launch)");
            return check_launch("train::dw512_kernel");
        }
        if (ak && bnc) hipLaunchKernelGGL((f16g::gemm_f16x3_kernel<true, true>), grid, dim3(256), 0, st, a);
        else if (ak && !bnc) hipLaunchKernelGGL((f16g::gemm_f16x3_kernel<true, false>), grid, dim3(256), 0, st, a);
        else if (!ak && bnc) hipLaunchKernelGGL((f16g::gemm_f16x3_kernel<false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((f16g::gemm_f16x3_kernel<false, false>), grid, dim3(256), 0, st, a);
        return check_launch("train::gemm_f16x3_kernel");
    }
    if (ak && bnc) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, st, a);
    else if (ak && !bnc) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, st, a);
    else if (!ak && bnc) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, st, a);
    return check_launch("train::gemm_kernel");
}

// column sums: db[n] += sum_m dY[m][n]  (one block per 64 columns, rows strided over the block, atomic tail)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ dY, int64_t M, int N, int64_t ld, float *__restrict__ db)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float s = 0.f;
    const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y, beg = blockIdx.y * rows_per, end = beg + rows_per < M ? beg + rows_per : M;
    if (n < N)
        for (int64_t m = beg + r; m < end; m += 4) s += dY[m * ld + n];
    part[r][c] = s;
    __syncthreads();
    if (r == 0 && n < N) atomicAdd(db + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// ---- per-(view, point) MLP inputs: in55 (padded to 56), bilinear latent z, and the footprint ------------------
// rows are view-major: row = v*P + p.  latent is the reference's NCHW tensor [NV,C,h,w] (its gradient has
// that layout too).  taps_out [R,8] = 4 texel indices (y*w+x, as int bits) + 4 weights.
template <bool NHWC>
__global__ __launch_bounds__(64) void point_inputs_kernel(DinerScene s, const float *__restrict__ latent_nchw,
                                                          const float *__restrict__ rays, const float *__restrict__ zsamp,
                                                          int64_t NR, int K, int sb, float *__restrict__ in56,
                                                          float *__restrict__ zlat, float *__restrict__ taps_out)
{
    const int64_t P = NR * (int64_t)K, row = blockIdx.x;
    const int v = (int)(row / P);
    const int64_t p = row - (int64_t)v * P;
    const int lane = threadIdx.x;
    const float *rp = rays + ((int64_t)sb * NR + p / K) * 8;
    const float zz = zsamp[(int64_t)sb * P + p];
    const float dwx = rp[3], dwy = rp[4], dwz = rp[5];
    const float wx = rp[0] + zz * dwx, wy = rp[1] + zz * dwy, wz = rp[2] + zz * dwz;  // nerf_renderer.py:304
    const View vw = load_view(s, sb, v);
    float px, py, pz, u, w, dcx, dcy, dcz;
    project(vw, s.image_w, s.image_h, wx, wy, wz, px, py, pz, u, w);                  // pixelnerf.py:91-93,105-108
    rotate(vw, dwx, dwy, dwz, dcx, dcy, dcz);
    const float4 *tex = (const float4 *)s.maps + ((int64_t)sb * s.NV + v) * s.H * s.W * 2;
    const int ddx = safe_idx(__builtin_rintf(clipf(unnorm(u, (float)s.W / 2.0f), (float)(s.W - 1))), s.W);
    const int ddy = safe_idx(__builtin_rintf(clipf(unnorm(w, (float)s.H / 2.0f), (float)(s.H - 1))), s.H);
    const float delta = tex[((int64_t)ddy * s.W + ddx) * 2].w - pz;
    if (lane < 56) {
        const int e = lane;
        const float half_pi = 1.5707963267948966f;
        float val;
        if (e < 3) val = e == 0 ? px : e == 1 ? py : pz;
        else if (e < 39) { const int j = (e - 3) / 3, i = (e - 3) % 3;
            val = pe_sin(__builtin_fmaf(i == 0 ? px : i == 1 ? py : pz, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
        else if (e < 42) val = e == 39 ? dcx : e == 40 ? dcy : dcz;
        else if (e == 42) val = delta;
        else if (e < 55) { const int j = e - 43;
            val = pe_sin(__builtin_fmaf(delta, s.freq_factor * (float)(1 << (j >> 1)), (j & 1) ? half_pi : 0.0f)); }
        else val = 0.0f;
        in56[row * 56 + e] = val;
    }
    // bilinear / border footprint (image_encoder.py:97-127)
    const float sxl = ((float)s.w - s.feature_padding * 2.0f) / (float)s.w, syl = ((float)s.h - s.feature_padding * 2.0f) / (float)s.h;
    const float ix = clipf(unnorm(u * sxl, (float)s.w / 2.0f), (float)(s.w - 1));
    const float iy = clipf(unnorm(w * syl, (float)s.h / 2.0f), (float)(s.h - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, ex = 1.0f - fx, fy = iy - y0f, ey = 1.0f - fy;
    const int x0 = safe_idx(x0f, s.w), y0 = safe_idx(y0f, s.h);
    const bool x1ok = x0 + 1 <= s.w - 1, y1ok = y0 + 1 <= s.h - 1;
    const int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
    const int o[4] = {y0 * s.w + x0, y0 * s.w + x1, y1 * s.w + x0, y1 * s.w + x1};
    const float wt[4] = {ey * ex, x1ok ? ey * fx : 0.0f, y1ok ? fy * ex : 0.0f, (x1ok && y1ok) ? fy * fx : 0.0f};
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { taps_out[row * 8 + i] = __int_as_float(o[i]); taps_out[row * 8 + 4 + i] = wt[i]; }
    }
    const int64_t plane = (int64_t)s.h * s.w;
    const float *lat = latent_nchw + ((int64_t)sb * s.NV + v) * s.C * plane;
    for (int ch = lane; ch < s.C; ch += 64) {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)  // NHWC (diner_pack_latent): a wave reads 256 contiguous bytes of a texel; NCHW: 64 planes
            t[i] = NHWC ? lat[(int64_t)o[i] * s.C + ch] : lat[ch * plane + o[i]];
        zlat[row * s.C + ch] = __builtin_fmaf(t[3], wt[3], __builtin_fmaf(t[2], wt[2], __builtin_fmaf(t[1], wt[1], t[0] * wt[0])));
    }
}

// dlatent_nhwc[v][texel][ch] += dz[row][ch] * w_tap  (bilinear backward; atomics: several points share a texel).
// The target is NHWC on purpose: a wave's 64 channels of one texel are 256 contiguous bytes, the shape in which
// float atomics run at the full memory-side rate (64 lanes in 64 different rows are ~17x slower: scattering
// straight into the reference's NCHW layout took 82 ms per step, this + the transpose below 5 ms).
// One wave walks SCATTER_RUN consecutive rows (= consecutive samples of a ray in one view, whose 2x2 footprints move by a
// fraction of a texel per sample): contributions are summed in registers while the four texel indices stay the same
// and flushed with one atomic per tap when they change -- float atomics run at a fixed memory-side rate (~1.3 TB/s of
// added bytes), so fewer of them is the only lever.
constexpr int SCATTER_RUN = 16;
__global__ __launch_bounds__(64) void bilinear_scatter_kernel(const float *__restrict__ dz, const float *__restrict__ taps,
                                                              int64_t P, int C, int h, int w, int NV, int sb,
                                                              float *__restrict__ dlatent_nhwc)
{
    const int64_t R = P * NV, row0 = (int64_t)blockIdx.x * SCATTER_RUN;
    const int64_t row1 = row0 + SCATTER_RUN < R ? row0 + SCATTER_RUN : R;
    const int lane = threadIdx.x;
    for (int ch = lane; ch < C; ch += 64) {
        int cur[4] = {-1, -1, -1, -1};
        int64_t cur_v = -1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        bool used[4] = {false, false, false, false};
        for (int64_t row = row0; row < row1; ++row) {
            const int64_t v = row / P;
            int o[4];
            float wt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i] = __float_as_int(taps[row * 8 + i]); wt[i] = taps[row * 8 + 4 + i]; }
            if (v != cur_v || o[0] != cur[0] || o[1] != cur[1] || o[2] != cur[2] || o[3] != cur[3]) {  // wave-uniform
                if (cur_v >= 0) {
                    float *lat = dlatent_nhwc + ((int64_t)sb * NV + cur_v) * h * w * C;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (used[i]) atomicAdd(lat + (int64_t)cur[i] * C + ch, acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) { cur[i] = o[i]; acc[i] = 0.f; used[i] = false; }
                cur_v = v;
            }
            const float g = dz[row * C + ch];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (wt[i] != 0.0f) { acc[i] += g * wt[i]; used[i] = true; }
        }
        if (cur_v >= 0) {
            float *lat = dlatent_nhwc + ((int64_t)sb * NV + cur_v) * h * w * C;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (used[i]) atomicAdd(lat + (int64_t)cur[i] * C + ch, acc[i]);
        }
    }
}

// [N,h*w,C] -> [N,C,h*w] (the layout of encoder.latent and of its gradient), tiled through LDS
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float *__restrict__ in, int64_t hw, int C, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int64_t img = blockIdx.z, p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = (p0 + i < hw && c0 + tx < C) ? in[(img * hw + p0 + i) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && p0 + tx < hw) out[(img * C + c0 + i) * hw + p0 + tx] = tile[tx][i];
}

// mean over views (resnetfc.py:146-149): x [NV,P,C] -> xbar [P,C];  backward: dx[v] = dxbar / NV
__global__ void view_mean_kernel(const float *__restrict__ x, int64_t PC, int NV, float *__restrict__ xbar)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PC) return;
    float s = x[i];
    for (int v = 1; v < NV; ++v) s = s + x[(int64_t)v * PC + i];
    xbar[i] = s / (float)NV;
}
__global__ void view_mean_bwd_kernel(const float *__restrict__ dxbar, int64_t PC, int NV, float *__restrict__ dx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PC) return;
    const float g = dxbar[i] / (float)NV;
    for (int v = 0; v < NV; ++v) dx[(int64_t)v * PC + i] = g;
}

// head (pixelnerf.py:139-143): out [P,4] -> rgbsigma [P,4];  backward: d_out = d_rgbsigma * (sigmoid', relu')
__global__ void head_kernel(const float *__restrict__ out, int64_t n4, float *__restrict__ rgbsigma)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float v = out[i];
    rgbsigma[i] = (i & 3) < 3 ? 1.0f / (1.0f + expf(-v)) : (v < 0.0f ? 0.0f : v);
}
__global__ void head_bwd_kernel(const float *__restrict__ out, const float *__restrict__ rgbsigma, const float *__restrict__ d_rgbsigma,
                                int64_t n4, float *__restrict__ d_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float y = rgbsigma[i], g = d_rgbsigma[i];
    d_out[i] = (i & 3) < 3 ? g * y * (1.0f - y) : (out[i] > 0.0f ? g : 0.0f);
}

// compositing backward (nerf_renderer.py:299-301, 341-360), one ray per thread, sequential over K (K <= a few hundred):
//   alpha_k = 1 - exp(-delta_k relu(sigma_k)), T_k = prod_{i<k}(1 - alpha_i + 1e-10), w_k = alpha_k T_k
//   rgb = sum w c (+ 1 - sum w), depth = sum w z
// given d_rgb [N,3], d_depth [N] (and optionally d_weights [N,K]) -> d_rgbsigma [N,K,4].  z carries no gradient.
__global__ void composite_bwd_kernel(const float *__restrict__ rays, const float *__restrict__ z, const float *__restrict__ rgbsigma,
                                     const float *__restrict__ d_rgb, const float *__restrict__ d_depth,
                                     const float *__restrict__ d_weights, int64_t N, int K, int white_bkgd,
                                     float *__restrict__ d_rgbsigma)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const float far = rays[r * 8 + 7];
    const float *zr = z + r * K, *cr = rgbsigma + r * K * 4;
    float *dr = d_rgbsigma + r * K * 4;
    const float gr = d_rgb[r * 3 + 0], gg = d_rgb[r * 3 + 1], gb = d_rgb[r * 3 + 2], gd = d_depth ? d_depth[r] : 0.0f;
    // dL/dw_k = g.c_k + gd z_k - white*(gr+gg+gb) + d_weights_k
    // w_k = alpha_k T_k;  T_{k+1} = T_k (1 - alpha_k + eps)
    // reverse sweep with S_k = sum_{j>k} dL/dw_j w_j  (dT_j/dalpha_k = -T_j/(1-alpha_k+eps) for j > k)
    // forward sweep: park T_k in the sigma slot of the output (overwritten by the reverse sweep); recomputing
    // T_k backwards by division would lose it once the transmittance underflows behind an opaque sample
    float T = 1.0f;
    for (int k = 0; k < K; ++k) {
        const float delta = (k + 1 < K) ? zr[k + 1] - zr[k] : far - zr[k];
        const float sg = cr[k * 4 + 3] > 0.0f ? cr[k * 4 + 3] : 0.0f;
        dr[k * 4 + 3] = T;
        T = T * (1.0f - (1.0f - expf(-delta * sg)) + 1e-10f);
    }
    float S = 0.0f;  // sum_{j>k} dLdw_j * w_j
    const float gwhite = white_bkgd ? (gr + gg + gb) : 0.0f;
    for (int k = K - 1; k >= 0; --k) {
        const float delta = (k + 1 < K) ? zr[k + 1] - zr[k] : far - zr[k];
        const float sraw = cr[k * 4 + 3], sg = sraw > 0.0f ? sraw : 0.0f;
        const float e = expf(-delta * sg), alpha = 1.0f - e, keep = 1.0f - alpha + 1e-10f;
        const float Tk = dr[k * 4 + 3];
        const float w = alpha * Tk;
        const float dLdw = gr * cr[k * 4 + 0] + gg * cr[k * 4 + 1] + gb * cr[k * 4 + 2] + gd * zr[k] - gwhite + (d_weights ? d_weights[r * K + k] : 0.0f);
        dr[k * 4 + 0] = gr * w; dr[k * 4 + 1] = gg * w; dr[k * 4 + 2] = gb * w;
        const float dLdalpha = dLdw * Tk - S / keep;
        // alpha = 1 - exp(-delta * relu(sigma)) -> d alpha / d sigma = delta * e  (0 where sigma <= 0)
        dr[k * 4 + 3] = sraw > 0.0f ? dLdalpha * delta * e : 0.0f;
        S += dLdw * w;
    }
}

}  // namespace train

using namespace train;

int launch_train_gemm(const float *A, const float *B, const float *bias, const float *S, float *C, int64_t M, int N, int K,
                      int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc, int64_t lds, int relu_a, int relu_b,
                      int accumulate, int atomic, int64_t k_chunk, int precision, const unsigned int *amax_a,
                      const unsigned int *amax_b, int exp_a, int exp_b, hipStream_t st)
{
    GemmArgs g{A, B, bias, S, C, M, N, K, sam, sak, sbk, sbn, ldc, lds, relu_a, relu_b, accumulate, atomic, k_chunk, amax_a, amax_b, exp_a, exp_b};
    return launch_gemm(g, precision, st);
}

int launch_train_split_panel(const float *W, int K, int64_t ld, int transpose, int exp_, void *hi, void *lo, hipStream_t st)
{
    const int64_t n = (int64_t)panel::PN * ((K + panel::PK - 1) / panel::PK * panel::PK);
    hipLaunchKernelGGL(panel::split_panel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, K, ld, transpose, exp_,
                       (_Float16 *)hi, (_Float16 *)lo);
    return check_launch("train::split_panel_kernel");
}

int launch_train_gemm_panel(const float *A, int64_t sam, const void *Bhi, const void *Blo, const float *bias, const float *S,
                            int64_t lds, const float *addend, int64_t ldadd, float *C, int64_t ldc, int64_t M, int K, int relu_a,
                            const unsigned int *amax_a, int exp_a, int exp_b, hipStream_t st)
{
    if (M == 0) return DINER_OK;
    panel::PanelArgs g{A, sam, (const _Float16 *)Bhi, (const _Float16 *)Blo, bias, S, addend, C, M, K, ldc, lds, ldadd, relu_a, amax_a, exp_a, exp_b};
    hipLaunchKernelGGL(panel::gemm_panel_kernel, dim3((unsigned)(2 * ((M + panel::PM - 1) / panel::PM))), dim3(512), 0, st, g);
    return check_launch("train::gemm_panel_kernel");
}

int launch_train_colsum_amax(const float *dY, int64_t M, int N, int64_t ld, float *db, unsigned int *amax, hipStream_t st)
{
    if (amax && hipMemsetAsync(amax, 0, sizeof(unsigned int), st) != hipSuccess) { set_error("train_colsum_amax: memset failed"); return DINER_E_LAUNCH; }
    if (M == 0) return DINER_OK;
    const int lanes = 256 / (N / 4);
    const int64_t blocks = (M + lanes - 1) / lanes;
    hipLaunchKernelGGL(f16g::colsum_amax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, dY, M, N, ld, db, amax);
    return check_launch("train::colsum_amax_kernel");
}

int launch_train_amax(const float *x, int64_t n, unsigned int *out, hipStream_t st)
{
    if (hipMemsetAsync(out, 0, sizeof(unsigned int), st) != hipSuccess) { set_error("train_amax: memset failed"); return DINER_E_LAUNCH; }
    if (n == 0) return DINER_OK;
    const int64_t blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(f16g::amax_kernel, dim3((unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048)), dim3(256), 0, st, x, n, out);
    return check_launch("train::amax_kernel");
}

int launch_train_colsum(const float *dY, int64_t M, int N, int64_t ld, float *db, hipStream_t st)
{
    if (M == 0 || N == 0) return DINER_OK;
    const unsigned chunks = (unsigned)(M / 4096 + 1);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64), chunks > 512 ? 512 : chunks), dim3(256), 0, st, dY, M, N, ld, db);
    return check_launch("train::colsum_kernel");
}

int launch_train_point_inputs(const DinerScene &s, const float *latent, int nhwc, const float *rays, const float *z, int64_t NR, int K,
                              int sb, float *in56, float *zlat, float *taps, hipStream_t st)
{
    const int64_t R = NR * (int64_t)K * s.NV;
    if (R == 0) return DINER_OK;
    if (nhwc) hipLaunchKernelGGL((point_inputs_kernel<true>), dim3((unsigned)R), dim3(64), 0, st, s, latent, rays, z, NR, K, sb, in56, zlat, taps);
    else hipLaunchKernelGGL((point_inputs_kernel<false>), dim3((unsigned)R), dim3(64), 0, st, s, latent, rays, z, NR, K, sb, in56, zlat, taps);
    return check_launch("train::point_inputs_kernel");
}

int launch_train_bilinear_scatter(const float *dz, const float *taps, int64_t P, int C, int h, int w, int NV, int sb,
                                  float *dlatent, hipStream_t st)
{
    if (P * NV == 0) return DINER_OK;
    hipLaunchKernelGGL(bilinear_scatter_kernel, dim3((unsigned)((P * NV + SCATTER_RUN - 1) / SCATTER_RUN)), dim3(64), 0, st, dz, taps, P, C, h, w, NV,
                       sb, dlatent);
    return check_launch("train::bilinear_scatter_kernel");
}

int launch_train_nhwc_to_nchw(const float *in, int64_t N, int C, int h, int w, float *out, hipStream_t st)
{
    const int64_t hw = (int64_t)h * w;
    if (N * hw == 0) return DINER_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((hw + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N), dim3(256), 0, st, in,
                       hw, C, out);
    return check_launch("train::nhwc_to_nchw_kernel");
}

int launch_train_view_mean(const float *x, int64_t PC, int NV, float *xbar, int backward, hipStream_t st)
{
    if (PC == 0) return DINER_OK;
    const dim3 grid((unsigned)((PC + 255) / 256));
    if (backward) hipLaunchKernelGGL(view_mean_bwd_kernel, grid, dim3(256), 0, st, x, PC, NV, xbar);
    else hipLaunchKernelGGL(view_mean_kernel, grid, dim3(256), 0, st, x, PC, NV, xbar);
    return check_launch("train::view_mean_kernel");
}

int launch_train_head(const float *out, const float *rgbsigma, const float *d_rgbsigma, int64_t n4, float *result, int backward,
                      hipStream_t st)
{
    if (n4 == 0) return DINER_OK;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (backward) hipLaunchKernelGGL(head_bwd_kernel, grid, dim3(256), 0, st, out, rgbsigma, d_rgbsigma, n4, result);
    else hipLaunchKernelGGL(head_kernel, grid, dim3(256), 0, st, out, n4, result);
    return check_launch("train::head_kernel");
}

int launch_train_composite_bwd(const float *rays, const float *z, const float *rgbsigma, const float *d_rgb, const float *d_depth,
                               const float *d_weights, int64_t N, int K, int white_bkgd, float *d_rgbsigma, hipStream_t st)
{
    if (N == 0) return DINER_OK;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, rays, z, rgbsigma, d_rgb, d_depth,
                       d_weights, N, K, white_bkgd, d_rgbsigma);
    return check_launch("train::composite_bwd_kernel");
}

}  // namespace diner
