// Training path: the 512 x 512 forward / dX GEMMs on the assembly GEMM core of the inference kernel.
//
//   C[m][n] = addend[m][n] + (sum_k opA(A[m][k]) * B[n][k] + bias[n]) * [S[m][n] > 0],   n, k < 512
//
// (reference: the Linear layers of ResnetFC, src/models/resnetfc.py:62-69,139-158, and their autograd transposes.)
// Same arithmetic as diner_train_gemm_panel (f16x3: fp16 hi/lo split of both operands, three MFMAs per product, fp32
// accumulation), different machine mapping -- the one of points_mlp_f16.hip:
//   * a workgroup (8 waves, one per CU, persistent) owns a tile of 64 rows x all 512 columns; wave w owns columns 64w..64w+63
//     (its accumulators: the core's x grid) and, in the S phase, the k-slice 64w..64w+63 of the operand image
//   * A is read ONCE per tile (the panel kernel reads and splits it twice), split into the fp16 hi/lo operand image in LDS
//     (128 KiB); the weights arrive pre-split in the core's stream layout (diner_train_pack_core) and go L2 -> registers
//     through the core's ring, never through LDS
//   * no workgroup barrier in the loop: the core's arrival counters (f16_core.inc, flow mode) let waves 0-3 and 4-7 drift half
//     a layer apart, so one group's loads / split / epilogue run under the other group's MFMAs
// Diagnostics and parity: tests/test_training.py (against float64 and against the exact-fp32 GEMM).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.hpp"

namespace diner {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace train {
namespace core {

#include "f16_core.inc"

constexpr int NWAVES = 8, TILE_M = 64, HID = 512, NKB = HID / 16;
constexpr int64_t W_HALFS = 16LL * NKB * 2 * 64 * 8;   // one 512 x 512 matrix, hi + lo
constexpr int IMG_BYTES = 128 * 1024;
constexpr int LDS_CTR = IMG_BYTES;                      // the core's arrival counters
constexpr int LDS_BYTES = LDS_CTR + 64;
constexpr int X = F16_X, NET = F16_NET;
#ifndef DINER_DEPHASE
#define DINER_DEPHASE 4
#endif
constexpr int DEPHASE = DINER_DEPHASE, DEPHASE_CYCLES = 65024;   // ~ one tile period (8 x 8128 cycles of s_sleep 127)

struct Args {
    const float *A;
    int64_t sam;
    const _Float16 *W;           // diner_train_pack_core
    const float *bias, *S, *addend;
    float *C;
    int64_t M, ldc, lds_, ldadd;
    int relu_a;
    const unsigned int *amax_a;
    int exp_a, exp_b;
    float *colsum;               // += column sums of C (NULL: skipped)
    unsigned int *amax_out;      // max= bits of max |C| (NULL: skipped)
};

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

template <int N> __device__ __forceinline__ float acc_read()
{
    float v;
    asm volatile("v_mov_b32 %0, v%c1" : "=v"(v) : "n"(N));
    return v;
}
template <int N> __device__ __forceinline__ void acc_zero() { asm volatile("v_mov_b32 v%c0, 0" ::"n"(N)); }
template <int I, int N, class F> __device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// B[n][k] = (transpose ? W[k*ld + n] : W[n*ld + k]) * 2^exp in the core's stream layout (points_mlp_f16.hip, "packed weight
// image"): [col_tile][kb][part hi/lo][lane][8], lane = h*32 + c holds B[32*col_tile + c][16*kb + 8*h + j]
__global__ void pack_core_kernel(const float *__restrict__ W, int64_t ld, int transpose, int exp_, _Float16 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W_HALFS) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), part = (int)((i >> 9) & 1);
    const int64_t blk = i >> 10;
    const int kb = (int)(blk % NKB), tile = (int)(blk / NKB);
    const int n = tile * 32 + (lane & 31), k = kb * 16 + 8 * (lane >> 5) + j;
    float sc, inv;
    scale_of(nullptr, exp_, sc, inv);
    const float v = (transpose ? W[(int64_t)k * ld + n] : W[(int64_t)n * ld + k]) * sc;
    const _Float16 hi = (_Float16)v;
    out[i] = part == 0 ? hi : (_Float16)(v - (float)hi);
}

// This lane's 16 float4 of A for tile `tile` -> the core's `net` grid (v[NET + 4*(8*tp + j) ..]: row 32*tp + c, k = 64w + 8j + 4h ..+3).
// Issued before a layer block: the block's first counted vmcnt wait covers them (loads return in order), so their latency is
// spent where the partner wave of the SIMD can use the MFMA pipe, and the next S phase finds the data in registers.
// core registers R..R+3 (x sc, relu'd if RELU: NaN kept) -> fp16 hi / lo pairs; see split4_pk in train.hip
template <int R, bool RELU> __device__ __forceinline__ void split4_reg(float sc, unsigned &h01, unsigned &h23, unsigned &l01, unsigned &l23)
{
    float t0, t1, t2, t3;
    if constexpr (RELU) {
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_mul_f32 %4, v%c12, %16\n\tv_mul_f32 %5, v%c13, %16\n\tv_mul_f32 %6, v%c14, %16\n\tv_mul_f32 %7, v%c15, %16\n\t"
                     "v_cmp_ngt_f32_e64 %8, 0, %4\n\tv_cmp_ngt_f32_e64 %9, 0, %5\n\tv_cmp_ngt_f32_e64 %10, 0, %6\n\tv_cmp_ngt_f32_e64 %11, 0, %7\n\t"
                     "v_cndmask_b32_e64 %4, 0, %4, %8\n\tv_cndmask_b32_e64 %5, 0, %5, %9\n\tv_cndmask_b32_e64 %6, 0, %6, %10\n\tv_cndmask_b32_e64 %7, 0, %7, %11\n\t"
                     "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
                     "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
                     : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3), "v"(sc));
    } else {
        asm volatile("v_mul_f32 %4, v%c8, %12\n\tv_mul_f32 %5, v%c9, %12\n\tv_mul_f32 %6, v%c10, %12\n\tv_mul_f32 %7, v%c11, %12\n\t"
                     "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
                     "v_fma_mixlo_f16 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %3, %1, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mixhi_f16 %2, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %3, %1, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3), "v"(sc));
    }
}

template <int REG, int OFF> __device__ __forceinline__ void load4_into(const float *ap)
{
    asm volatile("global_load_dwordx4 v[%c1:%c2], %0, off offset:%c3" ::"v"(ap), "n"(REG), "n"(REG + 3), "n"(OFF) : "memory");
}
template <int TP> __device__ __forceinline__ void prefetch_rows(const float *ap)
{
    load4_into<NET + 4 * (8 * TP + 0), 0>(ap);
    load4_into<NET + 4 * (8 * TP + 1), 32>(ap);
    load4_into<NET + 4 * (8 * TP + 2), 64>(ap);
    load4_into<NET + 4 * (8 * TP + 3), 96>(ap);
    load4_into<NET + 4 * (8 * TP + 4), 128>(ap);
    load4_into<NET + 4 * (8 * TP + 5), 160>(ap);
    load4_into<NET + 4 * (8 * TP + 6), 192>(ap);
    load4_into<NET + 4 * (8 * TP + 7), 224>(ap);
}
__device__ __forceinline__ void prefetch_a(const Args &g, int64_t tile, int wave, int c, int h)
{
    const int64_t m0 = tile * TILE_M + c, m1 = m0 + 32;
    prefetch_rows<0>(g.A + (m0 < g.M ? m0 : g.M - 1) * g.sam + wave * 64 + 4 * h);
    prefetch_rows<1>(g.A + (m1 < g.M ? m1 : g.M - 1) * g.sam + wave * 64 + 4 * h);
}

template <bool HAS_S, bool HAS_ADD>
__global__ __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_num_vgpr(F16_VGPR_CAP / 2))) void gemm_core_kernel(Args g)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *img = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    // this wave's two weight streams (column tiles 2*wave, 2*wave + 1); opaque at every use (see points_mlp_f16.hip)
    const uint64_t wbase = (uint64_t)(uintptr_t)g.W + (uint64_t)wave * (2 * NKB * 2048);
    auto wst = [&](int tn) -> uint64_t {
        uint64_t b = wbase;
        asm volatile("" : "+s"(b));
        return b + (uint64_t)(tn * (NKB * 2048));
    };
    const unsigned loff = lane * 16;
    const unsigned ab0 = (unsigned)(uintptr_t)img + h * 2048 + c * 16;
    Sync sy;
    sy.ctr = (unsigned)(uintptr_t)(lds + LDS_CTR);
    sy.ctrh = sy.ctr + 4 * (wave >> 2);
    sy.one = 1;
    sy.lay = 0;
    sy.half = wave >> 2;
    if (tid < 8) ((__attribute__((address_space(3))) volatile unsigned *)(lds + LDS_CTR))[tid] = 0;   // (typed: a generic volatile access becomes a FLAT instruction)
    ring_prologue(wst(0), wst(1), loff);
    if ((int64_t)blockIdx.x < (g.M + TILE_M - 1) / TILE_M) prefetch_a(g, blockIdx.x, wave, c, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (once: the first tile's A; later tiles are covered by the layer block's waits)
    __syncthreads();
    float sa, ia, sb, ib;
    scale_of(g.amax_a, g.exp_a, sa, ia);
    scale_of(nullptr, g.exp_b, sb, ib);
    const float unscale = ia * ib;
    const int64_t tiles = (g.M + TILE_M - 1) / TILE_M;
    float amax = 0.0f;
    float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                   // this lane's column sums (columns 64w + 4(lane & 15) ..+3, rows = lane>>4 mod 4)
    const f32x4 bias4 = g.bias ? *(const f32x4 *)(g.bias + wave * 64 + 4 * (lane & 15)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // De-phase the workgroups: every tile is a memory phase (epilogue + operand load: 256 KiB per CU) followed by a GEMM phase, and
    // workgroups that start together stay in lockstep -- HBM saturated during the memory phases and idle during the GEMMs.
    // A start delay of (blockIdx % DEPHASE) / DEPHASE of a tile period spreads the memory phases over the period.
    for (int d = 0; d < (int)(blockIdx.x % DEPHASE) * (DEPHASE_CYCLES / DEPHASE / 8128); ++d) __builtin_amdgcn_s_sleep(127);
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m0 = tile * TILE_M;
#ifdef DINER_CORE_TRACE
        const bool tr_ = blockIdx.x == 0 && tile == 3 * (int64_t)gridDim.x;
        unsigned long long t0_ = __builtin_amdgcn_s_memtime(), t1_ = 0, t2_ = 0, t3_ = 0;
#endif
        // ---- S: rows m0..m0+63, k = 64*wave .. +63 of A -> this wave's 8 unit-rows of the operand image.  Lane (c, h) holds
        //      the float4 k = 64w + 8j + 4h of rows c and 32 + c (a lane pair = 32 contiguous bytes of a row) in the core's idle
        //      `net` grid -- prefetched one tile ahead, see below -- and writes the half cell (4 halfs) of both planes: the layout
        //      store_relu() of the inference kernel writes
        sfor<0, 16>([&](auto TJ) {
            constexpr int tp = TJ.value / 8, j = TJ.value % 8, r0 = NET + 4 * TJ.value;
            u32x2 hi, lo;
            unsigned a, b, cc, d;
            if (g.relu_a) split4_reg<r0, true>(sa, a, b, cc, d);
            else split4_reg<r0, false>(sa, a, b, cc, d);
            hi.x = a; hi.y = b; lo.x = cc; lo.y = d;
            char *p = img + (wave * 8 + j) * 2048 + (tp * 32 + c) * 16 + 8 * h;
            *(u32x2 *)p = hi;
            *(u32x2 *)(p + 1024) = lo;
        });
        prefetch_a(g, tile + gridDim.x < tiles ? tile + gridDim.x : tile, wave, c, h);   // the next tile's A lands during this tile's GEMM
#ifdef DINER_CORE_TRACE
        t1_ = __builtin_amdgcn_s_memtime();
#endif
        sfor<0, 64>([&](auto I) { acc_zero<X + I>(); });
        layer_x_full(wst(0), wst(1), wst(0), wst(1), loff, ab0, (++sy.lay, sy));   // the ring runs on into the next tile (same weights)

#ifdef DINER_CORE_TRACE
        t2_ = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue.  The accumulators (register I of tile (tn, tp) = column 64w + 32tn + 8(I>>2) + 4h + (I&3), row 32tp + c) go
        //      through this wave's own 8 unit-rows of the image (16 KiB; free: the block returns once every wave has finished reading
        //      them) into the row-major layout -- 16 lanes = the 256 contiguous bytes of this wave's columns of one row -- so that
        //      mask, addend and result move in whole cache lines.  float4 q of row r sits at r*256 + ((q ^ (r & 15)) << 4): conflict-free
        //      both ways.
        {
            char *stage = img + wave * 16384;
            sfor<0, 16>([&](auto TG) {
                constexpr int t = TG.value / 4, gq = TG.value % 4, tn = t / 2, tp = t % 2, r0 = X + 16 * t + 4 * gq;
                const f32x4 v = {acc_read<r0>(), acc_read<r0 + 1>(), acc_read<r0 + 2>(), acc_read<r0 + 3>()};
                const int row = 32 * tp + c, q4 = 8 * tn + 2 * gq + h;
                *(f32x4 *)(stage + row * 256 + ((q4 ^ (row & 15)) << 4)) = v;
            });
            const int rq = lane >> 4, q4 = lane & 15, col = wave * 64 + 4 * q4;
            // the core's poison word (a flow wait that gave up: never in a correct build): the tile's C becomes NaN, amax_out inf -- a
            // silently wrong gradient is the one thing this kernel must not produce
            const bool poisoned = F16_FLOW && *(__attribute__((address_space(3))) volatile unsigned *)(lds + LDS_CTR + F16_POISON_OFF) != 0;
            if (poisoned) amax = __builtin_inff();
#pragma unroll 1
            for (int i0 = 0; i0 < 16; i0 += 4) {
                f32x4 msk[4], old[4];
                int64_t mrow[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = m0 + 4 * (i0 + i) + rq;
                    mrow[i] = m;
                    const int64_t mc = m < g.M ? m : g.M - 1;
                    if (HAS_S) msk[i] = *(const f32x4 *)(g.S + mc * g.lds_ + col);
                    if (HAS_ADD) old[i] = *(const f32x4 *)(g.addend + mc * g.ldadd + col);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * (i0 + i) + rq;
                    f32x4 v = *(const f32x4 *)(stage + row * 256 + ((q4 ^ (row & 15)) << 4));
                    const bool ok = mrow[i] < g.M;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = v[q] * unscale + bias4[q];
                        if (HAS_S) t = msk[i][q] > 0.0f ? t : 0.0f;
                        if (HAS_ADD) t = old[i][q] + t;
                        if (poisoned) t = __builtin_nanf("");
                        v[q] = t;
                        if (ok) {
                            cs[q] += t;
                            amax = __builtin_fmaxf(amax, __builtin_fabsf(t));
                        }
                    }
                    if (ok) *(f32x4 *)(g.C + mrow[i] * g.ldc + col) = v;
                }
            }
        }
#ifdef DINER_CORE_TRACE
        t3_ = __builtin_amdgcn_s_memtime();
        if (tr_ && lane == 0) printf("[core trace] wave %d: start %llu  S %llu  block %llu  epilogue %llu\n", wave, t0_ % 100000000ull, t1_ - t0_, t2_ - t1_, t3_ - t2_);
#endif
    }
    if (g.colsum) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = cs[q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) atomicAdd(g.colsum + wave * 64 + 4 * lane + q, v);
        }
    }
    if (g.amax_out) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, d, 64));
        if (lane == 0) atomicMax(g.amax_out, __float_as_uint(amax));
    }
}

}  // namespace core
}  // namespace train

int launch_train_pack_core(const float *W, int64_t ld, int transpose, int exp_, void *out, hipStream_t st)
{
    hipLaunchKernelGGL(train::core::pack_core_kernel, dim3((unsigned)((train::core::W_HALFS + 255) / 256)), dim3(256), 0, st, W, ld, transpose, exp_,
                       (_Float16 *)out);
    return check_launch("train::pack_core_kernel");
}

int launch_train_gemm_core(const float *A, int64_t sam, const void *Wcore, const float *bias, const float *S, int64_t lds,
                           const float *addend, int64_t ldadd, float *C, int64_t ldc, int64_t M, int relu_a, const unsigned int *amax_a,
                           int exp_a, int exp_b, float *colsum, unsigned int *amax_out, hipStream_t st)
{
    if (M == 0) return DINER_OK;
    const void *fn = S ? (addend ? (const void *)train::core::gemm_core_kernel<true, true> : (const void *)train::core::gemm_core_kernel<true, false>)
                       : (addend ? (const void *)train::core::gemm_core_kernel<false, true> : (const void *)train::core::gemm_core_kernel<false, false>);
    const int which = (S ? 2 : 0) + (addend ? 1 : 0);
    if (const int rc = ensure_dynamic_lds(fn, train::core::LDS_BYTES, LDS_SLOT_TRAIN_CORE0 + which)) return rc;   // per device (common.hpp)
    const int cus = device_cus();
    const int64_t tiles = (M + train::core::TILE_M - 1) / train::core::TILE_M;
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
    train::core::Args g{A, sam, (const _Float16 *)Wcore, bias, S, addend, C, M, ldc, lds, ldadd, relu_a, amax_a, exp_a, exp_b, colsum, amax_out};
    void *kargs[] = {(void *)&g};
    if (hipLaunchKernel(fn, dim3(grid), dim3(train::core::NWAVES * 64), kargs, train::core::LDS_BYTES, st) != hipSuccess)
        return check_launch("train::gemm_core_kernel(launch)");
    return check_launch("train::gemm_core_kernel");
}

}  // namespace diner
