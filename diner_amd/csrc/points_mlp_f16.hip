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
// An activation beyond the fp16 range becomes inf, the next product NaN, and every ReLU here keeps NaN (v < 0 ? 0 : v,
// like torch.relu), so the sample's rgb-sigma is NaN and the compositing kernel raises DINER_STATUS_NONFINITE.
// Parity with the reference stays inside the 1e-4 bar (tests/test_gpu_parity.py, both precisions).
//
// Why: fp32-input MFMA runs at the vector rate (157 TFLOP/s); fp16 MFMA at ~16x that per
// instruction, so three of them are ~5x faster per product.  What then bounds the kernel is the
// weight stream (1 MiB per layer per 64-point tile from L2) and the activation re-staging between
// layers, see DESIGN.md §4.
//
// Differences from the fp32 kernel, all driven by that:
//   * the GEMMs are evaluated TRANSPOSED (C^T[feature][point] = W * act^T: the weight fragment is the
//     MFMA A operand, the activation fragment the B operand).  A lane of the 32x32 accumulator then
//     holds 4 consecutive features of one point in 4 consecutive registers -- exactly 4 consecutive k
//     of the next layer -- so re-staging an activation tile is convert + one 8-byte LDS store per
//     image per 4 values, with no cross-lane traffic;
//   * weights stream through a 2-deep register ring written with inline-asm loads and counted waits
//     (hipcc sinks ordinary prefetch loads to their first use); measured with in-kernel stamps the GEMM
//     phases already run at the chip's sustained fp16-MFMA rate on random data (~1.5 PFLOP/s, clock-
//     limited), so a deeper ring buys nothing;
//   * the per-view hidden states wait for the mean over views in a per-workgroup global scratch
//     (write-only until the last view), not in 64 more registers: with 192 accumulator registers the
//     compiler spilled inside the activation-store and gather code;
//   * persistent workgroups (one per CU): all biases are staged in LDS once, and in each round the
//     workgroups of one XCD take a contiguous run of tiles, so neighbouring texels share one L2.
#include <stdio.h>
#include <stdlib.h>

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
constexpr int BIAS_FLOATS = 14 * 512 + 32;       // fp32 biases appended after the halfs
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

// ---- weight stream: inline-asm loads + counted waits ------------------------------------------------
// hipcc sinks ordinary prefetch loads down to their first use (and then waits vmcnt(0) every k-block),
// so the register ring is written with asm loads and asm waits.  All VMEM traffic inside the GEMM loop
// is these loads, so the counts below are exact; any compiler-issued load around the loop can only make
// a wait stricter, never too weak (vmcnt retires in issue order).
// The address is a per-lane VGPR pair on purpose: with an SGPR base the compiler may reload a spilled
// SGPR (v_readlane = VALU write) right in front of the asm, and the 5 wait states gfx9 requires between
// a VALU-written SGPR and a VMEM instruction reading it are not inserted for inline asm (seen as a
// memory fault on the box).  A VGPR address has no such software-managed hazard.
template <int OFF>
__device__ __forceinline__ void wload(h8 &dst, const char *ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wwait(h8 (&w)[CT][2])
{
    static_assert(CT == 2, "operand list");
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
}

// acc[tn][tp] += W[64 features of this wave x 16*NKB] * act^T[16*NKB x 64 points]: three fp16 MFMAs
// per product (W_lo*a_hi, W_hi*a_lo, W_hi*a_hi: small terms first), fp32 accumulate.
// Weight fragments stream from global memory through a 2-deep register ring (one k-block = 4 loads in
// flight behind the one being consumed); activation fragments come from the LDS images.
template <int NKB>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[CT][2], const h8 *Ahi, const h8 *Alo, const h8 *__restrict__ Wl,
                                          int wave, int lane)
{
    static_assert(NKB % 2 == 0 && CT == 2, "ring depth / tile count");
    const int r = lane & 31, hh = lane >> 5;
    // per column tile: this lane's pointer into the packed layer block, advanced one k-block (2 KiB) per step
    const char *wp0 = (const char *)(Wl + (int64_t)(wave * CT + 0) * NKB * 128 + lane);
    const char *wp1 = (const char *)(Wl + (int64_t)(wave * CT + 1) * NKB * 128 + lane);
    h8 w[2][CT][2];
    wload<0>(w[0][0][0], wp0); wload<1024>(w[0][0][1], wp0); wload<0>(w[0][1][0], wp1); wload<1024>(w[0][1][1], wp1);
#define DINER_F16_STEP(ST, LOADS, WAITN)                                                                        \
    {                                                                                                           \
        const int kb = kb0 + ST;                                                                                \
        if (LOADS) { /* k-block kb+1 -> the other ring slot */                                                  \
            wload<2048>(w[(ST + 1) & 1][0][0], wp0); wload<3072>(w[(ST + 1) & 1][0][1], wp0);                   \
            wload<2048>(w[(ST + 1) & 1][1][0], wp1); wload<3072>(w[(ST + 1) & 1][1][1], wp1);                   \
            wp0 += 2048; wp1 += 2048;                                                                           \
        }                                                                                                       \
        const int u = kb * 2 + hh, o0 = unit(u, r), o1 = unit(u, 32 + r);                                       \
        const h8 ah0 = Ahi[o0], ah1 = Ahi[o1], al0 = Alo[o0], al1 = Alo[o1];                                    \
        wwait<WAITN>(w[ST]);                                                                                    \
        _Pragma("unroll") for (int tn = 0; tn < CT; ++tn) {                                                     \
            acc[tn][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][1], ah0, acc[tn][0], 0, 0, 0);         \
            acc[tn][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][1], ah1, acc[tn][1], 0, 0, 0);         \
        }                                                                                                       \
        _Pragma("unroll") for (int tn = 0; tn < CT; ++tn) {                                                     \
            acc[tn][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][0], al0, acc[tn][0], 0, 0, 0);         \
            acc[tn][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][0], al1, acc[tn][1], 0, 0, 0);         \
        }                                                                                                       \
        _Pragma("unroll") for (int tn = 0; tn < CT; ++tn) {                                                     \
            acc[tn][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][0], ah0, acc[tn][0], 0, 0, 0);         \
            acc[tn][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ST][tn][0], ah1, acc[tn][1], 0, 0, 0);         \
        }                                                                                                       \
    }
    int kb0 = 0;
    for (; kb0 < NKB - 2; kb0 += 2) {
        DINER_F16_STEP(0, true, 4) DINER_F16_STEP(1, true, 4)
    }
    DINER_F16_STEP(0, true, 4) DINER_F16_STEP(1, false, 0)  // last pair: the ring drains
#undef DINER_F16_STEP
}

// Accumulator layout (transposed product): lane (c = l&31, h = l>>5), register i of acc[tn][tp] holds
// feature n = 64*wave + 32*tn + 8*(i>>2) + 4*h + (i&3) of point 32*tp + c.
template <bool ADD>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[CT][2], const float *bias, int wave, int lane)
{
    const int h = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < CT; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *(const f32x4 *)(bias + wave * (32 * CT) + tn * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ADD) acc[tn][tp][4 * g + j] += b[j];
                    else acc[tn][tp][4 * g + j] = b[j];
                }
        }
}

// relu(acc) -> split -> LDS images: registers 4g..4g+3 of a lane are features k..k+3 (k = 8u + 4h) of
// one point: half of the 16-byte unit (u, point) of each image, one 8-byte store each.
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[CT][2], _Float16 *Ahi, _Float16 *Alo, int wave, int lane)
{
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < CT; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u = wave * (4 * CT) + tn * 4 + g;
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                h4 vh, vl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[tn][tp][4 * g + j];
                    _Float16 hi, lo;
                    split(v < 0.0f ? 0.0f : v, hi, lo);  // relu that keeps NaN (torch.relu does): a blown-up activation stays visible
                    vh[j] = hi;
                    vl[j] = lo;
                }
                const int o = unit(u, tp * 32 + c) * 8 + 4 * h;
                *(h4 *)(Ahi + o) = vh;
                *(h4 *)(Alo + o) = vl;
            }
        }
}

constexpr int SLAB_FLOATS = TILE_P * HID;  // one view's x of one tile (128 KiB)
// rounds between two pacing points (power of two).  Kernel time in ms, same binary otherwise -- box A: 1 -> 1416, 2 -> 1405,
// 4 -> 1404; box B: 1 -> 1421, 4 -> 1412, 8 -> 1414, 16 -> 1423, 64 -> 1549.  Re-syncing every tile makes every round as slow
// as its slowest workgroup; hardly ever re-syncing loses the shared L2 hits on the weight stream.
constexpr int PACE_EVERY = 4;
constexpr int SYNC_WORDS = 256;            // head of the scratch: 8 pacing counters, one per 128-byte line

struct Tap {
    int o00, o01, o10, o11;  // float4 offsets of the 4 texels (clamped, always readable)
    float nw, ne, sw, se;    // weights; a tap outside the map has its weight forced to 0
};

// STAMP = diagnostic build (env DINER_F16_STAMP=1): per-phase s_memtime totals of workgroup 0 / wave 0
// go to `dbg`; never used by the product path.
enum { PH_GEOM, PH_GEMM, PH_GATHER, PH_STORE, PH_BIAS, PH_BARRIER, PH_VIEWSUM, PH_HEAD, PH_COUNT };
// LINZ: lin_z[b](z) comes from the pre-multiplied feature maps s.linz_maps (diner_pack_linz_maps) as a
// bilinear gather-add into the accumulators instead of a gather + GEMM per block.
template <bool STAMP, bool LINZ>
__global__ __launch_bounds__(NWAVES * 64) void points_mlp_f16_kernel(DinerScene s, const float *__restrict__ Wp,
                                                                     const float *__restrict__ rays,
                                                                     const float *__restrict__ zsamp, int64_t NR, int K,
                                                                     int64_t tiles, float *__restrict__ scratch,
                                                                     float *__restrict__ rgbsigma,
                                                                     unsigned long long *__restrict__ dbg)
{
    unsigned long long t_prev = 0, t_acc[PH_COUNT] = {0};
    if (STAMP) t_prev = __builtin_amdgcn_s_memtime();
#define PHASE(ID)                                                        \
    if (STAMP) {                                                         \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                              \
        t_acc[ID] += t_now - t_prev;                                     \
        t_prev = t_now;                                                  \
        __builtin_amdgcn_sched_barrier(0);                               \
    }
#define BARRIER()        \
    PHASE(cur_phase)     \
    __syncthreads();     \
    PHASE(PH_BARRIER)
    int cur_phase = PH_GEOM;
    (void)cur_phase;
    __shared__ h8 lds[2 * UNITS + TILE_P * 2 + BIAS_FLOATS / 4 + 1];  // A_hi | A_lo | one Tap per row | biases | pacing flag (ONE array)
    h8 *Ahi8 = lds, *Alo8 = lds + UNITS;
    _Float16 *Ahi = (_Float16 *)Ahi8, *Alo = (_Float16 *)Alo8;
    Tap *taps = (Tap *)(lds + 2 * UNITS);
    float *bias = (float *)(lds + 2 * UNITS + TILE_P * 2);  // all 14 bias vectors + lin_out's, staged once

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lets weight bases live in SGPRs
    const int sb = blockIdx.y;
    const int64_t P = NR * (int64_t)K;
    const _Float16 *Wh = (const _Float16 *)Wp;
    for (int i = threadIdx.x; i < BIAS_FLOATS; i += NWAVES * 64) bias[i] = (Wp + W_HALFS / 2)[i];
    // this lane's slice of the workgroup's scratch: [view][wave][tn][tp][i][lane] (coalesced 256-B rows)
    float *xs = scratch + SYNC_WORDS + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (int64_t)(s.NV - 1) * SLAB_FLOATS + wave * (CT * 2 * 16 * 64) + lane;
    const float sxl = ((float)s.w - s.feature_padding * 2.0f) / (float)s.w;  // image_encoder.py:113-114
    const float syl = ((float)s.h - s.feature_padding * 2.0f) / (float)s.h;
    const int row = tid & 63;

    // persistent workgroup: in round i the grid covers tiles [i*G, (i+1)*G); inside a round the
    // workgroups that share an XCD (equal blockIdx % 8) take a contiguous run of tiles, so neighbouring
    // samples/rays -- the same latent texels -- are gathered through one L2 (bijective for any G)
    int64_t slot;
    {
        const int64_t nwg = gridDim.x, b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    }
    // Pacing (speed only, no data is exchanged): the workgroups that share an XCD (equal blockIdx % 8) start
    // every PACE_EVERY-th tile together.  They all stream the same 1 MiB weight layers in the same order; kept in step, one
    // fetch from the Infinity Cache serves the whole XCD through its L2, while drifting apart they each
    // miss (measured: L2 hit rate 56 % unpaced).  A monotonic counter per group, bounded polling: a
    // workgroup that times out stops pacing for the rest of the launch and simply goes on, so residency or
    // placement can never turn this into a hang (or, on a shared GPU, into a repeated stall).
    unsigned int *pace = (unsigned int *)scratch + (blockIdx.x & 7) * 32;
    const unsigned int group = (gridDim.x + 7 - (blockIdx.x & 7)) / 8 * gridDim.y;  // workgroups with this residue
    const int64_t full_rounds = tiles / gridDim.x;                                   // rounds in which every workgroup has a tile
    int64_t round = 0;
    bool pace_on = true;
    volatile int *pace_flag = (volatile int *)(lds + 2 * UNITS + TILE_P * 2 + BIAS_FLOATS / 4);  // one LDS word
    for (int64_t tile = slot; tile < tiles; tile += gridDim.x, ++round) {
        if (pace_on && round < full_rounds && group > 1 && (round & (PACE_EVERY - 1)) == 0) {
            if (tid == 0) {
                __hip_atomic_fetch_add(pace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int want = (unsigned int)(round / PACE_EVERY + 1) * group;
                int spin = 0;
                for (; spin < 4000; ++spin) {
                    if (__hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                pace_flag[0] = spin < 4000;  // timed out once (peers not co-resident?): stop pacing, keep computing
            }
            __syncthreads();
            pace_on = pace_flag[0] != 0;
        }
        int64_t p = tile * TILE_P + row;
        if (p > P - 1) p = P - 1;  // tail tile: duplicate the last point, masked at the store
        const int64_t ray = p / K;
        const float *rp = rays + ((int64_t)sb * NR + ray) * 8;
        const float zz = zsamp[(int64_t)sb * P + p];
        const float dwx = rp[3], dwy = rp[4], dwz = rp[5];
        const float wx = rp[0] + zz * dwx, wy = rp[1] + zz * dwy, wz = rp[2] + zz * dwz;  // nerf_renderer.py:304

        f32x16 x[CT][2], net[CT][2];
        for (int v = 0; v < s.NV; ++v) {
            // ---- geometry + positional encodings -> images[:, 0:64] (55 real inputs); footprint -> taps ----
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
            cur_phase = PH_GEOM; BARRIER()
            acc_bias<false>(x, bias, wave, lane);
            PHASE(PH_BIAS)
            gemm_tile<NKB_IN>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_LIN_IN), wave, lane);   // resnetfc.py:139
            cur_phase = PH_GEMM; BARRIER()

            const f32x4 *lat = (const f32x4 *)s.latent + ((int64_t)sb * s.NV + v) * s.h * s.w * (HID / 4);
            for (int b = 0; b < DINER_COMBINE_LAYER; ++b) {
                if (LINZ) {
                    // ---- x += lin_z[b](z) = bilerp(G_b)(uv) (resnetfc.py:152-153): gather-add in the accumulator
                    //      layout -- a lane owns 4 consecutive features (one float4 per tap) of its two points
                    const f32x4 *G = (const f32x4 *)s.linz_maps + (((int64_t)b * s.SB + sb) * s.NV + v) * s.h * s.w * (HID / 4);
                    const int c = lane & 31, h = lane >> 5;
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        const Tap t = taps[32 * tp + c];
                        const float nw = t.nw * ACT_SCALE, ne = t.ne * ACT_SCALE, sw = t.sw * ACT_SCALE, se = t.se * ACT_SCALE;
#pragma unroll
                        for (int tn = 0; tn < CT; ++tn) {
                            f32x4 ta[4], tb[4], tc[4], td[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int q = wave * (8 * CT) + tn * 8 + 2 * g + h;
                                ta[g] = G[t.o00 + q]; tb[g] = G[t.o01 + q]; tc[g] = G[t.o10 + q]; td[g] = G[t.o11 + q];
                            }
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    x[tn][tp][4 * g + j] += __builtin_fmaf(td[g][j], se, __builtin_fmaf(tc[g][j], sw, __builtin_fmaf(tb[g][j], ne, ta[g][j] * nw)));
                        }
                    }
                    PHASE(PH_GATHER)
                } else {
                // ---- z = bilinear latent of the 64 points -> images (each wave gathers 8 rows);
                    //      lane l takes channels 8l..8l+7 = one 16-byte unit of each image
                    constexpr int GB = 2;  // rows per batch: GB*8 16-byte loads per lane in flight
                    for (int r0 = 0; r0 < TILE_P / NWAVES; r0 += GB) {
                        f32x4 tex[GB][2][4];
                        Tap t[GB];
    #pragma unroll
                        for (int rr = 0; rr < GB; ++rr) {
                            t[rr] = taps[wave * (TILE_P / NWAVES) + r0 + rr];
    #pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const int q = 2 * lane + half;
                                tex[rr][half][0] = lat[t[rr].o00 + q]; tex[rr][half][1] = lat[t[rr].o01 + q];
                                tex[rr][half][2] = lat[t[rr].o10 + q]; tex[rr][half][3] = lat[t[rr].o11 + q];
                            }
                        }
    #pragma unroll
                        for (int rr = 0; rr < GB; ++rr) {
                            h8 vh, vl;
    #pragma unroll
                            for (int half = 0; half < 2; ++half)
    #pragma unroll
                                for (int i = 0; i < 4; ++i) {  // ATen's accumulation order nw,ne,sw,se with contracted FMAs
                                    const float o = __builtin_fmaf(tex[rr][half][3][i], t[rr].se, __builtin_fmaf(tex[rr][half][2][i], t[rr].sw,
                                                    __builtin_fmaf(tex[rr][half][1][i], t[rr].ne, tex[rr][half][0][i] * t[rr].nw)));
                                    _Float16 hi, lo;
                                    split(o * ACT_SCALE, hi, lo);
                                    vh[half * 4 + i] = hi;
                                    vl[half * 4 + i] = lo;
                                }
                            const int o = unit(lane, wave * (TILE_P / NWAVES) + r0 + rr);
                            Ahi8[o] = vh;
                            Alo8[o] = vl;
                        }
                    }
                    cur_phase = PH_GATHER; BARRIER()
                    acc_bias<true>(x, bias + 512 * (1 + b), wave, lane);                         // :152-153 x = x + lin_z(z)
                    PHASE(PH_BIAS)
                    gemm_tile<NKB_FULL>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_LIN_Z + b * W_FULL), wave, lane);
                    cur_phase = PH_GEMM; BARRIER()
                }
                store_relu(x, Ahi, Alo, wave, lane);                                        // :62 fc_0(relu(x))
                cur_phase = PH_STORE; BARRIER()
                acc_bias<false>(net, bias + 512 * (4 + b), wave, lane);
                PHASE(PH_BIAS)
                gemm_tile<NKB_FULL>(net, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC0 + b * W_FULL), wave, lane);
                cur_phase = PH_GEMM; BARRIER()
                store_relu(net, Ahi, Alo, wave, lane);                                      // :63 fc_1(relu(net))
                cur_phase = PH_STORE; BARRIER()
                acc_bias<true>(x, bias + 512 * (9 + b), wave, lane);                         // :69 x + dx
                PHASE(PH_BIAS)
                gemm_tile<NKB_FULL>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC1 + b * W_FULL), wave, lane);
                cur_phase = PH_GEMM; BARRIER()
            }
            // ---- mean over views (resnetfc.py:146-149, combine()): views 0..NV-2 park their x in this
            //      workgroup's scratch slabs (fire-and-forget stores, nothing waits on them); the last view
            //      sums the slabs in the reference's order ((x0 + x1) + x2) + ... and divides
            if (s.NV > 1) {
                if (v < s.NV - 1) {
#pragma unroll
                    for (int tn = 0; tn < CT; ++tn)
#pragma unroll
                        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                            for (int i = 0; i < 16; ++i) xs[(int64_t)v * SLAB_FLOATS + ((tn * 2 + tp) * 16 + i) * 64] = x[tn][tp][i];
                } else {
                    const float nv = (float)s.NV;
#pragma unroll
                    for (int tn = 0; tn < CT; ++tn)
#pragma unroll
                        for (int tp = 0; tp < 2; ++tp) {
                            f32x16 sum;
#pragma unroll
                            for (int i = 0; i < 16; ++i) sum[i] = xs[((tn * 2 + tp) * 16 + i) * 64];
                            for (int u = 1; u < s.NV - 1; ++u) {
                                f32x16 t;
#pragma unroll
                                for (int i = 0; i < 16; ++i) t[i] = xs[(int64_t)u * SLAB_FLOATS + ((tn * 2 + tp) * 16 + i) * 64];
                                sum += t;
                            }
                            x[tn][tp] = (sum + x[tn][tp]) / nv;
                        }
                }
            }
            PHASE(PH_VIEWSUM)
        }
        for (int b = DINER_COMBINE_LAYER; b < DINER_N_BLOCKS; ++b) {
            store_relu(x, Ahi, Alo, wave, lane);
            cur_phase = PH_STORE; BARRIER()
            acc_bias<false>(net, bias + 512 * (4 + b), wave, lane);
            PHASE(PH_BIAS)
            gemm_tile<NKB_FULL>(net, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC0 + b * W_FULL), wave, lane);
            cur_phase = PH_GEMM; BARRIER()
            store_relu(net, Ahi, Alo, wave, lane);
            cur_phase = PH_STORE; BARRIER()
            acc_bias<true>(x, bias + 512 * (9 + b), wave, lane);
            PHASE(PH_BIAS)
            gemm_tile<NKB_FULL>(x, Ahi8, Alo8, (const h8 *)(Wh + OFF_FC1 + b * W_FULL), wave, lane);
            cur_phase = PH_GEMM; BARRIER()
        }
        store_relu(x, Ahi, Alo, wave, lane);                                                // :158 lin_out(relu(x))
        cur_phase = PH_STORE; BARRIER()
        if (wave < 2) {  // lin_out: one 32-feature tile (4 real outputs), wave w = points 32w..32w+31
            f32x16 o;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = 0.0f;
            const int r = lane & 31, hh = lane >> 5;
            const h8 *bp = (const h8 *)(Wh + OFF_LIN_OUT) + lane;
#pragma unroll 4
            for (int kb = 0; kb < NKB_FULL; ++kb) {
                const int oa = unit(kb * 2 + hh, wave * 32 + r);
                const h8 ah = Ahi8[oa], al = Alo8[oa], bh = bp[kb * 128], bl = bp[kb * 128 + 64];
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, o, 0, 0, 0);
            }
            const int64_t pp = tile * TILE_P + wave * 32 + r;
            if (hh == 0 && pp < P) {  // registers 0..3 of the h=0 half are outputs 0..3 of point r
                f32x4 out;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float val = (o[j] + bias[14 * 512 + j]) * (1.0f / ACT_SCALE);       // pixelnerf.py:139-143
                    out[j] = j < 3 ? 1.0f / (1.0f + expf(-val)) : (val < 0.0f ? 0.0f : val);
                }
                *(f32x4 *)(rgbsigma + ((int64_t)sb * P + pp) * 4) = out;
            }
        }
        cur_phase = PH_HEAD; BARRIER()  // the images are rewritten by the next tile's geometry phase
    }
    if (STAMP && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0)
        for (int i = 0; i < PH_COUNT; ++i) dbg[wave * PH_COUNT + i] = t_acc[i];
#undef PHASE
#undef BARRIER
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

// persistent grid: one workgroup per CU (the 130-KiB LDS image admits exactly one)
static int f16_grid_limit()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

int64_t points_mlp_f16_scratch_floats(int64_t SB, int NV)
{
    return f16x3::SYNC_WORDS + (NV > 1 ? (int64_t)f16_grid_limit() * SB * (NV - 1) * f16x3::SLAB_FLOATS : 0);
}

int launch_points_mlp_f16(const DinerScene &s, const float *mlp_packed, const float *rays, const float *z, int64_t NR,
                          int K, float *scratch, float *rgbsigma, hipStream_t st)
{
    using namespace f16x3;
    const int64_t P = NR * (int64_t)K;
    if (P == 0 || s.SB == 0) return DINER_OK;
    if (s.C != DINER_D_LATENT) { set_error("render_points: latent channels C=%d unsupported (need %d)", s.C, DINER_D_LATENT); return DINER_E_UNSUPPORTED; }
    if (s.num_freqs != 6) { set_error("render_points: num_freqs=%d unsupported (need 6)", s.num_freqs); return DINER_E_UNSUPPORTED; }
    if (!scratch) { set_error("render_points(f16x3): scratch is NULL"); return DINER_E_INVALID; }
    if (hipMemsetAsync(scratch, 0, SYNC_WORDS * sizeof(float), st) != hipSuccess) return check_launch("memset(pacing counters)");
    const int64_t tiles = (P + TILE_P - 1) / TILE_P;
    const int64_t grid = tiles < f16_grid_limit() ? tiles : f16_grid_limit();
    static const bool stamp = getenv("DINER_F16_STAMP") != nullptr;  // diagnostics only
    if (stamp && s.linz_maps) {
        static unsigned long long *dbg = nullptr;
        if (!dbg && hipMalloc(&dbg, NWAVES * PH_COUNT * sizeof(unsigned long long)) != hipSuccess) return DINER_E_LAUNCH;
        (void)hipMemsetAsync(dbg, 0, NWAVES * PH_COUNT * sizeof(unsigned long long), st);
        hipLaunchKernelGGL((points_mlp_f16_kernel<true, true>), dim3((unsigned)grid, (unsigned)s.SB), dim3(NWAVES * 64), 0, st, s,
                           mlp_packed, rays, z, NR, K, tiles, scratch, rgbsigma, dbg);
        unsigned long long h[NWAVES * PH_COUNT];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        static const char *names[PH_COUNT] = {"geom", "gemm", "gather", "store", "bias", "barrier", "viewsum", "head"};
        for (int w = 0; w < NWAVES; ++w) {
            unsigned long long tot = 0;
            for (int i = 0; i < PH_COUNT; ++i) tot += h[w * PH_COUNT + i];
            fprintf(stderr, "[f16 stamp] wg0 wave%d:", w);
            for (int i = 0; i < PH_COUNT; ++i) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * (double)h[w * PH_COUNT + i] / (double)(tot ? tot : 1));
            fprintf(stderr, " total=%llu\n", tot);
        }
        return check_launch("points_mlp_f16_kernel<stamp>");
    }
    if (s.linz_maps)
        hipLaunchKernelGGL((points_mlp_f16_kernel<false, true>), dim3((unsigned)grid, (unsigned)s.SB), dim3(NWAVES * 64), 0, st, s,
                           mlp_packed, rays, z, NR, K, tiles, scratch, rgbsigma, (unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((points_mlp_f16_kernel<false, false>), dim3((unsigned)grid, (unsigned)s.SB), dim3(NWAVES * 64), 0, st, s,
                           mlp_packed, rays, z, NR, K, tiles, scratch, rgbsigma, (unsigned long long *)nullptr);
    return check_launch("points_mlp_f16_kernel");
}

}  // namespace diner
