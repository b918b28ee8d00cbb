// Hardware probe for the GEMM chain of the fused point/MLP kernel (diner_amd/csrc/points_mlp_f16.hip):
// NL full 512x512 layers per 64-point tile, each = relu/split/store of the previous outputs into the LDS operand
// image + a split-fp16 GEMM (3 MFMAs per product), nothing else (no geometry, no gather).  Variants:
//   ref      round-1 structure: compiler-allocated accumulators, 2-deep VGPR weight ring, 2 barriers per layer
//   asm      the generated core (f16_core.inc): accumulators + weight ring in AGPRs, 3 barriers per layer, all 8 waves in step
//   asm+stg  the same with waves 4-7 delayed by one slot (half-layer stagger): their stores run beside the partner's MFMAs
// Prints ms per variant and the MFMA-pipe utilisation; checks that all variants produce bit-identical accumulators.
// Build: tools/build_chain_probe.sh     (run the binary on the GPU box)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#include "chain_probe_cores.inc"   // generated: namespaces d4 (ring 4), d2 (ring 2), d4nl (ring 4, no weight loads)

constexpr int NKB = 32;
constexpr int64_t W_LAYER = 16LL * NKB * 2 * 64 * 8;  // halfs per layer
constexpr int IMG_BYTES = 128 * 1024;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- the core's registers (high VGPRs reserved from hipcc by amdgpu_num_vgpr) accessed from C++ glue ---------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ float agpr_read()
{
    float v;
    asm volatile("v_mov_b32 %0, v%c1" : "=v"(v) : "n"(N));
    return v;
}
template <int N> __device__ __forceinline__ void agpr_write(float v)
{
    asm volatile("v_mov_b32 v%c0, %1" ::"n"(N), "v"(v));
}

__device__ __forceinline__ void split(float s, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
}

// new image layout: unit-row u (8 k) : hi at u*2048 + row*16, lo at +1024
template <int BASE, int G> __device__ __forceinline__ void store_group(char *img, int wave, int c, int h)
{
    // registers 4g..4g+3 of tile (tn,tp): features 64w + 32tn + 8g + 4h + j of point 32tp + c
    constexpr int tile = G >> 2, g = G & 3, tn = tile >> 1, tp = tile & 1;
    h4 vh, vl;
#define ONE(J)                                                         \
    {                                                                  \
        const float v = agpr_read<BASE + 16 * tile + 4 * g + J>();     \
        _Float16 hi, lo;                                               \
        split(v < 0.0f ? 0.0f : v, hi, lo);                            \
        vh[J] = hi;                                                    \
        vl[J] = lo;                                                    \
    }
    ONE(0) ONE(1) ONE(2) ONE(3)
#undef ONE
    const int u = wave * 8 + tn * 4 + g;
    char *p = img + u * 2048 + (tp * 32 + c) * 16 + 8 * h;
    *(h4 *)p = vh;
    *(h4 *)(p + 1024) = vl;
}
template <int BASE, int G = 0> __device__ __forceinline__ void store_relu_agpr(char *img, int wave, int c, int h)
{
    if constexpr (G < 16) {
        store_group<BASE, G>(img, wave, c, h);
        store_relu_agpr<BASE, G + 1>(img, wave, c, h);
    }
}
template <int BASE, int I = 0> __device__ __forceinline__ void zero_agpr()
{
    if constexpr (I < 64) {
        agpr_write<BASE + I>(0.0f);
        zero_agpr<BASE, I + 1>();
    }
}
template <int BASE, int I = 0> __device__ __forceinline__ void dump_agpr(float *dst)
{
    if constexpr (I < 64) {
        dst[I * 64] = agpr_read<BASE + I>();
        dump_agpr<BASE, I + 1>(dst);
    }
}

__device__ __forceinline__ void init_image_new(char *img, int tid)
{
    // deterministic pseudo-random activations in [0, 1): value depends on (k, row)
    for (int i = tid; i < 64 * 64; i += 512) {
        const int u = i >> 6, row = i & 63;
        h8 vh, vl;
        for (int j = 0; j < 8; ++j) {
            const unsigned k = u * 8 + j, x = (k * 2654435761u) ^ (row * 40503u + 12345u);
            const float v = (float)((x >> 8) & 0xffff) * (1.0f / 65536.0f);
            _Float16 hi, lo;
            split(v, hi, lo);
            vh[j] = hi;
            vl[j] = lo;
        }
        *(h8 *)(img + u * 2048 + row * 16) = vh;
        *(h8 *)(img + u * 2048 + 1024 + row * 16) = vl;
    }
}

#define DEFINE_CHAIN(NS)                                                                                                          \
    template <bool STAGGER, bool STORE>                                                                                           \
    __global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(NS::F16_VGPR_CAP / 2))) void chain_##NS(                         \
        const _Float16 *__restrict__ W, int NLW, int NL, int tiles, float *__restrict__ out, unsigned long long *__restrict__ clk) \
    {                                                                                                                             \
        __shared__ __attribute__((aligned(16))) char img[IMG_BYTES];                                                              \
        const int tid = threadIdx.x, lane = tid & 63;                                                                             \
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                                \
        const int c = lane & 31, h = lane >> 5;                                                                                   \
        init_image_new(img, tid);                                                                                                 \
        const unsigned loff = lane * 16;                                                                                          \
        const unsigned ab0 = (unsigned)(uintptr_t)img + h * 2048 + c * 16;                                                        \
        auto wptr = [&](int layer, int tn) -> uint64_t {                                                                          \
            return (uint64_t)(uintptr_t)(W + (int64_t)(layer % NLW) * W_LAYER) + (uint64_t)(wave * 2 + tn) * (NKB * 2048);        \
        };                                                                                                                        \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                        \
        NS::ring_prologue(wptr(0, 0), wptr(0, 1), loff);                                                                          \
        if (STAGGER && wave >= 4) __builtin_amdgcn_s_barrier(); /* waves 4-7 run one slot behind */                               \
        int layer = 0;                                                                                                            \
        for (int t = 0; t < tiles; ++t) {                                                                                         \
            for (int l = 0; l < NL; l += 2) {                                                                                     \
                zero_agpr<NS::F16_NET>();                                                                                                  \
                NS::layer_net_full(wptr(layer, 0), wptr(layer, 1), wptr(layer + 1, 0), wptr(layer + 1, 1), loff, ab0, NS::Sync{});            \
                ++layer;                                                                                                          \
                if (STORE) store_relu_agpr<NS::F16_NET>(img, wave, c, h);                                                                  \
                zero_agpr<NS::F16_X>();                                                                                                   \
                NS::layer_x_full(wptr(layer, 0), wptr(layer, 1), wptr(layer + 1, 0), wptr(layer + 1, 1), loff, ab0, NS::Sync{});              \
                ++layer;                                                                                                          \
                if (STORE) store_relu_agpr<NS::F16_X>(img, wave, c, h);                                                                   \
            }                                                                                                                     \
        }                                                                                                                         \
        if (STAGGER && wave < 4) __builtin_amdgcn_s_barrier();                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* drain the ring before the wave ends */                                \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
        if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }                                       \
        dump_agpr<NS::F16_X>(out + ((int64_t)blockIdx.x * 8 + wave) * 64 * 64 + lane);                                                    \
    }
DEFINE_CHAIN(d4)
DEFINE_CHAIN(d2)
DEFINE_CHAIN(d4nl)

// ---- reference: round-1 structure (compiler-managed registers) ---------------------------------------------------
namespace ref {
constexpr int TILE_P = 64, CT = 2;
constexpr int UNITS = 64 * TILE_P;
__device__ __forceinline__ int unit(int u, int row) { return u * TILE_P + (row ^ (u & 63)); }
template <int OFF> __device__ __forceinline__ void wload(h8 &dst, const char *ptr)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wwait(h8 (&w)[CT][2])
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
}
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[CT][2], const h8 *Ahi, const h8 *Alo, const h8 *__restrict__ Wl, int wave, int lane)
{
    const int r = lane & 31, hh = lane >> 5;
    const char *wp0 = (const char *)(Wl + (int64_t)(wave * CT + 0) * NKB * 128 + lane);
    const char *wp1 = (const char *)(Wl + (int64_t)(wave * CT + 1) * NKB * 128 + lane);
    h8 w[2][CT][2];
    wload<0>(w[0][0][0], wp0); wload<1024>(w[0][0][1], wp0); wload<0>(w[0][1][0], wp1); wload<1024>(w[0][1][1], wp1);
#define STEP(ST, LOADS, WAITN)                                                                                  \
    {                                                                                                           \
        const int kb = kb0 + ST;                                                                                \
        if (LOADS) {                                                                                            \
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
    for (; kb0 < NKB - 2; kb0 += 2) { STEP(0, true, 4) STEP(1, true, 4) }
    STEP(0, true, 4) STEP(1, false, 0)
#undef STEP
}
__device__ __forceinline__ void store_relu(const f32x16 (&acc)[CT][2], _Float16 *Ahi, _Float16 *Alo, int wave, int lane)
{
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
                    split(v < 0.0f ? 0.0f : v, hi, lo);
                    vh[j] = hi;
                    vl[j] = lo;
                }
                const int o = unit(u, tp * 32 + c) * 8 + 4 * h;
                *(h4 *)(Ahi + o) = vh;
                *(h4 *)(Alo + o) = vl;
            }
        }
}
__global__ __launch_bounds__(512) void chain_ref(const _Float16 *__restrict__ W, int NLW, int NL, int tiles, float *__restrict__ out)
{
    __shared__ h8 lds[2 * UNITS];
    h8 *Ahi8 = lds, *Alo8 = lds + UNITS;
    _Float16 *Ahi = (_Float16 *)Ahi8, *Alo = (_Float16 *)Alo8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 64 * 64; i += 512) {
        const int u = i >> 6, row = i & 63;
        h8 vh, vl;
        for (int j = 0; j < 8; ++j) {
            const unsigned k = u * 8 + j, x = (k * 2654435761u) ^ (row * 40503u + 12345u);
            const float v = (float)((x >> 8) & 0xffff) * (1.0f / 65536.0f);
            _Float16 hi, lo;
            split(v, hi, lo);
            vh[j] = hi;
            vl[j] = lo;
        }
        Ahi8[unit(u, row)] = vh;
        Alo8[unit(u, row)] = vl;
    }
    __syncthreads();
    f32x16 x[CT][2], net[CT][2];
    int layer = 0;
    for (int t = 0; t < tiles; ++t) {
        for (int l = 0; l < NL; l += 2) {
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) net[a][b][i] = 0.0f;
            gemm_tile(net, Ahi8, Alo8, (const h8 *)(W + (int64_t)(layer % NLW) * W_LAYER), wave, lane);
            ++layer;
            __syncthreads();
            store_relu(net, Ahi, Alo, wave, lane);
            __syncthreads();
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) x[a][b][i] = 0.0f;
            gemm_tile(x, Ahi8, Alo8, (const h8 *)(W + (int64_t)(layer % NLW) * W_LAYER), wave, lane);
            ++layer;
            __syncthreads();
            store_relu(x, Ahi, Alo, wave, lane);
            __syncthreads();
        }
    }
    float *dst = out + ((int64_t)blockIdx.x * 8 + wave) * 64 * 64 + lane;
    for (int tn = 0; tn < CT; ++tn) for (int tp = 0; tp < 2; ++tp) for (int i = 0; i < 16; ++i) dst[((tn * 2 + tp) * 16 + i) * 64] = x[tn][tp][i];
}
}  // namespace ref



int main(int argc, char **argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 32, NL = 28, NLW = 14, reps = argc > 2 ? atoi(argv[2]) : 3;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d, tiles/WG %d, layers/tile %d\n", cus, tiles, NL);
    // weights: kaiming-ish so that the relu chain keeps O(1) magnitudes: std = sqrt(2/512)
    std::vector<_Float16> hw((size_t)NLW * W_LAYER);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < hw.size(); i += 1024) {   // [..][part][lane][8]: hi block then lo block
        const bool lo_part = (i / 512) & 1;
        (void)lo_part;
        for (size_t j = 0; j < 1024; ++j) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const float u = (float)((s >> 11) & 0xfffff) * (1.0f / 1048576.0f) - 0.5f;   // uniform(-.5,.5): std .289
            const float w = u * 0.2165f;                                                   // std 0.0625 = sqrt(2/512)
            const bool is_lo = (j >= 512);
            hw[i + j] = (_Float16)(is_lo ? w * 4.8828125e-4f : w);                         // lo parts ~2^-11 of hi
        }
    }
    _Float16 *dW;
    float *d0, *d1, *d2, *d3;
    const size_t out_floats = (size_t)cus * 8 * 64 * 64;
    CK(hipMalloc(&dW, hw.size() * 2));
    CK(hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d0, out_floats * 4)); CK(hipMalloc(&d1, out_floats * 4)); CK(hipMalloc(&d2, out_floats * 4)); CK(hipMalloc(&d3, out_floats * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long *dclk;
    CK(hipMalloc(&dclk, cus * 16));
    const char *names[] = {"ref 32x32x16", "d4", "d4+stg", "d2", "d2+stg", "d4 nostore", "d4+stg nostore", "d4 noload", "d4+stg noload", "d4 noload nostore"};
    auto run = [&](int which, float *dst) {
        float best = 1e30f;
        double clock = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(dclk, 0, cus * 16));
            CK(hipEventRecord(e0));
            switch (which) {
            case 0: hipLaunchKernelGGL(ref::chain_ref, dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst); break;
            case 1: hipLaunchKernelGGL((chain_d4<false, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 2: hipLaunchKernelGGL((chain_d4<true, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 3: hipLaunchKernelGGL((chain_d2<false, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 4: hipLaunchKernelGGL((chain_d2<true, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 5: hipLaunchKernelGGL((chain_d4<false, false>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 6: hipLaunchKernelGGL((chain_d4<true, false>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 7: hipLaunchKernelGGL((chain_d4nl<false, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 8: hipLaunchKernelGGL((chain_d4nl<true, true>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            case 9: hipLaunchKernelGGL((chain_d4nl<false, false>), dim3(cus), dim3(512), 0, 0, dW, NLW, NL, tiles, dst, dclk); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) {
                best = ms;
                if (which) {
                    std::vector<unsigned long long> hc(cus * 2);
                    CK(hipMemcpy(hc.data(), dclk, cus * 16, hipMemcpyDeviceToHost));
                    std::vector<double> v;
                    for (int i = 0; i < cus; ++i) if (hc[2 * i + 1]) v.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1);
                    std::sort(v.begin(), v.end());
                    clock = v.empty() ? 0 : v[v.size() / 2];
                }
            }
        }
        const double mfma_cycles = (double)tiles * NL * 2 * 384 * 32;   // per SIMD
        const double ghz = clock > 0 ? clock : 2.4;
        printf("%-18s %8.3f ms  %6.3f us/layer  clock %.2f GHz  MFMA pipe busy %.1f %% of that clock  executed %.0f TFLOP/s\n", names[which], best,
               best * 1e3 / (tiles * NL), clock, 100.0 * mfma_cycles / (best * 1e-3 * ghz * 1e9), (double)cus * tiles * NL * 8 * 384 * 32768.0 / (best * 1e-3) / 1e12);
        return best;
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(0, d0); run(1, d1); run(2, d2);
        for (int w = 3; w < 10; ++w) run(w, d3);
    }
    std::vector<float> h0(out_floats), h1(out_floats), h2(out_floats);
    CK(hipMemcpy(h0.data(), d0, out_floats * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), d1, out_floats * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), d2, out_floats * 4, hipMemcpyDeviceToHost));
    size_t bad1 = 0, bad2 = 0, nz = 0;
    double amax = 0;
    for (size_t i = 0; i < out_floats; ++i) {
        bad1 += h0[i] != h1[i];
        bad2 += h0[i] != h2[i];
        nz += h0[i] != 0.0f;
        if (fabs(h0[i]) > amax) amax = fabs(h0[i]);
    }
    printf("outputs: %zu values, %zu non-zero, max |x| %.4g; d4 differs from ref in %zu, d4+stg in %zu\n", out_floats, nz, amax, bad1, bad2);
    if (bad1 || bad2)
        for (size_t i = 0, shown = 0; i < out_floats && shown < 8; ++i)
            if (h0[i] != h1[i] || h0[i] != h2[i]) { printf("  [%zu] ref %.9g asm %.9g stg %.9g\n", i, h0[i], h1[i], h2[i]); ++shown; }
    return (bad1 || bad2) ? 2 : 0;
}
