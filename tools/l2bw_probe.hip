// Hardware probe (not product code): what rate does a CU sustain when every workgroup of the chip streams the
// SAME weight image (L2/MALL-resident) straight into registers, the way points_mlp_f16_kernel streams its weights?
// Variants: loads in flight per wave, waves per CU, per-workgroup rotated start, dwordx4 vs dwordx2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// image = nlayers x 1 MiB; inside a layer wave w (of NW) owns a contiguous 1 MiB / NW slice, read as 1-KiB pieces
template <int INFLIGHT, bool ROT, typename V>
__global__ __launch_bounds__(1024) void stream_kernel(const char *img, int nlayers, int reps, unsigned *out)
{
    const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t slice = (1u << 20) / nw;
    const int pieces = (int)(slice / (64 * sizeof(V)));
    V acc = {};
    for (int r = 0; r < reps; ++r)
        for (int l0 = 0; l0 < nlayers; ++l0) {
            const int l = ROT ? (l0 + blockIdx.x) % nlayers : l0;
            const V *p = (const V *)(img + ((size_t)l << 20) + wave * slice) + lane;
            for (int i = 0; i < pieces; i += INFLIGHT) {
                V v[INFLIGHT];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) v[j] = p[(size_t)(i + j) * 64];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) acc ^= v[j];
            }
        }
    unsigned s = 0;
    for (int j = 0; j < (int)(sizeof(V) / 4); ++j) s ^= acc[j];
    if (s == 0x12345678u) out[blockIdx.x] = s;
}

template <int INFLIGHT, bool ROT, typename V>
static void run(const char *name, const char *img, unsigned *out, int threads, int nlayers, int reps, int grid)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    stream_kernel<INFLIGHT, ROT, V><<<grid, threads>>>(img, nlayers, 1, out);
    hipEventRecord(a);
    stream_kernel<INFLIGHT, ROT, V><<<grid, threads>>>(img, nlayers, reps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * nlayers * reps * (1 << 20);
    printf("%-34s threads %4d layers %2d: %7.2f ms  %6.2f TB/s chip  %6.1f GB/s per CU  %5.1f B/clk/CU @2.4GHz\n", name, threads, nlayers, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / grid, bytes / (ms * 1e-3) / grid / 2.4e9);
}

int main()
{
    const int nlayers = 14;
    char *img; unsigned *out;
    hipMalloc(&img, (size_t)nlayers << 20); hipMalloc(&out, 4096);
    hipMemset(img, 1, (size_t)nlayers << 20);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", grid, pr.clockRate);
    const int reps = 40;
    for (int threads : {256, 512, 1024}) {
        run<2, false, u4>("x4 inflight 2", img, out, threads, nlayers, reps, grid);
        run<4, false, u4>("x4 inflight 4", img, out, threads, nlayers, reps, grid);
        run<8, false, u4>("x4 inflight 8", img, out, threads, nlayers, reps, grid);
        run<16, false, u4>("x4 inflight 16", img, out, threads, nlayers, reps, grid);
        run<8, true, u4>("x4 inflight 8 rotated layers", img, out, threads, nlayers, reps, grid);
        run<16, false, u2>("x2 inflight 16", img, out, threads, nlayers, reps, grid);
    }
    // one layer only (1 MiB: fits every L2 trivially) -> the pure L2->CU rate
    run<8, false, u4>("x4 inflight 8, 1 layer", img, out, 512, 1, reps * 14, grid);
    run<16, false, u4>("x4 inflight 16, 1 layer", img, out, 1024, 1, reps * 14, grid);
    // half the chip (does the per-CU rate rise when fewer CUs pull on each L2?)
    run<8, false, u4>("x4 inflight 8, 128 WGs", img, out, 512, nlayers, reps, grid / 2);
    run<8, false, u4>("x4 inflight 8, 64 WGs", img, out, 512, nlayers, reps, grid / 4);
    return 0;
}
