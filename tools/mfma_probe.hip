// Hardware probe (not product code): does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs,
// what is the fragment layout, and what rate do back-to-back f16 MFMAs reach on random data?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void denorm_kernel(float *out, float aval, float bval)
{
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
    a[0] = (_Float16)aval;  // lane l: A[row l&31][k = 8*(l>>5) + 0]
    b[0] = (_Float16)bval;
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}

// layout check: A[i][k] = i*16+k (exact in fp16 up to 2048), B[k][j] = (k==kk) one-hot rows -> C[i][j] = A[i][kk]
__global__ void layout_kernel(float *out, int kk)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a[j] = (_Float16)(float)(r * 16 + k);
        b[j] = (_Float16)((k == kk) ? (float)(r + 1) : 0.0f);  // B[k][col r] = (k==kk)*(col+1)
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(const _Float16 *src, float *out, int iters)
{
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = src[(threadIdx.x * 8 + j) & 4095]; b[j] = src[(threadIdx.x * 8 + j + 777) & 4095]; }
    f16v c[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void rate16_kernel(const _Float16 *src, float *out, int iters)
{
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = src[(threadIdx.x * 8 + j) & 4095]; b[j] = src[(threadIdx.x * 8 + j + 777) & 4095]; }
    f4v c[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += c[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d; hipMalloc(&d, 1 << 22);
    float h[1024];
    // 1) denormals: 2^-20 is subnormal in fp16; times 2^10 -> 2^-10 if inputs are kept
    denorm_kernel<<<1, 64>>>(d, ldexpf(1.f, -20), 1024.f);
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    printf("denorm A: got %g expect %g\n", h[0], ldexpf(1.f, -10));
    denorm_kernel<<<1, 64>>>(d, 1024.f, ldexpf(1.f, -20));
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    printf("denorm B: got %g expect %g\n", h[0], ldexpf(1.f, -10));
    denorm_kernel<<<1, 64>>>(d, ldexpf(1.f, -24), 1.f);
    hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
    printf("denorm min A: got %g expect %g\n", h[0], ldexpf(1.f, -24));
    // 2) layout
    int bad = 0;
    for (int kk = 0; kk < 16; ++kk) {
        layout_kernel<<<1, 64>>>(d, kk);
        static float o[64 * 16];
        hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) {
            const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
            const float want = (float)(row * 16 + kk) * (float)(col + 1);
            if (o[l * 16 + i] != want) { if (bad < 5) printf("layout mismatch kk=%d lane=%d reg=%d got %g want %g\n", kk, l, i, o[l*16+i], want); ++bad; }
        }
    }
    printf("layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok: A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], C col=l&31 row=(i&3)+8*(i>>2)+4*(l>>5)", bad);
    // 3) rate on random data
    _Float16 *src; hipMalloc(&src, 4096 * 2);
    { _Float16 hs[4096]; srand(1); for (int i = 0; i < 4096; ++i) hs[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f); hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 2;  // 2 blocks of 4 waves per CU = 2 waves/SIMD
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        rate_kernel<4><<<blocks, 256>>>(src, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("f16 32x32x16 rate: %.1f TFLOP/s (%.2f ms)\n", flop / ms / 1e9, ms);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rate16_kernel<8><<<blocks, 256>>>(src, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 8 * 16384.0;
        printf("f16 16x16x32 rate: %.1f TFLOP/s (%.2f ms)\n", flop / ms / 1e9, ms);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rate_kernel<4><<<blocks, 256>>>(src, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
        printf("f16 32x32x16 rate (again): %.1f TFLOP/s (%.2f ms)\n", flop / ms / 1e9, ms);
    }
    return 0;
}
