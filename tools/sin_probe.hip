// Hardware probe (gfx950): accuracy of v_sin_f32 behind a compensated range reduction, over the argument range of DINER's positional
// encodings (|arg| <= ~700 rad: camera coordinates x 6.28 x 32 + pi/2), against sin() in double precision and against sinf().
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/sin_probe tools/sin_probe.hip && tools/sin_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__device__ __forceinline__ float sin_hw(float a)
{
    const float C_HI = 0.15915494f;                    // fp32(1 / 2pi)
    const float C_LO = (float)(0.15915494309189535 - (double)0.15915494f);
    const float r1 = a * C_HI;
    const float r2 = __builtin_fmaf(a, C_HI, -r1) + a * C_LO;   // the part of a / 2pi that r1 lost
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r1) + r2);   // v_sin_f32: sin(2 pi x)
}

__global__ void probe(const float *a, float *hw, float *lib, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { hw[i] = sin_hw(a[i]); lib[i] = sinf(a[i]); }
}

int main()
{
    const int n = 1 << 24;
    std::vector<float> a(n), hw(n), lib(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)(s >> 8) / 16777216.0f;          // [0, 1)
        const int oct = i % 6;                                   // the six octaves of the encoding
        const float x = (u * 2.0f - 1.0f) * 3.5f;                // camera coordinates / depth offsets
        a[i] = __builtin_fmaf(x, 6.28f * (float)(1 << oct), (i & 64) ? 1.5707963267948966f : 0.0f);
    }
    float *da, *dh, *dl;
    if (hipMalloc(&da, n * 4) != hipSuccess || hipMalloc(&dh, n * 4) != hipSuccess || hipMalloc(&dl, n * 4) != hipSuccess) return 1;
    (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, dh, dl, n);
    (void)hipMemcpy(hw.data(), dh, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(lib.data(), dl, n * 4, hipMemcpyDeviceToHost);
    double e_hw[6] = {}, e_lib[6] = {};
    for (int i = 0; i < n; ++i) {
        const double ref = sin((double)a[i]);
        const int oct = i % 6;
        e_hw[oct] = fmax(e_hw[oct], fabs((double)hw[i] - ref));
        e_lib[oct] = fmax(e_lib[oct], fabs((double)lib[i] - ref));
    }
    for (int o = 0; o < 6; ++o) printf("octave %d (|arg| <= %6.1f): max |v_sin path - sin| %.3e   max |sinf - sin| %.3e\n", o, 3.5 * 6.28 * (1 << o) + 1.6, e_hw[o], e_lib[o]);
    return 0;
}
