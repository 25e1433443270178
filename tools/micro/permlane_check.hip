// Checks the row-swap form of "maximum over the four 16-lane rows of a wave" (v_permlane32_swap + v_permlane16_swap,
// gfx950) against the __shfl_xor form, and prints what the two instructions do to lane ids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float rows_max_swap(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float m = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned v = __builtin_bit_cast(unsigned, m);
    const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float rows_max_shfl(float x) {
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
}
__global__ void k(const float* in, float* o1, float* o2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    o1[i] = rows_max_swap(in[i]);
    o2[i] = rows_max_shfl(in[i]);
}
int main() {
    const int n = 256 * 64;
    float *h = (float*)malloc(n * 4), *a = (float*)malloc(n * 4), *b = (float*)malloc(n * 4);
    srand(3);
    for (int i = 0; i < n; i++) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *d, *d1, *d2;
    if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&d1, n * 4) != hipSuccess || hipMalloc(&d2, n * 4) != hipSuccess) return 2;
    if (hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, d1, d2);
    if (hipMemcpy(a, d1, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b, d2, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int i = 0; i < n; i++) bad += a[i] != b[i];
    printf("row-swap maximum vs shuffle maximum: %d of %d lanes differ\n", bad, n);
    for (int i = 0; i < 64; i += 16) printf("  lane %2d: in %+.3f swap %+.3f shfl %+.3f\n", i, h[i], a[i], b[i]);
    return bad != 0;
}
