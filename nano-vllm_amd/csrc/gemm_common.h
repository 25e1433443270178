// gemm_common.h -- what the projection kernels share: the launch record and the
// element-wise pieces of their epilogues (see gemm.hip for what each epilogue does).
#pragma once

#include "nvllm_internal.h"

namespace nvllm {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const __half* W;
    long nb_stride;   // halves between 16-row blocks
    int NB;           // 16-row blocks
    int Kb;           // k tiles
    const __half* A;  // activation tiles [Tb][Kb][2][512]
    int Tb;
    int N, T;
    int kb_per_split;
    int ksplit;
    float* slab;      // [tile][slice][round][item] float4
    unsigned* counters;
    int close_sc1;    // K split: the closing workgroup loads the partials past the vector cache instead of invalidating it
                      // (launches of at most one workgroup per CU over a hipMalloc slab)
    GemmEpilogue e;
#ifdef NVLLM_STAMPS
    unsigned long long* stamps;   // diagnostic builds only: [workgroup][8] constant-rate clock readings
#endif
};

// In-kernel timeline for diagnostic builds (tools/gemm_stamps.py): wave 0 of every workgroup
// records the 100 MHz constant-rate clock once the value `dep` exists.  Compiled out otherwise.
#ifdef NVLLM_STAMPS
#define NVLLM_STAMP(k, dep)                                                                                   \
    do {                                                                                                      \
        unsigned long long t__;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) : "v"(dep) : "memory");            \
        if (p.stamps && threadIdx.x == 0)                                                                     \
            p.stamps[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = t__; \
    } while (0)
// the same inside gemm_tile(), whose caller says where the tile's eight readings go (`stamps_wg`)
#define NVLLM_TSTAMP(k, dep)                                                                        \
    do {                                                                                            \
        unsigned long long t__;                                                                     \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) : "v"(dep) : "memory");  \
        if (stamps_wg && tid == 0) stamps_wg[(k)] = t__;                                    \
    } while (0)
#else
#define NVLLM_STAMP(k, dep) do {} while (0)
#define NVLLM_TSTAMP(k, dep) do {} while (0)
#endif

__device__ __forceinline__ float silu_mul(float g, float u) {
    // reference layers.c:156-158: g / (1 + expf(-g)) * u
    return __fmul_rn(__fdiv_rn(g, __fadd_rn(1.0f, expf(-g))), u);
}

__device__ __forceinline__ void split16_4(const float* y, __half* hi, __half* lo) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float c = fminf(fmaxf(y[r], -65504.f), 65504.f);
        hi[r] = __float2half_rn(c);
        lo[r] = __float2half_rn(c - __half2float(hi[r]));
    }
}

// store 4 consecutive k of token t into hi/lo tiles (n % 4 == 0)
__device__ __forceinline__ void store_a16_4(__half* a16, int Kb_out, int t, int n, const float* y) {
    alignas(8) __half hi[4], lo[4];
    split16_4(y, hi, lo);
    const size_t tile = ((size_t)(t >> 4) * Kb_out + (n >> 5)) * 2;
    const size_t off = (size_t)((t & 15) + 16 * ((n >> 3) & 3)) * 8 + (n & 7);
    *reinterpret_cast<uint2*>(a16 + tile * TILE_HALVES + off) = *reinterpret_cast<uint2*>(hi);
    *reinterpret_cast<uint2*>(a16 + (tile + 1) * TILE_HALVES + off) = *reinterpret_cast<uint2*>(lo);
}

template <typename KVT>
__device__ __forceinline__ void kv_put4(KVT* dst, const float* y);
template <>
__device__ __forceinline__ void kv_put4<float>(float* dst, const float* y) {
    *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
}
__device__ __forceinline__ __half kv_sat_half(float v) {   // finite fp16 only: masked keys need 0 * finite = 0
    return __float2half_rn(fminf(fmaxf(v, -65504.f), 65504.f));
}
template <>
__device__ __forceinline__ void kv_put4<__half>(__half* dst, const float* y) {
    alignas(8) __half h[4] = {kv_sat_half(y[0]), kv_sat_half(y[1]), kv_sat_half(y[2]), kv_sat_half(y[3])};
    *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(h);
}
// four consecutive dims of one key in a V page: 4 halves apart
template <typename KVT>
__device__ __forceinline__ void kv_put4_strided(KVT* dst, const float* y);
template <>
__device__ __forceinline__ void kv_put4_strided<float>(float*, const float*) {}
template <>
__device__ __forceinline__ void kv_put4_strided<__half>(__half* dst, const float* y) {
#pragma unroll
    for (int e = 0; e < 4; e++) dst[4 * e] = kv_sat_half(y[e]);
}

// gemm_prefill.hip: 128 x 128 output tiles through LDS, for batches of hundreds of tokens
void launch_gemm_prefill(GemmArgs a);

}  // namespace nvllm
