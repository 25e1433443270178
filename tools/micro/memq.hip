// Micro-benchmark: what one CU can take in when a once-read HBM stream (weights) and a
// re-read, L2-resident stream (activations) share it -- the regime of the decode projections.
// One 512-thread workgroup per CU; every wave keeps two batches of 16-byte-per-lane loads in
// flight (1 KiB per wave instruction).  Variants differ only in which waves issue which
// stream and in which order:
//   W      every wave streams weights only (HBM, non-temporal)
//   A      every wave reads activations only (a 512 KiB buffer that every workgroup reads: L2)
//   MIX    every wave does both in every step (what k_gemm does)
//   SPEC   waves 0-3 stream weights only, waves 4-7 activations only (same totals per CU)
//   PHASE  every wave: four steps of activations, then four steps of weights, ...
//   ALDS   as MIX, but activations go global -> LDS by direct-to-LDS loads (never read back)
//   hipcc --offload-arch=gfx950 -O3 -o memq memq.hip && ./memq
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int A_TILES = 512;   // 512 KiB of activations, shared by all workgroups

// One batch: NW weight tiles + NA activation tiles per wave.
template <int NW, int NA>
struct Batch {
    u4 w[NW > 0 ? NW : 1];
    u4 a[NA > 0 ? NA : 1];
};

template <int NW, int NA>
__device__ __forceinline__ void issue(Batch<NW, NA>& b, const u4* wp, const u4* ap, int wtile, int atile, int lane) {
#pragma unroll
    for (int i = 0; i < NW; i++) b.w[i] = __builtin_nontemporal_load(wp + (size_t)(wtile + i) * 64 + lane);
#pragma unroll
    for (int i = 0; i < NA; i++) b.a[i] = ap[(size_t)((atile + i) & (A_TILES - 1)) * 64 + lane];
}
template <int NW, int NA>
__device__ __forceinline__ void consume(const Batch<NW, NA>& b, u4& acc) {
#pragma unroll
    for (int i = 0; i < NW; i++) acc ^= b.w[i];
#pragma unroll
    for (int i = 0; i < NA; i++) acc ^= b.a[i];
}

// `steps` steps of (NW, NA) tiles per wave; weight tiles of a wave are consecutive in its own range.
template <int NW, int NA>
__device__ __forceinline__ void stream(const u4* wp, const u4* ap, int& wtile, int& atile, int steps, int lane, u4& acc) {
    Batch<NW, NA> b0, b1;
    if (steps <= 0) return;
    issue(b0, wp, ap, wtile, atile, lane);
    wtile += NW; atile += NA;
    int s = 0;
    for (; s + 2 <= steps; s += 2) {
        issue(b1, wp, ap, wtile, atile, lane);
        wtile += NW; atile += NA;
        consume(b0, acc);
        if (s + 2 < steps) {
            issue(b0, wp, ap, wtile, atile, lane);
            wtile += NW; atile += NA;
        }
        consume(b1, acc);
    }
    if (s < steps) consume(b0, acc);
}

enum { M_W = 0, M_A, M_MIX, M_SPEC, M_PHASE, M_ALDS, M_COUNT };
static const char* kNames[M_COUNT] = {"W", "A", "MIX", "SPEC", "PHASE", "ALDS"};

// per step per CU: 32 KiB of weights, 64 KiB of activations (the 1:2 ratio of a 64-token projection)
template <int MODE>
__global__ __launch_bounds__(512) void k_memq(const u4* W, const u4* A, u4* sink, int steps, long w_tiles_per_wg) {
    extern __shared__ u4 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u4* wp = W + (size_t)blockIdx.x * w_tiles_per_wg * 64;
    u4 acc = {0, 0, 0, 0};
    int atile = (blockIdx.x * 5 + wave * 61) & (A_TILES - 1);
    if (MODE == M_W) {
        int wtile = wave * (int)(w_tiles_per_wg / 8);
        stream<4, 0>(wp, A, wtile, atile, steps, lane, acc);
    } else if (MODE == M_A) {
        int wtile = 0;
        stream<0, 8>(wp, A, wtile, atile, steps, lane, acc);
    } else if (MODE == M_MIX) {
        int wtile = wave * (int)(w_tiles_per_wg / 8);
        stream<4, 8>(wp, A, wtile, atile, steps, lane, acc);
    } else if (MODE == M_SPEC) {
        if (wave < 4) {
            int wtile = wave * (int)(w_tiles_per_wg / 4);
            stream<8, 0>(wp, A, wtile, atile, steps, lane, acc);
        } else {
            int wtile = 0;
            stream<0, 16>(wp, A, wtile, atile, steps, lane, acc);
        }
    } else if (MODE == M_PHASE) {
        int wtile = wave * (int)(w_tiles_per_wg / 8);
        for (int s = 0; s < steps; s += 4) {
            stream<0, 16>(wp, A, wtile, atile, 2, lane, acc);   // 2 steps x 16 KiB x 8 waves = 256 KiB of activations
            stream<8, 0>(wp, A, wtile, atile, 2, lane, acc);    // 2 steps x 8 KiB x 8 waves = 128 KiB of weights
        }
    } else {   // M_ALDS: activations by direct-to-LDS loads into a per-wave 8 KiB region, weights to registers
        int wtile = wave * (int)(w_tiles_per_wg / 8);
        u4 b0[4], b1[4];
        __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)lds + wave * 16384;
        auto dma = [&](int slot) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                __builtin_amdgcn_global_load_lds((const void*)(A + (size_t)((atile + i) & (A_TILES - 1)) * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(dst + slot * 8192 + i * 1024), 16, 0, 0);
            }
            atile += 8;
        };
        for (int s = 0; s < steps; s += 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) b0[i] = __builtin_nontemporal_load(wp + (size_t)(wtile + i) * 64 + lane);
            dma(0);
#pragma unroll
            for (int i = 0; i < 4; i++) b1[i] = __builtin_nontemporal_load(wp + (size_t)(wtile + 4 + i) * 64 + lane);
            dma(1);
            wtile += 8;
#pragma unroll
            for (int i = 0; i < 4; i++) { acc ^= b0[i]; acc ^= b1[i]; }
        }
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE>
static float run(const u4* W, const u4* A, u4* sink, int steps, long w_tiles_per_wg, int wgs, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    // the events carry the dispatch's own begin and end timestamps
    hipExtLaunchKernelGGL(k_memq<MODE>, dim3(wgs), dim3(512), MODE == M_ALDS ? 131072 : 0, st, e0, e1, 0, W, A, sink, steps, w_tiles_per_wg);
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int steps = argc > 2 ? atoi(argv[2]) : 12;   // multiple of 4
    const int regions = 10, reps = 20;
    const long w_tiles_per_wg = (long)steps * 32;       // 32 KiB per step per CU
    const size_t region_bytes = (size_t)wgs * w_tiles_per_wg * 1024;
    u4 *W, *A, *sink;
    CHECK(hipMalloc((void**)&W, region_bytes * regions));
    CHECK(hipMalloc((void**)&A, (size_t)A_TILES * 1024));
    CHECK(hipMalloc((void**)&sink, (size_t)wgs * 512 * 16));
    CHECK(hipMemset(W, 1, region_bytes * regions));
    CHECK(hipMemset(A, 2, (size_t)A_TILES * 1024));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)k_memq<M_ALDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const double wkb = steps * 32.0, akb = steps * 64.0;
    printf("memq: %d workgroups x 512 threads, per CU: %.0f KiB weights (HBM, %zu MB per launch over %d regions), %.0f KiB activations (L2)\n",
           wgs, wkb, region_bytes >> 20, regions, akb);
    printf("%-6s %10s %10s %14s %14s\n", "mode", "min us", "median us", "W GB/s per CU", "all GB/s per CU");
    for (int mode = 0; mode < M_COUNT; mode++) {
        std::vector<float> t;
        for (int r = 0; r < reps; r++) {
            const u4* w = (const u4*)((const char*)W + (size_t)(r % regions) * region_bytes);
            float us = 0.f;
            switch (mode) {
                case M_W: us = run<M_W>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
                case M_A: us = run<M_A>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
                case M_MIX: us = run<M_MIX>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
                case M_SPEC: us = run<M_SPEC>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
                case M_PHASE: us = run<M_PHASE>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
                default: us = run<M_ALDS>(w, A, sink, steps, w_tiles_per_wg, wgs, st, e0, e1); break;
            }
            if (r >= 2) t.push_back(us);
        }
        std::sort(t.begin(), t.end());
        const float mn = t.front(), med = t[t.size() / 2];
        const double wb = mode == M_A ? 0.0 : wkb * 1024, ab = mode == M_W ? 0.0 : akb * 1024;
        printf("%-6s %10.2f %10.2f %14.1f %14.1f\n", kNames[mode], mn, med, wb / med / 1e3, (wb + ab) / med / 1e3);
    }
    return 0;
}
