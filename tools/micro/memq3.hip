// Micro-benchmark: a projection's inner loop with the activations PHASED through LDS.
// One 512-thread workgroup per CU.  The K range is cut into chunks; per chunk every wave
// copies its share of the chunk's activation tiles global -> LDS (direct-to-LDS loads, L2
// hits), the workgroup meets at a barrier, then every wave streams its weight tiles of the
// chunk from HBM (non-temporal, one batch of NW tiles always in flight ahead, also across
// chunk boundaries) and reads RD activation tiles per weight tile back from LDS.  So the
// CU's memory pipe carries either L2 traffic or HBM traffic, rarely both.
// Compare with memq2's mixed loop at the same per-CU bytes.
//   hipcc --offload-arch=gfx950 -O3 -o memq3 memq3.hip && ./memq3
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int A_TILES = 512;   // 512 KiB of activations shared by all workgroups
typedef __attribute__((address_space(3))) char lds_char;

template <int NW, int RD>
__global__ __launch_bounds__(512) void k_phase(const u4* W, const u4* A, u4* sink, int chunks, int a_tiles, int steps, int share) {
    extern __shared__ u4 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int run = blockIdx.x / (8 * share), in_run = blockIdx.x % 8;
    const int region = run * 8 + in_run;
    const long per_wave = (long)chunks * steps * NW;
    const u4* wp = W + ((size_t)region * 8 + wave) * per_wave * 64 + lane;
    u4 acc = {0, 0, 0, 0};
    int atile = (blockIdx.x * 37) & (A_TILES - 1);
    u4 w[2][NW];
    long wt = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) w[0][i] = __builtin_nontemporal_load(wp + (size_t)(wt + i) * 64);
    wt += NW;
    const long total_steps = (long)chunks * steps;
    long done = 0;
    for (int c = 0; c < chunks; c++) {
        // ---- activations of this chunk: a_tiles KiB, the waves take tiles round-robin
        for (int t = wave; t < a_tiles; t += 8)
            __builtin_amdgcn_global_load_lds((const void*)(A + (size_t)((atile + t) & (A_TILES - 1)) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)((lds_char*)lds + (size_t)t * 1024), 16, 0, 0);
        atile += a_tiles;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- weights of this chunk
#pragma unroll 1
        for (int s = 0; s < steps; s += 2) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (done + 1 < total_steps) {
#pragma unroll
                    for (int i = 0; i < NW; i++) w[h ^ 1][i] = __builtin_nontemporal_load(wp + (size_t)(wt + i) * 64);
                    wt += NW;
                }
#pragma unroll
                for (int i = 0; i < NW; i++) {
                    acc ^= w[h][i];
#pragma unroll
                    for (int r = 0; r < RD; r++) acc ^= lds[(size_t)(((s + h) * NW * RD + i * RD + r) % a_tiles) * 64 + lane];
                }
                done++;
            }
        }
        __syncthreads();
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

struct Ctx {
    u4 *W, *A, *sink;
    size_t w_bytes;
    hipStream_t st;
    hipEvent_t e0, e1;
};

template <int NW, int RD>
static void bench(Ctx& c, int wgs, int chunks, int a_tiles, int steps, int share, const char* tag) {
    const int reps = 14;
    const size_t per_launch = (size_t)(wgs / share) * 8 * chunks * steps * NW * 1024;
    const int regions = (int)std::max<size_t>(1, std::min<size_t>(10, c.w_bytes / std::max<size_t>(per_launch, 1)));
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        const u4* w = (const u4*)((const char*)c.W + (size_t)(r % regions) * per_launch);
        hipExtLaunchKernelGGL((k_phase<NW, RD>), dim3(wgs), dim3(512), (size_t)a_tiles * 1024, c.st, c.e0, c.e1, 0, w, c.A, c.sink,
                              chunks, a_tiles, steps, share);
        CHECK(hipEventSynchronize(c.e1));
        CHECK(hipGetLastError());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
        if (r >= 2) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const float med = t[t.size() / 2], mn = t.front();
    const double wkb = 8.0 * chunks * steps * NW, akb = (double)chunks * a_tiles;
    printf("%-8s wgs %3d share %d chunks %d  per CU: W %5.0f KiB A %5.0f KiB (LDS reads %d/tile)  min %7.2f med %7.2f us  all %5.1f GB/s/CU  HBM %5.2f TB/s\n",
           tag, wgs, share, chunks, wkb, akb, RD, mn, med, (wkb + akb) * 1.024 / med, (double)per_launch / med / 1e6);
}

int main() {
    Ctx c;
    c.w_bytes = (size_t)2 << 30;
    CHECK(hipMalloc((void**)&c.W, c.w_bytes));
    CHECK(hipMalloc((void**)&c.A, (size_t)A_TILES * 1024));
    CHECK(hipMalloc((void**)&c.sink, (size_t)1024 * 512 * 16));
    CHECK(hipMemset(c.W, 1, c.w_bytes));
    CHECK(hipMemset(c.A, 2, (size_t)A_TILES * 1024));
    CHECK(hipStreamCreate(&c.st));
    CHECK(hipEventCreate(&c.e0));
    CHECK(hipEventCreate(&c.e1));
    CHECK(hipFuncSetAttribute((const void*)k_phase<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k_phase<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k_phase<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)k_phase<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // wo-like: 16 tokens x 32 rows x K 2048: A 128 KiB in one chunk, W 128 KiB shared by 4 workgroups
    bench<2, 2>(c, 256, 1, 128, 8, 4, "wo");
    // qkv-like: 16 tokens x 48 rows: A 128 KiB, W 192 KiB shared by 4 (216 workgroups)
    bench<2, 2>(c, 216, 1, 128, 12, 4, "qkv");
    // gate/up-like, 32 tokens x 96 rows: 2 chunks of 128 KiB, W 384 KiB shared by 2 (236 workgroups)
    bench<6, 4>(c, 236, 2, 128, 4, 2, "gu 96x32");
    // gate/up-like, 64 tokens x 64 rows: 4 chunks of 128 KiB, W 256 KiB unshared (176 workgroups)
    bench<4, 8>(c, 176, 4, 128, 2, 1, "gu 64x64");
    // down-like, 16 tokens x 32 rows x K 5632: 3 chunks of 120 KiB, W 360 KiB shared by 4 (4 x 15 x 6 steps ~ 360)
    bench<2, 2>(c, 256, 3, 120, 8, 4, "down");
    // classifier-like, 64 tokens x 128 rows: 4 chunks of 128 KiB, W 512 KiB unshared (250 workgroups)
    bench<4, 8>(c, 250, 4, 128, 4, 1, "head");
    bench<4, 2>(c, 250, 4, 128, 4, 1, "head rd2");
    return 0;
}
