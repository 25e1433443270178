// Micro-benchmark: the byte model of a decode projection's inner loop.  One 512-thread
// workgroup per CU; every wave keeps two batches of (NW weight tiles + NA activation tiles)
// in flight, 1 KiB per wave instruction.  Weights are a once-read HBM stream (non-temporal),
// activations a 512 KiB buffer that every workgroup re-reads (L2).  Sweeps:
//   ratio   (NW, NA) combinations at 256 workgroups: how the cost of a step depends on the mix
//   share   G workgroups with equal blockIdx.x % 8 (one XCD under round-robin placement) stream
//           the SAME weights, as token groups of one weight tile do
//   wgs     fewer workgroups than CUs: what one CU takes from HBM when HBM is not the limit
//   hipcc --offload-arch=gfx950 -O3 -o memq2 memq2.hip && ./memq2
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int A_TILES = 512;

template <int NW, int NA>
struct Batch {
    u4 w[NW > 0 ? NW : 1];
    u4 a[NA > 0 ? NA : 1];
};
template <int NW, int NA>
__device__ __forceinline__ void issue(Batch<NW, NA>& b, const u4* wp, const u4* ap, long wtile, int atile, int lane) {
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

// Each wave: `steps` steps; its weight tiles are consecutive in its own range of the
// workgroup's region.  wg_of maps the block to the weight region it streams.
template <int NW, int NA>
__global__ __launch_bounds__(512) void k_mix(const u4* W, const u4* A, u4* sink, int steps, int share) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blocks b, b+8, ..., b+8*(share-1) of each run of 8*share blocks stream the same weights
    const int run = blockIdx.x / (8 * share), in_run = blockIdx.x % 8;
    const int region = run * 8 + in_run;
    const long per_wave = (long)steps * (NW > 0 ? NW : 1);
    const u4* wp = W + ((size_t)region * 8 + wave) * per_wave * 64;
    u4 acc = {0, 0, 0, 0};
    int atile = (blockIdx.x * 5 + wave * 61) & (A_TILES - 1);
    long wtile = 0;
    Batch<NW, NA> b0, b1;
    issue(b0, wp, A, wtile, atile, lane);
    wtile += NW; atile += NA;
    int s = 0;
    for (; s + 2 <= steps; s += 2) {
        issue(b1, wp, A, wtile, atile, lane);
        wtile += NW; atile += NA;
        consume(b0, acc);
        if (s + 2 < steps) {
            issue(b0, wp, A, wtile, atile, lane);
            wtile += NW; atile += NA;
        }
        consume(b1, acc);
    }
    if (s < steps) consume(b0, acc);
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

struct Ctx {
    u4 *W, *A, *sink;
    size_t w_bytes;
    hipStream_t st;
    hipEvent_t e0, e1;
};

template <int NW, int NA>
static void bench(Ctx& c, int wgs, int steps, int share, const char* tag) {
    const int reps = 14;
    const size_t per_launch = (size_t)(wgs / share) * 8 * steps * (NW > 0 ? NW : 1) * 1024;
    const int regions = (int)std::max<size_t>(1, std::min<size_t>(10, c.w_bytes / std::max<size_t>(per_launch, 1)));
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        const u4* w = (const u4*)((const char*)c.W + (size_t)(r % regions) * per_launch);
        hipExtLaunchKernelGGL((k_mix<NW, NA>), dim3(wgs), dim3(512), 0, c.st, c.e0, c.e1, 0, w, c.A, c.sink, steps, share);
        CHECK(hipEventSynchronize(c.e1));
        CHECK(hipGetLastError());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
        if (r >= 2) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const float med = t[t.size() / 2], mn = t.front();
    const double wkb = 8.0 * steps * NW, akb = 8.0 * steps * NA;   // per CU
    printf("%-6s wgs %3d share %d  NW %d NA %d  per CU: W %5.0f KiB A %5.0f KiB  min %7.2f med %7.2f us  W %5.1f all %5.1f GB/s/CU  HBM %5.2f TB/s  ns/step %6.0f\n",
           tag, wgs, share, NW, NA, wkb, akb, mn, med, wkb * 1.024 / med, (wkb + akb) * 1.024 / med,
           (double)per_launch / med / 1e6, med * 1e3 / steps);
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
    // ---- ratio sweep, 256 workgroups, about 256 KiB of weights per CU
    bench<1, 0>(c, 256, 32, 1, "ratio");
    bench<2, 0>(c, 256, 16, 1, "ratio");
    bench<4, 0>(c, 256, 8, 1, "ratio");
    bench<8, 0>(c, 256, 4, 1, "ratio");
    bench<4, 1>(c, 256, 8, 1, "ratio");
    bench<4, 2>(c, 256, 8, 1, "ratio");
    bench<4, 4>(c, 256, 8, 1, "ratio");
    bench<4, 8>(c, 256, 8, 1, "ratio");
    bench<2, 8>(c, 256, 16, 1, "ratio");
    bench<1, 8>(c, 256, 32, 1, "ratio");
    bench<8, 4>(c, 256, 4, 1, "ratio");
    bench<8, 2>(c, 256, 4, 1, "ratio");
    bench<0, 8>(c, 256, 16, 1, "ratio");
    bench<0, 4>(c, 256, 32, 1, "ratio");
    bench<0, 2>(c, 256, 64, 1, "ratio");
    // longer runs of the main mixes (ramp share shrinks)
    bench<4, 0>(c, 256, 32, 1, "long");
    bench<4, 4>(c, 256, 32, 1, "long");
    bench<4, 8>(c, 256, 32, 1, "long");
    bench<2, 8>(c, 256, 32, 1, "long");
    // ---- weights shared by 2 / 4 workgroups of one XCD
    bench<4, 0>(c, 256, 16, 2, "share");
    bench<4, 0>(c, 256, 16, 4, "share");
    bench<4, 4>(c, 256, 16, 2, "share");
    bench<4, 4>(c, 256, 16, 4, "share");
    bench<4, 2>(c, 256, 16, 2, "share");
    bench<4, 2>(c, 256, 16, 4, "share");
    // ---- fewer workgroups than CUs
    bench<4, 0>(c, 32, 32, 1, "wgs");
    bench<4, 0>(c, 64, 32, 1, "wgs");
    bench<4, 0>(c, 128, 32, 1, "wgs");
    bench<8, 0>(c, 64, 16, 1, "wgs");
    bench<4, 8>(c, 64, 32, 1, "wgs");
    bench<4, 4>(c, 64, 32, 1, "wgs");
    bench<4, 4>(c, 128, 32, 1, "wgs");
    return 0;
}
