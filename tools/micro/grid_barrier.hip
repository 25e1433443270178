// Micro-benchmark: cost of a grid-wide barrier among one workgroup per CU (512 threads each),
// built from an agent-scope counter, with and without the release/acquire fences that make
// data written before the barrier visible after it on every XCD.  Every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: counter only, 1: + release/acquire fences, 2: + each WG writes 4 KiB (sc1-less) and reads a neighbour's after
__global__ __launch_bounds__(512) void k_barriers(unsigned* ctr, unsigned* err, int iters, float* data, float* sink) {
    extern __shared__ char pad[];
    (void)pad;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
            // slot by iteration parity: a fast workgroup's next write must not land under a slow reader
            data[(size_t)blockIdx.x * 1024 + (it & 1) * 512 + threadIdx.x] = (float)(it + blockIdx.x);
        }
        if (MODE >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned target = (unsigned)(it + 1) * gridDim.x;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int n = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++n > (1 << 22)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (MODE == 2) {
            const unsigned nb = (blockIdx.x + 37) % gridDim.x;   // another XCD's workgroup
            const float v = data[(size_t)nb * 1024 + (it & 1) * 512 + threadIdx.x];
            if (v != (float)(it + nb)) atomicAdd(err + 1, 1u);
            acc += v;
        }
    }
    if (acc == -1.f) sink[0] = acc;
}

// Two-level form: the workgroups that share an XCD (block id % 8) count on their own counter; the
// last of each group counts on a global one; the last of those publishes the generation, which
// everybody polls.  Data crosses through write-through (sc1) stores and sc1 loads, no fences.
__global__ __launch_bounds__(512) void k_barriers_2level(unsigned* ctr, unsigned* err, int iters, float* data, float* sink) {
    extern __shared__ char pad[];
    (void)pad;
    unsigned* local = ctr + 16 + (blockIdx.x & 7) * 16;    // one 64-byte line per XCD group
    unsigned* global = ctr + 16 + 8 * 16;
    unsigned* gen = ctr + 16 + 9 * 16;
    const unsigned per_group = (gridDim.x + 7 - (blockIdx.x & 7)) / 8;
    float acc = 0.f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(data, 0, (int)(gridDim.x * 4096u), 0x00020000);
    for (int it = 0; it < iters; it++) {
        const float v0 = (float)(it + blockIdx.x);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs, (blockIdx.x * 1024 + (it & 1) * 512 + threadIdx.x) * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(it + 1);
            const unsigned t = __hip_atomic_fetch_add(local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == target * per_group - 1) {
                const unsigned g = __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g == target * 8 - 1) __hip_atomic_store(gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int n = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++n > (1 << 22)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const unsigned nb = (blockIdx.x + 37) % gridDim.x;   // another XCD's workgroup
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (nb * 1024 + (it & 1) * 512 + threadIdx.x) * 4, 0, 16));
        if (v != (float)(it + nb)) atomicAdd(err + 1, 1u);
        acc += v;
    }
    if (acc == -1.f) sink[0] = acc;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    unsigned *ctr, *err;
    float *data, *sink;
    CHECK(hipMalloc(&ctr, 4096));
    CHECK(hipMalloc(&err, 8));
    CHECK(hipMalloc(&data, (size_t)cus * 1024 * 4));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int iters = 2000;
    const size_t shmem = 100 * 1024;   // one workgroup per CU
    auto run = [&](auto kernel, const char* name) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipMemset(ctr, 0, 4096));
            CHECK(hipMemset(err, 0, 8));
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(kernel, dim3(cus), dim3(512), shmem, 0, ctr, err, iters, data, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            unsigned h[2];
            CHECK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
            if (rep) printf("%-28s %d workgroups: %.2f us per barrier (timeouts %u, stale reads %u)\n", name, cus, ms * 1e3 / iters, h[0], h[1]);
        }
    };
    run(k_barriers<0>, "counter only");
    run(k_barriers<1>, "release + acquire");
    run(k_barriers<2>, "release + acquire + data");
    run(k_barriers_2level, "2-level, sc1 data, no fences");
    return 0;
}
