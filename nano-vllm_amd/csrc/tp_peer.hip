// tp_peer.hip -- the tensor-parallel sum all-reduce written for the node itself: every rank (one
// process per GPU) owns one exchange segment in its HBM, exported to the other ranks with
// hipIpcGetMemHandle and mapped by them, so that a rank's kernel stores straight into its peers'
// memory over xGMI (point-to-point links, every GPU one hop from every other) and reads theirs.
//
// What is exchanged: the partial sums [tokens, dim] fp32 after the output projection and after the
// MLP's down projection (reference call sites: src/model.c:385 and :392), 64 times per decode step
// for a 32-layer model, 2 MiB each at 128 rows x 4096 -- latency-bound, so ONE launch per all-reduce
// and a direct two-phase scheme instead of a ring:
//   A. push: rank r cuts its partial sums into `world` slices and stores slice j into rank j's
//      segment (stage[parity][r]); the copy out of the caller's buffer and the transfer are the
//      same stores.  Then it raises flag A(r) in every peer's segment.
//   B. rank j, once all A flags have arrived in ITS segment, sums the `world` staged slices in rank
//      order (every element is summed by exactly one rank in one fixed order: all ranks end up with
//      the same bits, run after run), writes the sum into its result slice `red[parity]` and into
//      its own buffer, and raises flag B(j) in every peer's segment.
//   C. pull: every rank, once all B flags have arrived, reads the other ranks' result slices into
//      its buffer.
// Per link and all-reduce that is 2 x count/world floats (0.5 MiB of 2 MiB at world 8) and two
// flag hops.  Flags are the call's serial number, data and flags are double-buffered by the
// serial's parity, and a rank can be at most one call ahead of any peer (it cannot finish call s
// without every peer's B(s)), so no buffer is overwritten while somebody still reads it.
//
// The kernel that follows every such all-reduce in the forward pass -- x += sum, x times the next norm's gain as
// fp16 tiles, sums of squares per 32 features -- can ride in phases B and C (RESID): every element of the sum
// already passes through a lane there, eight consecutive features at a time, so the sum never goes back to the
// caller's buffer and the stand-alone kernel (and its launch) is not needed.
//
// Memory: the segment is allocated uncached (fine-grained), as RCCL allocates the buffers it
// shares between processes; every access to it carries system scope (sc0 sc1) and data is fenced
// at system scope before a flag is raised.  A wait that does not end (a peer that never launches)
// gives up after 30 s of the constant 100 MHz clock and marks the exchange broken in
// page-locked host memory: the launch still ends, and the next call on the host stops the process
// with a message.
#include "nvllm_internal.h"
#include "resid_tail.h"

#include <cstring>

namespace nvllm {

namespace {

constexpr int PEER_MAX_WORLD = 8;
constexpr unsigned PEER_MAX_FLOATS = 1u << 20;     // per launch: 4 MiB (128 rows x 8192); longer buffers go in pieces
constexpr int PEER_WGS_DEFAULT = 128;              // workgroups per launch (a power of two: arrival tickets wrap cleanly)
constexpr int PEER_THREADS = 256;
constexpr unsigned long long PEER_TICKS_PER_S = 100000000ull;   // the constant clock of s_memrealtime
constexpr int PEER_WAIT_S_DEFAULT = 30;                         // ranks may be far apart at their first call (NVLLM_TP_PEER_WAIT_S overrides)

// segment layout, bytes
constexpr size_t FLAG_LINE = 128;
constexpr size_t FLAGS_BYTES = 2 /*phase*/ * 2 /*parity*/ * PEER_MAX_WORLD * FLAG_LINE;
constexpr size_t STAGE_FLOATS = PEER_MAX_FLOATS + 4 * PEER_MAX_WORLD;   // world slices of ceil(count / world) rounded to 4
constexpr size_t RED_FLOATS = PEER_MAX_FLOATS / 2 + 4;                  // one slice at world >= 2
constexpr size_t STAGE_OFF = FLAGS_BYTES;
constexpr size_t RED_OFF = STAGE_OFF + 2 * STAGE_FLOATS * sizeof(float);
constexpr size_t SEG_BYTES = RED_OFF + 2 * RED_FLOATS * sizeof(float);

constexpr int AUX_SYS = 17;   // raw buffer builtins on gfx950: bit 0 = sc0, bit 4 = sc1; both = system scope

typedef unsigned int pu4_t __attribute__((ext_vector_type(4)));
typedef float pf4_t __attribute__((ext_vector_type(4)));

struct PeerArgs {
    float* buf;
    unsigned count;      // floats, a multiple of 4
    unsigned per;        // floats per slice, a multiple of 4
    int rank, world;
    unsigned serial;     // this call's number (>= 1)
    unsigned char* seg[PEER_MAX_WORLD];   // every rank's segment as mapped here; seg[rank] is this rank's own
    unsigned* tickets;   // this rank's own: two arrival counters of its launches (never reset during a session), then the session's broken mark
    unsigned* status;    // page-locked host word: non-zero = a wait ran out
    unsigned wgs;        // workgroups of the launch (a power of two, the same for every launch of a session)
    unsigned long long wait_ticks;   // how long a wait may last
    // RESID: what the sums are added to (see resid_tail.h); `base` = index of buf[0] in the whole [tokens, D] array
    float* x;
    __half* a16;
    float* ss;
    const float* gain;
    unsigned D, base;
};

__device__ __forceinline__ unsigned* flag_of(unsigned char* seg, int phase, int parity, int src) {
    return reinterpret_cast<unsigned*>(seg + ((size_t)(phase * 2 + parity) * PEER_MAX_WORLD + src) * FLAG_LINE);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t seg_rsrc(unsigned char* seg) {
    return __builtin_amdgcn_make_buffer_rsrc(seg, 0, (int)SEG_BYTES, 0x00020000);
}

// Every wave waits until its stores into the segments have been acknowledged (the segments are
// uncached: an acknowledged store is in memory, the caches hold nothing of it to write back); the
// workgroup that draws the launch's last ticket then raises this rank's flag in every segment, its
// own included, behind one system-scope release.
__device__ __forceinline__ void arrive_and_publish(const PeerArgs& a, int phase, int parity) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.tickets + phase, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((t & (a.wgs - 1)) == a.wgs - 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            for (int j = 0; j < a.world; j++)
                __hip_atomic_store(flag_of(a.seg[j], phase, parity, a.rank), a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Until every rank's flag of this phase has arrived in this rank's own segment (lane r watches rank r).
__device__ __forceinline__ void wait_all(const PeerArgs& a, int phase, int parity) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool mine = lane < a.world;
        const unsigned* word = flag_of(a.seg[a.rank], phase, parity, mine ? lane : 0);
        // a session in which one wait has run out is broken: later waits do not wait again
        const bool broken = __hip_atomic_load(a.tickets + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned v = mine ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : a.serial;
            if (__all((int)(v - a.serial) >= 0)) break;
            __builtin_amdgcn_s_sleep(4);
            if (broken || __builtin_amdgcn_s_memrealtime() - t0 > a.wait_ticks) {   // give up: the launch must end
                if (lane == 0 && !broken) {
                    __hip_atomic_store(a.tickets + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.status, 1u + (unsigned)phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
        }
    }
    // (every load from a segment carries system scope and so never hits a cached line: no invalidate;
    // the barrier keeps the compiler from moving such loads above the wait)
    __syncthreads();
}

// eight sums (features E .. E+7 of the flat [tokens, D] array) -> the residual tail; the four lanes of a 32-feature
// group are active together (slices and pieces are cut at multiples of 32)
__device__ __forceinline__ void peer_resid_8(const PeerArgs& a, unsigned E, pf4_t b0, pf4_t b1) {
    const unsigned t = E / a.D, i = E - t * a.D;
    const unsigned Kb = a.D >> 5;
    float part = resid_tail_8(a.x + (size_t)t * a.D, make_float4(b0[0], b0[1], b0[2], b0[3]), make_float4(b1[0], b1[1], b1[2], b1[3]),
                              a.a16, a.gain, (int)t, (int)i, (int)Kb);
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    if ((threadIdx.x & 3) == 0) a.ss[(size_t)t * Kb + (i >> 5)] = part;
}

template <bool RESID>
__global__ __launch_bounds__(PEER_THREADS) void k_peer_allreduce(PeerArgs a) {
    const int parity = (int)(a.serial & 1u);
    const unsigned tid = blockIdx.x * PEER_THREADS + threadIdx.x;
    const unsigned nthr = a.wgs * PEER_THREADS;
    const unsigned slice_cap = a.per;   // floats reserved per (parity, source rank) in a stage area

    // ---- A. push slice j of this rank's partial sums into rank j's stage area
    for (int j = 0; j < a.world; j++) {
        const unsigned lo = min(a.count, (unsigned)j * a.per), hi = min(a.count, lo + a.per);
        const __amdgpu_buffer_rsrc_t rs = seg_rsrc(a.seg[j]);
        const unsigned base = (unsigned)STAGE_OFF + ((unsigned)parity * (unsigned)STAGE_FLOATS + (unsigned)a.rank * slice_cap) * 4u;
        for (unsigned e = 4 * tid; e < hi - lo; e += 4 * nthr) {
            const pf4_t v = *reinterpret_cast<const pf4_t*>(a.buf + lo + e);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4_t, v), rs, (int)(base + e * 4u), 0, AUX_SYS);
        }
    }
    arrive_and_publish(a, 0, parity);
    wait_all(a, 0, parity);

    // ---- B. this rank's slice: the sum over the ranks in rank order
    {
        const unsigned lo = min(a.count, (unsigned)a.rank * a.per), hi = min(a.count, lo + a.per);
        const __amdgpu_buffer_rsrc_t rs = seg_rsrc(a.seg[a.rank]);
        const unsigned stage0 = (unsigned)STAGE_OFF + ((unsigned)parity * (unsigned)STAGE_FLOATS) * 4u;
        const unsigned red0 = (unsigned)RED_OFF + ((unsigned)parity * (unsigned)RED_FLOATS) * 4u;
        auto sum4 = [&](unsigned e) {
            pf4_t part[PEER_MAX_WORLD];
#pragma unroll
            for (int i = 0; i < PEER_MAX_WORLD; i++)
                if (i < a.world)
                    part[i] = __builtin_bit_cast(pf4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(stage0 + ((unsigned)i * slice_cap + e) * 4u), 0, AUX_SYS));
            pf4_t sum = part[0];
#pragma unroll
            for (int i = 1; i < PEER_MAX_WORLD; i++)
                if (i < a.world) sum += part[i];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4_t, sum), rs, (int)(red0 + e * 4u), 0, AUX_SYS);
            return sum;
        };
        if constexpr (RESID) {
            for (unsigned e = 8 * tid; e < hi - lo; e += 8 * nthr) {
                const pf4_t b0 = sum4(e), b1 = sum4(e + 4);
                peer_resid_8(a, a.base + lo + e, b0, b1);
            }
        } else {
            for (unsigned e = 4 * tid; e < hi - lo; e += 4 * nthr) *reinterpret_cast<pf4_t*>(a.buf + lo + e) = sum4(e);
        }
    }
    arrive_and_publish(a, 1, parity);
    wait_all(a, 1, parity);

    // ---- C. pull the other ranks' result slices
    for (int j = 0; j < a.world; j++) {
        if (j == a.rank) continue;
        const unsigned lo = min(a.count, (unsigned)j * a.per), hi = min(a.count, lo + a.per);
        const __amdgpu_buffer_rsrc_t rs = seg_rsrc(a.seg[j]);
        const unsigned red0 = (unsigned)RED_OFF + ((unsigned)parity * (unsigned)RED_FLOATS) * 4u;
        if constexpr (RESID) {
            for (unsigned e = 8 * tid; e < hi - lo; e += 8 * nthr) {
                const pf4_t b0 = __builtin_bit_cast(pf4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(red0 + e * 4u), 0, AUX_SYS));
                const pf4_t b1 = __builtin_bit_cast(pf4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(red0 + (e + 4) * 4u), 0, AUX_SYS));
                peer_resid_8(a, a.base + lo + e, b0, b1);
            }
        } else {
            for (unsigned e = 4 * tid; e < hi - lo; e += 4 * nthr)
                *reinterpret_cast<pf4_t*>(a.buf + lo + e) =
                    __builtin_bit_cast(pf4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(red0 + e * 4u), 0, AUX_SYS));
        }
    }
}

struct PeerState {
    unsigned char* own = nullptr;        // this rank's segment (exported)
    bool own_in_use = false;             // handed to a session; the next handle request makes a fresh one
    unsigned char* seg[PEER_MAX_WORLD] = {};
    bool opened[PEER_MAX_WORLD] = {};
    unsigned* tickets = nullptr;
    unsigned* status = nullptr;          // page-locked host memory
    unsigned serial = 0;
    int rank = 0, world = 1;
    unsigned wgs = PEER_WGS_DEFAULT;
    int wait_s = PEER_WAIT_S_DEFAULT;
    bool active = false;
    unsigned long long calls = 0, floats = 0;
};
PeerState g_peer;

void peer_check() {
    if (g_peer.status && *reinterpret_cast<volatile unsigned*>(g_peer.status) != 0)
        NVLLM_FATAL("tensor-parallel peer exchange: rank %d waited %d s for a flag of phase %u that never came (a rank missing, or ranks that "
                    "do not make the same calls)", g_peer.rank, g_peer.wait_s, *g_peer.status - 1u);
}

struct PeerResid {   // what a RESID call adds its sums to
    float* x;
    __half* a16;
    float* ss;
    const float* gain;
    int D;
};

void peer_exchange(float* buf, size_t count, void* stream, const PeerResid* resid) {
    PeerState& s = g_peer;
    peer_check();
    s.calls++;
    s.floats += count;
    if (s.world == 1 || count == 0) return;
    const size_t grain = resid ? 32 : 4;   // slices end at whole granules (RESID: whole 32-feature groups)
    if (count % grain != 0) NVLLM_FATAL("tensor-parallel peer exchange: %zu floats is not a multiple of %zu", count, grain);
    for (size_t done = 0; done < count;) {
        const size_t n = count - done < PEER_MAX_FLOATS ? count - done : PEER_MAX_FLOATS;
        PeerArgs a;
        a.buf = buf + done;
        a.count = (unsigned)n;
        a.per = (unsigned)(((n + s.world - 1) / s.world + grain - 1) / grain * grain);
        a.rank = s.rank;
        a.world = s.world;
        a.serial = ++s.serial;
        for (int j = 0; j < PEER_MAX_WORLD; j++) a.seg[j] = j < s.world ? s.seg[j] : nullptr;
        a.tickets = s.tickets;
        a.status = s.status;
        a.wgs = s.wgs;
        a.wait_ticks = (unsigned long long)s.wait_s * PEER_TICKS_PER_S;
        a.x = resid ? resid->x : nullptr;
        a.a16 = resid ? resid->a16 : nullptr;
        a.ss = resid ? resid->ss : nullptr;
        a.gain = resid ? resid->gain : nullptr;
        a.D = resid ? (unsigned)resid->D : 1u;
        a.base = (unsigned)done;
        if (resid) hipLaunchKernelGGL(k_peer_allreduce<true>, dim3(s.wgs), dim3(PEER_THREADS), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(k_peer_allreduce<false>, dim3(s.wgs), dim3(PEER_THREADS), 0, (hipStream_t)stream, a);
        HIP_CHECK(hipGetLastError());
        done += n;
    }
}

void peer_allreduce(void*, float* buf, size_t count, void* stream) { peer_exchange(buf, count, stream, nullptr); }

}  // namespace

// (tp.hip) The all-reduce of the partial sums at `partial` ([T, D], this rank's) with the residual tail in its last
// phases: x += sum, a16 = (x * gain) tiles, ss = sums of squares per 32 features.  False: no peer session with more
// than one rank (or a width the tail does not take) -- the caller all-reduces and runs the tail as its own kernel.
bool tp_peer_allreduce_resid(float* partial, float* x, __half* a16, float* ss, const float* gain, int T, int D) {
    PeerState& s = g_peer;
    static const int fused_env = getenv("NVLLM_TP_PEER_FUSED") ? atoi(getenv("NVLLM_TP_PEER_FUSED")) : 1;
    if (!s.active || s.world == 1 || !fused_env || D % 32 != 0 || T <= 0) return false;
    if ((size_t)T * D >= (1ull << 32)) return false;
    const PeerResid r{x, a16, ss, gain, D};
    peer_exchange(partial, (size_t)T * D, (void*)rt().stream, &r);
    return true;
}

// (tp.hip) the session ends: unmap the peers' segments; this rank's own stays allocated (a peer may still be reading it)
void tp_peer_finalize() {
    PeerState& s = g_peer;
    if (!s.active) return;
    peer_check();
    for (int j = 0; j < s.world; j++)
        if (s.opened[j]) {
            HIP_CHECK(hipIpcCloseMemHandle(s.seg[j]));
            s.opened[j] = false;
        }
    s.active = false;
}

}  // namespace nvllm

using namespace nvllm;

extern "C" {

int nvllm_tp_peer_handle(void* handle) {
    if (!handle) return -1;
    ensure_init();
    PeerState& s = g_peer;
    if (s.active) {
        fprintf(stderr, "nvllm: the peer exchange is in use; call nvllm_tp_finalize first\n");
        return -1;
    }
    if (s.own && s.own_in_use) {
        // The last session's segment is retired, not freed: this rank's last all-reduce ended when every peer had
        // raised its second flag, but a peer may still be reading this rank's result slice, and nothing here tells
        // when it has finished.  12.6 MiB stay allocated per past session until the process ends.
        s.own = nullptr;
    }
    if (!s.own) {
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&s.own), SEG_BYTES, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "nvllm: cannot allocate %zu bytes of uncached device memory for the peer exchange\n", SEG_BYTES);
            return -1;
        }
        HIP_CHECK(hipMemset(s.own, 0, FLAGS_BYTES));   // serial numbers start at 1: no flag is up
        HIP_CHECK(hipDeviceSynchronize());
        s.own_in_use = false;
    }
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == NVLLM_TP_PEER_HANDLE_BYTES, "hipIpcMemHandle_t size");
    if (hipIpcGetMemHandle(&h, s.own) != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "nvllm: hipIpcGetMemHandle refused the exchange segment\n");
        return -1;
    }
    memcpy(handle, &h, sizeof h);
    return 0;
}

int nvllm_tp_init_peer(int rank, int world, const void* handles) {
    PeerState& s = g_peer;
    if (!handles || world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world || !s.own || s.own_in_use || s.active) {
        fprintf(stderr, "nvllm: nvllm_tp_init_peer(%d, %d): needs 1..%d ranks and this process's handle from nvllm_tp_peer_handle "
                        "among the %d handles\n", rank, world, PEER_MAX_WORLD, world);
        return -1;
    }
    ensure_init();
    const unsigned char* hs = static_cast<const unsigned char*>(handles);
    for (int j = 0; j < world; j++) {
        s.opened[j] = false;
        if (j == rank) {
            s.seg[j] = s.own;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, hs + (size_t)j * sizeof h, sizeof h);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "nvllm: hipIpcOpenMemHandle failed for rank %d's exchange segment\n", j);
            for (int i = 0; i < j; i++)
                if (s.opened[i]) {
                    (void)hipIpcCloseMemHandle(s.seg[i]);
                    s.opened[i] = false;
                }
            return -1;
        }
        s.seg[j] = static_cast<unsigned char*>(p);
        s.opened[j] = true;
    }
    if (!s.tickets) {
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&s.tickets), 3 * sizeof(unsigned)));
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.status), sizeof(unsigned), hipHostMallocDefault));
        HIP_CHECK(hipDeviceSynchronize());
    }
    *s.status = 0;
    HIP_CHECK(hipMemset(s.tickets, 0, 3 * sizeof(unsigned)));
    HIP_CHECK(hipDeviceSynchronize());
    s.wgs = PEER_WGS_DEFAULT;
    if (const char* e = getenv("NVLLM_TP_PEER_WGS")) {   // tuning knob: 32, 64, 128 or 256
        const int v = atoi(e);
        if (v == 32 || v == 64 || v == 128 || v == 256) s.wgs = (unsigned)v;
    }
    s.wait_s = PEER_WAIT_S_DEFAULT;
    if (const char* e = getenv("NVLLM_TP_PEER_WAIT_S")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 600) s.wait_s = v;
    }
    s.serial = 0;
    s.rank = rank;
    s.world = world;
    s.calls = s.floats = 0;
    if (nvllm_tp_init_custom(rank, world, peer_allreduce, nullptr) != 0) {
        for (int j = 0; j < world; j++)
            if (s.opened[j]) {
                (void)hipIpcCloseMemHandle(s.seg[j]);
                s.opened[j] = false;
            }
        return -1;
    }
    s.own_in_use = true;
    s.active = true;
    return 0;
}

void nvllm_tp_peer_stats(unsigned long long* calls, unsigned long long* floats) {
    if (calls) *calls = g_peer.calls;
    if (floats) *floats = g_peer.floats;
}

int nvllm_tp_allreduce(float* device_buf, size_t count) {
    if (!tp().active) return -1;
    tp_allreduce(device_buf, count);
    return 0;
}

}  // extern "C"
