// runtime.hip -- device selection, the library stream, the allocation/copy shim of
// include/backend.h, scratch buffers, the pinned staging ring, the weight registry,
// RoPE tables and per-kernel event timing.
#include <cstring>
#include <unordered_map>
#include <vector>

#include "nvllm_internal.h"

extern "C" void nvllm_host_rope_table(float* table, float theta, int head_dim, int first_pos, int n_pos);

namespace nvllm {

static Runtime g_rt;
Runtime& rt() { return g_rt; }

// ------------------------------------------------------------------ scratch
struct WsBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
static WsBuf g_ws[WS_COUNT];

void sync_streams() {
    HIP_CHECK(hipStreamSynchronize(g_rt.stream));
    if (g_rt.copy_stream) HIP_CHECK(hipStreamSynchronize(g_rt.copy_stream));
}

void* ws_get(WsSlot slot, size_t bytes) {
    ensure_init();
    WsBuf& b = g_ws[slot];
    if (bytes > b.bytes) {
        if (b.p) {
            sync_streams();
            HIP_CHECK(hipFree(b.p));
        }
        size_t want = bytes + bytes / 4;  // head-room: shapes creep up during prefill
        want = (want + 255) & ~(size_t)255;
        HIP_CHECK(hipMalloc(&b.p, want));
        b.bytes = want;
    }
    return b.p;
}

// ------------------------------------------------------- pinned staging ring
// One page-locked buffer cut into NSTAGE regions used round-robin; an event per
// region says when the H2D copy that read it has finished.
static constexpr int NSTAGE = 8;
static constexpr size_t STAGE_INTS = 1 << 20;  // 4 MiB per region
static int* g_stage_host = nullptr;
static hipEvent_t g_stage_ev[NSTAGE];
static bool g_stage_used[NSTAGE];
static int g_stage_cur = 0;
static size_t g_stage_fill = 0;

static void stage_next_region() {
    if (g_stage_fill > 0) {
        HIP_CHECK(hipEventRecord(g_stage_ev[g_stage_cur], g_rt.stream));
        g_stage_used[g_stage_cur] = true;
    }
    g_stage_cur = (g_stage_cur + 1) % NSTAGE;
    if (g_stage_used[g_stage_cur]) HIP_CHECK(hipEventSynchronize(g_stage_ev[g_stage_cur]));
    g_stage_fill = 0;
}

int* pinned_ints(size_t n) {
    ensure_init();
    if (n > STAGE_INTS) NVLLM_FATAL("staging request of %zu ints exceeds the ring region", n);
    if (g_stage_fill + n > STAGE_INTS) stage_next_region();
    int* p = g_stage_host + (size_t)g_stage_cur * STAGE_INTS + g_stage_fill;
    g_stage_fill += (n + 63) & ~(size_t)63;
    return p;
}

// ----------------------------------------------------------------- registry
static std::unordered_map<const void*, WeightEntry>& registry() {
    static std::unordered_map<const void*, WeightEntry> m;
    return m;
}
const WeightEntry* find_weight(const void* handle) {
    auto& m = registry();
    auto it = m.find(handle);
    return it == m.end() ? nullptr : &it->second;
}
void register_weight(const void* handle, const WeightEntry& e) { registry()[handle] = e; }
void unregister_weight(const void* handle) { registry().erase(handle); }

// -------------------------------------------------------------- RoPE tables
struct RopeTab {
    float theta;
    int head_dim;
    int n_pos;
    float2* dev;
};
static std::vector<RopeTab> g_rope;

const float2* rope_table(float theta, int head_dim, int min_positions) {
    ensure_init();
    RopeTab* hit = nullptr;
    for (auto& t : g_rope)
        if (t.theta == theta && t.head_dim == head_dim) hit = &t;
    if (hit && hit->n_pos >= min_positions) return hit->dev;
    int n = 1024;
    while (n < min_positions) n *= 2;
    const size_t floats = (size_t)n * (head_dim / 2) * 2;
    std::vector<float> host(floats);
    nvllm_host_rope_table(host.data(), theta, head_dim, 0, n);
    float2* dev = nullptr;
    HIP_CHECK(hipMalloc(&dev, floats * sizeof(float)));
    HIP_CHECK(hipMemcpyAsync(dev, host.data(), floats * sizeof(float), hipMemcpyHostToDevice, g_rt.stream));
    HIP_CHECK(hipStreamSynchronize(g_rt.stream));
    if (hit) {
        HIP_CHECK(hipFree(hit->dev));
        hit->dev = dev;
        hit->n_pos = n;
    } else {
        g_rope.push_back({theta, head_dim, n, dev});
    }
    return dev;
}

// ---------------------------------------------------------------- profiling
struct ProfRec {
    int cls;
    hipEvent_t a, b;
    double bytes, flops;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_pending;
static std::vector<hipEvent_t> g_prof_free;
static long long g_prof_launches[PC_COUNT];
static double g_prof_ms[PC_COUNT], g_prof_bytes[PC_COUNT], g_prof_flops[PC_COUNT];

static hipEvent_t prof_event() {
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}

// Timing uses the start/stop events of hipExtLaunchKernelGGL: they carry the
// dispatch's own begin and end timestamps (what a kernel trace reports), not the time
// at which separate event packets happen to be processed before and after it.
static hipEvent_t g_cur_start = nullptr, g_cur_stop = nullptr;
hipEvent_t prof_start_event() { return g_cur_start; }
hipEvent_t prof_stop_event() { return g_cur_stop; }

static int g_prof_suppress = 0;
ProfSuppress::ProfSuppress() { g_prof_suppress++; }
ProfSuppress::~ProfSuppress() { g_prof_suppress--; }

ProfScope::ProfScope(int cls, double bytes, double flops) : slot(-1) {
    if (!g_prof_on || g_prof_suppress > 0) return;
    ProfRec r{cls, prof_event(), prof_event(), bytes, flops};
    slot = (int)g_prof_pending.size();
    g_prof_pending.push_back(r);
    g_cur_start = r.a;   // consumed by the launch inside this scope (one launch per scope)
    g_cur_stop = r.b;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    g_cur_start = g_cur_stop = nullptr;
}

static void prof_drain() {
    if (g_prof_pending.empty()) return;
    HIP_CHECK(hipStreamSynchronize(g_rt.stream));
    for (auto& r : g_prof_pending) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        g_prof_launches[r.cls] += 1;
        g_prof_ms[r.cls] += ms;
        g_prof_bytes[r.cls] += r.bytes;
        g_prof_flops[r.cls] += r.flops;
        g_prof_free.push_back(r.a);
        g_prof_free.push_back(r.b);
    }
    g_prof_pending.clear();
}

static const char* kProfNames[PC_COUNT] = {
    "embed",      "rmsnorm",  "gemm_qkv", "gemm_wo",      "gemm_gateup", "gemm_down", "gemm_head",
    "gemm_other", "rope_kv",  "attn",     "attn_combine", "elementwise", "argmax",    "pack",
    "decode_chain"};

// --------------------------------------------------------------------- init
void ensure_init() {
    // HIP's current device is per host thread: bind every thread that enters the library once
    static thread_local bool bound = false;
    if (g_rt.inited) {
        if (!bound) {
            HIP_CHECK(hipSetDevice(g_rt.device));
            bound = true;
        }
        return;
    }
    HIP_CHECK(hipSetDevice(g_rt.device));
    bound = true;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, g_rt.device));
    g_rt.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_CHECK(hipStreamCreateWithFlags(&g_rt.stream, hipStreamNonBlocking));
    HIP_CHECK(hipHostMalloc((void**)&g_stage_host, NSTAGE * STAGE_INTS * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < NSTAGE; i++) {
        HIP_CHECK(hipEventCreateWithFlags(&g_stage_ev[i], hipEventDisableTiming));
        g_stage_used[i] = false;
    }
    const char* kv = getenv("NVLLM_KV_BYTES");
    if (kv && !g_rt.kv_elem_set && (atoi(kv) == 2 || atoi(kv) == 4)) g_rt.kv_elem_bytes = atoi(kv);
    noalloc_setup(g_rt.device);
    g_rt.inited = true;
}

}  // namespace nvllm

using namespace nvllm;

// ======================================================================= C ABI
extern "C" {

int nvllm_set_device(int ordinal) {
    if (g_rt.inited && ordinal != g_rt.device) {
        fprintf(stderr, "nvllm_set_device: library already initialised on device %d\n", g_rt.device);
        return -1;
    }
    g_rt.device = ordinal;
    return 0;
}

void* nvllm_stream(void) {
    ensure_init();
    return (void*)g_rt.stream;
}

void nvllm_set_logits_mode(int mode) { g_rt.logits_mode = mode == NVLLM_LOGITS_ARGMAX ? NVLLM_LOGITS_ARGMAX : NVLLM_LOGITS_HOST; }
int nvllm_get_logits_mode(void) { return g_rt.logits_mode; }
void nvllm_set_kv_elem_bytes(int bytes) {
    if (bytes != 2 && bytes != 4) NVLLM_FATAL("KV element size must be 2 or 4 bytes, got %d", bytes);
    g_rt.kv_elem_bytes = bytes;
    g_rt.kv_elem_set = true;
}
int nvllm_get_kv_elem_bytes(void) { return g_rt.kv_elem_bytes; }

void nvllm_prof_enable(int on) {
    ensure_init();
    if (!on) prof_drain();
    g_prof_on = on != 0;
}
void nvllm_prof_reset(void) {
    prof_drain();
    for (int i = 0; i < PC_COUNT; i++) {
        g_prof_launches[i] = 0;
        g_prof_ms[i] = g_prof_bytes[i] = g_prof_flops[i] = 0.0;
    }
}
int nvllm_prof_num_classes(void) { return PC_COUNT; }
const char* nvllm_prof_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
void nvllm_prof_get(int cls, long long* launches, double* total_ms, double* bytes, double* flops) {
    prof_drain();
    if (cls < 0 || cls >= PC_COUNT) return;
    if (launches) *launches = g_prof_launches[cls];
    if (total_ms) *total_ms = g_prof_ms[cls];
    if (bytes) *bytes = g_prof_bytes[cls];
    if (flops) *flops = g_prof_flops[cls];
}

// ---- include/backend.h ---------------------------------------------------------
// Page-locked buffers handed out by malloc_run_state are remembered here so that
// device_free() can release them with the right call.
static std::vector<void*>& pinned_list() {
    static std::vector<void*> v;
    return v;
}
void* nvllm_internal_pinned_alloc(size_t bytes) {
    ensure_init();
    void* p = nullptr;
    HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    pinned_list().push_back(p);
    return p;
}

int device_malloc(void** ptr, size_t size) {
    ensure_init();
    if (size == 0) size = 4;
    hipError_t e = hipMalloc(ptr, size);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return -1;
    }
    return 0;
}

void device_free(void* ptr) {
    if (!ptr) return;
    ensure_init();
    sync_streams();
    auto& pl = pinned_list();
    for (size_t i = 0; i < pl.size(); i++)
        if (pl[i] == ptr) {
            pl.erase(pl.begin() + (long)i);
            HIP_CHECK(hipHostFree(ptr));
            return;
        }
    HIP_CHECK(hipFree(ptr));
}

int device_memcpy(void* dst, const void* src, size_t size, DeviceMemcpyKind kind) {
    ensure_init();
    if (size == 0) return 0;
    switch (kind) {
        case DEVICE_TO_HOST:
            HIP_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, g_rt.stream));
            HIP_CHECK(hipStreamSynchronize(g_rt.stream));
            break;
        case HOST_TO_DEVICE:
            // pageable source: the runtime stages it, but only a synchronised stream
            // guarantees the caller may overwrite `src` afterwards
            HIP_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, g_rt.stream));
            HIP_CHECK(hipStreamSynchronize(g_rt.stream));
            break;
        case DEVICE_TO_DEVICE:
            HIP_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToDevice, g_rt.stream));
            break;
        default:
            NVLLM_FATAL("device_memcpy: unknown kind %d", kind);
    }
    return 0;
}

void device_sync(void) {
    ensure_init();
    sync_streams();
}

// ---- include/ops.h life-cycle ---------------------------------------------------
void init_ops(void) { ensure_init(); }

void free_ops(void) {
    if (!g_rt.inited) return;
    sync_streams();
    prof_drain();
    for (int i = 0; i < WS_COUNT; i++) {
        if (g_ws[i].p) HIP_CHECK(hipFree(g_ws[i].p));
        g_ws[i] = WsBuf();
    }
    for (auto& t : g_rope) HIP_CHECK(hipFree(t.dev));
    g_rope.clear();
    gemm_release();
    // the stream, staging ring and weight registry stay: free_kv_cache_manager and
    // free_run_state may still be called after free_ops (reference main.c:455-469)
}

}  // extern "C"
