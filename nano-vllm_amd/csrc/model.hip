// model.hip -- include/model.h on the GPU: the model file goes to HBM as fp16 in
// tile order, the per-call workspace is allocated once, and transformer_batch
// enqueues the fused per-layer kernel sequence and hands back logits / arg-max.
#include <algorithm>
#include <cstring>
#include <vector>

#include "nvllm_internal.h"
#include "step_arrays.h"

// The driver owns these (reference src/main.c:51-55); engine.c holds weak
// fall-back definitions for when the library runs without that driver.
extern "C" {
extern int g_paged_mode;
extern KVCacheManager g_kv_manager;
void* nvllm_internal_pinned_alloc(size_t bytes);
}

namespace nvllm {

struct LayerPriv {
    const __half* qkv;     // tiles of [QD + 2*KD, D]: wq rows, then wk, then wv
    const __half* wo;      // tiles of [D, QD]
    const __half* gateup;  // tiles of [2F, D], 16-row blocks alternating gate, up
    const __half* down;    // tiles of [D, F]
    const float* rms_att;
    const float* rms_ffn;
};

struct ModelPriv {
    Config cfg;
    NvllmTpShard sh;   // the slice of heads / hidden columns held here (everything when world == 1)
    void* arena = nullptr;
    size_t arena_bytes = 0;
    const __half* embed = nullptr;   // [V, D] row-major fp16
    const __half* head = nullptr;    // tiles of [V, D]
    const float* rms_final = nullptr;
    std::vector<LayerPriv> layers;
    std::vector<const void*> handles;
    // An fp32 file with values that fp16 cannot hold stays fp32: one row-major HBM buffer per
    // tensor (the public Weights pointers are those buffers, as in the reference) and the
    // forward pass runs the fp32 operators one by one.
    bool f32 = false;
    std::vector<void*> f32_bufs;
    ChainLayer* chain_layers = nullptr;   // per-layer weight pointers in HBM, for the one-launch decode step (chain.hip)
};

// Does fp16 hold this fp32 value exactly?  (Zero, or a normal/subnormal fp16 value; no NaN.)
static inline bool fp16_exact(uint32_t bits) {
    const uint32_t e = (bits >> 23) & 0xffu, man = bits & 0x7fffffu;
    if (e == 0) return man == 0;            // zero; fp32 subnormals are far below fp16's range
    if (e == 255) return man == 0;          // infinity
    const int E = (int)e - 127;
    if (E > 15) return false;
    if (E >= -14) return (man & 0x1fffu) == 0;
    if (E < -24) return false;
    const uint32_t full = man | 0x800000u;  // 1.m as a 24-bit integer: value = full * 2^(E-23), grid 2^-24
    return (full & ((1u << (-(E + 1))) - 1u)) == 0;
}

// Scan the matrices of an fp32 model file (everything but the norm vectors, which stay fp32
// anyway): true when every value survives a round trip through fp16.
static bool f32_file_fits_fp16(FILE* f, long start, const std::vector<std::pair<long long, bool>>& tensors) {
    const size_t CH = 1 << 20;
    std::vector<uint32_t> buf(CH);
    bool ok = true;
    fseek(f, start, SEEK_SET);
    for (const auto& t : tensors) {
        long long left = t.first;
        if (!t.second) {   // a norm vector: skip
            fseek(f, (long)(left * 4), SEEK_CUR);
            continue;
        }
        while (left > 0 && ok) {
            const size_t n = (size_t)std::min<long long>(left, (long long)CH);
            if (fread(buf.data(), 4, n, f) != n) { ok = false; break; }
            for (size_t i = 0; i < n; i++)
                if (!fp16_exact(buf[i])) { ok = false; break; }
            left -= (long long)n;
        }
        if (!ok) break;
    }
    fseek(f, start, SEEK_SET);
    return ok;
}

// What sample() and nvllm_last_argmax() may look at: the pass most recently COLLECTED.  A pass's view
// is put together when it is enqueued and becomes current only when it is collected, all fields at once,
// so between nvllm_forward_enqueue(n + 1) and its collect the callers still see pass n, consistently.
struct PassView {
    const int* argmax = nullptr;        // page-locked indices, one per output row
    int outputs = 0;
    const float* logits_host = nullptr; // RunState.logits of the pass: rows there NAME its outputs
    const float* logits_dev = nullptr;  // its logits in HBM -- a workspace the NEXT enqueued pass overwrites
    int vocab = 0;
    long ticket = -1;
    // tensor parallelism: the pass left only this rank's vocabulary slice (arg-max hand-off);
    // sample() with a temperature gathers the full rows when it needs them
    const float* tp_local = nullptr;
    bool tp_full_valid = true;
    int tp_slice = 0;
};
// Arg-max indices of the passes in flight: a small ring of page-locked arrays, each with the
// event recorded behind the copy that fills it.  A pass enqueued with nvllm_forward_enqueue owns
// slot (ticket % PASS_RING) until it is collected; transformer_batch collects at once.
constexpr int PASS_RING = 4;
struct PassSlot {
    int* host = nullptr;
    int cap = 0;
    hipEvent_t ev = nullptr;
    long ticket = -1;
    bool chained = false;   // the pass ran its layers as one launch: host[outputs] is that launch's status word
    PassView view;
};
static PassSlot g_pass[PASS_RING];
// Logits rows to the host (8 MB per 64-row pass at a 32,000-word vocabulary): the copies of a pass run on a
// stream of their own, behind an event that follows the pass's kernels, so that the next pass's layers do
// not queue behind 0.16 ms of PCIe traffic.  What the copies read (the logits and index workspaces) is
// overwritten by the next pass's classifier, which therefore waits for `g_rows_copied` first.
static hipEvent_t g_kernels_done = nullptr;
static hipEvent_t g_rows_copied = nullptr;   // the event of the latest such copy that nobody has waited for yet
static void wait_for_row_copies() {
    if (!g_rows_copied) return;
    HIP_CHECK(hipStreamWaitEvent(rt().stream, g_rows_copied, 0));
    g_rows_copied = nullptr;
}
static long g_next_ticket = 0;
static PassView g_seen;                // the most recently collected pass

static void read_exact(FILE* f, void* dst, size_t bytes, const char* what) {
    if (fread(dst, 1, bytes, f) != bytes) {
        fprintf(stderr, "Failed to read %s from file (expected %zu bytes)\n", what, bytes);
        exit(1);
    }
}

// every rank's slice [rows, slice] -> full logits rows [rows, V] on every rank
static void tp_gather_logits(float* full, const float* local, int rows, int slice, int V) {
    const int W = tp().world;
    float* parts = (float*)ws_get(WS_TP_GATHER, (size_t)W * rows * slice * sizeof(float));
    tp_allgather(local, parts, (size_t)rows * slice);
    launch_logits_unshard(full, parts, W, rows, slice, V);
}

// End of a forward pass: the copies that carry its results to the host are enqueued behind its
// kernels and an event behind them; returns the pass's ticket.  `view` arrives with the tensor-parallel
// fields set.
// The slot the next finish_pass will use, with room for B indices: a pass whose arg-max kernel writes the
// indices into the slot's page-locked array itself needs the address before that kernel is launched.
static PassSlot& next_slot(int B) {
    PassSlot& slot = g_pass[g_next_ticket % PASS_RING];
    if (slot.ticket >= 0) NVLLM_FATAL("forward pass %ld was never collected: at most %d passes may be in flight", slot.ticket, PASS_RING);
    // the slot's previous pass (PASS_RING passes ago) ends here: nobody may go on reading its indices
    if (g_seen.argmax == slot.host) g_seen = PassView{};
    if (B + 1 > slot.cap) {
        if (slot.host) HIP_CHECK(hipHostFree(slot.host));
        slot.cap = B + 256;
        HIP_CHECK(hipHostMalloc((void**)&slot.host, (size_t)slot.cap * sizeof(int), hipHostMallocDefault));
    }
    if (!slot.ev) HIP_CHECK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
    return slot;
}

// `host_written`: the pass's last kernel has stored the indices into next_slot(B).host itself.
static long finish_pass(RunState* s, float* d_logits, int* d_argmax, int B, int V, PassView view, bool chained = false,
                        bool host_written = false) {
    PassSlot& slot = next_slot(B);
    const long ticket = g_next_ticket++;
    hipStream_t stream = rt().stream;
    if (rt().logits_mode == NVLLM_LOGITS_HOST) {
        if (!rt().copy_stream) HIP_CHECK(hipStreamCreateWithFlags(&rt().copy_stream, hipStreamNonBlocking));
        if (!g_kernels_done) HIP_CHECK(hipEventCreateWithFlags(&g_kernels_done, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(g_kernels_done, stream));
        stream = rt().copy_stream;
        HIP_CHECK(hipStreamWaitEvent(stream, g_kernels_done, 0));
    }
    // (with the status word of a one-launch step, which sits right behind the indices)
    // (a copy operation between two passes costs the stream ~10 us: the copy kernel and the switch around it)
    if (!host_written)
        HIP_CHECK(hipMemcpyAsync(slot.host, d_argmax, (size_t)(B + (chained ? 1 : 0)) * sizeof(int), hipMemcpyDeviceToHost, stream));
    slot.chained = chained;
    if (rt().logits_mode == NVLLM_LOGITS_HOST)
        HIP_CHECK(hipMemcpyAsync(s->logits, d_logits, (size_t)B * V * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipEventRecord(slot.ev, stream));
    if (rt().logits_mode == NVLLM_LOGITS_HOST) g_rows_copied = slot.ev;
    slot.ticket = ticket;
    view.argmax = slot.host;
    view.outputs = B;
    view.logits_host = s->logits;
    view.logits_dev = d_logits;
    view.vocab = V;
    view.ticket = ticket;
    slot.view = view;
    return ticket;
}

// Wait for a pass's results and make it the pass sample() and nvllm_last_argmax() answer for; its
// indices stay readable at the returned address until the slot's next use (PASS_RING passes later).
static const int* collect_pass(long ticket, int* num_outputs) {
    PassSlot& slot = g_pass[ticket % PASS_RING];
    if (ticket < 0 || slot.ticket != ticket) NVLLM_FATAL("forward pass %ld is not in flight", ticket);
    HIP_CHECK(hipEventSynchronize(slot.ev));
    const int B = slot.view.outputs;
    if (slot.chained && slot.host[B] != 1) {
        const int st = slot.host[B];
        NVLLM_FATAL("the one-launch decode step did not complete (status %d: %s); set NVLLM_DECODE_CHAIN=0 to run the layers as separate kernels",
                    st, (st & 4) ? "a workgroup waited for a hand-off longer than 0.2 s"
                                 : (st & 2) ? "not all of its workgroups were resident at once -- is another process using this GPU?"
                                            : "it never started");
    }
    slot.ticket = -1;
    if (num_outputs) *num_outputs = B;
    g_seen = slot.view;
    return slot.host;
}

// ---- fp32 files whose values fp16 cannot hold (e.g. a trained llama2.c stories15M.bin) ----
// Loaded as the reference does (model.c:87-108, 230-279): one fp32 row-major buffer per
// tensor, in file order; the classifier aliases the embedding when the file shares them.
static void load_model_f32(Weights* w, Config* p, FILE* f, bool shared) {
    if (tp().world > 1) {
        fprintf(stderr, "fp32 model files with values outside fp16 are not split over tensor-parallel ranks; convert the file to fp16\n");
        exit(1);
    }
    const size_t D = p->dim, F = p->hidden_dim, V = p->vocab_size;
    const size_t QD = (size_t)p->n_heads * p->head_dim, KD = (size_t)p->n_kv_heads * p->head_dim;
    const int L = p->n_layers;
    ModelPriv* m = new ModelPriv();
    m->cfg = *p;
    nvllm_tp_shard(p, 0, 1, &m->sh);
    m->f32 = true;
    const size_t biggest = std::max(std::max(V * D, D * F), D * QD);
    float* host = (float*)malloc(biggest * sizeof(float));
    if (!host) NVLLM_FATAL("no memory for the %zu-byte load buffer", biggest * sizeof(float));
    float dbg0 = 0.f, dbg100 = 0.f;
    bool first = true;
    auto load = [&](size_t n, const char* what) -> float* {   // model.c:87-108
        read_exact(f, host, n * sizeof(float), what);
        if (first) {
            dbg0 = host[0];
            dbg100 = host[std::min<size_t>(100, n - 1)];
            first = false;
        }
        void* d = nullptr;
        if (device_malloc(&d, n * sizeof(float)) != 0) NVLLM_FATAL("cannot allocate %zu bytes of HBM for %s", n * sizeof(float), what);
        device_memcpy(d, host, n * sizeof(float), HOST_TO_DEVICE);   // synchronous: `host` is reused
        m->f32_bufs.push_back(d);
        return (float*)d;
    };
    w->layers = (LayerWeights*)malloc((size_t)L * sizeof(LayerWeights));
    w->token_embedding_table = load(V * D, "token_embedding_table");
    for (int l = 0; l < L; l++) w->layers[l].rms_att_weight = load(D, "rms_att_weight");
    for (int l = 0; l < L; l++) w->layers[l].wq = load(D * QD, "wq");
    for (int l = 0; l < L; l++) w->layers[l].wk = load(D * KD, "wk");
    for (int l = 0; l < L; l++) w->layers[l].wv = load(D * KD, "wv");
    for (int l = 0; l < L; l++) w->layers[l].wo = load(QD * D, "wo");
    for (int l = 0; l < L; l++) w->layers[l].rms_ffn_weight = load(D, "rms_ffn_weight");
    for (int l = 0; l < L; l++) w->layers[l].w_gate = load(D * F, "w_gate");
    for (int l = 0; l < L; l++) w->layers[l].w_down = load(F * D, "w_down");   // file order: down before up (model.c:262-269)
    for (int l = 0; l < L; l++) w->layers[l].w_up = load(D * F, "w_up");
    w->rms_final_weight = load(D, "rms_final_weight");
    if (shared) {
        w->lm_head = w->token_embedding_table;
        printf("Weights shared: lm_head -> token_embedding_table\n");
    } else {
        w->lm_head = load(V * D, "lm_head");
    }
    fclose(f);
    free(host);
    w->_data_buffer = m;
    printf("Model loaded.\n");
    printf("DEBUG: Embed[0]: %f\n", dbg0);
    printf("DEBUG: Embed[100]: %f\n", dbg100);
}

// The reference's layer loop (model.c:352-418) on fp32 weights, one operator launch per
// reference call; leaves the logits of the output rows in d_logits.
static void forward_f32(const StepArrays& st, int T, int B, Config* p, RunState* s, Weights* w, const KvPoolView& pool,
                        const float2* rope, float* d_logits) {
    const int D = p->dim, F = p->hidden_dim, hd = p->head_dim, H = p->n_heads, KVH = p->n_kv_heads;
    const int QD = H * hd, KD = KVH * hd, V = p->vocab_size;
    const float eps = 1e-5f;
    launch_gather_rows_f32(s->x, w->token_embedding_table, st.d_tokens, T, D);
    for (int l = 0; l < p->n_layers; l++) {
        const LayerWeights& lw = w->layers[l];
        launch_rmsnorm_f32(s->xb, s->x, lw.rms_att_weight, D, T, eps);
        launch_matmul_f32w(s->q, s->xb, lw.wq, T, D, QD);
        launch_matmul_f32w(s->k, s->xb, lw.wk, T, D, KD);
        launch_matmul_f32w(s->v, s->xb, lw.wv, T, D, KD);
        launch_rope(s->q, s->k, st.d_pos, 0, rope, T, H, KVH, hd);
        launch_kv_write(s->k, s->v, st.d_pos, 0, st.d_seq, st.d_tbl_off, st.d_tables, pool, l, st.write_limit, KVH, hd);
        AttnArgs at{};
        at.q = s->q;
        at.out_f32 = s->xb2;
        at.pool = pool;
        at.layer = l;
        at.tables = st.d_tables;
        at.T = T; at.H = H; at.KVH = KVH; at.hd = hd;
        at.max_pos = st.max_pos;
        at.runs = st.d_runs;
        at.n_runs = st.n_runs;
        at.kv_positions = st.kv_positions;
        launch_paged_attention(at);
        launch_matmul_f32w(s->xb, s->xb2, lw.wo, T, QD, D);
        launch_accum(s->x, s->xb, (long)T * D);
        launch_rmsnorm_f32(s->xb, s->x, lw.rms_ffn_weight, D, T, eps);
        launch_matmul_f32w(s->hb, s->xb, lw.w_gate, T, D, F);
        launch_matmul_f32w(s->hb2, s->xb, lw.w_up, T, D, F);
        launch_swiglu(s->hb, s->hb, s->hb2, (long)T * F);
        launch_matmul_f32w(s->xb, s->hb, lw.w_down, T, F, D);
        launch_accum(s->x, s->xb, (long)T * D);
    }
    launch_rmsnorm_f32(s->x, s->x, w->rms_final_weight, D, T, eps);
    launch_gather_rows_f32(s->xb, s->x, st.d_out_idx, B, D);
    launch_matmul_f32w(d_logits, s->xb, w->lm_head, B, D, V);
}

}  // namespace nvllm

using namespace nvllm;

extern "C" {

// ------------------------------------------------------------- load_model
// File format and tensor order: reference src/model.c:118-279.
void load_model(Weights* w, Config* p, const char* checkpoint_path) {
    ensure_init();
    FILE* f = fopen(checkpoint_path, "rb");
    if (!f) {
        fprintf(stderr, "Failed to open model file: %s\n", checkpoint_path);
        exit(1);
    }
    if (fread(p, sizeof(int), 7, f) != 7) exit(1);
    const long header = ftell(f);
    fseek(f, 0, SEEK_END);
    const long long payload = ftell(f) - header;
    fseek(f, header, SEEK_SET);
    if (p->dim <= 0 || p->n_heads <= 0 || p->n_kv_heads <= 0 || p->dim % p->n_heads != 0 ||
        p->n_heads % p->n_kv_heads != 0) {
        fprintf(stderr, "model header is inconsistent (dim %d, heads %d/%d)\n", p->dim, p->n_heads, p->n_kv_heads);
        exit(1);
    }
    p->head_dim = p->dim / p->n_heads;  // model.c:144
    p->rope_theta = 10000.0f;           // model.c:146

    // the reference's report of what it found (model.c:148-157)
    printf("Model Config:\n");
    printf("  dim: %d\n", p->dim);
    printf("  n_layers: %d\n", p->n_layers);
    printf("  n_heads: %d\n", p->n_heads);
    printf("  n_kv_heads: %d\n", p->n_kv_heads);
    printf("  vocab_size: %d\n", p->vocab_size);
    printf("  max_seq_len: %d\n", p->max_seq_len);
    printf("  hidden_dim: %d\n", p->hidden_dim);
    printf("  head_dim: %d\n", p->head_dim);
    printf("  rope_theta: %f\n", p->rope_theta);

    const long long D = p->dim, F = p->hidden_dim, L = p->n_layers, V = p->vocab_size;
    const long long QD = (long long)p->n_heads * p->head_dim, KD = (long long)p->n_kv_heads * p->head_dim;
    const long long body = V * D + L * (2 * D + D * QD + 2 * D * KD + QD * D + 3 * D * F) + D;
    const long long headw = V * D;
    const long long rope_tail = (long long)p->max_seq_len * p->head_dim * 4;
    printf("File data size: %ld bytes\n", (long)payload);                       // model.c:194-197
    printf("Expected FP32 Shared (Raw): %lld\n", body * 4);
    printf("Expected FP32 Shared (+RoPE): %lld\n", body * 4 + rope_tail);
    printf("Expected FP16 Unshared: %lld\n", (body + headw) * 2);
    int fp32, shared;  // size-based detection, model.c:199-220
    if (payload == body * 4 || payload == body * 4 + rope_tail) {
        fp32 = 1; shared = 1;
        printf("Detected format: FP32 Shared Weights (Legacy%s)\n", payload == body * 4 + rope_tail ? " + RoPE" : "");
    } else if (payload == (body + headw) * 4) {
        fp32 = 1; shared = 0;
        printf("Detected format: FP32 Unshared Weights\n");
    } else if (payload == (body + headw) * 2) {
        fp32 = 0; shared = 0;
        printf("Detected format: FP16 Unshared Weights (Nano Export)\n");
    } else if (payload == body * 2) {
        fp32 = 0; shared = 1;
        printf("Detected format: FP16 Shared Weights\n");
    } else {
        printf("Warning: File size matches none of the expected formats! Trying FP16 Unshared as fallback...\n");
        fp32 = 0; shared = 0;
    }
    if (D % 32 || F % 32 || QD % 32 || KD % 16 || p->head_dim % 8) {
        fprintf(stderr, "unsupported shape: dim %lld, hidden %lld, q width %lld must be multiples of 32, kv width %lld of 16, head_dim %d of 8\n",
                D, F, QD, KD, p->head_dim);
        exit(1);
    }
    printf("Loading weights...\n");                                             // model.c:226
    // An fp32 file goes to the fp16 tile layout only if that loses nothing; otherwise it
    // stays fp32 in HBM and the forward pass runs on the fp32 operators (load_model_f32).
    if (fp32) {
        std::vector<std::pair<long long, bool>> tensors;   // (elements, is a matrix) in file order
        tensors.push_back({V * D, true});
        tensors.push_back({L * D, false});
        tensors.push_back({L * D * QD, true});
        tensors.push_back({2 * L * D * KD, true});
        tensors.push_back({L * QD * D, true});
        tensors.push_back({L * D, false});
        tensors.push_back({3 * L * D * F, true});
        tensors.push_back({D, false});
        if (!shared) tensors.push_back({V * D, true});
        if (!f32_file_fits_fp16(f, header, tensors)) {
            load_model_f32(w, p, f, shared != 0);
            return;
        }
    }
    // tensor parallelism: this process keeps the heads / hidden columns of its rank
    NvllmTpShard sh;
    if (nvllm_tp_shard(p, tp().rank, tp().world, &sh) != 0) {
        fprintf(stderr, "model shape (heads %d/%d, hidden %d) does not divide over %d tensor-parallel ranks\n",
                p->n_heads, p->n_kv_heads, p->hidden_dim, tp().world);
        exit(1);
    }
    const long long QDl = (long long)sh.n_heads * p->head_dim, KDl = (long long)sh.n_kv_heads * p->head_dim, Fl = sh.hidden;
    const long long q0 = (long long)sh.head0 * p->head_dim, k0 = (long long)sh.kv_head0 * p->head_dim, f0 = sh.hidden0;

    ModelPriv* m = new ModelPriv();
    m->cfg = *p;
    m->sh = sh;
    // ---- arena plan (all offsets 256-byte aligned)
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t Kb_D = D / 32, Kb_QD = QDl / 32, Kb_F = Fl / 32;
    const size_t t_qkv = (size_t)((QDl + 2 * KDl) / 16) * Kb_D * TILE_HALVES * 2;
    const size_t t_wo = (size_t)(D / 16) * Kb_QD * TILE_HALVES * 2;
    const size_t t_gu = (size_t)(2 * Fl / 16) * Kb_D * TILE_HALVES * 2;
    const size_t t_dn = (size_t)(D / 16) * Kb_F * TILE_HALVES * 2;
    const long long Vh = tp().world > 1 ? sh.vocab : V, v0 = tp().world > 1 ? sh.vocab0 : 0;   // classifier rows held here
    const size_t t_head = (size_t)((Vh + 15) / 16 + 1) * Kb_D * TILE_HALVES * 2;
    const size_t per_layer = align(t_qkv) + align(t_wo) + align(t_gu) + align(t_dn) + 2 * align(D * 4);
    const size_t total = align(V * D * 2) + L * per_layer + align(D * 4) + align(t_head);
    if (device_malloc(&m->arena, total) != 0) NVLLM_FATAL("cannot allocate %zu bytes of HBM for the weights", total);
    m->arena_bytes = total;
    char* base = (char*)m->arena;
    size_t off = 0;
    auto carve = [&](size_t bytes) { char* q = base + off; off += align(bytes); return q; };
    m->embed = (const __half*)carve(V * D * 2);
    m->layers.resize(L);
    for (int l = 0; l < L; l++) {
        LayerPriv& lp = m->layers[l];
        lp.qkv = (const __half*)carve(t_qkv);
        lp.wo = (const __half*)carve(t_wo);
        lp.gateup = (const __half*)carve(t_gu);
        lp.down = (const __half*)carve(t_dn);
        lp.rms_att = (const float*)carve(D * 4);
        lp.rms_ffn = (const float*)carve(D * 4);
    }
    m->rms_final = (const float*)carve(D * 4);
    m->head = (const __half*)carve(t_head);

    // ---- stream the file through one host buffer and one HBM staging buffer
    const size_t biggest = (size_t)std::max<long long>(std::max(V * D, D * F), D * QD);
    const size_t esz = fp32 ? 4 : 2;
    void* host = malloc(biggest * esz);
    void* stage = nullptr;       // raw file elements
    __half* stage16 = nullptr;   // fp16 row-major (== stage for fp16 files)
    if (!host || device_malloc(&stage, biggest * esz) != 0) NVLLM_FATAL("no memory for the %zu-byte load buffer", biggest * esz);
    if (fp32) {
        void* s16 = nullptr;
        if (device_malloc(&s16, biggest * 2) != 0) NVLLM_FATAL("no memory for the fp16 staging buffer");
        stage16 = (__half*)s16;
    } else {
        stage16 = (__half*)stage;
    }
    hipStream_t st = rt().stream;
    auto fetch = [&](size_t n, const char* what) {  // file -> stage16 (fp16 row-major on device)
        read_exact(f, host, n * esz, what);
        HIP_CHECK(hipMemcpyAsync(stage, host, n * esz, hipMemcpyHostToDevice, st));
        if (fp32) launch_f32_to_f16(stage16, (const float*)stage, n);
        HIP_CHECK(hipStreamSynchronize(st));  // `host` is reused by the next tensor
    };
    // rows [r0, r0 + nr) of a [rows, K] tensor: only that part of the file is read
    auto fetch_rows = [&](size_t rows, size_t K, size_t r0, size_t nr, const char* what) {
        if (r0 && fseek(f, (long)(r0 * K * esz), SEEK_CUR) != 0) NVLLM_FATAL("seek failed in %s", what);
        read_exact(f, host, nr * K * esz, what);
        if (rows > r0 + nr && fseek(f, (long)((rows - r0 - nr) * K * esz), SEEK_CUR) != 0) NVLLM_FATAL("seek failed in %s", what);
        HIP_CHECK(hipMemcpyAsync(stage, host, nr * K * esz, hipMemcpyHostToDevice, st));
        if (fp32) launch_f32_to_f16(stage16, (const float*)stage, nr * K);
        HIP_CHECK(hipStreamSynchronize(st));
    };
    // columns [c0, c0 + nc) of a [rows, K] tensor, compacted to [rows, nc] by the upload
    auto fetch_cols = [&](size_t rows, size_t K, size_t c0, size_t nc, const char* what) {
        read_exact(f, host, rows * K * esz, what);
        HIP_CHECK(hipMemcpy2DAsync(stage, nc * esz, (const char*)host + c0 * esz, K * esz, nc * esz, rows,
                                   hipMemcpyHostToDevice, st));
        if (fp32) launch_f32_to_f16(stage16, (const float*)stage, rows * nc);
        HIP_CHECK(hipStreamSynchronize(st));
    };
    auto fetch_vec = [&](const float* dst, size_t n, const char* what) {  // norm gains stay fp32
        read_exact(f, host, n * esz, what);
        HIP_CHECK(hipMemcpyAsync(stage, host, n * esz, hipMemcpyHostToDevice, st));
        if (fp32) HIP_CHECK(hipMemcpyAsync((void*)dst, stage, n * 4, hipMemcpyDeviceToDevice, st));
        else launch_f16_to_f32((float*)dst, (const __half*)stage, n);
        HIP_CHECK(hipStreamSynchronize(st));
    };
    auto tile_into = [&](const __half* dst, long long rows, long long K, long nb_stride, long long first_row = 0) {
        if (rows > 0) launch_tile_weights(const_cast<__half*>(dst), stage16 + first_row * K, (int)rows, (int)K, nb_stride);
        HIP_CHECK(hipStreamSynchronize(st));
    };
    const long str_D = (long)Kb_D * TILE_HALVES, str_QD = (long)Kb_QD * TILE_HALVES, str_F = (long)Kb_F * TILE_HALVES;

    fetch(V * D, "token_embedding_table");
    // the two values the reference prints through its "device" pointer (model.c:283-284); `host` still holds the table
    auto host_elem = [&](size_t i) { return fp32 ? ((const float*)host)[i] : __half2float(((const __half*)host)[i]); };
    const float dbg0 = host_elem(0), dbg100 = host_elem((size_t)std::min<long long>(100, V * D - 1));
    HIP_CHECK(hipMemcpyAsync((void*)m->embed, stage16, V * D * 2, hipMemcpyDeviceToDevice, st));
    if (shared) tile_into(m->head, Vh, D, str_D, v0);
    HIP_CHECK(hipStreamSynchronize(st));
    for (int l = 0; l < L; l++) fetch_vec(m->layers[l].rms_att, D, "rms_att_weight");
    for (int l = 0; l < L; l++) { fetch_rows(QD, D, q0, QDl, "wq"); tile_into(m->layers[l].qkv, QDl, D, str_D); }
    for (int l = 0; l < L; l++) { fetch_rows(KD, D, k0, KDl, "wk"); tile_into(m->layers[l].qkv + (QDl / 16) * str_D, KDl, D, str_D); }
    for (int l = 0; l < L; l++) { fetch_rows(KD, D, k0, KDl, "wv"); tile_into(m->layers[l].qkv + ((QDl + KDl) / 16) * str_D, KDl, D, str_D); }
    for (int l = 0; l < L; l++) { fetch_cols(D, QD, q0, QDl, "wo"); tile_into(m->layers[l].wo, D, QDl, str_QD); }
    for (int l = 0; l < L; l++) fetch_vec(m->layers[l].rms_ffn, D, "rms_ffn_weight");
    for (int l = 0; l < L; l++) { fetch_rows(F, D, f0, Fl, "w_gate"); tile_into(m->layers[l].gateup, Fl, D, 2 * str_D); }
    for (int l = 0; l < L; l++) { fetch_cols(D, F, f0, Fl, "w_down"); tile_into(m->layers[l].down, D, Fl, str_F); }  // file order: down before up (model.c:262-269)
    for (int l = 0; l < L; l++) { fetch_rows(F, D, f0, Fl, "w_up"); tile_into(m->layers[l].gateup + str_D, Fl, D, 2 * str_D); }
    fetch_vec(m->rms_final, D, "rms_final_weight");
    if (shared) printf("Weights shared: lm_head -> token_embedding_table\n");   // model.c:273-276
    else { fetch_rows(V, D, v0, Vh, "lm_head"); tile_into(m->head, Vh, D, str_D); }
    fclose(f);
    printf("Model loaded.\n");
    printf("DEBUG: Embed[0]: %f\n", dbg0);
    printf("DEBUG: Embed[100]: %f\n", dbg100);
    free(host);
    device_free(stage);
    if (fp32) device_free(stage16);

    // ---- public handles
    w->layers = (LayerWeights*)malloc((size_t)L * sizeof(LayerWeights));
    auto reg = [&](const __half* tiles, long long N, long long K, long nb_stride) -> float* {
        WeightEntry e{tiles, (int)N, (int)K, (int)(K / 32), nb_stride, nullptr};
        register_weight(tiles, e);
        m->handles.push_back(tiles);
        return (float*)tiles;
    };
    for (int l = 0; l < L; l++) {
        LayerPriv& lp = m->layers[l];
        LayerWeights& lw = w->layers[l];
        lw.rms_att_weight = (float*)lp.rms_att;
        lw.wq = reg(lp.qkv, QDl, D, str_D);
        lw.wk = reg(lp.qkv + (QDl / 16) * str_D, KDl, D, str_D);
        lw.wv = reg(lp.qkv + ((QDl + KDl) / 16) * str_D, KDl, D, str_D);
        lw.wo = reg(lp.wo, D, QDl, str_QD);
        lw.rms_ffn_weight = (float*)lp.rms_ffn;
        lw.w_gate = reg(lp.gateup, Fl, D, 2 * str_D);
        lw.w_up = reg(lp.gateup + str_D, Fl, D, 2 * str_D);
        lw.w_down = reg(lp.down, D, Fl, str_F);
    }
    {
        WeightEntry e{nullptr, (int)V, (int)D, (int)Kb_D, 0, m->embed};
        register_weight(m->embed, e);
        m->handles.push_back(m->embed);
    }
    w->token_embedding_table = (float*)m->embed;
    w->rms_final_weight = (float*)m->rms_final;
    w->lm_head = reg(m->head, Vh, D, str_D);
    w->_data_buffer = m;
    {
        std::vector<ChainLayer> tab((size_t)L);
        for (int l = 0; l < L; l++) {
            const LayerPriv& lp = m->layers[l];
            tab[l] = ChainLayer{lp.qkv, lp.wo, lp.gateup, lp.down, lp.rms_ffn, l + 1 < L ? m->layers[l + 1].rms_att : m->rms_final};
        }
        void* d = nullptr;
        if (device_malloc(&d, tab.size() * sizeof(ChainLayer)) != 0) NVLLM_FATAL("cannot allocate the layer table");
        device_memcpy(d, tab.data(), tab.size() * sizeof(ChainLayer), HOST_TO_DEVICE);
        m->chain_layers = (ChainLayer*)d;
    }
}

void nvllm_free_model(Weights* w) {
    if (!w || !w->_data_buffer) return;
    ModelPriv* m = (ModelPriv*)w->_data_buffer;
    for (const void* h : m->handles) unregister_weight(h);
    if (m->arena) device_free(m->arena);
    if (m->chain_layers) device_free(m->chain_layers);
    for (void* q : m->f32_bufs) device_free(q);
    delete m;
    free(w->layers);
    memset(w, 0, sizeof *w);
}

// ------------------------------------------------------------- workspace
// Reference src/model.c:287-336.  Same buffers, same capacities; `logits` is
// page-locked host memory (the driver reads it; sample() with a temperature rewrites
// the row it is given to probabilities, as the reference's does).
void malloc_run_state(RunState* s, Config* p) {
    ensure_init();
    const size_t S = (size_t)p->max_seq_len, D = p->dim, F = p->hidden_dim;
    auto dev = [&](size_t floats) {
        void* q = nullptr;
        if (device_malloc(&q, floats * sizeof(float)) != 0) NVLLM_FATAL("cannot allocate %zu floats of workspace", floats);
        return (float*)q;
    };
    s->x = dev(S * D);
    s->xb = dev(S * D);
    s->xb2 = dev(S * D);
    s->hb = dev(S * F);
    s->hb2 = dev(S * F);
    s->q = dev(S * D);
    s->k = dev(S * D);
    s->v = dev(S * D);
    s->att = dev((size_t)p->n_heads * S);
    s->logits = (float*)nvllm_internal_pinned_alloc(S * (size_t)p->vocab_size * sizeof(float));
    s->key_cache = s->value_cache = nullptr;
    if (!g_paged_mode) {
        const size_t n = (size_t)p->n_layers * S * p->n_kv_heads * p->head_dim;
        s->key_cache = dev(n);
        s->value_cache = dev(n);
    }
}

void free_run_state(RunState* s) {
    device_free(s->x); device_free(s->xb); device_free(s->xb2); device_free(s->hb); device_free(s->hb2);
    device_free(s->q); device_free(s->k); device_free(s->v); device_free(s->att); device_free(s->logits);
    if (s->key_cache) device_free(s->key_cache);
    if (s->value_cache) device_free(s->value_cache);
    if (g_seen.logits_host == s->logits) g_seen.logits_host = nullptr;   // its rows name nothing any more
    memset(s, 0, sizeof *s);
}

void transformer(int* tokens, int num_tokens, int pos, Config* p, RunState* s, Weights* w, BlockTable* bt) {
    (void)tokens; (void)num_tokens; (void)pos; (void)p; (void)s; (void)w; (void)bt;
    printf("Error: Old transformer called in Continuous Batching mode.\n");  // model.c:342
    exit(1);
}

// ------------------------------------------------------- transformer_batch
// One forward pass, enqueued on the library stream; returns its ticket.  `token_src` (optional, one
// entry per token): -1 = the token id is tokens[t]; k >= 0 = it is the arg-max of output row k of
// the pass enqueued just before this one, taken on the GPU (the host need not have seen it yet).
static long enqueue_pass(int* tokens, int num_tokens, int* pos_arr, int* seq_ids, int* output_indices,
                         int num_outputs, Config* p, RunState* s, Weights* w, BlockTable** block_tables,
                         const int* token_src) {
    ensure_init();
    if (!g_paged_mode) {  // model.c:374-377
        printf("Error: Continuous batching requires paged mode.\n");
        exit(1);
    }
    ModelPriv* m = (ModelPriv*)w->_data_buffer;
    if (!m) NVLLM_FATAL("transformer_batch: these weights were not produced by this library's load_model");
    const int T = num_tokens, B = num_outputs;
    if (T <= 0) return -1;
    if (T > p->max_seq_len || B > p->max_seq_len) NVLLM_FATAL("batch of %d tokens exceeds the workspace (%d)", T, p->max_seq_len);
    // H, KVH, F, QD, KD: what THIS process computes (its tensor-parallel slice; the whole model when world == 1)
    const int D = p->dim, F = m->sh.hidden, hd = p->head_dim, H = m->sh.n_heads, KVH = m->sh.n_kv_heads;
    const int QD = H * hd, KD = KVH * hd, V = p->vocab_size;
    const bool partial = tp().active;   // wo / w_down leave partial sums that are all-reduced
    if (partial && (m->sh.world != tp().world || m->sh.rank != tp().rank))
        NVLLM_FATAL("these weights were loaded as rank %d of %d, the process is now rank %d of %d", m->sh.rank,
                    m->sh.world, tp().rank, tp().world);
    if (!partial && m->sh.world != 1) NVLLM_FATAL("weights of a %d-way split need nvllm_tp_init before the forward pass", m->sh.world);
    const float eps = 1e-5f;  // model.c:361,388,396

    StepArrays st = stage_step(tokens, pos_arr, seq_ids, T, output_indices, B, block_tables, nullptr, 0,
                               g_kv_manager.block_size, attn_tokens_per_item(H, KVH), token_src);
    // The reference prints "Block table too small" once per layer and then attends over block ids
    // read past the table (attention.c:159-162, 285); here the pass ends before any kernel reads them.
    if (st.table_short) NVLLM_FATAL("transformer_batch: a position lies beyond its sequence's block table (allocate the block first, main.c:320-335)");
    const float2* rope = rope_table(p->rope_theta, hd, st.max_pos + 1);
    const KvPoolView pool = pool_view(&g_kv_manager, KVH, hd);

    const int ssn = gemm_feature_tiles(D);   // 32-feature groups of the residual width
    __half* a16_x = (__half*)ws_get(WS_A16_X, a16_halves(T, D) * sizeof(__half));
    __half* a16_att = (__half*)ws_get(WS_A16_ATT, a16_halves(T, QD) * sizeof(__half));
    __half* a16_h = (__half*)ws_get(WS_A16_H, a16_halves(T, F) * sizeof(__half));
    float* ss = (float*)ws_get(WS_MISC, (size_t)2 * T * ssn * sizeof(float));
    float* ss_a = ss;                       // sums of squares of x entering an attention block
    float* ss_b = ss + (size_t)T * ssn;     // ... entering an MLP block
    float* d_logits = (float*)ws_get(WS_LOGITS, (size_t)B * V * sizeof(float));
    // the arg-max indices live in a buffer that never moves (the next pass may read them as its tokens)
    int* d_argmax = (int*)ws_get(WS_ARGMAX, (size_t)(p->max_seq_len + 64) * sizeof(int));
    if (m->f32) {   // fp32 weights: the reference's operator sequence
        if (st.d_tok_src) launch_patch_tokens(st.d_tokens, st.d_tok_src, d_argmax, T);
        wait_for_row_copies();
        forward_f32(st, T, B, p, s, w, pool, rope, d_logits);
        launch_argmax(d_argmax, d_logits, B, V);
        return finish_pass(s, d_logits, d_argmax, B, V, PassView{});
    }

    // One launch for all layers (chain.hip) when the step fits it: a decode-sized batch on an fp16 page pool,
    // the whole model on this GPU.  Its arrival counters and status word are cleared by the embedding kernel.
    const bool try_chain = chain_enabled() && !partial && T <= 64 && pool.elem_bytes == 2 && m->chain_layers;
    unsigned* chain_sync = nullptr;
    unsigned* chain_status = (unsigned*)(d_argmax + B);
    size_t sync_words = 0;
    if (try_chain) {
        wait_for_row_copies();   // (the embedding kernel clears the status word, which sits behind the indices being copied)
        sync_words = chain_sync_words(p->n_layers);
        chain_sync = (unsigned*)ws_get(WS_CHAIN_SYNC, sync_words * sizeof(unsigned));
    }
    // x = embedding rows; a16_x = x * rms_att[0]; ss_a = per-tile sums of x^2   (model.c:352-355,361)
    launch_embed_prescale(s->x, a16_x, ss_a, m->embed, st.d_tokens, m->layers[0].rms_att, T, D, st.d_tok_src, d_argmax,
                          chain_sync, (int)(sync_words / 32), try_chain ? chain_status : nullptr);
    bool chained = false;
    if (try_chain) {
        ChainStep cs{};
        cs.layers = m->chain_layers;
        cs.n_layers = p->n_layers;
        cs.T = T; cs.D = D; cs.F = F; cs.H = H; cs.KVH = KVH; cs.hd = hd;
        cs.eps = eps;
        cs.x = s->x; cs.a16_x = a16_x; cs.a16_att = a16_att; cs.a16_h = a16_h; cs.ss_a = ss_a; cs.ss_b = ss_b; cs.q = s->q;
        cs.pos = st.d_pos; cs.kv_row = st.d_kv_row; cs.rope = rope;
        cs.pool_k = pool.k; cs.pool_v = pool.v; cs.kv_elem_bytes = pool.elem_bytes; cs.block_size = pool.block_size;
        cs.layer_stride = pool.layer_stride;
        cs.tables = st.d_tables; cs.runs = st.d_runs; cs.n_runs = st.n_runs;
        for (int i = 0; i < 4; i++) cs.runs_in_tb[i] = st.runs_in_tb[i];
        cs.one_token_per_seq = st.one_token_per_seq;
        cs.sync = chain_sync;
        cs.status = chain_status;
        const double wbytes = 2.0 * p->n_layers * ((double)D * QD + 2.0 * D * KD + (double)QD * D + 3.0 * D * F);
        cs.bytes = wbytes + (double)st.kv_positions * 2.0 * KD * pool.elem_bytes * p->n_layers;
        cs.flops = wbytes * T + 4.0 * (double)st.kv_positions * H * hd * p->n_layers;
        chained = launch_decode_chain(cs);
    }

    for (int l = 0; l < p->n_layers && !chained; l++) {
        const LayerPriv& lp = m->layers[l];
        // q,k,v = rmsnorm(x) . [wq;wk;wv]^T, RoPE, K/V rows into the pool      (model.c:361-377)
        WeightEntry wqkv{lp.qkv, QD + 2 * KD, D, D / 32, (long)(D / 32) * TILE_HALVES, nullptr};
        GemmEpilogue e1;
        e1.kind = GemmEpilogue::ROPE_KV;
        e1.ss_in = ss_a; e1.ss_tiles = ssn; e1.d_in = D; e1.eps = eps;
        e1.q_out = s->q; e1.QD = QD; e1.KD = KD; e1.hd = hd;
        e1.pos = st.d_pos; e1.kv_row = st.d_kv_row;
        e1.rope = rope;
        e1.pool_k = pool.k; e1.pool_v = pool.v; e1.kv_elem_bytes = pool.elem_bytes; e1.bs = pool.block_size;
        e1.layer_off = (long)l * pool.layer_stride;
        launch_gemm(wqkv, lp.qkv, QD + 2 * KD, a16_x, T, e1, 0, PC_GEMM_QKV);
        // attention over the block table -> fp16 hi/lo tiles                     (model.c:380-383)
        AttnArgs at{};
        at.q = s->q;
        at.out_a16 = a16_att;
        at.pool = pool;
        at.layer = l;
        at.tables = st.d_tables;
        at.T = T; at.H = H; at.KVH = KVH; at.hd = hd;
        at.max_pos = st.max_pos;
        at.runs = st.d_runs;
        at.n_runs = st.n_runs;
        at.kv_positions = st.kv_positions;
        launch_paged_attention(at);
        // x += att . wo^T; a16_x = x * rms_ffn; ss_b                              (model.c:385-388)
        WeightEntry wwo{lp.wo, D, QD, QD / 32, (long)(QD / 32) * TILE_HALVES, nullptr};
        GemmEpilogue e2;
        if (!partial) {
            e2.kind = GemmEpilogue::RESID_A16;
            e2.x = s->x; e2.gain_next = lp.rms_ffn; e2.a16_out = a16_x; e2.ss_out = ss_b;
            launch_gemm(wwo, lp.wo, D, a16_att, T, e2, 0, PC_GEMM_WO);
        } else {   // this rank's heads only: sum over the ranks, then the same tail as a separate launch
            e2.kind = GemmEpilogue::F32;
            e2.out = s->xb; e2.ldo = D;
            launch_gemm(wwo, lp.wo, D, a16_att, T, e2, 0, PC_GEMM_WO);
            if (!tp_peer_allreduce_resid(s->xb, s->x, a16_x, ss_b, lp.rms_ffn, T, D)) {
                tp_allreduce(s->xb, (size_t)T * D);
                launch_resid_prescale(s->x, s->xb, a16_x, ss_b, lp.rms_ffn, T, D);
            }
        }
        // hb = silu(rmsnorm(x).w_gate^T) * (rmsnorm(x).w_up^T)                    (model.c:388-391)
        WeightEntry wgu{lp.gateup, 2 * F, D, D / 32, (long)(D / 32) * TILE_HALVES, nullptr};
        GemmEpilogue e3;
        e3.kind = GemmEpilogue::SWIGLU_A16;
        e3.ss_in = ss_b; e3.ss_tiles = ssn; e3.d_in = D; e3.eps = eps;
        e3.a16_out = a16_h;
        launch_gemm(wgu, lp.gateup, 2 * F, a16_x, T, e3, 0, PC_GEMM_GATEUP);
        // x += hb . w_down^T; a16_x = x * (next norm's gain); ss_a                (model.c:392-393,361|396)
        WeightEntry wdn{lp.down, D, F, F / 32, (long)(F / 32) * TILE_HALVES, nullptr};
        GemmEpilogue e4;
        const float* gain_next = l + 1 < p->n_layers ? m->layers[l + 1].rms_att : m->rms_final;
        if (!partial) {
            e4.kind = GemmEpilogue::RESID_A16;
            e4.x = s->x;
            e4.gain_next = gain_next;
            e4.a16_out = a16_x; e4.ss_out = ss_a;
            launch_gemm(wdn, lp.down, D, a16_h, T, e4, 0, PC_GEMM_DOWN);
        } else {   // this rank's hidden columns only
            e4.kind = GemmEpilogue::F32;
            e4.out = s->xb; e4.ldo = D;
            launch_gemm(wdn, lp.down, D, a16_h, T, e4, 0, PC_GEMM_DOWN);
            if (!tp_peer_allreduce_resid(s->xb, s->x, a16_x, ss_a, gain_next, T, D)) {
                tp_allreduce(s->xb, (size_t)T * D);
                launch_resid_prescale(s->x, s->xb, a16_x, ss_a, gain_next, T, D);
            }
        }
    }
    // logits only for the rows named by output_indices                            (model.c:396,411-418)
    wait_for_row_copies();   // the previous pass's rows have left the workspaces the classifier writes
    bool identity = B == T;
    for (int k = 0; identity && k < B; k++) identity = output_indices[k] == k;
    const __half* a16_head = a16_x;
    const float* ss_head = ss_a;
    if (!identity) {
        __half* g16 = (__half*)ws_get(WS_A16_ATT, a16_halves(B > T ? B : T, QD > D ? QD : D) * sizeof(__half));
        float* gss = (float*)ws_get(WS_MISC2, (size_t)B * ssn * sizeof(float));
        launch_gather_a16(g16, gss, a16_x, ss_a, st.d_out_idx, B, D, ssn);
        a16_head = g16;
        ss_head = gss;
    }
    GemmEpilogue e5;
    e5.kind = GemmEpilogue::F32;
    e5.ss_in = ss_head; e5.ss_tiles = ssn; e5.d_in = D; e5.eps = eps;
    PassView view;
    bool host_written = false;
    // no K split for the classifier (tens of thousands of rows fill the chip); from 112 output rows the
    // launcher may take the dense-tile kernel instead (0 = its own choice)
    const int head_split = B >= 112 ? 0 : 1;
    if (!partial) {
        WeightEntry wh{m->head, V, D, D / 32, (long)(D / 32) * TILE_HALVES, nullptr};
        e5.out = d_logits;
        e5.ldo = V;
        // the classifier's tiles also report their (maximum, index) pairs: the greedy token comes from
        // a merge of those instead of a second pass over the logits
        e5.amax_out = d_argmax;
        // only token ids leave the GPU: the kernel that picks them also stores them into the pass's page-locked array
        static const int direct_env = getenv("NVLLM_ARGMAX_DIRECT") ? atoi(getenv("NVLLM_ARGMAX_DIRECT")) : 1;
        host_written = direct_env && rt().logits_mode == NVLLM_LOGITS_ARGMAX && !chained;
        if (host_written) e5.amax_out_host = next_slot(B).host;
        e5.amax_ld = (B + 63) & ~63;
        e5.amax_pairs = (float2*)ws_get(WS_AMAX_PAIRS, (size_t)cdiv(V, 32) * e5.amax_ld * sizeof(float2));
        launch_gemm(wh, m->head, V, a16_head, B, e5, head_split, PC_GEMM_HEAD);
    } else {
        // this rank's vocabulary rows; then either one (max, index) pair per row from every rank,
        // or the slices themselves when full rows are wanted
        const int W = tp().world, slice = m->sh.vocab_slice, mine = W > 1 ? m->sh.vocab : V, base = W > 1 ? m->sh.vocab0 : 0;
        const int ld = W > 1 ? slice : V;
        float* local = (float*)ws_get(WS_TP_LOCAL, ((size_t)B * ld + 2 * (size_t)B + 64) * sizeof(float));
        if (mine > 0) {
            WeightEntry wh{m->head, mine, D, D / 32, (long)(D / 32) * TILE_HALVES, nullptr};
            e5.out = local;
            e5.ldo = ld;
            launch_gemm(wh, m->head, mine, a16_head, B, e5, head_split, PC_GEMM_HEAD);
        }
        view.tp_local = local;
        view.tp_slice = ld;
        if (rt().logits_mode == NVLLM_LOGITS_ARGMAX) {
            float2* pairs = (float2*)(local + (((size_t)B * ld + 1) & ~(size_t)1));
            launch_argmax_pairs(pairs, local, B, mine, ld, base);
            float2* all = (float2*)ws_get(WS_TP_GATHER, (size_t)W * B * sizeof(float2));
            tp_allgather((const float*)pairs, (float*)all, 2 * (size_t)B);
            launch_argmax_merge(d_argmax, all, W, B);
            view.tp_full_valid = false;   // d_logits was not produced; sample() gathers it if it needs rows
        } else {
            tp_gather_logits(d_logits, local, B, ld, V);
            launch_argmax(d_argmax, d_logits, B, V);
        }
    }

    return finish_pass(s, d_logits, d_argmax, B, V, view, chained, host_written);
}

void transformer_batch(int* tokens, int num_tokens, int* pos_arr, int* seq_ids, int* output_indices,
                       int num_outputs, Config* p, RunState* s, Weights* w, BlockTable** block_tables) {
    const long ticket = enqueue_pass(tokens, num_tokens, pos_arr, seq_ids, output_indices, num_outputs, p, s, w,
                                     block_tables, nullptr);
    if (ticket < 0) return;
    collect_pass(ticket, nullptr);
    HIP_CHECK(hipStreamSynchronize(rt().stream));  // the call is synchronous, like the CPU reference
}

// include/nvllm.h: the forward pass in two halves, for a driver that keeps the GPU busy across steps
long nvllm_forward_enqueue(int* tokens, int num_tokens, int* pos_arr, int* seq_ids, int* output_indices, int num_outputs,
                           const int* token_src, Config* p, RunState* s, Weights* w, BlockTable** block_tables) {
    // (with host logits rows, every pass in flight needs a RunState of its own: its rows land in s->logits)
    if (num_tokens <= 0 || num_outputs < 0 || !tokens || !pos_arr || !seq_ids || !p || !s || !w || !block_tables) return -1;
    return enqueue_pass(tokens, num_tokens, pos_arr, seq_ids, output_indices, num_outputs, p, s, w, block_tables, token_src);
}

int nvllm_forward_collect(long ticket, int* argmax_out, int cap) {
    int n = 0;
    const int* r = collect_pass(ticket, &n);
    for (int i = 0; i < n && i < cap; i++) argmax_out[i] = r[i];
    return n;
}

// The arg-max kernels of the forward pass on caller-supplied rows (include/nvllm.h).
int nvllm_argmax_rows(int* out, const float* logits, int num_rows, int vocab_size, int slices) {
    ensure_init();
    if (num_rows <= 0 || vocab_size <= 0 || !out || !logits) return -1;
    const int B = num_rows, V = vocab_size;
    int* d_out = (int*)ws_get(WS_MISC2, (size_t)(B + 64) * sizeof(int));   // not the passes' own index buffer
    if (slices <= 1) {
        launch_argmax(d_out, logits, B, V);
    } else {
        // slices of whole 16-row blocks, as nvllm_tp_shard cuts the classifier over ranks
        const int per = ((V + 15) / 16 + slices - 1) / slices * 16;
        float2* pairs = (float2*)ws_get(WS_TP_GATHER, (size_t)slices * B * sizeof(float2));
        for (int r = 0; r < slices; r++) {
            const int v0 = (long)r * per < V ? r * per : V;
            const int n = V - v0 < per ? V - v0 : per;   // may be 0: that slice reports (-inf, v0)
            launch_argmax_pairs(pairs + (size_t)r * B, logits + v0, B, n, V, v0);
        }
        launch_argmax_merge(d_out, pairs, slices, B);
    }
    HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, rt().stream));
    HIP_CHECK(hipStreamSynchronize(rt().stream));
    return 0;
}

const int* nvllm_last_argmax(int* num_outputs) {
    if (num_outputs) *num_outputs = g_seen.outputs;
    return g_seen.argmax;
}

// ------------------------------------------------------------------ sampler
void build_sampler(Sampler* s, int vocab_size, float temperature, float topp, unsigned long long seed) {
    s->state = seed;
    s->temperature = temperature;
    s->topp = topp;
    s->vocab_size = vocab_size;
    s->probindex = (ProbIndex*)malloc((size_t)vocab_size * sizeof(ProbIndex));
    srand((unsigned int)seed);  // reference sampler.c:26
}

// ---- temperature > 0 (reference sampler.c:100-139).  The GPU turns every row of the
// most recent forward pass into probabilities and filters the nucleus candidates
// (one launch for all rows, on the first sample() call after the pass); the host
// receives the few candidates of each row, sorts them with the reference's comparator
// and walks the cumulative sums with the reference's libc rand() coin.
constexpr int SAMPLE_CAP = 1024;          // candidates kept per row; fuller rows take the whole-row path
// One prepared set of rows: probabilities in HBM, the nucleus candidates of every row in page-locked
// memory, and (when something needs whole rows on the host) the probabilities there too.  Two sets:
// all rows of a pass, prepared together, and a lone row the caller brought.  Buffers only ever grow.
struct Prepared {
    float* probs_dev = nullptr;       // [rows, V] (workspace)
    ProbIndex* cand_host = nullptr;   // pinned [cand_rows, SAMPLE_CAP]
    int* count_host = nullptr;        // pinned [cand_rows]
    int cand_rows = 0;
    float* probs_host = nullptr;      // pinned [rows, V], filled on demand ...
    size_t probs_host_cap = 0;
    bool probs_on_host = false;       // ... for the current set
    int rows = 0, V = 0;
    long ticket = -1;                 // pass whose rows are prepared (-1: none) ...
    float temp = 0.f, topp = 0.f;     // ... with these settings ...
    bool host_rows = false;           // ... from the rows in RunState.logits (else from the logits in HBM)
};
static Prepared g_prep_pass, g_prep_lone;
// logits mode HOST: the rows of the pass as they stood in RunState.logits when they were uploaded together
static float* g_rows_snapshot = nullptr;   // pinned
static size_t g_rows_snapshot_cap = 0;

static int by_prob_desc(const void* a, const void* b) {   // sampler.c:29-35
    const float pa = ((const ProbIndex*)a)->prob, pb = ((const ProbIndex*)b)->prob;
    if (pa > pb) return -1;
    if (pa < pb) return 1;
    return 0;
}

// sampler.c:54-81 on an already filtered candidate list
static int nucleus_pick(ProbIndex* c, int n, float topp, float coin) {
    qsort(c, (size_t)n, sizeof(ProbIndex), by_prob_desc);
    float mass = 0.0f;
    int last = n - 1;
    for (int i = 0; i < n; i++) {
        mass += c[i].prob;
        if (mass > topp) {
            last = i;
            break;
        }
    }
    const float r = coin * mass;
    float cdf = 0.0f;
    for (int i = 0; i <= last; i++) {
        cdf += c[i].prob;
        if (r < cdf) return c[i].index;
    }
    return c[last].index;
}

// all rows' probabilities -> page-locked host memory, one copy
static void fetch_probs(Prepared& pr) {
    const size_t n = (size_t)pr.rows * pr.V;
    if (n > pr.probs_host_cap) {
        if (pr.probs_host) HIP_CHECK(hipHostFree(pr.probs_host));
        pr.probs_host_cap = n + n / 4;
        HIP_CHECK(hipHostMalloc((void**)&pr.probs_host, pr.probs_host_cap * sizeof(float), hipHostMallocDefault));
    }
    HIP_CHECK(hipMemcpyAsync(pr.probs_host, pr.probs_dev, n * sizeof(float), hipMemcpyDeviceToHost, rt().stream));
    HIP_CHECK(hipStreamSynchronize(rt().stream));
    pr.probs_on_host = true;
}

// `probs_ws` / `cand_ws`: the workspace slots that hold this set's device side.
static void prepare_rows(Prepared& pr, WsSlot probs_ws, WsSlot cand_ws, const float* logits_dev, int rows, int V,
                         float temperature, float topp, bool want_probs_on_host) {
    const bool nucleus = topp > 0.f && topp < 1.f;
    const float cutoff = nucleus ? (1.0f - topp) / (float)(V - 1) : -1.0f;   // sampler.c:47
    if (rows > pr.cand_rows) {
        if (pr.cand_host) HIP_CHECK(hipHostFree(pr.cand_host));
        if (pr.count_host) HIP_CHECK(hipHostFree(pr.count_host));
        pr.cand_rows = rows + 64;
        HIP_CHECK(hipHostMalloc((void**)&pr.cand_host, (size_t)pr.cand_rows * SAMPLE_CAP * sizeof(ProbIndex), hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void**)&pr.count_host, (size_t)pr.cand_rows * sizeof(int), hipHostMallocDefault));
    }
    pr.probs_dev = (float*)ws_get(probs_ws, (size_t)rows * V * sizeof(float));
    char* scratch = (char*)ws_get(cand_ws, (size_t)rows * (SAMPLE_CAP * sizeof(ProbIndex) + sizeof(int)) + 256);
    ProbIndex* cand_dev = (ProbIndex*)scratch;
    int* count_dev = (int*)(scratch + (size_t)rows * SAMPLE_CAP * sizeof(ProbIndex));
    launch_sample_prep(pr.probs_dev, logits_dev, rows, V, temperature, cutoff, cand_dev, SAMPLE_CAP, count_dev);
    if (nucleus) {
        HIP_CHECK(hipMemcpyAsync(pr.count_host, count_dev, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, rt().stream));
        HIP_CHECK(hipMemcpyAsync(pr.cand_host, cand_dev, (size_t)rows * SAMPLE_CAP * sizeof(ProbIndex), hipMemcpyDeviceToHost, rt().stream));
    }
    pr.rows = rows;
    pr.V = V;
    pr.temp = temperature;
    pr.topp = topp;
    pr.probs_on_host = false;
    // plain sampling walks whole rows on the host (sampler.c:126-134): they travel now, together
    if (want_probs_on_host || !nucleus) fetch_probs(pr);
    else HIP_CHECK(hipStreamSynchronize(rt().stream));
}

// one prepared row -> token, with the reference's coin
static int pick_from_row(Sampler* s, Prepared& pr, int row) {
    const float coin = (float)rand() / (float)RAND_MAX;   // sampler.c:124
    const bool nucleus = s->topp > 0.f && s->topp < 1.f;
    if (nucleus && pr.count_host[row] <= SAMPLE_CAP)
        return nucleus_pick(pr.cand_host + (size_t)row * SAMPLE_CAP, pr.count_host[row], s->topp, coin);
    // whole-row path: plain sampling (sampler.c:126-134), or a nucleus too wide for the candidate buffer
    // (then every row of the set comes over, once: its neighbours are likely as wide)
    if (!pr.probs_on_host) fetch_probs(pr);
    const float* p = pr.probs_host + (size_t)row * pr.V;
    const int V = pr.V;
    if (!nucleus) {
        float cdf = 0.0f;
        for (int i = 0; i < V; i++) {
            cdf += p[i];
            if (coin < cdf) return i;
        }
        return V - 1;
    }
    const float cutoff = (1.0f - s->topp) / (float)(V - 1);
    int n = 0;
    for (int i = 0; i < V; i++)
        if (p[i] >= cutoff) {
            s->probindex[n].index = i;
            s->probindex[n].prob = p[i];
            n++;
        }
    return nucleus_pick(s->probindex, n, s->topp, coin);
}

// Reference src/sampler.c:84-140.  Which memory answers depends on the hand-off mode:
//   NVLLM_LOGITS_HOST    the row the caller passes is host memory that the forward pass filled and the
//                        caller may have edited since (logit bias, banned tokens ...): it is READ as it
//                        stands, and with a temperature it is rewritten to probabilities, like the
//                        reference's in-place softmax (sampler.c:100-121).
//   NVLLM_LOGITS_ARGMAX  RunState.logits was never filled; a pointer to row k of it only NAMES output k
//                        of the most recently collected pass, whose logits are still in HBM.
int sample(Sampler* s, float* logits) {
    ensure_init();
    const int V = s->vocab_size;
    int row = -1;   // output row of the most recently collected forward pass that `logits` points at, if any
    if (g_seen.logits_host && V == g_seen.vocab && logits >= g_seen.logits_host) {
        const size_t d = (size_t)(logits - g_seen.logits_host);
        if (d % (size_t)V == 0 && d / (size_t)V < (size_t)g_seen.outputs) row = (int)(d / (size_t)V);
    }
    const bool named = row >= 0 && rt().logits_mode == NVLLM_LOGITS_ARGMAX;   // the pointer only names the row
    if (s->temperature == 0.0f) {
        if (named) return g_seen.argmax[row];   // the arg-max the GPU already computed
        // sampler.c:86-97 on the caller's row: the strict '>' scan from index 0 ends on the first
        // index that attains the maximum.  Two passes the compiler can vectorise: the maximum (a
        // NaN never wins a '>' and an initial NaN is never displaced, as in the reference), then
        // its first occurrence.
        if (logits[0] != logits[0]) return 0;   // logits[0] is NaN: the reference's scan never leaves index 0
        // (sixteen independent running maxima, each started from logits[0]: the host has 64 rows of
        // 32,000 floats to scan per step, and one dependent compare per element is what bounds it)
        float acc[16];
        for (int j = 0; j < 16; j++) acc[j] = logits[0];
        int i = 1;
        for (; i + 16 <= V; i += 16)
            for (int j = 0; j < 16; j++) acc[j] = logits[i + j] > acc[j] ? logits[i + j] : acc[j];
        float top = acc[0];
        for (int j = 1; j < 16; j++) top = acc[j] > top ? acc[j] : top;
        for (; i < V; i++) top = logits[i] > top ? logits[i] : top;
        // its first occurrence: whole chunks are tested without a branch per element
        for (i = 0; i + 64 <= V; i += 64) {
            int hit = 0;
            for (int j = 0; j < 64; j++) hit |= (logits[i + j] == top);
            if (hit) break;
        }
        for (; i < V; i++)
            if (logits[i] == top) return i;
        return 0;
    }
    // temperature > 0: softmax and the nucleus filter run on the GPU, for all rows of the pass at once
    Prepared& pp = g_prep_pass;
    const bool fresh = pp.ticket == g_seen.ticket && pp.temp == s->temperature && pp.topp == s->topp && pp.host_rows == !named;
    if (named) {
        if (!fresh) {
            // the logits live in a workspace that the next pass's classifier overwrites
            if (g_seen.ticket != g_next_ticket - 1)
                NVLLM_FATAL("sample(): pass %ld's logits are gone, pass %ld was enqueued since; sample a pass's rows before the next nvllm_forward_enqueue",
                            g_seen.ticket, g_next_ticket - 1);
            if (!g_seen.tp_full_valid) {   // the logits are vocabulary slices on the ranks: every rank makes this call
                tp_gather_logits(const_cast<float*>(g_seen.logits_dev), g_seen.tp_local, g_seen.outputs, g_seen.tp_slice, V);
                g_seen.tp_full_valid = true;
            }
            prepare_rows(pp, WS_MISC, WS_MISC2, g_seen.logits_dev, g_seen.outputs, V, s->temperature, s->topp, false);
            pp.ticket = g_seen.ticket;
            pp.host_rows = false;
        }
        return pick_from_row(s, pp, row);
    }
    if (row >= 0 && g_seen.outputs > 1) {
        // HOST mode, a row of the pass: the first such call uploads ALL its rows as they stand now and
        // prepares them together; a row the caller has changed since then is handled alone, below.
        const size_t n = (size_t)g_seen.outputs * V;
        if (!fresh) {
            if (n > g_rows_snapshot_cap) {
                if (g_rows_snapshot) HIP_CHECK(hipHostFree(g_rows_snapshot));
                g_rows_snapshot_cap = n + n / 4;
                HIP_CHECK(hipHostMalloc((void**)&g_rows_snapshot, g_rows_snapshot_cap * sizeof(float), hipHostMallocDefault));
            }
            memcpy(g_rows_snapshot, g_seen.logits_host, n * sizeof(float));
            float* dev = (float*)ws_get(WS_SAMPLE_ROWS, n * sizeof(float));
            HIP_CHECK(hipMemcpyAsync(dev, g_rows_snapshot, n * sizeof(float), hipMemcpyHostToDevice, rt().stream));
            prepare_rows(pp, WS_SAMPLE_PROBS, WS_SAMPLE_CAND, dev, g_seen.outputs, V, s->temperature, s->topp, true);
            pp.ticket = g_seen.ticket;
            pp.host_rows = true;
        }
        if (memcmp(logits, g_rows_snapshot + (size_t)row * V, (size_t)V * sizeof(float)) == 0) {
            memcpy(logits, pp.probs_host + (size_t)row * V, (size_t)V * sizeof(float));   // in place, like sampler.c:100-121
            return pick_from_row(s, pp, row);
        }
    }
    // a lone host row: upload it, prepare it, hand the probabilities back in place
    Prepared& pl = g_prep_lone;
    float* dev = (float*)ws_get(WS_SAMPLE, (size_t)V * sizeof(float));
    HIP_CHECK(hipMemcpyAsync(dev, logits, (size_t)V * sizeof(float), hipMemcpyHostToDevice, rt().stream));
    prepare_rows(pl, WS_SAMPLE_PROBS1, WS_SAMPLE_CAND1, dev, 1, V, s->temperature, s->topp, true);
    memcpy(logits, pl.probs_host, (size_t)V * sizeof(float));
    return pick_from_row(s, pl, 0);
}

}  // extern "C"
