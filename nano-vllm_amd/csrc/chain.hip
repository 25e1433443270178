// chain.hip -- one decode step's layers as ONE launch (the reference's layer loop,
// src/model.c:358-394, with src/kernels/layers.c:44-67 and src/kernels/attention.c:256-336 inside).
//
// Why.  Launched one by one, the five kernels of a layer (q/k/v projection, attention, output
// projection, gate/up, down) each start cold: a kernel boundary, a ramp until the first weight
// bytes arrive, and at the end a reduction, an epilogue and a store drain during which nothing
// streams from HBM anywhere on the chip.  Weights, block tables and every K/V page but the newest
// depend on nothing the step computes, so here a workgroup requests the first ring of its NEXT
// tile's weights as soon as it has handed its current tile over -- they travel while the other
// workgroups finish theirs -- and attention requests its first pages before the projection that
// produces q has finished.
//
// What it buys (measured on MI355X, in-kernel clock stamps, tools/chain_stamps.py).  The overlap
// happens -- the first ring of a tile has landed before the last workgroup of the previous phase
// has drained its stores -- but a hand-off through memory costs what a kernel boundary costs: the
// phase ends when its SLOWEST workgroup has stored (3 to 5 us behind the median), detecting that
// takes another 1 to 2 us, and issuing a ring holds a wave for its bytes / ~50 GB/s.  TinyLlama-1.1B
// at 64 rows: 70 us per layer here against 65 us as five kernels; the 42M shape at 32 rows: 0.28 ms
// per step against 0.27.  The switch is therefore OFF unless asked for (NVLLM_DECODE_CHAIN=1 or
// nvllm_set_decode_chain); the results are the same bits either way.
//
// How.  A grid of one 512-thread workgroup per CU runs, per layer, five phases with the work
// decompositions of the separate kernels (gemm_tile.h, attn_tile.h; workgroup b takes tile b of
// every phase, so the token groups of one weight tile still share an XCD's L2).  Between phases
// there is no grid barrier: every 16-token block has an arrival counter per phase; a tile's
// producer stores write-through (sc1), drains, and adds one to the counters of the token blocks it
// covers; a consumer polls the counters of the token blocks it reads (one lane, relaxed sc1 loads),
// then loads past its vector cache (sc1; the attention phase, whose inputs include 2-byte stores, also
// invalidates that cache first).  All waits are bounded: a workgroup that waits longer
// than a fifth of a second marks the step faulty, stops waiting and finishes, so the launch always
// ends.  Static tile assignment needs every workgroup resident at once, which a census at the
// start checks before anything is written; if it fails (another process holds CUs) no workgroup
// writes anything, the launch reports it in its status word and the pass ends with a fatal error
// that names the switch (collect_pass in model.hip).
#include "attn_tile.h"
#include "gemm_tile.h"

extern "C" float nvllm_host_attn_scale(int head_dim);

namespace nvllm {

constexpr int CH_NW = 8;              // waves per workgroup
constexpr int CH_PHASES = 5;
enum ChainPhase { PH_QKV = 0, PH_ATT = 1, PH_WO = 2, PH_GU = 3, PH_DN = 4 };
constexpr int CH_TBMAX = 4;           // 16-token blocks per step (64 tokens)
constexpr int CH_LINE = 32;           // words between two counters (one 128-byte line each)
constexpr int CH_REP = 8;             // replicas of every arrival counter: a producer adds to all of them with one wave
                                      // instruction, a consumer polls the one of its workgroup number modulo 8, so that
                                      // the chip's pollers are spread over eight lines (memory channels) per counter
constexpr unsigned CH_GO = 1, CH_ABORT = 2;
constexpr unsigned long long CH_WAIT_TICKS = 20000000ull;    // 0.2 s of the 100 MHz constant clock
constexpr unsigned long long CH_CENSUS_TICKS = 5000000ull;   // 50 ms

struct ChainProj {            // the grid one projection would be launched with
    int NB, Kb, N;
    long nb_stride;
    int kb_per_split, ksplit;
    int gx, gz;               // feature tiles per token group (padding included), token groups
};

struct ChainArgs {
    const ChainLayer* layers;
    int n_layers;
    int T, Tb;
    ChainProj qkv, wo, gu, dn;
    float* x;
    __half* a16_x;
    __half* a16_att;
    __half* a16_h;
    float* ss_a;
    float* ss_b;
    float* q;
    int ss_tiles, D, F;
    float eps;
    int QD, KD, hd;
    const int* pos;
    const int* kv_row;
    const float2* rope;
    void* pool_k;
    void* pool_v;
    int bs;
    long layer_stride;
    AttnKArgs att;            // layer_off is set per layer
    int fresh_rule;           // attn_tile.h: 1 = every sequence has exactly one token in the step
    int n_items;
    float* slab;
    unsigned* tile_counters;
    unsigned* sync;           // census, verdict, then one counter line per (layer, phase, token block); zero at launch
    unsigned* status;         // 0 at launch; 1 = ran, 2 = not every workgroup was resident (nothing done), |4 = a wait timed out
    unsigned expect[CH_PHASES][CH_TBMAX];   // arrivals that complete a token block of a phase
    unsigned long long* stamps;             // diagnostic builds: [layer * 5 + phase][workgroup][8]
};

size_t chain_sync_words(int n_layers) { return 64 + (size_t)n_layers * CH_PHASES * CH_TBMAX * CH_REP * CH_LINE; }

// The launch record is read where it is needed, straight from the kernel-argument segment (scalar
// loads).  Held as an ordinary by-value parameter its hundred-odd words would sit in scalar
// registers for the whole launch, most of them spilled; `chain_kargs()` returns an opaque copy of
// the segment's address, so that the fields one phase reads are not kept for the next.
typedef const ChainArgs __attribute__((address_space(4))) * KArgs;
__device__ __forceinline__ KArgs chain_kargs() {
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <typename T>
__device__ __forceinline__ T karg(const T __attribute__((address_space(4))) * p) {
    T v;
    __builtin_memcpy(&v, p, sizeof(T));
    return v;
}

__device__ __forceinline__ unsigned* chain_counter(KArgs k, int layer, int phase) {
    return k->sync + 64 + (size_t)(layer * CH_PHASES + phase) * CH_TBMAX * CH_REP * CH_LINE;
}
__device__ __forceinline__ unsigned long long chain_clock() { return __builtin_amdgcn_s_memrealtime(); }

// Every workgroup of the grid is resident (they all arrived) -> true; otherwise nobody does anything.
__device__ __forceinline__ bool chain_census(KArgs k, unsigned* s_word) {
    if (threadIdx.x == 0) {
        unsigned* census = k->sync;
        unsigned* verdict = k->sync + CH_LINE;
        unsigned* status = k->status;
        const unsigned old = __hip_atomic_fetch_add(census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned v = 0;
        if (old + 1 == gridDim.x) {
            unsigned expected = 0;
            if (__hip_atomic_compare_exchange_strong(verdict, &expected, CH_GO, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                v = CH_GO;
                __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                v = expected;
            }
        } else {
            const unsigned long long t0 = chain_clock();
            for (;;) {
                v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != 0) break;
                __builtin_amdgcn_s_sleep(8);
                if (chain_clock() - t0 > CH_CENSUS_TICKS) {
                    unsigned expected = 0;
                    if (__hip_atomic_compare_exchange_strong(verdict, &expected, CH_ABORT, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        v = CH_ABORT;
                        __hip_atomic_fetch_or(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        v = expected;
                    }
                    break;
                }
            }
        }
        s_word[0] = v;
        s_word[1] = 0;   // "this workgroup has given up waiting"
    }
    __syncthreads();
    return s_word[0] == CH_GO;
}

// Wait until the token blocks of `mask` have all their arrivals of one phase.  Called by every
// thread; wave 0 polls (lane i watches token block i), the others wait at the barrier.
__device__ __forceinline__ void chain_wait(KArgs k, const unsigned* counters, int phase, unsigned mask, unsigned* s_word) {
    if (mask != 0 && threadIdx.x < 64 && s_word[1] == 0) {
        const int lane = threadIdx.x;
        const unsigned want = lane == 0 ? k->expect[phase][0] : lane == 1 ? k->expect[phase][1] : lane == 2 ? k->expect[phase][2] : k->expect[phase][3];
        const bool mine = lane < CH_TBMAX && ((mask >> lane) & 1u);
        const unsigned* word = counters + ((lane < CH_TBMAX ? lane : 0) * CH_REP + (blockIdx.x % CH_REP)) * CH_LINE;
        const unsigned long long t0 = chain_clock();
        for (;;) {
            const unsigned v = mine ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
            if (__all((int)(v - want) >= 0)) break;
            __builtin_amdgcn_s_sleep(2);
            if (chain_clock() - t0 > CH_WAIT_TICKS) {   // give up: the step is marked faulty, the launch still ends
                if (lane == 0) {
                    __hip_atomic_fetch_or(k->status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_word[1] = 1;
                }
                break;
            }
        }
    }
    __syncthreads();
}

// The tile's stores are on their way: drain them (every wave), then one add per covered token block.
__device__ __forceinline__ void chain_publish(unsigned* counters, bool wrote, int tb_lo, int tb_hi) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wrote && (int)threadIdx.x < (tb_hi - tb_lo) * CH_REP)   // lane = (token block, replica)
        __hip_atomic_fetch_add(counters + (tb_lo * CH_REP + threadIdx.x) * CH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// An opaque copy: the compiler cannot tell that it equals its argument, so nothing computed from it
// is hoisted out of the layer loop (where it would occupy registers for the whole launch).
__device__ __forceinline__ int chain_opaque_v(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

struct ChainTile {
    TileIdx ix;
    bool on;         // this workgroup has a tile in the phase
    unsigned mask;   // the 16-token blocks the tile covers
    int tb_lo, tb_hi;
};
// tile of workgroup b in a projection's grid (x fastest, as the hardware numbers a 3-d grid)
__device__ __forceinline__ ChainTile chain_tile(const ChainProj& g, int b, int TB, int Tb) {
    ChainTile t;
    const unsigned per_z = (unsigned)g.gx * (unsigned)g.ksplit;
    t.on = (unsigned)b < per_z * (unsigned)g.gz;
    const unsigned z = (unsigned)b / per_z, r = (unsigned)b % per_z;
    t.ix = TileIdx{(int)(r % (unsigned)g.gx), (int)(r / (unsigned)g.gx), (int)z, g.gx};
    t.tb_lo = (int)z * TB;
    t.tb_hi = min(Tb, t.tb_lo + TB);
    t.mask = t.on && t.tb_hi > t.tb_lo ? ((1u << t.tb_hi) - 1u) & ~((1u << t.tb_lo) - 1u) : 0u;
    return t;
}

__device__ __forceinline__ GemmArgs chain_gemm_args(KArgs k, const ChainProj& g, const __half* W, const __half* A) {
    GemmArgs p;
    p.W = W;
    p.nb_stride = g.nb_stride;
    p.NB = g.NB;
    p.Kb = g.Kb;
    p.A = A;
    p.Tb = k->Tb;
    p.N = g.N;
    p.T = k->T;
    p.kb_per_split = g.kb_per_split;
    p.ksplit = g.ksplit;
    p.slab = k->slab;
    p.counters = k->tile_counters;
    p.close_sc1 = 1;   // one workgroup per CU, slab from hipMalloc
#ifdef NVLLM_STAMPS
    p.stamps = nullptr;
#endif
    return p;
}
// Diagnostic builds: eight clock readings per (layer, phase, workgroup); a projection tile writes 0..6
// (gemm_tile.h), the chain adds 7 = the tile's arrival has been published.
#ifdef NVLLM_STAMPS
#define CHAIN_STAMPS(p, k, l, ph) do { if ((k)->stamps) (p).stamps = (k)->stamps + (size_t)((l) * CH_PHASES + (ph)) * gridDim.x * 8; } while (0)
#define CHAIN_STAMP_END(k, l, ph)                                                                                           \
    do {                                                                                                                    \
        if ((k)->stamps && threadIdx.x == 0)                                                                                \
            (k)->stamps[((size_t)((l) * CH_PHASES + (ph)) * gridDim.x + blockIdx.x) * 8 + 7] = chain_clock();                \
    } while (0)
#else
#define CHAIN_STAMPS(p, k, l, ph) do {} while (0)
#define CHAIN_STAMP_END(k, l, ph) do {} while (0)
#endif

// S: the tile shapes of the four projections (row blocks per tile, token blocks per group, weight
// ring depth), the head width and the pool's element type
template <typename S>
__global__ __launch_bounds__(CH_NW * 64) void k_decode_chain(ChainArgs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    unsigned* s_word = reinterpret_cast<unsigned*>(lds_dyn + S::LDS - 16);
    if (!chain_census(chain_kargs(), s_word)) return;

    const int b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int n_layers, n_items;
    {
        KArgs k = chain_kargs();
        n_layers = k->n_layers;
        n_items = k->n_items;
    }
    const bool on_a = b < n_items;            // this workgroup's attention item
    const bool att_wave = on_a && wave < 4;   // (four waves share an item)

    // The attention item of this workgroup is the same in every layer: its run descriptor is looked up once.
    int4 run = make_int4(0, 1, 0, 0);
    int tb_a0 = 0, tb_a1 = 0;
    unsigned ma = 0;
    {
        KArgs k = chain_kargs();
        if (on_a) {
            run = k->att.runs[(unsigned)b / (unsigned)k->att.NG];
            run = make_int4(__builtin_amdgcn_readfirstlane(run.x), __builtin_amdgcn_readfirstlane(run.y),
                            __builtin_amdgcn_readfirstlane(run.z), __builtin_amdgcn_readfirstlane(run.w));
            tb_a0 = run.x >> 4;
            tb_a1 = ((run.x + run.y - 1) >> 4) + 1;
            // one token per sequence: the item needs its own token's q and K/V row only; otherwise keys of
            // the step's other tokens too (a prompt chunk): everything the q/k/v phase writes
            ma = k->fresh_rule == 1 ? ((1u << tb_a1) - 1u) & ~((1u << tb_a0) - 1u) : (1u << k->Tb) - 1u;
        }
    }
    // Weight rings: each is declared where its prefetch is issued, so that no value of an earlier
    // layer stays live in registers; only the q/k/v ring crosses from one layer to the next.
    h8_t wq[S::Q_RW][S::Q_NBW];
    {
        KArgs k = chain_kargs();
        const ChainProj g = karg(&k->qkv);
        const ChainTile t = chain_tile(g, b, S::Q_TB, k->Tb);
        const GemmArgs p0 = chain_gemm_args(k, g, k->layers[0].qkv, k->a16_x);
        gemm_tile_prefetch<S::Q_NBW, CH_NW, S::Q_RW>(p0, t.ix, wq, (int)threadIdx.x, t.on);
    }
    for (int l = 0; l < n_layers; l++) {
        // ---------------------------------------------------------------- q, k, v (+ RoPE, K/V rows into the pool)
        h8_t ww[S::W_RW][S::W_NBW];
        {
            const int tid = chain_opaque_v((int)threadIdx.x);
            KArgs k = chain_kargs();
            const ChainProj g = karg(&k->qkv);
            const ChainTile t = chain_tile(g, b, S::Q_TB, k->Tb);
            GemmArgs p = chain_gemm_args(k, g, k->layers[l].qkv, k->a16_x);
            p.e.kind = GemmEpilogue::ROPE_KV;
            p.e.ss_in = k->ss_a; p.e.ss_tiles = k->ss_tiles; p.e.d_in = k->D; p.e.eps = k->eps;
            p.e.q_out = k->q; p.e.QD = k->QD; p.e.KD = k->KD; p.e.hd = k->hd;
            p.e.pos = k->pos; p.e.kv_row = k->kv_row; p.e.rope = k->rope;
            p.e.pool_k = k->pool_k; p.e.pool_v = k->pool_v; p.e.kv_elem_bytes = sizeof(typename S::KVT); p.e.bs = k->bs;
            p.e.layer_off = (long)l * k->layer_stride;
            CHAIN_STAMPS(p, k, l, PH_QKV);
            if (l > 0) chain_wait(k, chain_counter(k, l - 1, PH_DN), PH_DN, t.mask, s_word);
            bool wrote = false;
            if (t.on)
                wrote = gemm_tile<S::Q_NBW, S::Q_TB, CH_NW, S::Q_RW, 2, GemmEpilogue::ROPE_KV, typename S::KVT, true, 1>(
                    p, t.ix, lds_dyn, wq, [](int) {}, tid);
            chain_publish(chain_counter(k, l, PH_QKV), wrote, t.tb_lo, t.tb_hi);
            CHAIN_STAMP_END(k, l, PH_QKV);
        }
        // ---------------------------------------------------------------- attention over the block table
        {
            const int tid = chain_opaque_v((int)threadIdx.x);
            KArgs k = chain_kargs();
            const ChainProj gw = karg(&k->wo);
            const ChainTile tw = chain_tile(gw, b, S::W_TB, k->Tb);
            const GemmArgs pw = chain_gemm_args(k, gw, k->layers[l].wo, k->a16_att);
            // the first pages are requested before the q/k/v projection has finished, if none of them can
            // change in this launch
            Attn16Wave<S::HD, 2, 4, true> aw;
            bool att_started = false;
            AttnKArgs ak = karg(&k->att);
            ak.layer_off = (long)l * k->layer_stride;
            aw.setup(ak, (unsigned)(on_a ? b : 0), wave & 3, tid & 63, k->fresh_rule, run);
            if (att_wave && aw.start_is_stable()) {
                aw.start();
                att_started = true;
            } else {
                aw.clear_ring();
            }
            chain_wait(k, chain_counter(k, l, PH_QKV), PH_QKV, ma, s_word);
            // The V rows of this step reach the pages as 2-byte write-through stores and q is read in
            // 8-byte pieces: sizes for which loads past the vector cache alone are not known to suffice.
            // So this phase also invalidates the CU's vector cache (agent acquire) before it reads them --
            // issued by a wave that has no page loads in flight (the fence drains its issuer's loads).
            if (threadIdx.x == 4 * 64) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed
            }
            __syncthreads();
#ifdef NVLLM_STAMPS
            unsigned long long* const st = k->stamps ? k->stamps + ((size_t)(l * CH_PHASES + PH_ATT) * gridDim.x + b) * 8 : nullptr;
            if (st && threadIdx.x == 0) st[0] = chain_clock();
#endif
            if (att_wave) {
                aw.load_q();
                if (!att_started) aw.start();
#ifdef NVLLM_STAMPS
                if (st && threadIdx.x == 0) st[1] = chain_clock();
#endif
                aw.loop();
            }
#ifdef NVLLM_STAMPS
            if (st && threadIdx.x == 0) st[3] = chain_clock();
#endif
            if (att_wave) aw.deposit(lds_dyn, wave);
            __syncthreads();
            if (att_wave) aw.merge(lds_dyn, wave);
#ifdef NVLLM_STAMPS
            if (st && threadIdx.x == 0) st[6] = chain_clock();
#endif
            chain_publish(chain_counter(k, l, PH_ATT), on_a, tb_a0, tb_a1);
#ifdef NVLLM_STAMPS
            if (st && threadIdx.x == 0) st[7] = chain_clock();
#endif
            // the output projection's weights travel while the other items finish
            gemm_tile_prefetch<S::W_NBW, CH_NW, S::W_RW>(pw, tw.ix, ww, tid, tw.on);
        }
        h8_t wg[S::G_RW][S::G_NBW], wd[S::D_RW][S::D_NBW];
        // ---------------------------------------------------------------- x += att . wo^T; x * gain -> tiles
        {
            const int tid = chain_opaque_v((int)threadIdx.x);
            KArgs k = chain_kargs();
            const ChainLayer L = k->layers[l];
            const ChainProj g = karg(&k->wo);
            const ChainTile t = chain_tile(g, b, S::W_TB, k->Tb);
            GemmArgs p = chain_gemm_args(k, g, L.wo, k->a16_att);
            p.e.kind = GemmEpilogue::RESID_A16;
            p.e.x = k->x; p.e.gain_next = L.gain_ffn; p.e.a16_out = k->a16_x; p.e.ss_out = k->ss_b;
            p.e.Kb_out = (k->D + 31) >> 5;
            CHAIN_STAMPS(p, k, l, PH_WO);
            const ChainProj gn = karg(&k->gu);
            const ChainTile tn = chain_tile(gn, b, S::G_TB, k->Tb);
            const GemmArgs pn = chain_gemm_args(k, gn, L.gateup, k->a16_x);
            chain_wait(k, chain_counter(k, l, PH_ATT), PH_ATT, t.mask, s_word);
            bool wrote = false;
            h8_t wg_new[S::G_RW][S::G_NBW];
            if (t.on)
                wrote = gemm_tile<S::W_NBW, S::W_TB, CH_NW, S::W_RW, 2, GemmEpilogue::RESID_A16, float, true, 1>(p, t.ix, lds_dyn, ww, [](int) {}, tid);
            chain_publish(chain_counter(k, l, PH_WO), wrote, t.tb_lo, t.tb_hi);
            CHAIN_STAMP_END(k, l, PH_WO);
            // the next tile's weights travel while the other workgroups finish theirs
            gemm_tile_prefetch<S::G_NBW, CH_NW, S::G_RW>(pn, tn.ix, wg_new, tid, tn.on);
#pragma unroll
            for (int s = 0; s < S::G_RW; s++)
#pragma unroll
                for (int i = 0; i < S::G_NBW; i++) wg[s][i] = wg_new[s][i];
        }
        // ---------------------------------------------------------------- silu(gate) * up -> tiles
        {
            const int tid = chain_opaque_v((int)threadIdx.x);
            KArgs k = chain_kargs();
            const ChainLayer L = k->layers[l];
            const ChainProj g = karg(&k->gu);
            const ChainTile t = chain_tile(g, b, S::G_TB, k->Tb);
            GemmArgs p = chain_gemm_args(k, g, L.gateup, k->a16_x);
            p.e.kind = GemmEpilogue::SWIGLU_A16;
            p.e.ss_in = k->ss_b; p.e.ss_tiles = k->ss_tiles; p.e.d_in = k->D; p.e.eps = k->eps;
            p.e.a16_out = k->a16_h;
            p.e.Kb_out = (k->F + 31) >> 5;
            CHAIN_STAMPS(p, k, l, PH_GU);
            const ChainProj gn = karg(&k->dn);
            const ChainTile tn = chain_tile(gn, b, S::D_TB, k->Tb);
            const GemmArgs pn = chain_gemm_args(k, gn, L.down, k->a16_h);
            chain_wait(k, chain_counter(k, l, PH_WO), PH_WO, t.mask, s_word);
            bool wrote = false;
            h8_t wd_new[S::D_RW][S::D_NBW];
            if (t.on)
                wrote = gemm_tile<S::G_NBW, S::G_TB, CH_NW, S::G_RW, 2, GemmEpilogue::SWIGLU_A16, float, true, 1>(p, t.ix, lds_dyn, wg, [](int) {}, tid);
            chain_publish(chain_counter(k, l, PH_GU), wrote, t.tb_lo, t.tb_hi);
            CHAIN_STAMP_END(k, l, PH_GU);
            gemm_tile_prefetch<S::D_NBW, CH_NW, S::D_RW>(pn, tn.ix, wd_new, tid, tn.on);
#pragma unroll
            for (int s = 0; s < S::D_RW; s++)
#pragma unroll
                for (int i = 0; i < S::D_NBW; i++) wd[s][i] = wd_new[s][i];
        }
        // ---------------------------------------------------------------- x += h . w_down^T; x * next gain -> tiles
        {
            const int tid = chain_opaque_v((int)threadIdx.x);
            KArgs k = chain_kargs();
            const ChainLayer L = k->layers[l];
            const ChainProj g = karg(&k->dn);
            const ChainTile t = chain_tile(g, b, S::D_TB, k->Tb);
            GemmArgs p = chain_gemm_args(k, g, L.down, k->a16_h);
            p.e.kind = GemmEpilogue::RESID_A16;
            p.e.x = k->x; p.e.gain_next = L.gain_next; p.e.a16_out = k->a16_x; p.e.ss_out = k->ss_a;
            p.e.Kb_out = (k->D + 31) >> 5;
            CHAIN_STAMPS(p, k, l, PH_DN);
            const bool more = l + 1 < n_layers;
            const ChainProj gn = karg(&k->qkv);
            const ChainTile tn = chain_tile(gn, b, S::Q_TB, k->Tb);
            const GemmArgs pn = chain_gemm_args(k, gn, k->layers[more ? l + 1 : l].qkv, k->a16_x);
            chain_wait(k, chain_counter(k, l, PH_GU), PH_GU, t.mask, s_word);
            bool wrote = false;
            h8_t wq_new[S::Q_RW][S::Q_NBW];
            if (t.on)
                wrote = gemm_tile<S::D_NBW, S::D_TB, CH_NW, S::D_RW, 2, GemmEpilogue::RESID_A16, float, true, 1>(p, t.ix, lds_dyn, wd, [](int) {}, tid);
            chain_publish(chain_counter(k, l, PH_DN), wrote, t.tb_lo, t.tb_hi);
            CHAIN_STAMP_END(k, l, PH_DN);
            gemm_tile_prefetch<S::Q_NBW, CH_NW, S::Q_RW>(pn, tn.ix, wq_new, tid, more && tn.on);
#pragma unroll
            for (int s = 0; s < S::Q_RW; s++)
#pragma unroll
                for (int i = 0; i < S::Q_NBW; i++) wq[s][i] = wq_new[s][i];
        }
    }
}

// ------------------------------------------------------------------ shapes
template <int QN, int QT, int QR, int WN, int WT, int WR, int GN, int GT_, int GR, int DN, int DT, int DR, int HD_, typename KVT_>
struct ChainShape {
    static constexpr int Q_NBW = QN, Q_TB = QT, Q_RW = QR, W_NBW = WN, W_TB = WT, W_RW = WR;
    static constexpr int G_NBW = GN, G_TB = GT_, G_RW = GR, D_NBW = DN, D_TB = DT, D_RW = DR, HD = HD_;
    typedef KVT_ KVT;
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int LDS = cmax(cmax(cmax(gemm_tile_lds<QN, QT, CH_NW>(), gemm_tile_lds<WN, WT, CH_NW>()),
                                         cmax(gemm_tile_lds<GN, GT_, CH_NW>(), gemm_tile_lds<DN, DT, CH_NW>())),
                                    attn16_lds<HD_>()) + 16;
};
// 49..64 tokens at the TinyLlama widths (dim 2048, hidden 5632, 32 / 4 heads of 64)
typedef ChainShape<2, 2, 8, 2, 1, 8, 6, 2, 4, 4, 2, 4, 64, __half> ShapeWide64;
// up to 32 tokens at narrow widths (dim 512, hidden 1376, heads of 64): token groups of 16 everywhere but gate/up
typedef ChainShape<2, 1, 4, 2, 1, 4, 4, 2, 4, 2, 1, 4, 64, __half> ShapeNarrow32;

template <typename S>
static bool shape_fits(const GemmPlan& q, const GemmPlan& w, const GemmPlan& g, const GemmPlan& d, int hd) {
    return q.nbw == S::Q_NBW && q.TB == S::Q_TB && w.nbw == S::W_NBW && w.TB == S::W_TB && g.nbw == S::G_NBW && g.TB == S::G_TB &&
           d.nbw == S::D_NBW && d.TB == S::D_TB && hd == S::HD && q.nw == CH_NW && w.nw == CH_NW && g.nw == CH_NW && d.nw == CH_NW;
}

template <typename S>
static void launch_shape(const ChainArgs& c, int grid) {
    static bool enabled = false;
    if (!enabled) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_chain<S>), hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS));
        enabled = true;
    }
    NVLLM_LAUNCH((k_decode_chain<S>), dim3(grid), dim3(CH_NW * 64), S::LDS, c);
}

static ChainProj proj_of(const GemmPlan& pl, int N, int K, long nb_stride) {
    ChainProj g;
    g.NB = cdiv(N, 16);
    g.Kb = cdiv(K, 32);
    g.N = N;
    g.nb_stride = nb_stride;
    g.kb_per_split = pl.kb_per_split;
    g.ksplit = pl.ks;
    g.gx = pl.tiles_n;
    g.gz = pl.tiles_t;
    return g;
}

// diagnostic builds: where the workgroups' clock readings of the next chain launches go
static unsigned long long* g_chain_stamps = nullptr;
extern "C" void nvllm_debug_set_chain_stamps(unsigned long long* buf) { g_chain_stamps = buf; }

static int g_chain_on = -1;   // -1: not decided yet (the environment decides at first use)
static long g_chain_launches = 0;
int chain_enabled() {
    if (g_chain_on < 0) g_chain_on = getenv("NVLLM_DECODE_CHAIN") ? (atoi(getenv("NVLLM_DECODE_CHAIN")) != 0) : 0;
    return g_chain_on;
}
extern "C" void nvllm_set_decode_chain(int on) { g_chain_on = on != 0; }
extern "C" int nvllm_get_decode_chain(void) { return chain_enabled(); }
extern "C" long nvllm_decode_chain_launches(void) { return g_chain_launches; }

// Enqueue the layers of one step as one launch; false = this step does not fit the chain (the
// caller launches the layers one kernel at a time), with nothing enqueued.
bool launch_decode_chain(const ChainStep& s) {
    if (!chain_enabled() || s.T <= 0 || s.T > CH_TBMAX * 16 || s.kv_elem_bytes != 2 || s.block_size != 16) return false;
    const int G = rt().num_cus;
    const int Tb = cdiv(s.T, 16);
    const int QD = s.H * s.hd, KD = s.KVH * s.hd;
    const long big = 1l << 30;
    const GemmPlan pq = gemm_plan(QD + 2 * KD, s.D, s.T, GemmEpilogue::ROPE_KV, 0, big, 0);
    const GemmPlan pw = gemm_plan(s.D, QD, s.T, GemmEpilogue::RESID_A16, 0, big, 0);
    const GemmPlan pg = gemm_plan(2 * s.F, s.D, s.T, GemmEpilogue::SWIGLU_A16, 0, big, 0);
    const GemmPlan pd = gemm_plan(s.D, s.F, s.T, GemmEpilogue::RESID_A16, 0, big, 0);
    auto wgs = [](const GemmPlan& p) { return (long)p.tiles_n * p.ks * p.tiles_t; };
    if (wgs(pq) > G || wgs(pw) > G || wgs(pg) > G || wgs(pd) > G) return false;
    if (s.D % 32 || s.F % 32 || (2 * s.F / 16) % 2) return false;
    const int CT = attn_tokens_per_item(s.H, s.KVH), GT = 16 / CT, NG = s.H / GT;
    const long items = (long)s.n_runs * NG;
    if (items > G || items <= 0) return false;
    // the pool slice of one layer and every activation buffer are addressed with 32-bit byte offsets
    if ((double)s.layer_stride * 2.0 >= 2147483648.0 || (double)s.T * (2 * s.F) * 8.0 >= 2147483648.0) return false;
    const bool wide = shape_fits<ShapeWide64>(pq, pw, pg, pd, s.hd), narrow = shape_fits<ShapeNarrow32>(pq, pw, pg, pd, s.hd);
    if (!wide && !narrow) return false;

    ChainArgs c{};
    c.layers = (const ChainLayer*)s.layers;
    c.n_layers = s.n_layers;
    c.T = s.T;
    c.Tb = Tb;
    c.qkv = proj_of(pq, QD + 2 * KD, s.D, (long)(s.D / 32) * TILE_HALVES);
    c.wo = proj_of(pw, s.D, QD, (long)(QD / 32) * TILE_HALVES);
    c.gu = proj_of(pg, 2 * s.F, s.D, (long)(s.D / 32) * TILE_HALVES);
    c.dn = proj_of(pd, s.D, s.F, (long)(s.F / 32) * TILE_HALVES);
    c.x = s.x; c.a16_x = s.a16_x; c.a16_att = s.a16_att; c.a16_h = s.a16_h; c.ss_a = s.ss_a; c.ss_b = s.ss_b; c.q = s.q;
    c.ss_tiles = gemm_feature_tiles(s.D);
    c.D = s.D; c.F = s.F; c.eps = s.eps;
    c.QD = QD; c.KD = KD; c.hd = s.hd;
    c.pos = s.pos; c.kv_row = s.kv_row; c.rope = s.rope;
    c.pool_k = s.pool_k; c.pool_v = s.pool_v; c.bs = s.block_size; c.layer_stride = s.layer_stride;
    AttnKArgs& k = c.att;
    k.q = s.q;
    k.pool_k = s.pool_k;
    k.pool_v = s.pool_v;
    k.layer_off = 0;
    k.rowstride = (long)s.KVH * s.hd;
    k.page = 16;
    k.tables = s.tables;
    k.runs = (const int4*)s.runs;
    k.n_runs = s.n_runs;
    k.H = s.H; k.KVH = s.KVH; k.GT = GT; k.CT = CT; k.NG = NG;
    k.group = s.H / s.KVH;
    k.gt_shift = 0;
    while ((1 << k.gt_shift) < GT) k.gt_shift++;
    k.nsplit = 1;
    k.scale = nvllm_host_attn_scale(s.hd);
    k.out_f32 = nullptr;
    k.out_a16 = s.a16_att;
    k.Kb_out = cdiv(QD, 32);
    k.part_acc = nullptr;
    k.part_ml = nullptr;
    k.bs_shift = 0;
    c.fresh_rule = s.one_token_per_seq ? 1 : 0;
    c.n_items = (int)items;
    // split-K partial tiles of the down projection
    c.slab = nullptr;
    for (const GemmPlan* p : {&pq, &pw, &pg, &pd})
        if (p->ks > 1) {
            const size_t need = (size_t)p->tiles_n * p->tiles_t * p->ks * (size_t)(p->nbw / 2) * 2 * p->TB * 64 * 16;
            c.slab = (float*)ws_get(WS_SLAB, need);   // (one buffer: the phases run one after the other)
        }
    c.tile_counters = gemm_tile_counters();
    c.sync = s.sync;
    c.status = s.status;
    for (int tb = 0; tb < CH_TBMAX; tb++) {
        const bool live = tb < Tb;
        c.expect[PH_QKV][tb] = live ? cdiv(c.qkv.NB, pq.nbw) : 0;
        c.expect[PH_WO][tb] = live ? cdiv(c.wo.NB, pw.nbw) : 0;
        c.expect[PH_GU][tb] = live ? cdiv(c.gu.NB, pg.nbw) : 0;
        c.expect[PH_DN][tb] = live ? cdiv(c.dn.NB, pd.nbw) : 0;
        c.expect[PH_ATT][tb] = live ? (unsigned)s.runs_in_tb[tb] * NG : 0;
    }
    c.stamps = g_chain_stamps;
    ProfScope ps(PC_CHAIN, s.bytes, s.flops);
    if (wide) launch_shape<ShapeWide64>(c, G);
    else launch_shape<ShapeNarrow32>(c, G);
    g_chain_launches++;
    return true;
}

}  // namespace nvllm
