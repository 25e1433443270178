// gemm.hip -- projections: out[T, N] = act[T, K] x W[N, K]^T with fp16 weights
// streamed once from HBM in tile order, fp32 activations carried as fp16 hi/lo
// tile pairs, fp32 accumulation on the matrix cores, and the element-wise work
// that follows each projection fused into its epilogue.
//
// Shape of the problem at decode: T <= 64 tokens, so every weight byte is used
// for only 2*T flops and the kernel is bound by how fast bytes reach the CUs.
// One kernel template, instantiated per tile shape (one launch per projection):
//   grid.x  tiles of NBW 16-row blocks of W
//   grid.y  split of K across workgroups (the last resort: its partial sums meet
//           through memory, a chain of dependent round trips)
//   grid.z  groups of TB*16 tokens
//   inside a workgroup the NW waves take the K-steps round-robin, each with all
//   NBW row blocks, and their accumulators are summed through LDS at the end.
// launch_gemm() picks (NBW, TB, K split) per projection so that one wave of
// workgroups covers the chip: token groups instead of K slices for short-K narrow
// projections, 96 x 32 tiles for the wide one, 64-feature tiles x 32-token groups x
// a shallow K split for long K; grids with token groups are padded to a multiple of 8
// so that the groups of one weight tile run behind the same XCD's L2.
// Per K-step (32 k) a wave loads NBW weight tiles (1 KiB each, one fully
// coalesced dwordx4 per lane, non-temporal) and 2*TB activation tiles (L2), and
// issues 2*NBW*TB v_mfma_f32_16x16x32_f16.  Weight loads run RW steps ahead,
// activation loads RA steps ahead, in register rings.
//
// Split-K is closed INSIDE the launch: every K-slice writes its partial tile to a
// slab with write-through (sc1) stores, drains them, and one lane takes a ticket
// from a per-tile counter; the workgroup that draws the last ticket acquires,
// adds the slabs in slice order (so the sum does not depend on who arrived last)
// and runs the epilogue.  That removes a kernel boundary and a slab re-read per
// projection, and lets the epilogue see complete sums:
//   F32         logits (scaled by the token's 1/rms)
//   SWIGLU_A16  silu(gate) * up             -> fp16 hi/lo tiles for w_down
//   RESID_A16   x += result; x * next_gain  -> tiles for the next projection,
//               plus sums of x^2 per 32 features from which the consumer forms 1/rms
//   ROPE_KV     rotate q,k; q -> fp32 rows for attention; k,v -> block pool
// RMSNorm is thereby split in two: the producer stores x*g (no reduction needed),
// the consumer multiplies its result by 1/rms (a per-token scalar commutes with
// the projection).
#include <vector>

#include "gemm_tile.h"

namespace nvllm {

#ifndef NVLLM_RW
#define NVLLM_RW 4
#endif
#ifndef NVLLM_RA
#define NVLLM_RA 2
#endif

// One workgroup = one tile (gemm_tile.h); weight ring of NVLLM_RW K-steps, activation ring of NVLLM_RA.
template <int NBW, int TB, int NW, int EPI, typename KVT, int RW = NVLLM_RW>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    h8_t w[RW][NBW];
    const TileIdx ix{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x};
    gemm_tile<NBW, TB, NW, RW, NVLLM_RA, EPI, KVT, false, 0>(p, ix, lds_dyn, w, [](int) {}, (int)threadIdx.x);
}

// ------------------------------------------------------------------ launch
constexpr int GEMM_NBW = 4;
constexpr int GEMM_NW = 8;

int gemm_feature_tiles(int N) { return cdiv(N, 32); }

int gemm_pick_ksplit(int N, int K, int T) {
    const int NB = cdiv(N, 16), Kb = cdiv(K, 32);
    const int groups = cdiv(NB, GEMM_NBW) * cdiv(cdiv(T, 16), 4);
    static const int target_wgs = getenv("NVLLM_GEMM_WGS") ? atoi(getenv("NVLLM_GEMM_WGS")) : 256;
    int ks = (target_wgs + groups / 2) / groups;      // aim for a few hundred workgroups
    const int ks_max = Kb / GEMM_NW > 0 ? Kb / GEMM_NW : 1;  // at least one K-step per wave
    if (ks > ks_max) ks = ks_max;
    // never a few workgroups more than CUs: they would run alone in a second round (86 groups x 3 slices
    // = 258 workgroups took as long as two rounds)
    if (ks > 1 && groups * ks > rt().num_cus) ks = rt().num_cus / groups;
    if (ks < 1) ks = 1;
    const int per = cdiv(Kb, ks);
    return cdiv(Kb, per);
}

// arrival counters: zero when allocated, every launch leaves them zero again
static unsigned* g_counters = nullptr;
#ifdef NVLLM_STAMPS
// Diagnostic builds only (tools/gemm_stamps.py): launch i of the projection kernel records its
// workgroups' clock readings in slot i % cap of the harness's buffer, STAMP_WGS x 8 words per slot.
constexpr int STAMP_WGS = 1024;
static unsigned long long* g_gemm_stamps = nullptr;
static int g_stamp_cap = 0, g_stamp_next = 0;
extern "C" void nvllm_debug_set_gemm_stamps(unsigned long long* buf, int cap) {
    g_gemm_stamps = buf;
    g_stamp_cap = cap;
    g_stamp_next = 0;
}
extern "C" int nvllm_debug_gemm_stamp_launches(void) { return g_stamp_next; }
#endif
constexpr int MAX_COUNTERS = 1 << 16;

// One instantiation: its dynamic LDS is the waves' partial tiles (1 KiB per wave, row block and
// token block); sizes beyond the 64 KiB default are enabled once.
template <int NBW, int TB, int NW, int EPI, typename KVT, int RW = NVLLM_RW>
static void launch_k(const GemmArgs& a, dim3 grid) {
    constexpr int shmem = gemm_tile_lds<NBW, TB, NW>();
    static_assert(shmem <= 160 * 1024, "partial tiles must fit the CU's LDS");
    static bool enabled = false;
    if (shmem > 64 * 1024 && !enabled) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<NBW, TB, NW, EPI, KVT, RW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
        enabled = true;
    }
    NVLLM_LAUNCH((k_gemm<NBW, TB, NW, EPI, KVT, RW>), grid, dim3(NW * 64), shmem, a);
}

template <int NBW, int TB, typename KVT>
static void launch_tb(const GemmArgs& a, dim3 grid) {
    switch (a.e.kind) {
        case GemmEpilogue::F32: launch_k<NBW, TB, GEMM_NW, GemmEpilogue::F32, KVT>(a, grid); break;
        case GemmEpilogue::SWIGLU_A16: launch_k<NBW, TB, GEMM_NW, GemmEpilogue::SWIGLU_A16, KVT>(a, grid); break;
        case GemmEpilogue::RESID_A16: launch_k<NBW, TB, GEMM_NW, GemmEpilogue::RESID_A16, KVT>(a, grid); break;
        default: launch_k<NBW, TB, GEMM_NW, GemmEpilogue::ROPE_KV, KVT>(a, grid); break;
    }
}

// The decomposition of a decode-sized projection (T below the dense-tile threshold): row blocks per
// tile, token blocks per group, K slices, and the grid they make.  `slab_cap` > 0: a caller's slab of
// that many bytes (a split whose partial tiles do not fit it is not taken).
GemmPlan gemm_plan(int N, int K, int T, int epi_kind, int force_ksplit, long max_counters, size_t slab_cap) {
    const int NB = cdiv(N, 16), Kb = cdiv(K, 32), Tb = cdiv(T, 16);
    int TB = T <= 16 ? 1 : (T <= 32 ? 2 : 4);
    int nbw = GEMM_NBW;
    int ks = force_ksplit > 0 ? force_ksplit : gemm_pick_ksplit(N, K, T);
    // Narrow projections with a short K (wo, qkv at decode): instead of cutting K across
    // workgroups -- whose partial sums then meet through memory, a chain of dependent
    // round trips -- cut the TOKENS: 32-feature tiles x 16-token groups, every workgroup
    // sweeps the whole K and finishes its own outputs.  The weight tile is read by the
    // token groups' workgroups (one read from HBM, the rest from cache); worth it while a
    // full sweep is short.
    static const int tsplit_env = getenv("NVLLM_GEMM_TSPLIT") ? atoi(getenv("NVLLM_GEMM_TSPLIT")) : 1;
    const bool tsplit = tsplit_env && force_ksplit <= 0 && ks > 1 && TB > 1 && Kb <= 64 && Tb <= 8 &&
                        epi_kind != GemmEpilogue::SWIGLU_A16;
    if (tsplit) {
        // 16-token groups unless that makes more workgroups than CUs (a CU with two of
        // them takes twice as long: measured 12.3 us against 10.0 for qkv); then 32-token groups
        const bool wide = cdiv(NB, 2) * Tb > rt().num_cus;
        TB = wide ? 2 : 1;
        nbw = 2;
        ks = 1;
    }
    // Long-K projections that still need a K split (w_down): half the split, two 32-token
    // groups -- the same number of workgroups, but each tile's reduction reads half as
    // many, half as large partials (measured 14.9 -> 13.8 us; more workgroups than CUs,
    // or a shallower split, measured slower).
    if (!tsplit && force_ksplit <= 0 && TB == 4 && ks >= 4 && ks % 2 == 0 && Tb <= 4) {
        TB = 2;
        ks /= 2;
    }
    // The wide gate/up projection at full decode width: 64 x 64 tiles give 176 workgroups
    // (80 CUs idle) pulling 768 KiB each; 96-feature x 32-token tiles give 236 workgroups
    // pulling 640 KiB each, still in one wave of workgroups.
    static const int gu_env = getenv("NVLLM_GEMM_GU96") ? atoi(getenv("NVLLM_GEMM_GU96")) : 1;
    const bool gu96 = gu_env && !tsplit && force_ksplit <= 0 && epi_kind == GemmEpilogue::SWIGLU_A16 && ks == 1 &&
                      Tb > 2 && Tb <= 4 && cdiv(NB, 6) * 2 <= rt().num_cus && cdiv(NB, 4) < rt().num_cus;
    if (gu96) {
        TB = 2;
        nbw = 6;
    }
    int tiles_n = cdiv(NB, nbw);
    const int tiles_t = cdiv(Tb, TB);
    // Token groups of one tile read the same weights: with workgroups dealt round-robin to
    // the 8 XCDs, a grid width that is a multiple of 8 puts them behind the same L2 (speed
    // only; surplus workgroups exit at once).
    if (tiles_t > 1 && ks == 1) tiles_n = (tiles_n + 7) & ~7;
    if ((long)tiles_n * tiles_t > max_counters) ks = 1;
    int kb_per_split = cdiv(Kb, ks);
    ks = cdiv(Kb, kb_per_split);
    const size_t slab_tile_bytes = (size_t)(nbw / 2) * 2 * TB * 64 * 16;
    // a caller's slab that cannot hold the partial tiles: every workgroup sweeps the whole K instead
    if (slab_cap > 0 && ks > 1 && (size_t)tiles_n * tiles_t * ks * slab_tile_bytes > slab_cap) {
        ks = 1;
        kb_per_split = Kb;
    }
    GemmPlan pl;
    pl.nbw = nbw;
    pl.TB = TB;
    pl.ks = ks;
    pl.kb_per_split = kb_per_split;
    pl.tiles_n = tiles_n;
    pl.tiles_t = tiles_t;
    // more tiles than CUs without a K split (the classifier): four waves per workgroup, so that two
    // workgroups share a CU and one's start-up overlaps the other's streaming (41 -> 35.5 us)
    pl.nw = (epi_kind == GemmEpilogue::F32 && nbw == GEMM_NBW && TB == 4 && ks == 1 && (long)tiles_n * tiles_t > rt().num_cus) ? 4 : GEMM_NW;
    return pl;
}

int launch_gemm(const WeightEntry& w, const __half* tiles, int N, const __half* a16, int T,
                const GemmEpilogue& epi, int force_ksplit, int prof_cls, const GemmWorkspace* wsp) {
    if (T <= 0 || N <= 0) return 1;
    ensure_init();
    // the caller's workspace (counters already zero), or the library's own grow-only buffers
    if (!wsp) gemm_tile_counters();
    const long max_counters = wsp ? wsp->n_counters : MAX_COUNTERS;
    GemmArgs a;
    a.W = tiles;
    a.nb_stride = w.nb_stride;
    a.NB = cdiv(N, 16);
    a.Kb = w.Kb;
    a.A = a16;
    a.Tb = cdiv(T, 16);
    a.N = N;
    a.T = T;
    a.e = epi;
    if (epi.kind == GemmEpilogue::SWIGLU_A16) {
        if ((N / 2) % 32 != 0 || a.NB % 2 != 0) NVLLM_FATAL("fused gate/up projection needs hidden_dim %% 32 == 0, got %d", N / 2);
        a.e.Kb_out = cdiv(N / 2, 32);
    } else if (epi.kind == GemmEpilogue::RESID_A16) {
        if (N % 32 != 0) NVLLM_FATAL("fused residual epilogue needs a width that is a multiple of 32, got %d", N);
        a.e.Kb_out = cdiv(N, 32);
    }
    const double out_bytes = epi.kind == GemmEpilogue::SWIGLU_A16 ? (double)T * N * 2.0
                             : epi.kind == GemmEpilogue::RESID_A16 ? (double)T * N * 12.0
                                                                   : (double)T * N * 4.0;
    const double bytes = (double)N * w.K * 2.0 + (double)T * w.K * 4.0 + out_bytes;
    // Hundreds of tokens (prefill chunks): the contraction is dense enough for the
    // LDS-shared 128 x 128 tiles of gemm_prefill.hip.
    static const int pf_min_t = getenv("NVLLM_GEMM_PREFILL_T") ? atoi(getenv("NVLLM_GEMM_PREFILL_T")) : 256;
    // ... and already for a wide decode batch (about 128 rows) when the projection is wide enough that its
    // 128 x 128 tiles occupy most of the chip without a K split (the 7B gate/up projection at batch 128:
    // 172 tiles, 78 us against 104 us for the streaming kernel in two 64-token groups).
    // A projection half that wide (7B q/k/v: 96 tiles) takes it with K cut in two: 192 workgroups, and
    // the tile's second half arrives through a 64 KiB slab (70 -> about 45 us).
    const int tiles128 = cdiv(a.NB, 8);
    const bool wide_decode = T >= 112 && T <= 128 && tiles128 * 5 >= rt().num_cus * 3;
    const bool wide_decode_k2 = T >= 112 && T <= 128 && !wide_decode && tiles128 * 2 * 5 >= rt().num_cus * 3 &&
                                tiles128 * 2 <= rt().num_cus && a.Kb >= 64 && tiles128 <= max_counters;
    if (force_ksplit <= 0 && (T >= pf_min_t || wide_decode || wide_decode_k2) && !wsp) {
        a.ksplit = 1;
        a.kb_per_split = a.Kb;
        a.slab = nullptr;
        a.counters = g_counters;
        a.close_sc1 = 0;
        if (wide_decode_k2 && T < pf_min_t) a.ksplit = 2;   // every tile's K in two (slab: the launcher's)
#ifdef NVLLM_STAMPS
        a.stamps = nullptr;
        if (g_gemm_stamps && g_stamp_cap > 0) {
            a.stamps = g_gemm_stamps + (size_t)(g_stamp_next % g_stamp_cap) * STAMP_WGS * 8;
            g_stamp_next++;
        }
#endif
        a.e.amax_pairs = nullptr;   // the dense-tile kernel does not emit pairs: the rows are scanned afterwards
        {
            ProfScope ps(prof_cls, bytes, 2.0 * T * (double)N * w.K);
            launch_gemm_prefill(a);
        }
        if (epi.kind == GemmEpilogue::F32 && epi.amax_out) launch_argmax_ld(epi.amax_out, epi.out, T, N, epi.ldo, epi.amax_out_host);
        return a.ksplit;
    }
    const GemmPlan pl = gemm_plan(N, w.K, T, epi.kind, force_ksplit, max_counters, wsp ? wsp->slab_bytes : 0);
    const int TB = pl.TB, nbw = pl.nbw, tiles_n = pl.tiles_n, tiles_t = pl.tiles_t;
    int ks = pl.ks;
    a.kb_per_split = pl.kb_per_split;
    const size_t slab_tile_bytes = (size_t)(nbw / 2) * 2 * TB * 64 * 16;
    a.ksplit = ks;
    a.counters = wsp ? wsp->counters : g_counters;
#ifdef NVLLM_STAMPS
    a.stamps = nullptr;
    if (g_gemm_stamps && g_stamp_cap > 0) {
        a.stamps = g_gemm_stamps + (size_t)(g_stamp_next % g_stamp_cap) * STAMP_WGS * 8;
        g_stamp_next++;
    }
#endif
    a.slab = nullptr;
    if (ks > 1) {
        const size_t need = (size_t)tiles_n * tiles_t * ks * slab_tile_bytes;
        a.slab = wsp ? wsp->slab : (float*)ws_get(WS_SLAB, need);
    }
    const dim3 grid(tiles_n, ks, tiles_t);
    static const int sc1_env = getenv("NVLLM_GEMM_CLOSE_SC1") ? atoi(getenv("NVLLM_GEMM_CLOSE_SC1")) : 1;
    a.close_sc1 = sc1_env && ks > 1 && !wsp && (long)tiles_n * ks * tiles_t <= rt().num_cus;
    const bool amax = epi.kind == GemmEpilogue::F32 && epi.amax_out != nullptr;
    if (amax && (!epi.amax_pairs || epi.amax_ld < T)) NVLLM_FATAL("fused arg-max needs a pair buffer of at least %d rows per tile", T);
    {
    ProfScope ps(prof_cls, bytes, 2.0 * T * (double)N * w.K);
    const bool kv16 = epi.kind == GemmEpilogue::ROPE_KV && epi.kv_elem_bytes == 2;
    if (nbw == 6) {
        launch_k<6, 2, GEMM_NW, GemmEpilogue::SWIGLU_A16, float>(a, grid);
    } else if (nbw == 2) {   // token-split configuration
        if (kv16) { if (TB == 1) launch_tb<2, 1, __half>(a, grid); else launch_tb<2, 2, __half>(a, grid); }
        else { if (TB == 1) launch_tb<2, 1, float>(a, grid); else launch_tb<2, 2, float>(a, grid); }
    } else if (pl.nw == 4) {   // the classifier's four-wave form
        launch_k<GEMM_NBW, 4, 4, GemmEpilogue::F32, float>(a, grid);
    } else if (kv16) {
        if (TB == 1) launch_tb<GEMM_NBW, 1, __half>(a, grid);
        else if (TB == 2) launch_tb<GEMM_NBW, 2, __half>(a, grid);
        else launch_tb<GEMM_NBW, 4, __half>(a, grid);
    } else {
        if (TB == 1) launch_tb<GEMM_NBW, 1, float>(a, grid);
        else if (TB == 2) launch_tb<GEMM_NBW, 2, float>(a, grid);
        else launch_tb<GEMM_NBW, 4, float>(a, grid);
    }
    }
    // the tiles' (maximum, index) pairs -> one index per row (tiles that are grid padding wrote nothing)
    if (amax) launch_argmax_tiles(epi.amax_out, epi.amax_pairs, cdiv(a.NB, nbw), T, epi.amax_ld, epi.amax_out_host);
    return ks;
}

unsigned* gemm_tile_counters() {
    if (!g_counters) {
        HIP_CHECK(hipMalloc((void**)&g_counters, MAX_COUNTERS * sizeof(unsigned)));
        HIP_CHECK(hipMemsetAsync(g_counters, 0, MAX_COUNTERS * sizeof(unsigned), rt().stream));
    }
    return g_counters;
}

void gemm_release() {
    if (g_counters) HIP_CHECK(hipFree(g_counters));
    g_counters = nullptr;
}

// --------------------------------------------- raw fp32 weights (ops.h matmul)
// One wave per output feature, eight tokens at a time; used only when matmul() is
// handed a plain fp32 [N, K] device buffer instead of a load_model handle.
__global__ __launch_bounds__(256) void k_matmul_f32w(float* __restrict__ out, const float* __restrict__ in,
                                                     const float* __restrict__ w, int T, int K, int N) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t0 = blockIdx.y * 8;
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.f;
    const float* wr = w + (size_t)n * K;
    for (int k = lane; k < K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int t = min(t0 + i, T - 1);
            acc[i] = fmaf(in[(size_t)t * K + k], wv, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && t0 + i < T) out[(size_t)(t0 + i) * N + n] = v;
    }
}

void launch_matmul_f32w(float* out, const float* in, const float* w, int T, int K, int N) {
    if (T <= 0 || N <= 0) return;
    ProfScope ps(PC_GEMM_OTHER, (double)N * K * 4.0 + (double)T * (K + N) * 4.0, 2.0 * T * (double)N * K);
    NVLLM_LAUNCH(k_matmul_f32w, dim3(cdiv(N, 4), cdiv(T, 8)), dim3(256), 0, out, in, w, T, K, N);
}

}  // namespace nvllm
