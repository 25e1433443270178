// gemm_tile.h -- one output tile of a decode-sized projection as a device function: what a
// workgroup of the projection kernel does (gemm.hip describes the decomposition and the
// epilogues).  Two callers:
//   k_gemm           one launch per projection; operands written by earlier launches (COH = false)
//   k_decode_chain   every projection and attention of a decode step in ONE launch; a tile's
//                    activations were written by other workgroups of the same launch (COH = true)
//
// COH = true changes how bytes that another workgroup of the SAME launch wrote or will read move:
// every such store is write-through (sc1) and every such load bypasses this CU's vector cache
// (sc1), so that the caller can hand tiles over with a drained counter add on the producer and a
// relaxed poll on the consumer, without cache write-backs or invalidates.  Weights, gains, RoPE
// angles and the integer arrays of the step are read-only during the launch and stay on plain loads.
#pragma once

#include "gemm_common.h"

namespace nvllm {

// Weight tiles are read once per launch: non-temporal loads (a build knob for A/B runs)
#ifdef NVLLM_W_PLAIN
#define NVLLM_WLOAD(p) (*(p))
#else
#define NVLLM_WLOAD(p) __builtin_nontemporal_load(p)
#endif

constexpr int AUX_SC1 = 16;   // cache-policy bits of the raw buffer builtins on gfx950: bit 4 = sc1

typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

// position of a tile in its projection's grid (what blockIdx / gridDim are to k_gemm)
struct TileIdx {
    int x, y, z;   // feature tile, K slice, token group
    int gx;        // feature tiles per token group (grid width, padding included)
};

// bytes of dynamic LDS a tile needs: the waves' partial tiles, then four small arrays
template <int NBW, int TB, int NW>
constexpr int gemm_tile_lds() {
    return NW * NBW * TB * 64 * 16 + (TB * 16 + NW * 64 + 2 * TB * 16) * 4 + 16;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// 4 consecutive k of token t into hi/lo tiles with write-through stores (n % 4 == 0), 16 bytes per
// lane: the lane that holds k..k+3 and the lane 16 further on, which holds k+4..k+7 of the same
// token, are both active here; the first gets both hi halves and stores the hi plane's 16-byte
// chunk, the second both lo halves and stores the lo plane's (narrow write-through stores each cost
// a transaction of their own on the way to memory).  `rs` covers the whole a16 buffer.
__device__ __forceinline__ void store_a16_4_sc1(__amdgpu_buffer_rsrc_t rs, int Kb_out, int t, int n, const float* y, int lane) {
    alignas(8) __half hi[4], lo[4];
    split16_4(y, hi, lo);
    const u2_t H = *reinterpret_cast<const u2_t*>(hi), L = *reinterpret_cast<const u2_t*>(lo);
    const bool first = (lane & 16) == 0;          // k % 8 == 0
    const u2_t send = first ? L : H;
    u2_t recv;
    recv[0] = (unsigned)__shfl_xor((int)send[0], 16, 64);
    recv[1] = (unsigned)__shfl_xor((int)send[1], 16, 64);
    const u4_t out = first ? u4_t{H[0], H[1], recv[0], recv[1]} : u4_t{recv[0], recv[1], L[0], L[1]};
    const int k8 = n & ~7;
    const int tile = ((t >> 4) * Kb_out + (k8 >> 5)) * 2 + (first ? 0 : 1);
    const int off = ((t & 15) + 16 * ((k8 >> 3) & 3)) * 8;
    __builtin_amdgcn_raw_buffer_store_b128(out, rs, (tile * TILE_HALVES + off) * 2, 0, AUX_SC1);
}

template <typename KVT>
__device__ __forceinline__ void kv_put4_sc1(KVT* dst, const float* y);
template <>
__device__ __forceinline__ void kv_put4_sc1<float>(float* dst, const float* y) {
    alignas(8) float a[2] = {y[0], y[1]}, b[2] = {y[2], y[3]};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), *reinterpret_cast<unsigned long long*>(a),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + 2), *reinterpret_cast<unsigned long long*>(b),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <>
__device__ __forceinline__ void kv_put4_sc1<__half>(__half* dst, const float* y) {
    alignas(8) __half h[4] = {kv_sat_half(y[0]), kv_sat_half(y[1]), kv_sat_half(y[2]), kv_sat_half(y[3])};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), *reinterpret_cast<unsigned long long*>(h),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename KVT>
__device__ __forceinline__ void kv_put4_strided_sc1(KVT* dst, const float* y);
template <>
__device__ __forceinline__ void kv_put4_strided_sc1<float>(float*, const float*) {}
template <>
__device__ __forceinline__ void kv_put4_strided_sc1<__half>(__half* dst, const float* y) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const __half h = kv_sat_half(y[e]);
        __hip_atomic_store(reinterpret_cast<unsigned short*>(dst + 4 * e), *reinterpret_cast<const unsigned short*>(&h),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The first RW K-steps of a wave's weight stream: they depend on nothing the step computes, so a
// caller that knows its next tile may issue them long before that tile's activations exist.
// Must be followed by gemm_tile<..., PRE = true> with the same arguments.
// `tid` = threadIdx.x (a caller that runs many tiles in a loop passes it through an opaque copy, so that
// the compiler does not hoist every tile's per-lane index arithmetic out of that loop and keep it in registers).
// part / nparts: only the ring entries s with s * nparts / RW == part (a caller that spreads the
// requests over several places calls once per part); part < 0: all of them.
template <int NBW, int NW, int RW>
__device__ __forceinline__ void gemm_tile_prefetch(const GemmArgs& p, const TileIdx ix, h8_t (&w)[RW][NBW], const int tid, const bool on = true,
                                                   const int part = -1, const int nparts = 1) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb0 = ix.x * NBW;
    const int kb_begin = ix.y * p.kb_per_split;
    const int kb_end = min(p.Kb, kb_begin + p.kb_per_split);
    const int first = kb_begin + wave;
    // `on` = the workgroup has a tile at all; every ring entry is written either way (zeros where there
    // is nothing to load), so that no entry carries a value over from an earlier use of the registers
    const int nsteps = on && nb0 < p.NB && first < kb_end ? (kb_end - first + NW - 1) / NW : 0;
    const h8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < RW; s++) {
        if (part >= 0 && (s * nparts) / RW != part) continue;
        if (s < nsteps) {
            const int kb = first + NW * s;
#pragma unroll
            for (int i = 0; i < NBW; i++)
                w[s][i] = NVLLM_WLOAD(reinterpret_cast<const h8_t*>(p.W + (size_t)min(nb0 + i, p.NB - 1) * p.nb_stride) + lane +
                                      (size_t)kb * 64);
        } else {
#pragma unroll
            for (int i = 0; i < NBW; i++) w[s][i] = zero;
        }
    }
}

// One tile.  Returns true when this workgroup wrote the tile's results (false: grid padding, or a
// K slice that was not the last to arrive).  `slot_free(r)` is called exactly once for every r in
// [0, RW), as soon as the wave will not load ring entry r again for this tile -- inside the K loop,
// spread over its last RW steps: the place to start loads for later work, a few at a time.
// PRE: 0 = the tile requests its first RW K-steps of weights itself; 1 = the caller has (gemm_tile_prefetch);
// 2 = the caller says at run time (`prefetched`, uniform over the workgroup).
template <int NBW, int TB, int NW, int RW, int RA, int EPI, typename KVT, bool COH, int PRE, typename Hook>
__device__ __forceinline__ bool gemm_tile(const GemmArgs& p, const TileIdx ix, unsigned char* lds_dyn, h8_t (&w)[RW][NBW],
                                          Hook&& slot_free, const int tid, const bool prefetched = true) {
    static_assert(NBW % 2 == 0, "row blocks are reduced in pairs");
    static_assert(RW % RA == 0, "the activation ring must divide the weight ring");
    constexpr int ROUNDS = NBW / 2;    // LDS reduction rounds, two row blocks each
    constexpr int ITEMS = 2 * TB * 64; // float4 results per round
    constexpr int WITEMS = ROUNDS * ITEMS;                        // float4 results of the tile: item w = round * ITEMS + e
    constexpr int FPT = (WITEMS + NW * 64 - 1) / (NW * 64);     // items per thread: thread tid takes w = tid + s * NW*64
    constexpr int ITEMS_SHIFT = TB == 1 ? 7 : (TB == 2 ? 8 : 9);
    static_assert((1 << ITEMS_SHIFT) == ITEMS, "2, 4 or 8 row-block x token-block pairs of 64 lanes per round");
    constexpr int PARTS = (NW * 64) / (TB * 16);
    // the waves' partial tiles, all row blocks at once: red[wave][row block][token block][lane]
    typedef f4_t RedTile[NBW][TB][64];
    RedTile* red = reinterpret_cast<RedTile*>(lds_dyn);
    float* s_rinv = reinterpret_cast<float*>(lds_dyn + NW * NBW * TB * 64 * 16);   // [TB * 16]
    float (*s_part)[TB * 16] = reinterpret_cast<float (*)[TB * 16]>(s_rinv + TB * 16);   // [PARTS][TB * 16]
    unsigned* s_ticket = reinterpret_cast<unsigned*>(s_rinv + TB * 16 + NW * 64 + 2 * TB * 16);

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb0 = ix.x * NBW;
    if (nb0 >= p.NB) {   // grid padded for XCD alignment
#pragma unroll
        for (int r = 0; r < RW; r++) slot_free(r);
        return false;
    }
#ifdef NVLLM_STAMPS
    unsigned long long* const stamps_wg = p.stamps ? p.stamps + (size_t)((ix.z * p.ksplit + ix.y) * ix.gx + ix.x) * 8 : nullptr;
#endif
    NVLLM_TSTAMP(0, lane);
    const int kb_begin = ix.y * p.kb_per_split;
    const int kb_end = min(p.Kb, kb_begin + p.kb_per_split);
    const int tb0 = ix.z * TB;
    const int first = kb_begin + wave;
    const int nsteps = first < kb_end ? (kb_end - first + NW - 1) / NW : 0;
    const h8_t* wp[NBW];
#pragma unroll
    for (int i = 0; i < NBW; i++)
        wp[i] = reinterpret_cast<const h8_t*>(p.W + (size_t)min(nb0 + i, p.NB - 1) * p.nb_stride) + lane;
    // activations: pointers (plain loads), or byte offsets into a buffer descriptor (sc1 loads)
    const h8_t* ap[TB];
    int aoff[TB];
    __amdgpu_buffer_rsrc_t rs_a = gemm_rsrc(p.A, (unsigned)p.Tb * (unsigned)p.Kb * 2048u);
#pragma unroll
    for (int j = 0; j < TB; j++) {
        ap[j] = reinterpret_cast<const h8_t*>(p.A + (size_t)min(tb0 + j, p.Tb - 1) * p.Kb * 2 * TILE_HALVES) + lane;
        aoff[j] = min(tb0 + j, p.Tb - 1) * p.Kb * 2048 + lane * 16;
    }
    auto load_act = [&](int j, int kb, h8_t& hi, h8_t& lo) {
        if constexpr (COH) {
            hi = __builtin_bit_cast(h8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff[j], kb * 2048, AUX_SC1));
            lo = __builtin_bit_cast(h8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_a, aoff[j] + 1024, kb * 2048, AUX_SC1));
        } else {
            hi = ap[j][(size_t)kb * 128];
#ifdef NVLLM_ABLATE_LO   // diagnostic builds only: what the second activation plane costs (results are then fp16-activation results)
            lo = hi;
#else
            lo = ap[j][(size_t)kb * 128 + 64];
#endif
        }
    };

    f4_t acc[NBW][TB];
#pragma unroll
    for (int i = 0; i < NBW; i++)
#pragma unroll
        for (int j = 0; j < TB; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    h8_t ah[RA][TB], al[RA][TB];
    auto kb_of = [&](int s) { return first + NW * s; };

    // Loads are issued only for steps that exist (wave-uniform branches): a clamped
    // re-load of the last tile would be real HBM traffic, since streamed weights are
    // loaded non-temporally and do not stay in L2.
    if (PRE == 0 || (PRE == 2 && !prefetched)) {
#pragma unroll
        for (int s = 0; s < RW; s++)
            if (s < nsteps) {
                const int kb = kb_of(s);
#pragma unroll
                for (int i = 0; i < NBW; i++) w[s][i] = NVLLM_WLOAD(wp[i] + (size_t)kb * 64);
            }
    }
#pragma unroll
    for (int s = 0; s < RA; s++)
        if (s < nsteps) {
            const int kb = kb_of(s);
#pragma unroll
            for (int j = 0; j < TB; j++) load_act(j, kb, ah[s][j], al[s][j]);
        }
    // 1/rms of this token group, from the producer's per-tile sums of squares: every
    // thread sums a slice of one token's tiles (independent loads, issued while the
    // first weight tiles are in flight); the slices meet in LDS before the epilogue.
    if (p.e.ss_in != nullptr) {
        const int tl = tid % (TB * 16), part = tid / (TB * 16);
        const int t = min(tb0 * 16 + tl, p.T - 1);
        const int src = p.e.ss_gather ? p.e.ss_gather[t] : t;
        const float* sp = p.e.ss_in + (size_t)src * p.e.ss_tiles;
        float ss = 0.f;
        if constexpr (COH) {
            for (int j = part; j < p.e.ss_tiles; j += PARTS)
                ss += __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(sp + j), __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT));
        } else {
            for (int j = part; j < p.e.ss_tiles; j += PARTS) ss += sp[j];
        }
        s_part[part][tl] = ss;
    }
    NVLLM_TSTAMP(1, lane);
    for (int s0 = 0; s0 < nsteps; s0 += RW) {
#pragma unroll
        for (int r = 0; r < RW; r++) {
            const int s = s0 + r;
            if (s < nsteps) {
#pragma unroll
                for (int i = 0; i < NBW; i++)
#pragma unroll
                    for (int j = 0; j < TB; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[r][i], ah[r % RA][j], acc[i][j], 0, 0, 0);
#ifndef NVLLM_ABLATE_LO
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[r][i], al[r % RA][j], acc[i][j], 0, 0, 0);
#endif
                    }
                if (s == 0) NVLLM_TSTAMP(2, acc[NBW - 1][TB - 1][0]);
                if (s + RW < nsteps) {
                    const int kbw = kb_of(s + RW);
#pragma unroll
                    for (int i = 0; i < NBW; i++) w[r][i] = NVLLM_WLOAD(wp[i] + (size_t)kbw * 64);
                } else {
                    slot_free(r);
                }
                if (s + RA < nsteps) {
                    const int kba = kb_of(s + RA);
#pragma unroll
                    for (int j = 0; j < TB; j++) load_act(j, kba, ah[r % RA][j], al[r % RA][j]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; r++)
        if (r >= nsteps) slot_free(r);   // entries a short K slice never used
    NVLLM_TSTAMP(3, acc[NBW - 1][TB - 1][0]);
    // ---- 0. epilogue operands that do not depend on the sums: issue their loads now,
    // so that they travel while the K split is being closed (every workgroup does it;
    // only the one that ends up running the epilogue uses them).  Indices are clamped,
    // not predicated: unconditional loads.
    const GemmEpilogue& ep = p.e;
    float4 pf_x[FPT], pf_g[FPT];
    int pf_pos = 0, pf_row = -1;
    __amdgpu_buffer_rsrc_t rs_x = gemm_rsrc(ep.x, (unsigned)p.T * (unsigned)p.N * 4u);
    __amdgpu_buffer_rsrc_t rs_o = gemm_rsrc(ep.a16_out, (unsigned)p.Tb * (unsigned)ep.Kb_out * 2048u);
    __amdgpu_buffer_rsrc_t rs_q = gemm_rsrc(ep.q_out, (unsigned)p.T * (unsigned)ep.QD * 4u);
    if (EPI == GemmEpilogue::RESID_A16) {
#pragma unroll
        for (int s = 0; s < FPT; s++) {
            const int w_ = min(tid + s * NW * 64, WITEMS - 1);
            const int h = w_ >> ITEMS_SHIFT, e = w_ & (ITEMS - 1);
            const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
            const int t = min(tb0 * 16 + j * 16 + (l & 15), p.T - 1);
            const int n = min((nb0 + 2 * h + i) * 16 + 4 * (l >> 4), p.N - 4);
            if constexpr (COH)
                pf_x[s] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (t * p.N + n) * 4, 0, AUX_SC1));
            else
                pf_x[s] = *reinterpret_cast<const float4*>(ep.x + (size_t)t * p.N + n);
            pf_g[s] = *reinterpret_cast<const float4*>(ep.gain_next + n);
        }
    }
    if (EPI == GemmEpilogue::ROPE_KV) {   // all items of a thread belong to one token (NW*64 is a multiple of TB*64)
        const int e = min(tid, WITEMS - 1) & (ITEMS - 1);
        const int t = min(tb0 * 16 + ((e >> 6) % TB) * 16 + (e & 15), p.T - 1);
        pf_pos = ep.pos[t];
        pf_row = ep.kv_row[t];
    }
    // ---- 1. sum the waves' partial tiles through LDS: every wave stores all its tiles, ONE
    // barrier, then each thread collects its items w = (round h, e): the float4 for token t(e) and
    // features n(h, e) .. n+3, in wave order.  (The items of all rounds are dealt over all threads.)
    f4_t v[FPT];
#pragma unroll
    for (int i = 0; i < NBW; i++)
#pragma unroll
        for (int j = 0; j < TB; j++) red[wave][i][j][lane] = acc[i][j];
    __syncthreads();
    if (ep.ss_in != nullptr && tid < TB * 16) {
        // 1/rms per token from the slices gathered at the top (s_part: visible since the barrier
        // above; s_rinv is read by the epilogue after at least one more barrier)
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; q++) ss += s_part[q][tid];
        s_rinv[tid] = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss, (float)ep.d_in), ep.eps)));
    }
#pragma unroll
    for (int s = 0; s < FPT; s++) {
        const int w_ = tid + s * NW * 64;
        f4_t sum = f4_t{0.f, 0.f, 0.f, 0.f};
        if (w_ < WITEMS) {
            const int h = w_ >> ITEMS_SHIFT, e = w_ & (ITEMS - 1);
            const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
            sum = red[0][2 * h + i][j][l];
#pragma unroll
            for (int wv = 1; wv < NW; wv++) sum += red[wv][2 * h + i][j][l];
        }
        v[s] = sum;
    }
    __syncthreads();   // everybody has read the partial tiles (the epilogue reuses the array) and s_rinv is visible
    NVLLM_TSTAMP(4, v[FPT - 1][0]);
    // ---- 2. close the K split: write-through slab, ticket, last arriver reduces.
    const unsigned tile_id = ix.z * ix.gx + ix.x;
    if (p.ksplit > 1) {
        float* my = p.slab + ((size_t)tile_id * p.ksplit + ix.y) * (ROUNDS * ITEMS * 4);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(my, 0, ROUNDS * ITEMS * 16, 0x00020000);
#pragma unroll
        for (int s = 0; s < FPT; s++) {
            const int w_ = tid + s * NW * 64;
            if (w_ < WITEMS)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v[s]), rs, w_ * 16, 0, /*sc1: write through*/ AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
        __syncthreads();
        if (tid == 0)
            *s_ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*s_ticket != (unsigned)p.ksplit - 1) return false;     // not the last slice of this tile
        // The partials were stored write-through, every storing wave drained, and this workgroup's add came
        // last (its returned value says so, read behind the barrier above).  With one workgroup per CU the
        // partials may then be loaded past the vector cache (sc1) with no invalidate; otherwise an agent
        // acquire comes first.
        const bool past_l1 = p.close_sc1;
        if (!past_l1) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed
            __syncthreads();
        }
        const float* base = p.slab + (size_t)tile_id * p.ksplit * (ROUNDS * ITEMS * 4);
        __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                                                           p.ksplit * ROUNDS * ITEMS * 16, 0x00020000);
        // Every slab load of both rounds is issued before the first add: one round trip.
        // Slice indices past the end are clamped (a repeated load), not predicated, so the
        // loads stay unconditional; sums are taken in slice order.
#pragma unroll
        for (int s = 0; s < FPT; s++) v[s] = f4_t{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.ksplit; s0 += 8) {
            f4_t part[FPT][8];
#pragma unroll
            for (int s = 0; s < FPT; s++) {
                const int w_ = min(tid + s * NW * 64, WITEMS - 1);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int sl = min(s0 + u, p.ksplit - 1);
                    if (past_l1)
                        part[s][u] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_all, (sl * WITEMS + w_) * 16, 0, AUX_SC1));
                    else
                        part[s][u] = *reinterpret_cast<const f4_t*>(base + ((size_t)sl * WITEMS + w_) * 4);
                }
            }
#pragma unroll
            for (int s = 0; s < FPT; s++)
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (s0 + u < p.ksplit) v[s] += part[s][u];
        }
        // ready for the next launch (nobody else touches this tile's counter any more)
        if (tid == 0) __hip_atomic_store(p.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    NVLLM_TSTAMP(5, v[FPT - 1][0]);
    // ---- 3. epilogue on complete sums
    float ssq[FPT];
#pragma unroll
    for (int s = 0; s < FPT; s++) ssq[s] = 0.f;
    if (EPI == GemmEpilogue::SWIGLU_A16) {
        // gate (even row block, i = 0) and up (odd, i = 1) sit in different threads: swap through LDS so that
        // the i = 0 thread has both.  All rounds at once, each in its own part of the array (the partial tiles
        // in it have been consumed: a barrier back), so one barrier serves them all.
#pragma unroll
        for (int s = 0; s < FPT; s++) {
            const int w_ = tid + s * NW * 64;
            const int h = w_ >> ITEMS_SHIFT, e = w_ & (ITEMS - 1);
            if (w_ < WITEMS) red[0][2 * h + (e >> 6) / TB][(e >> 6) % TB][e & 63] = v[s];
        }
        __syncthreads();
    }
    {
#pragma unroll
        for (int s = 0; s < FPT; s++) {
            const int w_ = tid + s * NW * 64;
            if (w_ >= WITEMS) continue;
            const int h = w_ >> ITEMS_SHIFT, e = w_ & (ITEMS - 1);
            const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
            const int tl = j * 16 + (l & 15);
            const int t = tb0 * 16 + tl;
            const int nb = nb0 + 2 * h + i;
            const int n = nb * 16 + 4 * (l >> 4);
            const bool live = t < p.T && tb0 + j < p.Tb && nb < p.NB;
            const float rinv = ep.ss_in ? s_rinv[tl] : 1.0f;
            float y[4] = {v[s][0] * rinv, v[s][1] * rinv, v[s][2] * rinv, v[s][3] * rinv};
            if (EPI == GemmEpilogue::F32) {
                if (live) {
                    float* o = ep.out + (size_t)t * ep.ldo + n;
                    if (n + 3 < p.N && (ep.ldo & 3) == 0) {
                        *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (n + r < p.N) o[r] = y[r];
                    }
                }
                if (ep.amax_pairs != nullptr) {
                    // this item's four outputs: first index of their maximum (a NaN never wins, -inf may)
                    float bv = -INFINITY, bi = -1.0f;
                    if (live) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (n + r < p.N && (y[r] > bv || (bi < 0.f && y[r] == bv))) {
                                bv = y[r];
                                bi = (float)(n + r);
                            }
                        // the reference's scan starts from logits[0], and nothing displaces a NaN there
                        // (sampler.c:86-97): +inf at index 0 wins every comparison downstream
                        if (n == 0 && y[0] != y[0]) {
                            bv = INFINITY;
                            bi = 0.f;
                        }
                    }
                    // (the partial tiles in `red` have been consumed: two barriers back)
                    float2* am = reinterpret_cast<float2*>(&red[0][0][0][0]);
                    am[((2 * h + i) * 4 + (l >> 4)) * (TB * 16) + tl] = make_float2(bv, bi);
                }
            } else if (EPI == GemmEpilogue::SWIGLU_A16) {
                if (i == 0 && live) {
                    const f4_t u = red[0][2 * h + 1][j][l];
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) z[r] = silu_mul(y[r], u[r] * rinv);
                    const int hid = (nb >> 1) * 16 + 4 * (l >> 4);
                    if constexpr (COH) store_a16_4_sc1(rs_o, ep.Kb_out, t, hid, z, l);
                    else store_a16_4(ep.a16_out, ep.Kb_out, t, hid, z);
                }
            } else if (EPI == GemmEpilogue::RESID_A16) {
                if (live) {
                    float* xr = ep.x + (size_t)t * p.N + n;
                    const float4 xo = pf_x[s];
                    const float4 g = pf_g[s];
                    // residual add as the reference's accum(): x = x + branch (layers.c:163-168)
                    const float xn[4] = {__fadd_rn(xo.x, y[0]), __fadd_rn(xo.y, y[1]), __fadd_rn(xo.z, y[2]), __fadd_rn(xo.w, y[3])};
                    if constexpr (COH)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, make_float4(xn[0], xn[1], xn[2], xn[3])), rs_x,
                                                               (t * p.N + n) * 4, 0, AUX_SC1);
                    else
                        *reinterpret_cast<float4*>(xr) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                    const float z[4] = {xn[0] * g.x, xn[1] * g.y, xn[2] * g.z, xn[3] * g.w};
                    if constexpr (COH) store_a16_4_sc1(rs_o, ep.Kb_out, t, n, z, l);
                    else store_a16_4(ep.a16_out, ep.Kb_out, t, n, z);
                    ssq[s] += xn[0] * xn[0] + xn[1] * xn[1] + xn[2] * xn[2] + xn[3] * xn[3];
                }
            } else {  // ROPE_KV
                if (live) {
                    if (n < ep.QD + ep.KD) {   // q or k: rotate the two pairs of this float4 (layers.c:118-141)
                        const int d = (n < ep.QD ? n : n - ep.QD) % ep.hd;
                        const float2* tab = ep.rope + (size_t)pf_pos * (ep.hd >> 1) + (d >> 1);
                        const float2 c0 = tab[0], c1 = tab[1];
                        const float a0 = y[0], b0 = y[1], a1 = y[2], b1 = y[3];
                        y[0] = __fsub_rn(__fmul_rn(a0, c0.x), __fmul_rn(b0, c0.y));
                        y[1] = __fadd_rn(__fmul_rn(a0, c0.y), __fmul_rn(b0, c0.x));
                        y[2] = __fsub_rn(__fmul_rn(a1, c1.x), __fmul_rn(b1, c1.y));
                        y[3] = __fadd_rn(__fmul_rn(a1, c1.y), __fmul_rn(b1, c1.x));
                    }
                    if (n < ep.QD) {
                        if constexpr (COH)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, make_float4(y[0], y[1], y[2], y[3])), rs_q,
                                                                   (t * ep.QD + n) * 4, 0, AUX_SC1);
                        else *reinterpret_cast<float4*>(ep.q_out + (size_t)t * ep.QD + n) = make_float4(y[0], y[1], y[2], y[3]);
                    } else if (pf_row >= 0) {
                        const long block = pf_row / ep.bs;
                        const int key = pf_row % ep.bs;
                        const bool is_k = n < ep.QD + ep.KD;
                        const int c = is_k ? n - ep.QD : n - ep.QD - ep.KD;
                        KVT* pool = reinterpret_cast<KVT*>(is_k ? ep.pool_k : ep.pool_v) + ep.layer_off;
                        if (sizeof(KVT) == 4) {   // the reference's rows (memory.c:37-43)
                            if constexpr (COH) kv_put4_sc1<KVT>(pool + (block * ep.bs + key) * ep.KD + c, y);
                            else kv_put4<KVT>(pool + (block * ep.bs + key) * ep.KD + c, y);
                        } else {                  // fp16 pages in matrix-core operand order
                            const int kvh = c / ep.hd, d = c % ep.hd;
                            KVT* page = pool + kv16_page(block, kvh, ep.KD / ep.hd, ep.hd);
                            if constexpr (COH) {
                                if (is_k) kv_put4_sc1<KVT>(page + kv16_k_off(key, d), y);
                                else kv_put4_strided_sc1<KVT>(page + kv16_v_off(key, d), y);
                            } else {
                                if (is_k) kv_put4<KVT>(page + kv16_k_off(key, d), y);
                                else kv_put4_strided<KVT>(page + kv16_v_off(key, d), y);
                            }
                        }
                    }
                }
            }
        }
    }
    if (EPI == GemmEpilogue::F32 && ep.amax_pairs != nullptr) {
        // per token, over the tile's features in ascending order: the first maximum
        __syncthreads();
        if (tid < TB * 16) {
            const float2* am = reinterpret_cast<const float2*>(&red[0][0][0][0]);
            float2 best = am[tid];
#pragma unroll
            for (int s = 1; s < NBW * 4; s++) {
                const float2 c = am[s * (TB * 16) + tid];
                if (c.y >= 0.f && (c.x > best.x || best.y < 0.f)) best = c;
            }
            const int t = tb0 * 16 + tid;
            if (t < p.T) ep.amax_pairs[(size_t)ix.x * ep.amax_ld + t] = best;
        }
    }
    if (EPI == GemmEpilogue::RESID_A16) {
        // per-token sum of x^2 over each 32-feature group (= one round: two row blocks):
        // over the 4 lane groups (l>>4), then over the round's two row blocks (i)
        const int ss_cols = (p.N + 31) >> 5;
        // all rounds' halves meet in the array of the partial tiles (consumed: a barrier back), [round][half][token]:
        // one barrier for all rounds
        float (*ss_lds)[2][TB * 16] = reinterpret_cast<float (*)[2][TB * 16]>(lds_dyn);
#pragma unroll
        for (int s = 0; s < FPT; s++) {
            float sq = ssq[s];
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const int w_ = tid + s * NW * 64;
            const int h = w_ >> ITEMS_SHIFT, e = w_ & (ITEMS - 1);
            if (w_ < WITEMS && (e & 63) < 16) ss_lds[h][(e >> 6) / TB][((e >> 6) % TB) * 16 + (e & 15)] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < ROUNDS; h++) {
            if (tid < TB * 16) {
                const int t = tb0 * 16 + tid;
                const int col = ix.x * ROUNDS + h;
                if (t < p.T && col < ss_cols) {
                    const float sum = ss_lds[h][0][tid] + ss_lds[h][1][tid];
                    if constexpr (COH)
                        __hip_atomic_store(reinterpret_cast<unsigned*>(ep.ss_out + (size_t)t * ss_cols + col),
                                           __builtin_bit_cast(unsigned, sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        ep.ss_out[(size_t)t * ss_cols + col] = sum;
                }
            }
        }
    }
#ifdef NVLLM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores issued so far have left the CU
#endif
    NVLLM_TSTAMP(6, ssq[0]);
    return true;
}

}  // namespace nvllm
