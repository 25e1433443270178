// gemm_prefill.hip -- the projections when a batch carries hundreds of tokens (prefill
// chunks): out[T, N] = act[T, K] x W[N, K]^T as a dense contraction on the matrix cores.
//
// At 64 tokens (gemm.hip) every weight byte feeds 2*64 flops and the waves of a
// workgroup split K, each pulling its own operands.  From a few hundred tokens on the
// work is bound by the matrix cores and by the operand traffic INTO the CU, so here a
// workgroup owns an output tile of 128 features x 128 tokens and its waves (2 x 2, each
// 64 x 64 = 4 x 4 accumulator tiles) walk K together: per K-step the workgroup's 8 weight
// tiles and 8 x 2 activation tiles (hi, lo planes) are copied global -> LDS once, by
// direct-to-LDS loads, and every wave reads the 4 + 8 tiles it needs from LDS.  That is
// 24 KiB into the CU per 128 MFMAs instead of 48 KiB when each wave loads for itself.
// Two forms:
//   4 waves: for projections that make more workgroups than CUs; up to three share a CU
//            and cover each other's waits;
//   8 waves: for projections whose workgroups fit the chip at one per CU.  The second
//            group of four waves takes the odd K-steps, so every SIMD has two waves to
//            alternate between LDS reads and MFMAs and a barrier covers two K-steps; the
//            two groups add their accumulators through LDS before the epilogue.
// (256-feature tiles have the better operand traffic per MFMA and measured slower at
// prefill sizes: too few workgroups, and the last round of them runs nearly empty.)
//
// Both operands are stored in matrix-core operand order, one 1-KiB tile per (16 rows,
// 32 k), so a tile is copied by ONE wave instruction (64 lanes x 16 B, lane-linear in
// LDS) and read back by one conflict-free ds_read_b128 per lane: no swizzle, no
// transposition.  The copies run ahead of the multiplication in a ring of LDS buffers
// (24 KiB per K-step): each wave waits for its own copies of the step it is about to
// read with a COUNTED s_waitcnt (later steps stay in flight), a raw s_barrier publishes
// the step to the other waves and at the same time retires the buffer read one step
// earlier, which is then refilled.  The ring depth follows the launch: a wide
// projection makes several workgroups per CU, which cover each other's waits, so two
// buffers (three workgroups fit a CU) are enough; fewer workgroups per CU get deeper
// rings (up to five buffers of 24 KiB, or three of 48 KiB in the 8-wave form).
//
// The epilogues are those of gemm.hip (F32, SWIGLU_A16, RESID_A16, ROPE_KV) applied
// straight from the accumulators: a lane holds, per accumulator tile, four consecutive
// features of one token, the gate and up rows of a hidden unit sit in the same lane, and
// the per-token sums of squares need two cross-lane adds -- no LDS pass.
#include "gemm_common.h"
#include <type_traits>
#include <algorithm>

namespace nvllm {

namespace {

typedef __attribute__((address_space(3))) void lds_void;

// FW = 16-row weight blocks per wave (4 or 8), KG = wave groups that split the K-steps, WR = rows of
// the wave grid (WR x 2 waves per group: 2 -> 128-feature tiles; 4 -> 256-feature tiles compiles but measured
// slower on this model's shapes and is not instantiated), TW = 16-token
// blocks per wave (4, 3 or 2: the tile's height is chosen per launch so that its rounds of workgroups fill)
template <int FW, int KG, int WR = 2, int TW = 4>
struct PfGeom {
    static constexpr int FB = WR * FW;                 // weight blocks per workgroup tile
    static constexpr int TB = 2 * TW;                  // 16-token blocks per workgroup tile: two wave columns of TW (128, 96 or 64 tokens)
    static constexpr int TPK = FB + 2 * TB;            // 1-KiB tiles per K-step: weights, then (hi, lo) per token block
    static constexpr int STAGE_TILES = KG * TPK;       // a stage = KG consecutive K-steps = what one barrier publishes
    static constexpr int STAGE_HALVES = STAGE_TILES * TILE_HALVES;
    static constexpr int WAVES = 2 * WR * KG;
    static constexpr int CPW = STAGE_TILES / WAVES;    // copies per wave per stage (4, 6 or 8)
    static_assert(STAGE_TILES % WAVES == 0, "every wave copies the same number of tiles");
    static_assert(TW == 4 || KG == 1, "the two-group form exchanges halves of four token blocks");
    static constexpr int RINV_HALVES = 256;            // 128 floats in front of the ring (1/rms of up to 128 tokens)
};

// One wave copies one 1-KiB tile global -> LDS without passing through registers: lane l's
// 16 bytes land at lds_byte_address + 16*l.  Written as an asm statement on purpose: the
// compiler then keeps no account of the copy, and the waits below are the only ones (with
// the builtin it drains every copy in flight before the next read of the LDS array).
// M0 carries the LDS address and is restored inside the statement.
__device__ __forceinline__ void copy_tile_to_lds(const __half* lane_src, unsigned lds_byte_address) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_src), "s"(lds_byte_address)
                 : "memory");
}

// The same copy for the steady state of the K loop: M0 is written and left (nothing else in
// this kernel reads it: the compiler sets M0 itself right before any use of its own), the LDS address is a wave-uniform
// base plus a constant.
// The address is summed by the compiler (its own scalar add, whose condition-code write it accounts
// for); the statement itself only moves it into M0, which touches no condition code.
template <int OFS>
__device__ __forceinline__ void copy_tile_to_lds_at(const __half* lane_src, unsigned lds_base) {
    const unsigned dst = lds_base + OFS;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(lane_src), "s"(dst)
                 : "memory");
}

// wait until at most `stages` stages of this wave's copies (CPW each) are still in flight;
// s_waitcnt takes an immediate
template <int CPW>
__device__ __forceinline__ void wait_copies_but(int stages) {
    switch (stages) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CPW) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CPW) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CPW) : "memory"); break;
    }
}

// The epilogues of gemm.hip applied to NI x NJ accumulator tiles of one wave.  v[i][j] of
// lane l: token t = (tb0 + tbl_first + j)*16 + (l & 15), features n .. n+3 with
// n = nb*16 + 4*(l >> 4), nb = nb_first + i.
template <int NI, int NJ, int EPI, typename KVT>
__device__ __forceinline__ void pf_epilogue(const GemmArgs& p, const f4_t (&v)[NI][NJ], int nb_first, int tbl_first, int tb0,
                                            const float* s_rinv, int lane) {
    const GemmEpilogue& ep = p.e;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int tbj = tb0 + tbl_first + j;
        const int tl = (tbl_first + j) * 16 + (lane & 15);
        const int t = tb0 * 16 + tl;
        const bool tok_live = t < p.T && tbj < p.Tb;
        const float rinv = ep.ss_in ? s_rinv[tl] : 1.0f;
        int pos = 0, kv_row = -1;
        if (EPI == GemmEpilogue::ROPE_KV && tok_live) {
            pos = ep.pos[t];
            kv_row = ep.kv_row[t];
        }
        float ssq[NI / 2];
#pragma unroll
        for (int h = 0; h < NI / 2; h++) ssq[h] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int nb = nb_first + i;
            const int n = nb * 16 + 4 * (lane >> 4);
            const bool live = tok_live && nb < p.NB;
            float y[4] = {v[i][j][0] * rinv, v[i][j][1] * rinv, v[i][j][2] * rinv, v[i][j][3] * rinv};
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
            } else if (EPI == GemmEpilogue::SWIGLU_A16) {
                // row blocks alternate gate (even), up (odd); nb_first is even, so even i are gates
                if ((i & 1) == 0 && live) {
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) z[r] = silu_mul(y[r], v[(i + 1) % NI][j][r] * rinv);
                    const int hid = (nb >> 1) * 16 + 4 * (lane >> 4);
                    store_a16_4(ep.a16_out, ep.Kb_out, t, hid, z);
                }
            } else if (EPI == GemmEpilogue::RESID_A16) {
                if (live) {
                    float* xr = ep.x + (size_t)t * p.N + n;
                    const float4 xo = *reinterpret_cast<const float4*>(xr);
                    const float4 g = *reinterpret_cast<const float4*>(ep.gain_next + n);
                    // residual add as the reference's accum(): x = x + branch (layers.c:163-168)
                    const float xn[4] = {__fadd_rn(xo.x, y[0]), __fadd_rn(xo.y, y[1]), __fadd_rn(xo.z, y[2]), __fadd_rn(xo.w, y[3])};
                    *reinterpret_cast<float4*>(xr) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                    const float z[4] = {xn[0] * g.x, xn[1] * g.y, xn[2] * g.z, xn[3] * g.w};
                    store_a16_4(ep.a16_out, ep.Kb_out, t, n, z);
                    ssq[i >> 1] += xn[0] * xn[0] + xn[1] * xn[1] + xn[2] * xn[2] + xn[3] * xn[3];
                }
            } else {  // ROPE_KV
                if (live) {
                    if (n < ep.QD + ep.KD) {   // q or k: rotate the two pairs of this float4 (layers.c:118-141)
                        const int d = (n < ep.QD ? n : n - ep.QD) % ep.hd;
                        const float2* tab = ep.rope + (size_t)pos * (ep.hd >> 1) + (d >> 1);
                        const float2 c0 = tab[0], c1 = tab[1];
                        const float a0 = y[0], b0 = y[1], a1 = y[2], b1 = y[3];
                        y[0] = __fsub_rn(__fmul_rn(a0, c0.x), __fmul_rn(b0, c0.y));
                        y[1] = __fadd_rn(__fmul_rn(a0, c0.y), __fmul_rn(b0, c0.x));
                        y[2] = __fsub_rn(__fmul_rn(a1, c1.x), __fmul_rn(b1, c1.y));
                        y[3] = __fadd_rn(__fmul_rn(a1, c1.y), __fmul_rn(b1, c1.x));
                    }
                    if (n < ep.QD) {
                        *reinterpret_cast<float4*>(ep.q_out + (size_t)t * ep.QD + n) = make_float4(y[0], y[1], y[2], y[3]);
                    } else if (kv_row >= 0) {
                        const long block = kv_row / ep.bs;
                        const int key = kv_row % ep.bs;
                        const bool is_k = n < ep.QD + ep.KD;
                        const int c = is_k ? n - ep.QD : n - ep.QD - ep.KD;
                        KVT* pool = reinterpret_cast<KVT*>(is_k ? ep.pool_k : ep.pool_v) + ep.layer_off;
                        if (sizeof(KVT) == 4) {   // the reference's rows (memory.c:37-43)
                            kv_put4<KVT>(pool + (block * ep.bs + key) * ep.KD + c, y);
                        } else {                  // fp16 pages in matrix-core operand order
                            const int kvh = c / ep.hd, d = c % ep.hd;
                            KVT* page = pool + kv16_page(block, kvh, ep.KD / ep.hd, ep.hd);
                            if (is_k) kv_put4<KVT>(page + kv16_k_off(key, d), y);
                            else kv_put4_strided<KVT>(page + kv16_v_off(key, d), y);
                        }
                    }
                }
            }
        }
        if (EPI == GemmEpilogue::RESID_A16) {
            // per-token sum of x^2 over each 32-feature group = two row blocks (same lane)
            // x four lane groups
            const int ss_cols = (p.N + 31) >> 5;
#pragma unroll
            for (int h = 0; h < NI / 2; h++) {
                float sq = ssq[h];
                sq += __shfl_xor(sq, 16, 64);
                sq += __shfl_xor(sq, 32, 64);
                const int col = (nb_first + 2 * h) >> 1;
                if (lane < 16 && tok_live && col < ss_cols) ep.ss_out[(size_t)t * ss_cols + col] = sq;
            }
        }
    }
}

// Close a K split (as in gemm.hip): every slice publishes its partial tile -- each thread its own
// NI x NJ float4 sums, stored lane-contiguously with write-through (sc1) stores -- drains the stores,
// and one lane takes a ticket from the tile's counter.  The workgroup that draws the last ticket
// acquires, adds the slices in slice order (its own from the slab too, so the sum does not depend
// on who arrived last), resets the counter and returns true: it runs the epilogue.
template <int NI, int NJ, int THREADS>
__device__ __forceinline__ bool pf_close_split(const GemmArgs& p, f4_t (&v)[NI][NJ], int tile_id, int slice) {
    __shared__ unsigned s_ticket;
    constexpr int PER_SLICE = NI * NJ * THREADS;   // float4 per slice of one tile
    float* base = p.slab + (size_t)tile_id * p.ksplit * PER_SLICE * 4;
    {
        float* my = base + (size_t)slice * PER_SLICE * 4;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(my, 0, PER_SLICE * 16, 0x00020000);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v[i][j]), rs,
                                                       ((i * NJ + j) * THREADS + (int)threadIdx.x) * 16, 0, /*sc1: write through*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)p.ksplit - 1) return false;
#ifdef NVLLM_PF_CLOSE_FENCE
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed
    __syncthreads();
#endif
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) v[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < p.ksplit; sl++) {
        // coherent (sc1) loads: they do not hit lines this XCD's L2 may hold from an earlier
        // launch's slab, and need no cache invalidate -- which would also drop the weights and
        // activations the other workgroups behind this L2 are sharing
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)sl * PER_SLICE * 4, 0, PER_SLICE * 16, 0x00020000);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
                v[i][j] += __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * NJ + j) * THREADS + (int)threadIdx.x) * 16, 0, /*sc1*/ 16));
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int FW, int KG, int EPI, typename KVT, int WR = 2, int TW = 4>
__global__ __launch_bounds__(128 * WR * KG, WR == 4 ? 4 : 2) void k_gemm_prefill(GemmArgs p, int tiles_n, int tiles_t, int nbuf, int split_from) {
    using G = PfGeom<FW, KG, WR, TW>;
#ifdef NVLLM_PF_ABLATE
    const int ablate = nbuf >> 8;   // diagnostic build: 1 no in-loop copies, 2 no multiplies, 4 no LDS reads, 8 no barrier, 16 no epilogue
    nbuf &= 255;
#endif
    // all LDS is one object: 128 floats of 1/rms, then the ring [nbuf][STAGE_TILES][512 halves]
    extern __shared__ __attribute__((aligned(16))) __half lds_all[];
    float* s_rinv = reinterpret_cast<float*>(lds_all);
    __half* lds = lds_all + G::RINV_HALVES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    NVLLM_STAMP(0, lane);
    const int kg = wave / (2 * WR);                   // which K-steps of a stage this wave multiplies
    const int wr = (wave >> 1) % WR, wc = wave & 1;   // its FW*16 features x 64 tokens inside the tile

    // Workgroups are dealt round-robin to the 8 XCDs.  Give every XCD a contiguous range
    // of the (feature tile, token tile) sequence, token tiles fastest: the workgroups that
    // run together behind one L2 then share a few weight tiles and all walk the same
    // activations.
    // Tiles [0, split_from) are one workgroup each; the tiles after them (the launch's last,
    // partly filled round of workgroup slots -- or every tile of a launch too small to fill
    // the chip) are cut into p.ksplit slices of K, one workgroup per slice, closed below.
    const int total = tiles_n * tiles_t;
    auto deal = [](int i, int n) {   // position i of n dealt to 8 XCDs -> place in the XCD-contiguous order
        const int xcd = i & 7, slot = i >> 3;
        const int q = n >> 3, r = n & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    };
    int id, slice = 0;
    const bool split = (int)blockIdx.x >= split_from;
    if (!split) {
        id = deal(blockIdx.x, split_from);
    } else {
        const int j = deal(blockIdx.x - split_from, (total - split_from) * p.ksplit);
        id = split_from + j / p.ksplit;
        slice = j % p.ksplit;
    }
    const int nb0 = (id / tiles_t) * G::FB;
    const int tb0 = (id % tiles_t) * G::TB;
    // K-steps [kb_lo, kb_hi) of this workgroup
    const int kb_lo = split ? slice * p.kb_per_split : 0;
    const int kb_hi = split ? min(p.Kb, kb_lo + p.kb_per_split) : p.Kb;

    // ---- who copies what: tile q of a stage belongs to K-step q / TPK of the stage and is
    // weight block r (r < FB) or activation (block, plane) of r - FB, r = q % TPK; wave w
    // copies tiles w*CPW .. w*CPW + CPW-1.  Every wave issues exactly CPW copies per stage
    // (a K-step past the end repeats the last one), so the counted waits hold for all.
    const __half* src[G::CPW];
    int step[G::CPW], kofs[G::CPW];
#pragma unroll
    for (int u = 0; u < G::CPW; u++) {
        const int q = wave * G::CPW + u;
        const int r = q % G::TPK;
        kofs[u] = q / G::TPK;
        if (r < G::FB) {
            src[u] = p.W + (size_t)min(nb0 + r, p.NB - 1) * p.nb_stride + lane * 8;
            step[u] = TILE_HALVES;
        } else {
            const int tb = min(tb0 + ((r - G::FB) >> 1), p.Tb - 1);
            src[u] = p.A + ((size_t)tb * p.Kb * 2 + ((r - G::FB) & 1)) * TILE_HALVES + lane * 8;
            step[u] = 2 * TILE_HALVES;
        }
    }
    const unsigned lds_tiles = (unsigned)(size_t)(lds_void*)lds + (unsigned)(wave * G::CPW) * (TILE_HALVES * 2);   // byte address in LDS
    auto stage = [&](int s, int buf) {
#pragma unroll
        for (int u = 0; u < G::CPW; u++) {
            const int kb = min(kb_lo + s * KG + kofs[u], kb_hi - 1);
            copy_tile_to_lds(src[u] + (size_t)kb * step[u],
                             __builtin_amdgcn_readfirstlane(lds_tiles + (unsigned)(buf * G::STAGE_TILES + u) * (TILE_HALVES * 2)));
        }
    };

    f4_t acc[FW][TW];
#pragma unroll
    for (int i = 0; i < FW; i++)
#pragma unroll
        for (int j = 0; j < TW; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    // 1/rms of the tile's tokens from the producer's sums of squares (two threads per token);
    // done before the first copy is issued, so that its barrier does not wait for copies
    if (p.e.ss_in != nullptr) {
        const int tl = threadIdx.x & 127, part = threadIdx.x >> 7;
        float ss = 0.f;
        if (part < 2) {
            const int t = min(tb0 * 16 + tl, p.T - 1);
            const int srow = p.e.ss_gather ? p.e.ss_gather[t] : t;
            const float* sp = p.e.ss_in + (size_t)srow * p.e.ss_tiles;
            for (int j = part; j < p.e.ss_tiles; j += 2) ss += sp[j];
        }
        // the two halves meet in the result array itself: the upper half writes, the lower adds
        if (part == 1) s_rinv[tl] = ss;
        __syncthreads();
        if (part == 0) s_rinv[tl] = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss + s_rinv[tl], (float)p.e.d_in), p.e.eps)));
    }

    const int n_stages = (kb_hi - kb_lo + KG - 1) / KG;
    const int ahead = nbuf - 1;   // stages in flight beyond the one being multiplied
    for (int s0 = 0; s0 < ahead && s0 < n_stages; s0++) stage(s0, s0);
    NVLLM_STAMP(1, lane);

    // Steady state: stage s + ahead is a whole stage inside [kb_lo, kb_hi), so its copies need
    // no clamp -- every copy keeps a running source pointer and adds a constant per stage
    // (four instructions per copy instead of the dozen of the general form; at one
    // instruction per wave every four to five cycles that is what the loop's time is made of).
    const __half* cp[G::CPW];
    long adv[G::CPW];
#pragma unroll
    for (int u = 0; u < G::CPW; u++) {
        cp[u] = src[u] + (size_t)(kb_lo + ahead * KG + kofs[u]) * step[u];
        adv[u] = (long)KG * step[u];
    }
    auto stage_steady = [&](int buf) {
        // (readfirstlane: a scalar register even where the compiler could fold the sum into a literal)
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_tiles + (unsigned)buf * (unsigned)(G::STAGE_TILES * TILE_HALVES * 2));
#pragma unroll
        for (int u = 0; u < G::CPW; u++) {
            if (u == 0) copy_tile_to_lds_at<0>(cp[u], base);
            if (u == 1) copy_tile_to_lds_at<1 * TILE_HALVES * 2>(cp[u], base);
            if (u == 2) copy_tile_to_lds_at<2 * TILE_HALVES * 2>(cp[u], base);
            if (u == 3) copy_tile_to_lds_at<3 * TILE_HALVES * 2>(cp[u], base);
            if (u == 4) copy_tile_to_lds_at<4 * TILE_HALVES * 2>(cp[u], base);
            if (u == 5) copy_tile_to_lds_at<5 * TILE_HALVES * 2>(cp[u], base);
            if (u == 6) copy_tile_to_lds_at<6 * TILE_HALVES * 2>(cp[u], base);
            if (u == 7) copy_tile_to_lds_at<7 * TILE_HALVES * 2>(cp[u], base);
            cp[u] += adv[u];
        }
    };

    const h8_t* lw = reinterpret_cast<const h8_t*>(lds) + (kg * G::TPK + wr * FW) * 64 + lane;             // weight tiles of this wave
    const h8_t* la = reinterpret_cast<const h8_t*>(lds) + (kg * G::TPK + G::FB + wc * TW * 2) * 64 + lane;  // its activation tiles
    int rb = 0;               // buffer of stage s
    int wb = ahead % nbuf;    // buffer of stage s + ahead (= the one read at stage s - 1)

    // the multiplies of one stage out of buffer rb
    auto multiply = [&]() {
        const h8_t* bw = lw + rb * (G::STAGE_HALVES / 8);
        const h8_t* ba = la + rb * (G::STAGE_HALVES / 8);
        // every fragment of the K-step is requested before the first multiply (the LDS returns
        // them in order, so each group of multiplies waits only for its own two): left to
        // itself the scheduler re-uses two fragment registers and exposes an LDS round trip
        // in front of every eight multiplies
        h8_t wf[FW], af[2 * TW];
#ifdef NVLLM_PF_ABLATE
        if (ablate & 4) {
#pragma unroll
            for (int i = 0; i < FW; i++) asm volatile("" : "=v"(wf[i]));
#pragma unroll
            for (int j = 0; j < 2 * TW; j++) asm volatile("" : "=v"(af[j]));
        } else
#endif
        {
#pragma unroll
            for (int i = 0; i < FW; i++) wf[i] = bw[i * 64];
#pragma unroll
            for (int j = 0; j < 2 * TW; j++) af[j] = ba[j * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef NVLLM_PF_ABLATE
        if (ablate & 2) {
#pragma unroll
            for (int i = 0; i < FW; i++) asm volatile("" ::"v"(wf[i]));
#pragma unroll
            for (int j = 0; j < 2 * TW; j++) asm volatile("" ::"v"(af[j]));
            return;
        }
#endif
        __builtin_amdgcn_s_setprio(1);   // a wave that is multiplying keeps the issue slot over one that is copying
#pragma unroll
        for (int j = 0; j < TW; j++) {
#pragma unroll
            for (int i = 0; i < FW; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[2 * j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < FW; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[2 * j + 1], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    int s = 0;
    // ---- steady state: stages s .. s + ahead - 1 in flight at the wait, all whole.  One copy
    // of the loop per ring depth, so that the wait is a single instruction with its immediate.
    const int n_steady = n_stages - ahead - 1;   // stage s + ahead must not be the last one (which may be ragged)
    auto steady = [&](auto in_flight) {
        constexpr int LEAVE = decltype(in_flight)::value * G::CPW;
        for (; s < n_steady; s++) {
            // this wave's copies of stage s have landed; the stages issued after it may still fly
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEAVE) : "memory");
#ifdef NVLLM_PF_ABLATE
            if (!(ablate & 8))
#endif
            __builtin_amdgcn_s_barrier();   // everybody's copies of stage s; and everybody is done reading stage s - 1
            asm volatile("" ::: "memory");
#ifdef NVLLM_PF_ABLATE
            if (!(ablate & 1))
#endif
            stage_steady(wb);
            if (s == 0) NVLLM_STAMP(2, lane);
            multiply();
            rb = rb + 1 == nbuf ? 0 : rb + 1;
            wb = wb + 1 == nbuf ? 0 : wb + 1;
        }
    };
    switch (ahead) {
        case 1: steady(std::integral_constant<int, 0>{}); break;
        case 2: steady(std::integral_constant<int, 1>{}); break;
        case 3: steady(std::integral_constant<int, 2>{}); break;
        case 4: steady(std::integral_constant<int, 3>{}); break;
        default: break;   // deeper rings run in the general loop below
    }
    // ---- the last stages: clamped copies, waits that count down
    for (; s < n_stages; s++) {
        wait_copies_but<G::CPW>(min(ahead - 1, n_stages - 1 - s));
        asm volatile("" ::: "memory");
#ifdef NVLLM_PF_ABLATE
        if (!(ablate & 8))
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef NVLLM_PF_ABLATE
        if (!(ablate & 1))
#endif
        if (s + ahead < n_stages) stage(s + ahead, wb);
        if (s == 0) NVLLM_STAMP(2, lane);
        if (KG == 1 || kb_lo + s * KG + kg < kb_hi) multiply();
        rb = rb + 1 == nbuf ? 0 : rb + 1;
        wb = wb + 1 == nbuf ? 0 : wb + 1;
    }
    NVLLM_STAMP(3, acc[FW - 1][TW - 1][0]);
    __syncthreads();   // the ring is free; s_rinv (written by a quarter of the threads) is visible
    NVLLM_STAMP(4, lane);

    if (KG == 1) {
        if (split && !pf_close_split<FW, TW, 128 * WR * KG>(p, acc, id - split_from, slice)) {
            NVLLM_STAMP(7, lane);
            return;
        }
        NVLLM_STAMP(7, acc[FW - 1][TW - 1][0]);
#ifdef NVLLM_PF_ABLATE
        if ((ablate & 16) && acc[0][0][0] != 123.25f) return;   // 16: no epilogue (the test keeps the sums alive)
#endif
        pf_epilogue<FW, TW, EPI, KVT>(p, acc, nb0 + wr * FW, wc * TW, tb0, s_rinv, lane);
        NVLLM_STAMP(5, lane);
#ifdef NVLLM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores issued so far have left the CU
#endif
        NVLLM_STAMP(6, lane);
    } else {
        // The two wave groups hold the sums over the even and the odd K-steps of the same
        // outputs.  Each wave keeps two of its four token blocks and hands the other two to
        // its partner (same wr, wc in the other group) through the ring's memory.
        f4_t* xch = reinterpret_cast<f4_t*>(lds);                  // [wave][FW][2][64]
        f4_t* mine = xch + (size_t)wave * (FW * 2 * 64) + lane;
        const f4_t* theirs = xch + (size_t)(wave ^ (2 * WR)) * (FW * 2 * 64) + lane;
        f4_t fin[FW][2];
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < FW; i++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++) mine[(i * 2 + jj) * 64] = acc[i][2 + jj];
        } else {
#pragma unroll
            for (int i = 0; i < FW; i++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++) mine[(i * 2 + jj) * 64] = acc[i][jj];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < FW; i++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++) fin[i][jj] = acc[i][jj] + theirs[(i * 2 + jj) * 64];
        } else {
#pragma unroll
            for (int i = 0; i < FW; i++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++) fin[i][jj] = theirs[(i * 2 + jj) * 64] + acc[i][2 + jj];
        }
        if (split && !pf_close_split<FW, 2, 128 * WR * KG>(p, fin, id - split_from, slice)) return;
        pf_epilogue<FW, 2, EPI, KVT>(p, fin, nb0 + wr * FW, wc * 4 + kg * 2, tb0, s_rinv, lane);
    }
}

template <int FW, int KG, int EPI, typename KVT, int TW>
void launch_one(const GemmArgs& a, int tiles_n, int tiles_t, int nbuf, int split_from) {
    using G = PfGeom<FW, KG, 2, TW>;
    const size_t shmem = (size_t)(G::RINV_HALVES + (nbuf & 255) * G::STAGE_HALVES) * sizeof(__half);   // (a diagnostic build keeps flags above bit 8)
    static size_t allowed = 0;   // per instantiation: the largest dynamic LDS size enabled so far
    if (shmem > allowed) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_prefill<FW, KG, EPI, KVT, 2, TW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        allowed = shmem;
    }
    NVLLM_LAUNCH((k_gemm_prefill<FW, KG, EPI, KVT, 2, TW>), dim3(split_from + (tiles_n * tiles_t - split_from) * a.ksplit), dim3(256 * KG), shmem, a,
                 tiles_n, tiles_t, nbuf, split_from);
}

template <int FW, int KG, typename KVT, int TW = 4>
void launch_pf(const GemmArgs& a, int tiles_n, int tiles_t, int nbuf, int split_from) {
    switch (a.e.kind) {
        case GemmEpilogue::F32: launch_one<FW, KG, GemmEpilogue::F32, KVT, TW>(a, tiles_n, tiles_t, nbuf, split_from); break;
        case GemmEpilogue::SWIGLU_A16: launch_one<FW, KG, GemmEpilogue::SWIGLU_A16, KVT, TW>(a, tiles_n, tiles_t, nbuf, split_from); break;
        case GemmEpilogue::RESID_A16: launch_one<FW, KG, GemmEpilogue::RESID_A16, KVT, TW>(a, tiles_n, tiles_t, nbuf, split_from); break;
        default: launch_one<FW, KG, GemmEpilogue::ROPE_KV, KVT, TW>(a, tiles_n, tiles_t, nbuf, split_from); break;
    }
}

}  // namespace

// `a` is complete but for the K split: a.ksplit > 1 asks for every tile to be cut that many ways
// (wide decode batches; slab and counters are the library's own), otherwise none is.
void launch_gemm_prefill(GemmArgs a) {
    const int cus = rt().num_cus;
    static const int shape_env = getenv("NVLLM_GEMM_PREFILL_SHAPE") ? atoi(getenv("NVLLM_GEMM_PREFILL_SHAPE")) : 0;   // 2: 8 waves, 3: 4 waves
    static const int ring_env = getenv("NVLLM_GEMM_PREFILL_RING") ? atoi(getenv("NVLLM_GEMM_PREFILL_RING")) : 0;
    static const int ksplit_env = getenv("NVLLM_GEMM_PREFILL_KSPLIT") ? atoi(getenv("NVLLM_GEMM_PREFILL_KSPLIT")) : 0;   // cut every tile this many ways
    static const int tw_env = getenv("NVLLM_GEMM_PREFILL_TW") ? atoi(getenv("NVLLM_GEMM_PREFILL_TW")) : 0;               // token blocks per wave: 4, 3, 2
#ifdef NVLLM_PF_ABLATE
    static const int ablate_env = getenv("NVLLM_PF_ABLATE") ? atoi(getenv("NVLLM_PF_ABLATE")) << 8 : 0;
#else
    constexpr int ablate_env = 0;
#endif
    const bool kv16 = a.e.kind == GemmEpilogue::ROPE_KV && a.e.kv_elem_bytes == 2;
    int want_split = a.ksplit > 1 ? a.ksplit : 1;
    if (ksplit_env > 1) want_split = ksplit_env;
    const int tiles_n = cdiv(a.NB, 8);
    // 128-feature tiles, 128, 96 or 64 tokens high.  A launch runs in rounds of co-resident
    // workgroups (three per CU with two ring buffers) and a partly filled last round costs
    // nearly a full one, so the height is the one whose rounds waste least: 128 tokens unless
    // 96 needs fewer round-heights (TinyLlama's gate/up at 1,591 tokens: 1,144 tiles = 1.5 rounds
    // against 1,496 = 1.95 rounds of 3/4 the height, 150 -> 147 us; a projection half as wide
    // 88 -> 74 us).  Projections whose 128-token tiles would not even give every CU one
    // workgroup (wo, w_down: 208) take 64-token tiles, two workgroups per CU with three buffers
    // (35 -> 32 and 80 -> 77 us against the eight-wave form, which wide decode batches keep:
    // there every weight tile must still be read only once).
    const int tiles4 = tiles_n * cdiv(a.Tb, 8);
    const bool wide_decode = a.Tb <= 8 || want_split > 1;
    int shape = wide_decode && tiles4 * want_split <= cus ? 2 : 3;
    if (shape_env == 2 || shape_env == 3) shape = shape_env;
    int tw = 4;
    if (shape == 3 && !wide_decode) {
        if (tiles4 <= cus) {
            tw = 2;
        } else if (tiles_n * cdiv(a.Tb, 4) <= 2 * cus) {
            // a few more 128-token tiles than CUs: some CUs would run two tall tiles in a row (or side by side)
            // while the rest run one; 64-token tiles give every CU at most two short ones (TinyLlama's
            // q/k/v at 1,591 tokens: 340 tiles of 96 tokens -> 500 of 64, 52 -> 47 us)
            tw = 2;
        } else {
            const int slots = 3 * cus;
            const int tiles3 = tiles_n * cdiv(a.Tb, 6);
            if (cdiv(tiles3, slots) * 3 < cdiv(tiles4, slots) * 4) tw = 3;
        }
    }
    if (shape == 3 && tw_env >= 2 && tw_env <= 4) tw = tw_env;
    static const int tw_qkv_env = getenv("NVLLM_GEMM_PREFILL_TW_QKV") ? atoi(getenv("NVLLM_GEMM_PREFILL_TW_QKV")) : 0;   // the same for q/k/v only
    if (shape == 3 && a.e.kind == GemmEpilogue::ROPE_KV && tw_qkv_env >= 2 && tw_qkv_env <= 4) tw = tw_qkv_env;
    const int tiles_t = cdiv(a.Tb, 2 * tw);
    const int total = tiles_n * tiles_t;
    int nbuf;
    if (shape == 2) {   // 48 KiB per stage (two K-steps); a ring of three fills the LDS
        nbuf = 3;
        if (ring_env >= 2 && ring_env <= 3) nbuf = ring_env;
    } else {            // 24 KiB per stage (20, 16 for the lower tiles); three waves per SIMD fit the register file
        const int per_cu = cdiv(total * want_split, cus);
        nbuf = per_cu >= 3 ? 2 : (per_cu == 2 ? 3 : 5);
        if (ring_env >= 2 && ring_env <= 5) nbuf = ring_env;
    }
    // ---- the K split
    int split_from = total;
    a.ksplit = 1;
    a.kb_per_split = a.Kb;
    if (want_split > 1) {
        const int kg = shape == 2 ? 2 : 1;
        a.kb_per_split = cdiv(cdiv(a.Kb, want_split), kg) * kg;   // whole stages per slice
        a.ksplit = cdiv(a.Kb, a.kb_per_split);
        if (a.ksplit > 1 && total <= (1 << 16)) {   // (the counter array's size, gemm.hip)
            split_from = 0;
            a.slab = (float*)ws_get(WS_SLAB, (size_t)total * a.ksplit * 128 * (32 * tw) * sizeof(float));
        } else {
            a.ksplit = 1;
            a.kb_per_split = a.Kb;
        }
    }
    const int nb = nbuf | ablate_env;
    if (shape == 2) {
        if (kv16) launch_pf<4, 2, __half>(a, tiles_n, tiles_t, nb, split_from);
        else launch_pf<4, 2, float>(a, tiles_n, tiles_t, nb, split_from);
    } else if (tw == 4) {
        if (kv16) launch_pf<4, 1, __half, 4>(a, tiles_n, tiles_t, nb, split_from);
        else launch_pf<4, 1, float, 4>(a, tiles_n, tiles_t, nb, split_from);
    } else if (tw == 3) {
        if (kv16) launch_pf<4, 1, __half, 3>(a, tiles_n, tiles_t, nb, split_from);
        else launch_pf<4, 1, float, 3>(a, tiles_n, tiles_t, nb, split_from);
    } else {
        if (kv16) launch_pf<4, 1, __half, 2>(a, tiles_n, tiles_t, nb, split_from);
        else launch_pf<4, 1, float, 2>(a, tiles_n, tiles_t, nb, split_from);
    }
}

}  // namespace nvllm
