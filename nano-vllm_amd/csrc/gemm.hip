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

#include "gemm_common.h"

namespace nvllm {

// Weight tiles are read once per launch: non-temporal loads (a build knob for A/B runs)
#ifdef NVLLM_W_PLAIN
#define NVLLM_WLOAD(p) (*(p))
#else
#define NVLLM_WLOAD(p) __builtin_nontemporal_load(p)
#endif

template <int NBW, int TB, int NW, int EPI, typename KVT>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs p) {
    static_assert(NBW % 2 == 0, "row blocks are reduced in pairs");
#ifndef NVLLM_RW
#define NVLLM_RW 4
#endif
#ifndef NVLLM_RA
#define NVLLM_RA 2
#endif
    constexpr int RW = NVLLM_RW;       // weight ring depth (K-steps in flight)
    constexpr int RA = NVLLM_RA;       // activation ring depth
    constexpr int ROUNDS = NBW / 2;    // LDS reduction rounds, two row blocks each
    constexpr int ITEMS = 2 * TB * 64; // float4 results per round
    constexpr int IPT = (ITEMS + NW * 64 - 1) / (NW * 64);  // items per thread per round
    // the waves' partial tiles, all row blocks at once: red[wave][row block][token block][lane]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    typedef f4_t RedTile[NBW][TB][64];
    RedTile* red = reinterpret_cast<RedTile*>(lds_dyn);
    __shared__ float s_rinv[TB * 16];
    __shared__ float s_part[(NW * 64) / (TB * 16)][TB * 16];
    __shared__ float s_ss[2][TB * 16];
    __shared__ unsigned s_ticket;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb0 = blockIdx.x * NBW;
    if (nb0 >= p.NB) return;   // grid padded for XCD alignment
    NVLLM_STAMP(0, lane);
    const int kb_begin = blockIdx.y * p.kb_per_split;
    const int kb_end = min(p.Kb, kb_begin + p.kb_per_split);
    const int tb0 = blockIdx.z * TB;
    const int first = kb_begin + wave;
    const int nsteps = first < kb_end ? (kb_end - first + NW - 1) / NW : 0;
    const h8_t* wp[NBW];
#pragma unroll
    for (int i = 0; i < NBW; i++)
        wp[i] = reinterpret_cast<const h8_t*>(p.W + (size_t)min(nb0 + i, p.NB - 1) * p.nb_stride) + lane;
    const h8_t* ap[TB];
#pragma unroll
    for (int j = 0; j < TB; j++)
        ap[j] = reinterpret_cast<const h8_t*>(p.A + (size_t)min(tb0 + j, p.Tb - 1) * p.Kb * 2 * TILE_HALVES) + lane;

    f4_t acc[NBW][TB];
#pragma unroll
    for (int i = 0; i < NBW; i++)
#pragma unroll
        for (int j = 0; j < TB; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    h8_t w[RW][NBW];
    h8_t ah[RA][TB], al[RA][TB];
    auto kb_of = [&](int s) { return first + NW * s; };

    // Loads are issued only for steps that exist (wave-uniform branches): a clamped
    // re-load of the last tile would be real HBM traffic, since streamed weights are
    // loaded non-temporally and do not stay in L2.
#pragma unroll
    for (int s = 0; s < RW; s++)
        if (s < nsteps) {
            const int kb = kb_of(s);
#pragma unroll
            for (int i = 0; i < NBW; i++) w[s][i] = NVLLM_WLOAD(wp[i] + (size_t)kb * 64);
        }
#pragma unroll
    for (int s = 0; s < RA; s++)
        if (s < nsteps) {
            const int kb = kb_of(s);
#pragma unroll
            for (int j = 0; j < TB; j++) {
                ah[s][j] = ap[j][(size_t)kb * 128];
                al[s][j] = ap[j][(size_t)kb * 128 + 64];
            }
        }
    // 1/rms of this token group, from the producer's per-tile sums of squares: every
    // thread sums a slice of one token's tiles (independent loads, issued while the
    // first weight tiles are in flight); the slices meet in LDS before the epilogue.
    if (p.e.ss_in != nullptr) {
        constexpr int PARTS = (NW * 64) / (TB * 16);
        const int tl = threadIdx.x % (TB * 16), part = threadIdx.x / (TB * 16);
        const int t = min(tb0 * 16 + tl, p.T - 1);
        const int src = p.e.ss_gather ? p.e.ss_gather[t] : t;
        const float* sp = p.e.ss_in + (size_t)src * p.e.ss_tiles;
        float ss = 0.f;
        for (int j = part; j < p.e.ss_tiles; j += PARTS) ss += sp[j];
        s_part[part][tl] = ss;
    }
    NVLLM_STAMP(1, lane);
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[r][i], al[r % RA][j], acc[i][j], 0, 0, 0);
                    }
                if (s == 0) NVLLM_STAMP(2, acc[NBW - 1][TB - 1][0]);
                if (s + RW < nsteps) {
                    const int kbw = kb_of(s + RW);
#pragma unroll
                    for (int i = 0; i < NBW; i++) w[r][i] = NVLLM_WLOAD(wp[i] + (size_t)kbw * 64);
                }
                if (s + RA < nsteps) {
                    const int kba = kb_of(s + RA);
#pragma unroll
                    for (int j = 0; j < TB; j++) {
                        ah[r % RA][j] = ap[j][(size_t)kba * 128];
                        al[r % RA][j] = ap[j][(size_t)kba * 128 + 64];
                    }
                }
            }
        }
    }
    NVLLM_STAMP(3, acc[NBW - 1][TB - 1][0]);
    // ---- 0. epilogue operands that do not depend on the sums: issue their loads now,
    // so that they travel while the K split is being closed (every workgroup does it;
    // only the one that ends up running the epilogue uses them).  Indices are clamped,
    // not predicated: unconditional loads.
    const GemmEpilogue& ep = p.e;
    float4 pf_x[ROUNDS][IPT], pf_g[ROUNDS][IPT];
    int pf_pos = 0, pf_row = -1;
    if (EPI == GemmEpilogue::RESID_A16) {
#pragma unroll
        for (int h = 0; h < ROUNDS; h++)
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                const int e = min((int)threadIdx.x + q * NW * 64, ITEMS - 1);
                const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
                const int t = min(tb0 * 16 + j * 16 + (l & 15), p.T - 1);
                const int n = min((nb0 + 2 * h + i) * 16 + 4 * (l >> 4), p.N - 4);
                pf_x[h][q] = *reinterpret_cast<const float4*>(ep.x + (size_t)t * p.N + n);
                pf_g[h][q] = *reinterpret_cast<const float4*>(ep.gain_next + n);
            }
    }
    if (EPI == GemmEpilogue::ROPE_KV) {   // all items of a thread belong to one token
        const int e = min((int)threadIdx.x, ITEMS - 1);
        const int t = min(tb0 * 16 + ((e >> 6) % TB) * 16 + (e & 15), p.T - 1);
        pf_pos = ep.pos[t];
        pf_row = ep.kv_row[t];
    }
    // ---- 1. sum the waves' partial tiles through LDS: every wave stores all its tiles, ONE
    // barrier, then thread `e` (item e = threadIdx.x + q * NW*64) collects, per round h (two row
    // blocks), the float4 for token t(e) and features n(h, e) .. n+3, in wave order.
    f4_t v[ROUNDS][IPT];
#pragma unroll
    for (int i = 0; i < NBW; i++)
#pragma unroll
        for (int j = 0; j < TB; j++) red[wave][i][j][lane] = acc[i][j];
    __syncthreads();
    if (ep.ss_in != nullptr && threadIdx.x < TB * 16) {
        // 1/rms per token from the slices gathered at the top (s_part: visible since the barrier
        // above; s_rinv is read by the epilogue after at least one more barrier)
        constexpr int PARTS = (NW * 64) / (TB * 16);
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; q++) ss += s_part[q][threadIdx.x];
        s_rinv[threadIdx.x] = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss, (float)ep.d_in), ep.eps)));
    }
#pragma unroll
    for (int h = 0; h < ROUNDS; h++) {
#pragma unroll
        for (int q = 0; q < IPT; q++) {
            const int e = threadIdx.x + q * NW * 64;
            f4_t sum = f4_t{0.f, 0.f, 0.f, 0.f};
            if (e < ITEMS) {
                const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
                sum = red[0][2 * h + i][j][l];
#pragma unroll
                for (int wv = 1; wv < NW; wv++) sum += red[wv][2 * h + i][j][l];
            }
            v[h][q] = sum;
        }
    }
    __syncthreads();   // everybody has read the partial tiles (the epilogue reuses the array) and s_rinv is visible
    NVLLM_STAMP(4, v[ROUNDS - 1][IPT - 1][0]);
    // ---- 2. close the K split: write-through slab, ticket, last arriver reduces.
    const unsigned tile_id = blockIdx.z * gridDim.x + blockIdx.x;
    if (p.ksplit > 1) {
        float* my = p.slab + ((size_t)tile_id * p.ksplit + blockIdx.y) * (ROUNDS * ITEMS * 4);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(my, 0, ROUNDS * ITEMS * 16, 0x00020000);
#pragma unroll
        for (int h = 0; h < ROUNDS; h++)
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                const int e = threadIdx.x + q * NW * 64;
                if (e < ITEMS)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v[h][q]), rs, (h * ITEMS + e) * 16, 0,
                                                           /*sc1: write through*/ 16);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0)
            s_ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (unsigned)p.ksplit - 1) return;     // not the last slice of this tile
        if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed
        __syncthreads();
        const float* base = p.slab + (size_t)tile_id * p.ksplit * (ROUNDS * ITEMS * 4);
        // Every slab load of both rounds is issued before the first add: one round trip.
        // Slice indices past the end are clamped (a repeated load), not predicated, so the
        // loads stay unconditional; sums are taken in slice order.
#pragma unroll
        for (int h = 0; h < ROUNDS; h++)
#pragma unroll
            for (int q = 0; q < IPT; q++) v[h][q] = f4_t{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.ksplit; s0 += 8) {
            f4_t part[ROUNDS][IPT][8];
#pragma unroll
            for (int h = 0; h < ROUNDS; h++)
#pragma unroll
                for (int q = 0; q < IPT; q++) {
                    const int e = min((int)threadIdx.x + q * NW * 64, ITEMS - 1);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int sl = min(s0 + u, p.ksplit - 1);
                        part[h][q][u] = *reinterpret_cast<const f4_t*>(base + ((size_t)sl * ROUNDS * ITEMS + h * ITEMS + e) * 4);
                    }
                }
#pragma unroll
            for (int h = 0; h < ROUNDS; h++)
#pragma unroll
                for (int q = 0; q < IPT; q++)
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (s0 + u < p.ksplit) v[h][q] += part[h][q][u];
        }
        // ready for the next launch (nobody else touches this tile's counter any more)
        if (threadIdx.x == 0) __hip_atomic_store(p.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    NVLLM_STAMP(5, v[ROUNDS - 1][IPT - 1][0]);
    // ---- 3. epilogue on complete sums
    float ssq[ROUNDS][IPT];
#pragma unroll
    for (int h = 0; h < ROUNDS; h++)
#pragma unroll
        for (int q = 0; q < IPT; q++) ssq[h][q] = 0.f;
#pragma unroll
    for (int h = 0; h < ROUNDS; h++) {
        if (EPI == GemmEpilogue::SWIGLU_A16) {
            // gate (even row block, i = 0) and up (odd, i = 1) sit in different threads:
            // swap through LDS so that the i = 0 thread has both
            __syncthreads();
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                const int e = threadIdx.x + q * NW * 64;
                if (e < ITEMS) red[0][(e >> 6) / TB][(e >> 6) % TB][e & 63] = v[h][q];
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < IPT; q++) {
            const int e = threadIdx.x + q * NW * 64;
            if (e >= ITEMS) continue;
            const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
            const int tl = j * 16 + (l & 15);
            const int t = tb0 * 16 + tl;
            const int nb = nb0 + 2 * h + i;
            const int n = nb * 16 + 4 * (l >> 4);
            const bool live = t < p.T && tb0 + j < p.Tb && nb < p.NB;
            const float rinv = ep.ss_in ? s_rinv[tl] : 1.0f;
            float y[4] = {v[h][q][0] * rinv, v[h][q][1] * rinv, v[h][q][2] * rinv, v[h][q][3] * rinv};
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
                if (i == 0 && live) {
                    const f4_t u = red[0][1][j][l];
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) z[r] = silu_mul(y[r], u[r] * rinv);
                    const int hid = (nb >> 1) * 16 + 4 * (l >> 4);
                    store_a16_4(ep.a16_out, ep.Kb_out, t, hid, z);
                }
            } else if (EPI == GemmEpilogue::RESID_A16) {
                if (live) {
                    float* xr = ep.x + (size_t)t * p.N + n;
                    const float4 xo = pf_x[h][q];
                    const float4 g = pf_g[h][q];
                    // residual add as the reference's accum(): x = x + branch (layers.c:163-168)
                    const float xn[4] = {__fadd_rn(xo.x, y[0]), __fadd_rn(xo.y, y[1]), __fadd_rn(xo.z, y[2]), __fadd_rn(xo.w, y[3])};
                    *reinterpret_cast<float4*>(xr) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                    const float z[4] = {xn[0] * g.x, xn[1] * g.y, xn[2] * g.z, xn[3] * g.w};
                    store_a16_4(ep.a16_out, ep.Kb_out, t, n, z);
                    ssq[h][q] += xn[0] * xn[0] + xn[1] * xn[1] + xn[2] * xn[2] + xn[3] * xn[3];
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
                        *reinterpret_cast<float4*>(ep.q_out + (size_t)t * ep.QD + n) = make_float4(y[0], y[1], y[2], y[3]);
                    } else if (pf_row >= 0) {
                        const long block = pf_row / ep.bs;
                        const int key = pf_row % ep.bs;
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
    }
    if (EPI == GemmEpilogue::RESID_A16) {
        // per-token sum of x^2 over each 32-feature group (= one round: two row blocks):
        // over the 4 lane groups (l>>4), then over the round's two row blocks (i)
        const int ss_cols = (p.N + 31) >> 5;
#pragma unroll
        for (int h = 0; h < ROUNDS; h++) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < IPT; q++) {
                float sq = ssq[h][q];
                sq += __shfl_xor(sq, 16, 64);
                sq += __shfl_xor(sq, 32, 64);
                const int e = threadIdx.x + q * NW * 64;
                if (e < ITEMS && (e & 63) < 16) s_ss[(e >> 6) / TB][((e >> 6) % TB) * 16 + (e & 15)] = sq;
            }
            __syncthreads();
            if (threadIdx.x < TB * 16) {
                const int t = tb0 * 16 + threadIdx.x;
                const int col = blockIdx.x * ROUNDS + h;
                if (t < p.T && col < ss_cols) ep.ss_out[(size_t)t * ss_cols + col] = s_ss[0][threadIdx.x] + s_ss[1][threadIdx.x];
            }
        }
    }
#ifdef NVLLM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores issued so far have left the CU
#endif
    NVLLM_STAMP(6, ssq[0][0]);
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
template <int NBW, int TB, int NW, int EPI, typename KVT>
static void launch_k(const GemmArgs& a, dim3 grid) {
    constexpr int shmem = NW * NBW * TB * 64 * 16;
    static_assert(shmem <= 144 * 1024, "partial tiles must fit the CU's LDS next to the static arrays");
    static bool enabled = false;
    if (shmem > 64 * 1024 && !enabled) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<NBW, TB, NW, EPI, KVT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
        enabled = true;
    }
    NVLLM_LAUNCH((k_gemm<NBW, TB, NW, EPI, KVT>), grid, dim3(NW * 64), shmem, a);
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

int launch_gemm(const WeightEntry& w, const __half* tiles, int N, const __half* a16, int T,
                const GemmEpilogue& epi, int force_ksplit, int prof_cls, const GemmWorkspace* wsp) {
    if (T <= 0 || N <= 0) return 1;
    ensure_init();
    // the caller's workspace (counters already zero), or the library's own grow-only buffers
    if (!wsp && !g_counters) {
        HIP_CHECK(hipMalloc((void**)&g_counters, MAX_COUNTERS * sizeof(unsigned)));
        HIP_CHECK(hipMemsetAsync(g_counters, 0, MAX_COUNTERS * sizeof(unsigned), rt().stream));
    }
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
        if (wide_decode_k2 && T < pf_min_t) a.ksplit = 2;   // every tile's K in two (slab: the launcher's)
#ifdef NVLLM_STAMPS
        a.stamps = nullptr;
        if (g_gemm_stamps && g_stamp_cap > 0) {
            a.stamps = g_gemm_stamps + (size_t)(g_stamp_next % g_stamp_cap) * STAMP_WGS * 8;
            g_stamp_next++;
        }
#endif
        ProfScope ps(prof_cls, bytes, 2.0 * T * (double)N * w.K);
        launch_gemm_prefill(a);
        return a.ksplit;
    }
    int TB = T <= 16 ? 1 : (T <= 32 ? 2 : 4);
    int nbw = GEMM_NBW;
    int ks = force_ksplit > 0 ? force_ksplit : gemm_pick_ksplit(N, w.K, T);
    // Narrow projections with a short K (wo, qkv at decode): instead of cutting K across
    // workgroups -- whose partial sums then meet through memory, a chain of dependent
    // round trips -- cut the TOKENS: 32-feature tiles x 16-token groups, every workgroup
    // sweeps the whole K and finishes its own outputs.  The weight tile is read by the
    // token groups' workgroups (one read from HBM, the rest from cache); worth it while a
    // full sweep is short.
    static const int tsplit_env = getenv("NVLLM_GEMM_TSPLIT") ? atoi(getenv("NVLLM_GEMM_TSPLIT")) : 1;
    const bool tsplit = tsplit_env && force_ksplit <= 0 && ks > 1 && TB > 1 && a.Kb <= 64 && a.Tb <= 8 &&
                        epi.kind != GemmEpilogue::SWIGLU_A16;
    if (tsplit) {
        // 16-token groups unless that makes more workgroups than CUs (a CU with two of
        // them takes twice as long: measured 12.3 us against 10.0 for qkv); then 32-token groups
        const bool wide = cdiv(a.NB, 2) * a.Tb > rt().num_cus;
        TB = wide ? 2 : 1;
        nbw = 2;
        ks = 1;
    }
    // Long-K projections that still need a K split (w_down): half the split, two 32-token
    // groups -- the same number of workgroups, but each tile's reduction reads half as
    // many, half as large partials (measured 14.9 -> 13.8 us; more workgroups than CUs,
    // or a shallower split, measured slower).
    if (!tsplit && force_ksplit <= 0 && TB == 4 && ks >= 4 && ks % 2 == 0 && a.Tb <= 4) {
        TB = 2;
        ks /= 2;
    }
    // The wide gate/up projection at full decode width: 64 x 64 tiles give 176 workgroups
    // (80 CUs idle) pulling 768 KiB each; 96-feature x 32-token tiles give 236 workgroups
    // pulling 640 KiB each, still in one wave of workgroups.
    static const int gu_env = getenv("NVLLM_GEMM_GU96") ? atoi(getenv("NVLLM_GEMM_GU96")) : 1;
    const bool gu96 = gu_env && !tsplit && force_ksplit <= 0 && epi.kind == GemmEpilogue::SWIGLU_A16 && ks == 1 &&
                      a.Tb > 2 && a.Tb <= 4 && cdiv(a.NB, 6) * 2 <= rt().num_cus && cdiv(a.NB, 4) < rt().num_cus;
    if (gu96) {
        TB = 2;
        nbw = 6;
    }
    int tiles_n = cdiv(a.NB, nbw);
    const int tiles_t = cdiv(a.Tb, TB);
    // Token groups of one tile read the same weights: with workgroups dealt round-robin to
    // the 8 XCDs, a grid width that is a multiple of 8 puts them behind the same L2 (speed
    // only; surplus workgroups exit at once).
    if (tiles_t > 1 && ks == 1) tiles_n = (tiles_n + 7) & ~7;
    if ((long)tiles_n * tiles_t > max_counters) ks = 1;
    a.kb_per_split = cdiv(a.Kb, ks);
    ks = cdiv(a.Kb, a.kb_per_split);
    const size_t slab_tile_bytes = (size_t)(nbw / 2) * 2 * TB * 64 * 16;
    // a caller's slab that cannot hold the partial tiles: every workgroup sweeps the whole K instead
    if (wsp && ks > 1 && (size_t)tiles_n * tiles_t * ks * slab_tile_bytes > wsp->slab_bytes) {
        ks = 1;
        a.kb_per_split = a.Kb;
    }
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
    ProfScope ps(prof_cls, bytes, 2.0 * T * (double)N * w.K);
    const bool kv16 = epi.kind == GemmEpilogue::ROPE_KV && epi.kv_elem_bytes == 2;
    if (nbw == 6) {
        launch_k<6, 2, GEMM_NW, GemmEpilogue::SWIGLU_A16, float>(a, grid);
    } else if (nbw == 2) {   // token-split configuration
        if (kv16) { if (TB == 1) launch_tb<2, 1, __half>(a, grid); else launch_tb<2, 2, __half>(a, grid); }
        else { if (TB == 1) launch_tb<2, 1, float>(a, grid); else launch_tb<2, 2, float>(a, grid); }
    } else if (epi.kind == GemmEpilogue::F32 && TB == 4 && ks == 1 && (long)tiles_n * tiles_t > rt().num_cus) {
        // more tiles than CUs (the classifier): four waves per workgroup, so that two
        // workgroups share a CU and one's start-up overlaps the other's streaming (41 -> 35.5 us)
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
    return ks;
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
