// gemm_prefill.hip -- the projections when a batch carries hundreds of tokens (prefill
// chunks): out[T, N] = act[T, K] x W[N, K]^T as a dense contraction on the matrix cores.
//
// At 64 tokens (gemm.hip) every weight byte feeds 2*64 flops and the waves of a
// workgroup split K, each pulling its own operands.  From a few hundred tokens on the
// work is bound by the matrix cores and by the operand traffic INTO the CU, so here a
// workgroup owns a 128-feature x 128-token output tile and its four waves (2 x 2, each
// 64 x 64 = 4 x 4 accumulator tiles) walk K together: per K-step the workgroup's 8 weight
// tiles and 8 x 2 activation tiles (hi, lo planes) are copied global -> LDS once, by
// direct-to-LDS loads, and every wave reads the 4 + 8 tiles it needs from LDS.  That is
// 24 KiB into the CU per 128 MFMAs instead of 48 KiB when each wave loads for itself.
//
// Both operands are stored in matrix-core operand order, one 1-KiB tile per (16 rows,
// 32 k), so a tile is copied by ONE wave instruction (64 lanes x 16 B, lane-linear in
// LDS) and read back by one conflict-free ds_read_b128 per lane: no swizzle, no
// transposition.  Two LDS buffers: the copy of step k+1 runs while step k is
// multiplied; two workgroups fit a CU, so one computes while the other waits at its
// barrier.
//
// The epilogues are those of gemm.hip (F32, SWIGLU_A16, RESID_A16, ROPE_KV) applied
// straight from the accumulators: a lane holds, per accumulator tile, four consecutive
// features of one token, the gate and up rows of a hidden unit sit in the same lane, and
// the per-token sums of squares need two cross-lane adds -- no LDS pass.
#include "gemm_common.h"

namespace nvllm {

namespace {

constexpr int PF_FB = 8;   // 16-row weight blocks per workgroup tile (128 features)
constexpr int PF_TB = 8;   // 16-token blocks per workgroup tile (128 tokens)
constexpr int PF_STAGE_TILES = PF_FB + 2 * PF_TB;          // 24 tiles of 1 KiB per K-step
constexpr int PF_STAGE_HALVES = PF_STAGE_TILES * TILE_HALVES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int EPI, typename KVT>
__global__ __launch_bounds__(256, 2) void k_gemm_prefill(GemmArgs p, int tiles_n, int tiles_t) {
    // one object holds all LDS: [2 buffers][24 tiles][512 halves], then 128 floats of 1/rms
    __shared__ __attribute__((aligned(16))) __half lds[2 * PF_STAGE_HALVES + 2 * PF_TB * 16];
    float* s_rinv = reinterpret_cast<float*>(lds + 2 * PF_STAGE_HALVES);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;   // this wave's 64 features x 64 tokens inside the tile

    // Workgroups are dealt round-robin to the 8 XCDs.  Give every XCD a contiguous range
    // of the (feature tile, token tile) sequence, token tiles fastest: the workgroups that
    // run together behind one L2 then share a few weight tiles and all walk the same
    // activations.
    const int total = tiles_n * tiles_t;
    int id = blockIdx.x;
    {
        const int xcd = id & 7, slot = id >> 3;
        const int q = total >> 3, r = total & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int nb0 = (id / tiles_t) * PF_FB;
    const int tb0 = (id % tiles_t) * PF_TB;

    // ---- who copies what: tile q of a stage (0..7 weights, 8..23 activations) is copied
    // by wave q / 6; per copy a lane-dependent source address and a wave-uniform LDS slot
    const __half* src[6];
    int step[6];
#pragma unroll
    for (int u = 0; u < 6; u++) {
        const int q = wave * 6 + u;
        if (q < PF_FB) {
            src[u] = p.W + (size_t)min(nb0 + q, p.NB - 1) * p.nb_stride + lane * 8;
            step[u] = TILE_HALVES;
        } else {
            const int tb = min(tb0 + ((q - PF_FB) >> 1), p.Tb - 1);
            src[u] = p.A + ((size_t)tb * p.Kb * 2 + ((q - PF_FB) & 1)) * TILE_HALVES + lane * 8;
            step[u] = 2 * TILE_HALVES;
        }
    }
    auto stage = [&](int kb, int buf) {
#pragma unroll
        for (int u = 0; u < 6; u++)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src[u] + (size_t)kb * step[u]),
                                             (lds_void*)(lds + buf * PF_STAGE_HALVES + (wave * 6 + u) * TILE_HALVES), 16, 0, 0);
    };

    f4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    // 1/rms of the tile's tokens from the producer's sums of squares (two threads per token)
    if (p.e.ss_in != nullptr) {
        const int tl = threadIdx.x & 127, part = threadIdx.x >> 7;
        const int t = min(tb0 * 16 + tl, p.T - 1);
        const int srow = p.e.ss_gather ? p.e.ss_gather[t] : t;
        const float* sp = p.e.ss_in + (size_t)srow * p.e.ss_tiles;
        float ss = 0.f;
        for (int j = part; j < p.e.ss_tiles; j += 2) ss += sp[j];
        // the two halves meet in the result array itself: the upper half writes, the lower adds
        if (part == 1) s_rinv[tl] = ss;
        __syncthreads();
        if (part == 0) s_rinv[tl] = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss + s_rinv[tl], (float)p.e.d_in), p.e.eps)));
    }

    const h8_t* lw = reinterpret_cast<const h8_t*>(lds) + (wr * 4) * 64 + lane;                 // weight tiles of this wave
    const h8_t* la = reinterpret_cast<const h8_t*>(lds) + (PF_FB + wc * 4 * 2) * 64 + lane;     // its activation tiles
    for (int kb = 0; kb < p.Kb; kb++) {
        const int buf = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies of step kb have landed
        __syncthreads();                                    // ... everybody's; and step kb-1 has been read by all
        if (kb + 1 < p.Kb) stage(kb + 1, buf ^ 1);
        const h8_t* bw = lw + buf * (PF_STAGE_HALVES / 8);
        const h8_t* ba = la + buf * (PF_STAGE_HALVES / 8);
        h8_t wf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) wf[i] = bw[i * 64];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const h8_t ah = ba[(2 * j) * 64], al = ba[(2 * j + 1) * 64];
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], ah, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], al, acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue, straight from the accumulators.  acc[i][j] of lane l: token
    // t = (tb0 + wc*4 + j)*16 + (l & 15), features n .. n+3 with n = nb*16 + 4*(l >> 4),
    // nb = nb0 + wr*4 + i.
    __syncthreads();   // s_rinv written before the loop is visible to all (also when Kb == 0)
    const GemmEpilogue& ep = p.e;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int tbj = tb0 + wc * 4 + j;
        const int tl = (wc * 4 + j) * 16 + (lane & 15);
        const int t = tb0 * 16 + tl;
        const bool tok_live = t < p.T && tbj < p.Tb;
        const float rinv = ep.ss_in ? s_rinv[tl] : 1.0f;
        int pos = 0, kv_row = -1;
        if (EPI == GemmEpilogue::ROPE_KV && tok_live) {
            pos = ep.pos[t];
            kv_row = ep.kv_row[t];
        }
        float ssq[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int nb = nb0 + wr * 4 + i;
            const int n = nb * 16 + 4 * (lane >> 4);
            const bool live = tok_live && nb < p.NB;
            float y[4] = {acc[i][j][0] * rinv, acc[i][j][1] * rinv, acc[i][j][2] * rinv, acc[i][j][3] * rinv};
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
                // row blocks alternate gate (even), up (odd): i = 0,2 are gates, i+1 their ups
                if ((i & 1) == 0 && live) {
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) z[r] = silu_mul(y[r], acc[(i + 1) & 3][j][r] * rinv);
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
            for (int h = 0; h < 2; h++) {
                float sq = ssq[h];
                sq += __shfl_xor(sq, 16, 64);
                sq += __shfl_xor(sq, 32, 64);
                const int col = (nb0 + wr * 4 + 2 * h) >> 1;
                if (lane < 16 && tok_live && col < ss_cols) ep.ss_out[(size_t)t * ss_cols + col] = sq;
            }
        }
    }
}

template <typename KVT>
void launch_pf(const GemmArgs& a, int tiles_n, int tiles_t) {
    const dim3 grid(tiles_n * tiles_t), block(256);
    switch (a.e.kind) {
        case GemmEpilogue::F32:
            NVLLM_LAUNCH((k_gemm_prefill<GemmEpilogue::F32, KVT>), grid, block, 0, a, tiles_n, tiles_t);
            break;
        case GemmEpilogue::SWIGLU_A16:
            NVLLM_LAUNCH((k_gemm_prefill<GemmEpilogue::SWIGLU_A16, KVT>), grid, block, 0, a, tiles_n, tiles_t);
            break;
        case GemmEpilogue::RESID_A16:
            NVLLM_LAUNCH((k_gemm_prefill<GemmEpilogue::RESID_A16, KVT>), grid, block, 0, a, tiles_n, tiles_t);
            break;
        default:
            NVLLM_LAUNCH((k_gemm_prefill<GemmEpilogue::ROPE_KV, KVT>), grid, block, 0, a, tiles_n, tiles_t);
            break;
    }
}

}  // namespace

// `a` is complete except for the K split, which this kernel does not use.
void launch_gemm_prefill(GemmArgs a) {
    a.ksplit = 1;
    a.kb_per_split = a.Kb;
    const int tiles_n = cdiv(a.NB, PF_FB), tiles_t = cdiv(a.Tb, PF_TB);
    if (a.e.kind == GemmEpilogue::ROPE_KV && a.e.kv_elem_bytes == 2) launch_pf<__half>(a, tiles_n, tiles_t);
    else launch_pf<float>(a, tiles_n, tiles_t);
}

}  // namespace nvllm
