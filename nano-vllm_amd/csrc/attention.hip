// attention.hip -- causal attention of a ragged batch over the block pool
// (reference src/kernels/attention.c:256-336 and its single-sequence / contiguous
// siblings) on the matrix cores, in two variants selected by the pool's format.
//
// Common to both: K and V are streamed from HBM exactly once per (sequence, kv head)
// and go straight from global memory into MFMA operands -- no LDS staging, no
// cross-lane broadcast.  A work item is
//     (run of up to CT consecutive tokens of one sequence) x (group of GT query
//     heads sharing a KV head) [x (slice of the key pages)],   CT * GT = 16;
// the host describes the runs (first token, count, last position, block-table offset).
// The 16 (token, head) pairs are the 16 "columns" of every MFMA; a decode token
// fills GT of them, a prefill chunk all 16, so prefill re-reads K/V CT times less.
//
// fp32 pool (the reference's rows), k_paged_attn: one WAVE per item and key slice,
// exact fp32 on v_mfma_f32_16x16x4_f32.  Per page (16 keys = the pool's block size):
//   S^T[key, col] = sum_d K[key, d] * Q[col, d]      A = K (lane: key = l&15, 4 dims
//                                                     chosen by l>>4), B = Q^T
//       -> lane (col = l&15, g = l>>4) holds the scores of keys 4g..4g+3 of ITS column,
//          so max / sum / rescale are lane-local plus two 16-lane-stride shuffles;
//   O^T[d, col]  += sum_key V[key, d] * P[col, key]   A = V^T, B = P^T: the score
//          registers are already in B-operand order (k = key 4g+s  <->  register s),
//          and the result keeps `col` on the lane, next to its running max and sum.
// fp32 MFMA is an exact fma chain, so the arithmetic is fp32 throughout like the
// reference; only the summation order differs.  Key slices of one column leave
// (acc, max, sum) partials that k_attn_combine merges.
//
// fp16 pool (pages in operand order), k_paged_attn16: see the comment above it.
#include "attn_tile.h"

namespace nvllm {

typedef float f4_t __attribute__((ext_vector_type(4)));

struct alignas(16) H8 {
    __half v[8];
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
    return make_float4(a.x, a.y, b.x, b.y);
}

template <int HD, typename KVT>
struct PageRegs {
    static constexpr int NC = HD / 16;        // 16-dim chunks of a key row
    static constexpr int NV = (HD + 63) / 64;  // 64-dim groups of a value row
    float4 k[NC];
    float4 v[NV][4];
};

// One page of operands.  `live` = number of rows of the page that hold positions
// <= the item's last position.  Row indices are clamped to live-1 instead of being
// predicated: a clamped row is a written (finite) row, its score is replaced by
// -inf and its probability is exactly 0, so it contributes 0 * finite = 0.  All
// loads are unconditional, which lets the compiler wait with counted vmcnt and
// keep the next page in flight while this one is consumed.
template <int HD, typename KVT>
__device__ __forceinline__ void load_page(PageRegs<HD, KVT>& r, const KVT* __restrict__ kpage,
                                          const KVT* __restrict__ vpage, int live, int rowstride, int lane) {
    constexpr int NC = HD / 16, NV = (HD + 63) / 64;
    const int x = lane & 15, g = lane >> 4;
    // K as the A operand: lane (key = x, dims 16c + 4g .. +3)
    const int koff = min(x, live - 1) * rowstride + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; c++) r.k[c] = ld4(kpage + koff + 16 * c);
    // V^T as the A operand: lane (dims 4x .. 4x+3 [+64n], key 4g + s)
    const int dx = 4 * x < HD ? 4 * x : 0;   // lanes past the head width re-read dims 0..3 (result rows unused)
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int voff = min(4 * g + s, live - 1) * rowstride + dx;
#pragma unroll
        for (int n = 0; n < NV; n++) r.v[n][s] = ld4(vpage + voff + (64 * n + 4 * x < HD ? 64 * n : 0));
    }
}

// NSW = 0: every wave is an (item, key slice) of its own, four per workgroup; slices leave partials in
// memory for k_attn_combine.  NSW = 4 or 8: the NSW waves of a workgroup are the key slices of ONE item
// (a.nsplit = NSW) and merge through LDS -- no partials in memory, no second launch.
template <int HD, typename KVT, int NSW = 0>
__global__ __launch_bounds__(NSW == 8 ? 512 : 256) void k_paged_attn(AttnKArgs a) {
    constexpr int NC = HD / 16, NV = (HD + 63) / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // 32-bit index arithmetic throughout (64-bit division is a long software loop)
    const unsigned NG = (unsigned)a.NG;
    const unsigned widx = NSW ? blockIdx.x * (unsigned)NSW + (unsigned)wave : blockIdx.x * 4u + (unsigned)wave;
    if (NSW == 0 && widx >= (unsigned)a.n_runs * NG * (unsigned)a.nsplit) return;
    // NSW = 0: head group fastest -- the waves of a workgroup read neighbouring 256-byte
    // pieces of the same K/V rows at the same time (DRAM page locality); NSW > 0: slice fastest
    const int hg = NSW ? (int)((widx / (unsigned)NSW) % NG) : (int)(widx % NG);
    const unsigned rest = NSW ? (widx / (unsigned)NSW) / NG : widx / NG;
    const int split = NSW ? wave : (int)(rest % (unsigned)a.nsplit);
    const int4 run = a.runs[NSW ? rest : rest / (unsigned)a.nsplit];
    const int t0 = run.x, ntok = run.y, plast = run.z;   // positions rise by one within a run
    const int h0 = hg * a.GT;
    const int kvh = (int)((unsigned)h0 / (unsigned)a.group);

    // this lane's column
    const int x = lane & 15, g = lane >> 4;
    const int ctok = x >> a.gt_shift, chead = x & (a.GT - 1);   // GT is a power of two
    const bool cvalid = ctok < ntok;
    const int tcol = t0 + (cvalid ? ctok : 0);
    const int plim = cvalid ? plast - (ntok - 1 - ctok) : -1;   // last key this column may see (none for padding)
    const int npages = (plast >> 4) + 1;
    const int per = (int)((unsigned)(npages + a.nsplit - 1) / (unsigned)a.nsplit);
    const int pg_begin = split * per;
    const int pg_end = min(npages, pg_begin + per);
    const int* table = a.tables ? a.tables + run.w : nullptr;
    const KVT* pk = reinterpret_cast<const KVT*>(a.pool_k) + a.layer_off + (size_t)kvh * HD;
    const KVT* pv = reinterpret_cast<const KVT*>(a.pool_v) + a.layer_off + (size_t)kvh * HD;

    // Q^T as the B operand: lane (col = x, dims 16c + 4g .. +3)
    float4 qv[NC];
    {
        const float* qr = a.q + ((size_t)tcol * a.H + h0 + chead) * HD + 4 * g;
#pragma unroll
        for (int c = 0; c < NC; c++) qv[c] = cvalid ? ld4(qr + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    float m = -INFINITY, lsum = 0.f;
    f4_t oacc[NV][4];
#pragma unroll
    for (int n = 0; n < NV; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) oacc[n][e] = f4_t{0.f, 0.f, 0.f, 0.f};

    const int rs = (int)a.rowstride;
    // page -> operand base pointers (wave-uniform: scalar loads of the block table)
    // (a block of 16 * 2^bs_shift positions holds 2^bs_shift consecutive pages: memory.c:74-82)
    const int sh = a.bs_shift;
    auto row_of = [&](int pg) -> long {
        const int q = min(pg, pg_end - 1);
        return table ? (((long)table[q >> sh]) << (4 + sh)) + ((long)(q & ((1 << sh) - 1)) << 4) : (long)q * 16;
    };
    auto kbase = [&](int pg) -> const KVT* { return pk + row_of(pg) * rs; };
    auto vbase = [&](int pg) -> const KVT* { return pv + row_of(pg) * rs; };
    auto live_of = [&](int pg) -> int { return min(16, plast - min(pg, pg_end - 1) * 16 + 1); };

    auto compute = [&](const PageRegs<HD, KVT>& r, int pg) {
        // ---- scores: two accumulators so consecutive MFMAs are independent
        f4_t s0 = f4_t{0.f, 0.f, 0.f, 0.f}, s1 = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; c++) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.k[c].x, qv[c].x, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.k[c].y, qv[c].y, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.k[c].z, qv[c].z, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.k[c].w, qv[c].w, s1, 0, 0, 0);
        }
        float sc[4];
        float mx = -INFINITY;
        const int kfirst = pg * 16 + 4 * g;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const float v = (s0[rr] + s1[rr]) * a.scale;   // scale after the dot, like the reference
            sc[rr] = (kfirst + rr <= plim) ? v : -INFINITY;
            mx = fmaxf(mx, sc[rr]);
        }
#ifdef NVLLM_ATTN_SHFL_MAX
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
        mx = rows_max(mx);
#endif
        if (__any(mx > m)) {   // the running maximum of some column rises: rescale (rare after the first pages)
            const float mn = fmaxf(m, mx);
            const float alpha = mn == -INFINITY ? 1.f : expf(m - mn);
            lsum *= alpha;
#pragma unroll
            for (int n = 0; n < NV; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) oacc[n][e] *= alpha;
            m = mn;
        }
        float p[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            p[rr] = expf(sc[rr] - m);   // masked keys: exp(-inf) = 0; column with nothing visible: m = -inf -> nan-safe below
            if (m == -INFINITY) p[rr] = 0.f;
            lsum += p[rr];
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int n = 0; n < NV; n++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                oacc[n][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.v[n][s].x, p[s], oacc[n][0], 0, 0, 0);
                oacc[n][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.v[n][s].y, p[s], oacc[n][1], 0, 0, 0);
                oacc[n][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.v[n][s].z, p[s], oacc[n][2], 0, 0, 0);
                oacc[n][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.v[n][s].w, p[s], oacc[n][3], 0, 0, 0);
            }
    };

    // ---- page loop: the next page's loads are always in flight while the current
    // page is consumed.  Prefetches past the slice re-read its last page (an L2 hit).
    PageRegs<HD, KVT> ra, rb;
    if (pg_begin < pg_end) {
        load_page<HD, KVT>(ra, kbase(pg_begin), vbase(pg_begin), live_of(pg_begin), rs, lane);
        for (int pg = pg_begin; pg < pg_end; pg += 2) {
            load_page<HD, KVT>(rb, kbase(pg + 1), vbase(pg + 1), live_of(pg + 1), rs, lane);
            compute(ra, pg);
            load_page<HD, KVT>(ra, kbase(pg + 2), vbase(pg + 2), live_of(pg + 2), rs, lane);
            if (pg + 1 < pg_end) compute(rb, pg + 1);
        }
    }

    // ---- results: lane (col x, g) holds dims 64n + 16g + 4r + e of its column
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if constexpr (NSW > 0) {
        // the slices of the item meet in LDS; wave w then finishes the units (n, r) with 4n + r = w (mod NSW):
        // four consecutive dims of every column, summed over the slices in slice order as k_attn_combine does
        __shared__ float s_m[NSW][16], s_l[NSW][16];
        __shared__ f4_t s_o[NSW][NV * 4][64];
        if (g == 0) {
            s_m[wave][x] = m;
            s_l[wave][x] = lsum;
        }
#pragma unroll
        for (int n = 0; n < NV; n++)
#pragma unroll
            for (int e = 0; e < 4; e++) s_o[wave][n * 4 + e][lane] = oacc[n][e];
        __syncthreads();
        if (!cvalid) return;
        const int h = h0 + chead;
        float M = -INFINITY;
#pragma unroll
        for (int sl = 0; sl < NSW; sl++) M = fmaxf(M, s_m[sl][x]);
        float f[NSW], L = 0.f;
#pragma unroll
        for (int sl = 0; sl < NSW; sl++) {
            f[sl] = expf(s_m[sl][x] - M);   // empty slices carry max = -inf -> weight 0
            L = fmaf(s_l[sl][x], f[sl], L);
        }
        const float inv = 1.0f / L;
        for (int u = wave; u < NV * 4; u += NSW) {
            const int n = u >> 2, rr = u & 3;
            const int d0 = 64 * n + 16 * g + 4 * rr;
            if (d0 >= HD) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float acc = 0.f;
#pragma unroll
                for (int sl = 0; sl < NSW; sl++) acc = fmaf(s_o[sl][n * 4 + e][lane][rr], f[sl], acc);
                o[e] = acc * inv;
            }
            if (a.out_f32)
                *reinterpret_cast<float4*>(a.out_f32 + ((size_t)tcol * a.H + h) * HD + d0) = make_float4(o[0], o[1], o[2], o[3]);
            if (a.out_a16) {
                const int kk = h * HD + d0;
                alignas(8) __half hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float xv = fminf(fmaxf(o[e], -65504.f), 65504.f);
                    hi[e] = __float2half_rn(xv);
                    lo[e] = __float2half_rn(xv - __half2float(hi[e]));
                }
                const size_t tl = ((size_t)(tcol >> 4) * a.Kb_out + (kk >> 5)) * 2;
                const size_t off = (size_t)((tcol & 15) + 16 * ((kk >> 3) & 3)) * 8 + (kk & 7);
                *reinterpret_cast<uint2*>(a.out_a16 + tl * TILE_HALVES + off) = *reinterpret_cast<const uint2*>(hi);
                *reinterpret_cast<uint2*>(a.out_a16 + (tl + 1) * TILE_HALVES + off) = *reinterpret_cast<const uint2*>(lo);
            }
        }
        return;
    }
    if (!cvalid) return;
    const int h = h0 + chead;
    if (a.nsplit > 1) {
        float* pa = a.part_acc + (((size_t)tcol * a.H + h) * a.nsplit + split) * HD;
#pragma unroll
        for (int n = 0; n < NV; n++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int d = 64 * n + 16 * g + 4 * rr;
                if (d < HD)
                    *reinterpret_cast<float4*>(pa + d) =
                        make_float4(oacc[n][0][rr], oacc[n][1][rr], oacc[n][2][rr], oacc[n][3][rr]);
            }
        if (g == 0) {
            float* pm = a.part_ml + (((size_t)tcol * a.H + h) * a.nsplit + split) * 2;
            pm[0] = m;
            pm[1] = lsum;
        }
        return;
    }
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int n = 0; n < NV; n++) {
        const int d0 = 64 * n + 16 * g;
        if (d0 < HD) {
            float o[16];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int e = 0; e < 4; e++) o[4 * rr + e] = oacc[n][e][rr] * inv;
            if (a.out_f32) {
                float* dst = a.out_f32 + ((size_t)tcol * a.H + h) * HD + d0;
#pragma unroll
                for (int rr = 0; rr < 4; rr++)
                    *reinterpret_cast<float4*>(dst + 4 * rr) = make_float4(o[4 * rr], o[4 * rr + 1], o[4 * rr + 2], o[4 * rr + 3]);
            }
            if (a.out_a16) {
#pragma unroll
                for (int half8 = 0; half8 < 2; half8++) {
                    const int kk = h * HD + d0 + 8 * half8;
                    H8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float xv = fminf(fmaxf(o[8 * half8 + j], -65504.f), 65504.f);
                        hi.v[j] = __float2half_rn(xv);
                        lo.v[j] = __float2half_rn(xv - __half2float(hi.v[j]));
                    }
                    const size_t tl = ((size_t)(tcol >> 4) * a.Kb_out + (kk >> 5)) * 2;
                    const size_t off = (size_t)((tcol & 15) + 16 * ((kk >> 3) & 3)) * 8;
                    *reinterpret_cast<H8*>(a.out_a16 + tl * TILE_HALVES + off) = hi;
                    *reinterpret_cast<H8*>(a.out_a16 + (tl + 1) * TILE_HALVES + off) = lo;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- fp16 pool
// Same work decomposition for a pool in page order (see kv16_page): a page of one
// kv head is 32*HD contiguous bytes per pool, already in v_mfma_f32_16x16x16_f16
// operand order, so per page the wave issues 2*HD/16 contiguous 512-byte loads and
// HD/4 fp16 MFMAs and converts nothing:
//   S^T[key, col] = K[key, :] . Q[col, :]       A = K page, B = Q (fp16 hi + lo: two MFMAs)
//       -> lane (col = l&15, g = l>>4) holds keys 4g..4g+3 of its column, as above;
//   O[col, dim]  += P[col, key] V[key, dim]     A = P (those registers, fp16 hi + lo),
//                                               B = V page
//       -> lane (dim = 16n + (l&15), g) holds columns 4g..4g+3.
// fp16 products are exact in the fp32 accumulator and hi + lo carries 22 bits, so the
// arithmetic stays at fp32 accuracy.  Scores are kept in the log2 domain (one
// v_exp_f32 per probability).  Pages beyond the slice are requested through a
// zero-length buffer descriptor: the hardware returns zeros without touching memory,
// so the prefetch ring needs no branches and no clamped re-reads.
// The work of one wave on one item is Attn16Wave (attn_tile.h).
//
// SOLO: every wave takes an item of its own and walks all of its pages -- for short contexts with many
// items (a 7B-shape decode batch: 4,096 items of 11 pages), where four waves sharing an item would
// spend their time in the start-up and the merge rather than in the page loop.
template <int HD, int PF, bool SOLO = false, bool ONCE = false>
__global__ __launch_bounds__(256) void k_paged_attn16(AttnKArgs a) {
    constexpr int STRIDE = SOLO ? 1 : 4;   // pages between two consecutive pages of one wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one WORKGROUP per (item, key slice); its four waves take the slice's pages
    // round-robin and merge through LDS at the end
    const unsigned widx = SOLO ? blockIdx.x * 4u + (unsigned)wave : blockIdx.x;
    if (SOLO && widx >= (unsigned)a.NG * (unsigned)a.nsplit * (unsigned)a.n_runs) return;   // (no barrier in this form)
    Attn16Wave<HD, PF, STRIDE, false, ONCE> w;
    w.setup(a, widx, SOLO ? 0 : wave, lane, 0);
    w.load_q();
    w.start();
    w.loop();
    if constexpr (SOLO) {
        w.finish_solo();
    } else {
        __shared__ __attribute__((aligned(16))) unsigned char s_merge[attn16_lds<HD>()];
        w.deposit(s_merge, wave);
        __syncthreads();
        w.merge(s_merge, wave);
    }
}

// Merge the key-slice partials of each (token, head): thread = 8 consecutive dims.
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ part_acc,
                                                      const float* __restrict__ part_ml, int T, int H, int HD,
                                                      int nsplit, float* out_f32, __half* out_a16, int Kb_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = HD / 8;
    if (idx >= (long)T * H * c8) return;
    const int c = (int)(idx % c8);
    const long th = idx / c8;           // t * H + h
    const int h = (int)(th % H);
    const int t = (int)(th / H);
    const float* ml = part_ml + th * nsplit * 2;
    const float* pa = part_acc + th * nsplit * HD + c * 8;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; s++) M = fmaxf(M, ml[2 * s]);
    float L = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = 0.f;
    for (int s = 0; s < nsplit; s++) {
        const float w = expf(ml[2 * s] - M);   // empty slices carry max = -inf -> weight 0
        L = fmaf(ml[2 * s + 1], w, L);
        const float4 a0 = *reinterpret_cast<const float4*>(pa + (size_t)s * HD);
        const float4 a1 = *reinterpret_cast<const float4*>(pa + (size_t)s * HD + 4);
        o[0] = fmaf(a0.x, w, o[0]); o[1] = fmaf(a0.y, w, o[1]); o[2] = fmaf(a0.z, w, o[2]); o[3] = fmaf(a0.w, w, o[3]);
        o[4] = fmaf(a1.x, w, o[4]); o[5] = fmaf(a1.y, w, o[5]); o[6] = fmaf(a1.z, w, o[6]); o[7] = fmaf(a1.w, w, o[7]);
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] *= inv;
    if (out_f32) {
        float* dst = out_f32 + th * HD + c * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (out_a16) {
        const int kk = h * HD + c * 8;
        H8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float xv = fminf(fmaxf(o[j], -65504.f), 65504.f);
            hi.v[j] = __float2half_rn(xv);
            lo.v[j] = __float2half_rn(xv - __half2float(hi.v[j]));
        }
        const size_t tl = ((size_t)(t >> 4) * Kb_out + (kk >> 5)) * 2;
        const size_t off = (size_t)((t & 15) + 16 * ((kk >> 3) & 3)) * 8;
        *reinterpret_cast<H8*>(out_a16 + tl * TILE_HALVES + off) = hi;
        *reinterpret_cast<H8*>(out_a16 + (tl + 1) * TILE_HALVES + off) = lo;
    }
}

// ------------------------------------------------------------------ launch
template <int HD>
static void launch_hd(const AttnKArgs& k, long units, int elem_bytes, bool solo, bool once, int merge_waves) {
    const dim3 block(256);
    if (elem_bytes == 4) {   // fp32 rows: `units` are waves
        if (merge_waves == 8)        // the eight (four) key slices of an item in one workgroup
            NVLLM_LAUNCH((k_paged_attn<HD, float, 8>), dim3((unsigned)(units / 8)), dim3(512), 0, k);
        else if (merge_waves == 4)
            NVLLM_LAUNCH((k_paged_attn<HD, float, 4>), dim3((unsigned)(units / 4)), block, 0, k);
        else                         // four independent (item, slice) waves per workgroup
            NVLLM_LAUNCH((k_paged_attn<HD, float>), dim3((unsigned)((units + 3) / 4)), block, 0, k);
        return;
    }
    // fp16 pages: `units` are items.  Four cooperating waves per item, or -- many items of few pages --
    // a wave per item
    if (solo) {
        // (two pages in flight per wave; four measured slower: 76.8 against 73.9 us at the 7B shape)
        if (once) NVLLM_LAUNCH((k_paged_attn16<HD, 2, true, true>), dim3((unsigned)((units + 3) / 4)), block, 0, k);
        else NVLLM_LAUNCH((k_paged_attn16<HD, 2, true>), dim3((unsigned)((units + 3) / 4)), block, 0, k);
        return;
    }
    const dim3 grid((unsigned)units);
    // pages in flight per wave: two already cover the per-CU fetch rate (one workgroup
    // per CU, ~10 B/clk); deeper rings measured slower
    static const int pf = getenv("NVLLM_ATTN_PF") ? atoi(getenv("NVLLM_ATTN_PF")) : 2;
    if (HD <= 64 && pf == 8)
        NVLLM_LAUNCH((k_paged_attn16<HD, (HD <= 64 ? 8 : 2)>), grid, block, 0, k);
    else if ((pf == 2 || HD > 64) && once)
        NVLLM_LAUNCH((k_paged_attn16<HD, 2, false, true>), grid, block, 0, k);
    else if (pf == 2 || HD > 64)
        NVLLM_LAUNCH((k_paged_attn16<HD, 2>), grid, block, 0, k);
    else
        NVLLM_LAUNCH((k_paged_attn16<HD, 4>), grid, block, 0, k);
}

extern "C" float nvllm_host_attn_scale(int head_dim);

int attn_tokens_per_item(int H, int KVH) {
    int gt = 1;
    const int group = H / KVH;
    while (gt < 16 && group % (gt * 2) == 0) gt *= 2;   // gcd(group, 16)
    return 16 / gt;
}

void launch_paged_attention(const AttnArgs& a) {
    if (a.T <= 0) return;
    const int CT = attn_tokens_per_item(a.H, a.KVH), GT = 16 / CT;
    const int NG = a.H / GT;
    // The kernel walks pages of 16 positions; a pool with another block size is
    // addressed position by position through a page size of 1 ... not supported.
    // pages of 16 positions: a block of fp32 rows may hold 1, 2, 4 ... of them; fp16 pages are blocks of 16
    int bs_shift = 0;
    while ((16 << bs_shift) < a.pool.block_size) bs_shift++;
    if (a.tables && ((16 << bs_shift) != a.pool.block_size || (a.pool.elem_bytes == 2 && bs_shift != 0)))
        NVLLM_FATAL("paged attention supports block sizes 16, 32, 64 ... for fp32 rows and 16 for fp16 pages, got %d", a.pool.block_size);
    if (a.pool.elem_bytes == 2 && !a.tables) NVLLM_FATAL("an fp16 KV pool is always paged");
    const long base_items = (long)a.n_runs * NG;
    const int pages = (a.max_pos + 16) / 16;
    // Work items are cut into key slices until the chip is full.  fp32 rows: a unit is
    // a wave (the kernel is bound by its own instruction stream: 2 waves per SIMD).
    // fp16 pages: a unit is a workgroup whose four waves share an item and merge in
    // LDS; one workgroup per CU streams at the HBM rate, so slices -- whose partials
    // cost a round trip through memory and a merge launch -- are only cut when there
    // are fewer items than CUs.
    static const int env_units = getenv("NVLLM_ATTN_WAVES") ? atoi(getenv("NVLLM_ATTN_WAVES")) : 0;
    const bool wg_items = a.pool.elem_bytes == 2;
    const int target = env_units > 0 ? env_units : (wg_items ? rt().num_cus : 2048);
    int nsplit = (int)((target + base_items / 2) / base_items);
    const int min_pages = wg_items ? 8 : 2;   // pages per slice worth a slice
    if (nsplit > (pages + min_pages - 1) / min_pages) nsplit = (pages + min_pages - 1) / min_pages;
    if (nsplit < 1) nsplit = 1;
    // fp32 rows, 2..8 slices per item: the slices of an item share a workgroup (4 or 8 waves) and merge in LDS
    // instead of going through memory and k_attn_combine (NVLLM_ATTN_MERGE=0 keeps the two launches)
    static const int env_merge = getenv("NVLLM_ATTN_MERGE") ? atoi(getenv("NVLLM_ATTN_MERGE")) : 1;
    int merge_waves = 0;
    if (!wg_items && env_merge && nsplit > 1 && nsplit <= 8) {
        merge_waves = nsplit <= 4 ? 4 : 8;
        nsplit = merge_waves;
    }
    const long waves = base_items * nsplit;
    // a wave per item: at least eight items per CU and at most 16 pages each
    static const int env_solo = getenv("NVLLM_ATTN_SOLO") ? atoi(getenv("NVLLM_ATTN_SOLO")) : -1;
    bool solo = wg_items && nsplit == 1 && pages <= 16 && waves >= (long)rt().num_cus * 8;
    if (env_solo >= 0) solo = wg_items && env_solo > 0;

    AttnKArgs k;
    k.q = a.q;
    k.pool_k = a.pool.k;
    k.pool_v = a.pool.v;
    k.layer_off = (long)a.layer * a.pool.layer_stride;
    k.rowstride = (long)a.KVH * a.hd;
    k.page = 16;
    k.tables = a.tables;
    k.runs = (const int4*)a.runs;
    k.n_runs = a.n_runs;
    k.H = a.H;
    k.KVH = a.KVH;
    k.GT = GT;
    k.CT = CT;
    k.NG = NG;
    k.group = a.H / a.KVH;
    k.gt_shift = 0;
    while ((1 << k.gt_shift) < GT) k.gt_shift++;
    k.nsplit = nsplit;
    k.scale = nvllm_host_attn_scale(a.hd);
    k.out_f32 = a.out_f32;
    k.out_a16 = a.out_a16;
    k.Kb_out = cdiv(a.H * a.hd, 32);
    k.part_acc = nullptr;
    k.part_ml = nullptr;
    k.bs_shift = bs_shift;
    static const int nt_env = getenv("NVLLM_ATTN_NT") ? atoi(getenv("NVLLM_ATTN_NT")) : 1;
    const bool once = nt_env && a.n_runs == a.T;   // one token per run: no page is read by two items of a kv head
    if (nsplit > 1 && !merge_waves) {
        const size_t rows = (size_t)a.T * a.H * nsplit;
        float* part = (float*)ws_get(WS_ATT_PART, rows * (a.hd + 2) * sizeof(float));
        k.part_acc = part;
        k.part_ml = part + rows * a.hd;
    }
    {
        // algorithmic bytes: each (sequence run, kv head) reads its keys and values once
        const double kvb = (double)a.kv_positions * 2.0 * a.KVH * a.hd * a.pool.elem_bytes;
        ProfScope ps(PC_ATTN, kvb + (double)a.T * a.H * a.hd * 8.0, 4.0 * (double)a.kv_positions * CT * a.H * a.hd);
        switch (a.hd) {
            case 16: launch_hd<16>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 32: launch_hd<32>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 48: launch_hd<48>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 64: launch_hd<64>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 80: launch_hd<80>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 96: launch_hd<96>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            case 128: launch_hd<128>(k, waves, a.pool.elem_bytes, solo, once, merge_waves); break;
            default: NVLLM_FATAL("head_dim %d is not supported (16, 32, 48, 64, 80, 96, 128)", a.hd);
        }
    }
    if (nsplit > 1 && !merge_waves) {
        ProfScope ps(PC_ATTN_COMBINE, (double)a.T * a.H * nsplit * (a.hd + 2) * 4.0, 0);
        const long threads = (long)a.T * a.H * (a.hd / 8);
        NVLLM_LAUNCH(k_attn_combine, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                           k.part_acc, k.part_ml, a.T, a.H, a.hd, nsplit, a.out_f32, a.out_a16, k.Kb_out);
    }
}

}  // namespace nvllm
