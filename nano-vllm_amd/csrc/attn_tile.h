// attn_tile.h -- one attention work item over an fp16 page pool as device code: what a workgroup
// of k_paged_attn16 does (attention.hip describes the item, the page layout and the arithmetic).
// Two callers:
//   k_paged_attn16   one launch per layer; q and the newest K/V rows were written by the launch before
//   k_decode_chain   the whole decode step in one launch; q and the newest K/V rows were written by
//                    other workgroups of the same launch (COH = true)
//
// COH = true: q is loaded past this CU's vector cache (sc1), and so is every page from `fresh_page`
// on -- the pages that may hold rows written during this launch; results are stored write-through.
// Older pages were complete before the launch began and stay on plain loads, which is what lets the
// chain request them (start()) before the projection that produces q has finished.
#pragma once

#include "nvllm_internal.h"

namespace nvllm {

typedef float af4_t __attribute__((ext_vector_type(4)));
typedef _Float16 ah4_t __attribute__((ext_vector_type(4)));
typedef unsigned int au2_t __attribute__((ext_vector_type(2)));

struct AttnKArgs {
    const float* q;
    const void* pool_k;
    const void* pool_v;
    long layer_off;   // elements
    long rowstride;   // elements between consecutive positions of a page (KVH * hd)
    int page;         // positions per page as seen by the kernel (16)
    const int* tables;
    const int4* runs; // (first token, token count, last position, block-table offset) per item
    int n_runs;
    int H, KVH, GT, CT;
    int NG, group, gt_shift;
    int nsplit;
    float scale;
    float* out_f32;
    __half* out_a16;
    int Kb_out;
    float* part_acc;  // [t][h][split][HD]
    float* part_ml;   // [t][h][split][2]
    int bs_shift;     // fp32 rows: log2(block_size / 16) -- a block holds 2^bs_shift pages of 16 positions
};

__device__ __forceinline__ ah4_t as_ah4(au2_t v) { return __builtin_bit_cast(ah4_t, v); }

// Maximum / sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), the result in all four:
// two row swaps in the vector unit (v_permlane32_swap, v_permlane16_swap: rows 2,3 <-> 0,1 of a copy, then odd
// <-> even rows) instead of two trips through the LDS crossbar per reduction.
__device__ __forceinline__ float rows_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float m = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned v = __builtin_bit_cast(unsigned, m);
    const auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

// LDS an item's four waves need to merge: (max, sum) per wave and column, accumulators per wave
template <int HD>
constexpr int attn16_lds() {
    return 2 * 4 * 16 * 4 + 4 * (HD / 16) * 64 * 16;
}

// The state of one wave working on an item.  STRIDE = pages between two consecutive pages of the
// wave (4: four waves share the item; 1: the wave owns it).
template <int HD, int PF, int STRIDE, bool COH, bool ONCE = false>
struct Attn16Wave {
    static constexpr int NC = HD / 16;
    AttnKArgs a;   // (a copy: a pointer to the caller's record would keep that record in memory)
    int lane, x, g;
    int t0, ntok, plast, h0, kvh;
    int ctok, chead, tcol, plim;
    bool cvalid;
    int split, pg_begin, pg_end, fresh_page;
    const int* table;
    const __half* pk;
    const __half* pv;
    au2_t kr[PF][NC], vr[PF][NC];
    int nblk[PF];
    ah4_t qh[NC], ql[NC];
    float m, lsum;
    af4_t oacc[NC];

    // `wave_in_item`: 0..3 when four waves share the item, 0 when the wave owns it.  fresh_rule (COH
    // only) says which pages may hold rows written during this launch: 0 = any page, 1 = only the
    // page of the item's last position (a batch in which every sequence has exactly one token).
    __device__ __forceinline__ void setup(const AttnKArgs& args, unsigned widx, int wave_in_item, int lane_, int fresh_rule) {
        const unsigned rest = widx / (unsigned)args.NG;
        setup(args, widx, wave_in_item, lane_, fresh_rule, args.runs[rest / (unsigned)args.nsplit]);
    }
    // the same for a caller that already holds the item's run descriptor
    __device__ __forceinline__ void setup(const AttnKArgs& args, unsigned widx, int wave_in_item, int lane_, int fresh_rule, const int4 run) {
        a = args;
        lane = lane_;
        const unsigned NG = (unsigned)args.NG;
        const int hg = (int)(widx % NG);
        const unsigned rest = widx / NG;
        split = (int)(rest % (unsigned)args.nsplit);
        t0 = run.x, ntok = run.y, plast = run.z;
        h0 = hg * args.GT;
        kvh = (int)((unsigned)h0 / (unsigned)args.group);
        x = lane & 15, g = lane >> 4;
        ctok = x >> args.gt_shift, chead = x & (args.GT - 1);
        cvalid = ctok < ntok;
        tcol = t0 + (cvalid ? ctok : 0);
        plim = cvalid ? plast - (ntok - 1 - ctok) : -1;
        const int npages = (plast >> 4) + 1;
        const int per = (int)((unsigned)(npages + args.nsplit - 1) / (unsigned)args.nsplit);
        pg_begin = split * per + wave_in_item;   // this wave: pages pg_begin, +STRIDE, ...
        pg_end = min(npages, split * per + per);
        fresh_page = !COH ? 0x7fffffff : (fresh_rule == 0 ? 0 : (plast >> 4));
        table = args.tables + run.w;
        begin(args.layer_off);
    }
    // a new pass over the item's pages, in the pool slice that starts `layer_off` elements in (a caller
    // that works through the layers calls setup() once and begin() per layer)
    __device__ __forceinline__ void begin(long layer_off) {
        pk = reinterpret_cast<const __half*>(a.pool_k) + layer_off;
        pv = reinterpret_cast<const __half*>(a.pool_v) + layer_off;
        m = -INFINITY, lsum = 0.f;
#pragma unroll
        for (int n = 0; n < NC; n++) oacc[n] = af4_t{0.f, 0.f, 0.f, 0.f};
    }

    // every register of the page ring defined (for a caller that may not call start() on every path)
    __device__ __forceinline__ void clear_ring() {
#pragma unroll
        for (int s = 0; s < PF; s++) {
            nblk[s] = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) kr[s][c] = vr[s][c] = au2_t{0u, 0u};
        }
    }

    // block id of a page of this wave (a scalar load); looked up one ring turn before the
    // page is requested, so that the request never waits for the table
    __device__ __forceinline__ int blk_of(int pg) const { return table[pg < pg_end ? pg : 0]; }

    __device__ __forceinline__ void issue(int slot, int pg, int blk) {
        const bool ok = pg < pg_end;   // wave-uniform
        const long block = blk;
        const long off = kv16_page(block, kvh, a.KVH, HD);
        const int bytes = ok ? 32 * HD : 0;
        const int voff = lane * 8;
        auto rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(pk + off), 0, bytes, 0x00020000);
        auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(pv + off), 0, bytes, 0x00020000);
        if (COH && pg >= fresh_page) {   // wave-uniform
#pragma unroll
            for (int c = 0; c < NC; c++) kr[slot][c] = __builtin_amdgcn_raw_buffer_load_b64(rk, voff + 512 * c, 0, 16);
#pragma unroll
            for (int c = 0; c < NC; c++) vr[slot][c] = __builtin_amdgcn_raw_buffer_load_b64(rv, voff + 512 * c, 0, 16);
        } else {
            // ONCE: a decode batch reads every page once -> non-temporal (aux 2 = nt); the items of a prompt
            // chunk read the same pages and meet in L2 -> default policy
            constexpr int POLICY = ONCE ? 2 : 0;
#pragma unroll
            for (int c = 0; c < NC; c++) kr[slot][c] = __builtin_amdgcn_raw_buffer_load_b64(rk, voff + 512 * c, 0, POLICY);
#pragma unroll
            for (int c = 0; c < NC; c++) vr[slot][c] = __builtin_amdgcn_raw_buffer_load_b64(rv, voff + 512 * c, 0, POLICY);
        }
    }

    // Q as the B operand: lane (col = x, dims 16c + 4g .. +3), split into fp16 hi + lo
    __device__ __forceinline__ void load_q() {
        const float* qr = a.q + ((size_t)tcol * a.H + h0 + chead) * HD + 4 * g;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cvalid) {
                if constexpr (COH) {
                    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(qr + 16 * c),
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(qr + 16 * c + 2),
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    f = make_float4(__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                                    __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32)));
                } else {
                    f = *reinterpret_cast<const float4*>(qr + 16 * c);
                }
            }
            const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float cl = fminf(fmaxf(fv[e], -65504.f), 65504.f);
                qh[c][e] = (_Float16)cl;
                ql[c][e] = (_Float16)(cl - (float)qh[c][e]);
            }
        }
    }

    __device__ __forceinline__ void compute(int slot, int pg, float sl2) {
        // ---- scores: independent accumulator chains for the hi and lo halves of Q
        af4_t s0 = af4_t{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x16f16(as_ah4(kr[slot][c]), qh[c], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x16f16(as_ah4(kr[slot][c]), ql[c], s1, 0, 0, 0);
            if (c + 1 < NC) {
                s2 = __builtin_amdgcn_mfma_f32_16x16x16f16(as_ah4(kr[slot][c + 1]), qh[c + 1], s2, 0, 0, 0);
                s3 = __builtin_amdgcn_mfma_f32_16x16x16f16(as_ah4(kr[slot][c + 1]), ql[c + 1], s3, 0, 0, 0);
            }
        }
        float sc[4];
        float mx = -INFINITY;
        const int kfirst = pg * 16 + 4 * g;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const float v = ((s0[rr] + s2[rr]) + (s1[rr] + s3[rr])) * sl2;
            sc[rr] = (kfirst + rr <= plim) ? v : -INFINITY;
            mx = fmaxf(mx, sc[rr]);
        }
#ifdef NVLLM_ATTN_SHFL_MAX
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
        mx = rows_max(mx);
#endif
        if (__any(mx > m)) {   // some column's running maximum rises (rare after the first pages)
            const float mn = fmaxf(m, mx);
            const float alpha = mn == -INFINITY ? 1.f : __builtin_amdgcn_exp2f(m - mn);
            lsum *= alpha;
            m = mn;
            // the accumulators hold columns 4g..4g+3: fetch their factors from lanes 4g+r
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float ar = __shfl(alpha, 4 * g + rr, 64);
#pragma unroll
                for (int n = 0; n < NC; n++) oacc[n][rr] *= ar;
            }
        }
        const float msafe = fmaxf(m, -1.0e30f);   // nothing visible yet: exp2(-inf - finite) = 0, no NaN
        ah4_t ph, pl;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const float p = __builtin_amdgcn_exp2f(sc[rr] - msafe);
            lsum += p;
            ph[rr] = (_Float16)p;
            pl[rr] = (_Float16)(p - (float)ph[rr]);
        }
        // ---- O += P V
#pragma unroll
        for (int n = 0; n < NC; n++)
            oacc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(ph, as_ah4(vr[slot][n]), oacc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NC; n++)
            oacc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(pl, as_ah4(vr[slot][n]), oacc[n], 0, 0, 0);
    }

    // the first PF pages of the wave are requested, and the block ids of the next PF looked up
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int s = 0; s < PF; s++) {
            issue(s, pg_begin + STRIDE * s, blk_of(pg_begin + STRIDE * s));
            __builtin_amdgcn_sched_barrier(0);   // keep issue order = consumption order (counted vmcnt waits)
        }
#pragma unroll
        for (int s = 0; s < PF; s++) nblk[s] = blk_of(pg_begin + STRIDE * (s + PF));
    }
    // true when start() touches no page that may change during this launch
    __device__ __forceinline__ bool start_is_stable() const { return pg_begin + STRIDE * (PF - 1) < fresh_page; }

    // ---- page loop: PF pages in flight per wave
    __device__ __forceinline__ void loop() {
        const float sl2 = a.scale * 1.44269504088896340736f;   // scores in the log2 domain
        for (int pg = pg_begin; pg < pg_end; pg += STRIDE * PF) {
#pragma unroll
            for (int s = 0; s < PF; s++) {
                if (pg + STRIDE * s < pg_end) compute(s, pg + STRIDE * s, sl2);
                issue(s, pg + STRIDE * (s + PF), nblk[s]);
                nblk[s] = blk_of(pg + STRIDE * (s + 2 * PF));
            }
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
    }

    // one output element (token t, head h, dim d) of the normalised result
    __device__ __forceinline__ void put(int t, int h, int d, float o) const {
        if (a.out_f32) a.out_f32[((size_t)t * a.H + h) * HD + d] = o;
        if (a.out_a16) {
            const int kk = h * HD + d;
            const float xv = fminf(fmaxf(o, -65504.f), 65504.f);
            const __half hi = __float2half_rn(xv);
            const __half lo = __float2half_rn(xv - __half2float(hi));
            const size_t tl = ((size_t)(t >> 4) * a.Kb_out + (kk >> 5)) * 2;
            const size_t off = (size_t)((t & 15) + 16 * ((kk >> 3) & 3)) * 8 + (kk & 7);
            a.out_a16[tl * TILE_HALVES + off] = hi;
            a.out_a16[(tl + 1) * TILE_HALVES + off] = lo;
        }
    }
    // The same for the 16 lanes x = 0..15 of one g, which hold dims 16n + x of ONE (token, head) and
    // are all active here: every eight lanes gather their halves into one 16-byte chunk of the hi
    // plane and one of the lo plane, stored by the lane with x % 8 == 0 (write-through when COH).
    __device__ __forceinline__ void put16(int t, int h, int n, float o) const {
        if (a.out_f32) a.out_f32[((size_t)t * a.H + h) * HD + 16 * n + x] = o;
        if (!a.out_a16) return;
        const float xv = fminf(fmaxf(o, -65504.f), 65504.f);
        const __half hi = __float2half_rn(xv);
        const __half lo = __float2half_rn(xv - __half2float(hi));
        unsigned pk[2] = {(unsigned)__half_as_ushort(hi), (unsigned)__half_as_ushort(lo)};
        unsigned long long q0[2], q1[2];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const unsigned two = pk[pl] | ((unsigned)__shfl_down((int)pk[pl], 1, 64) << 16);                 // x even: dims x, x+1
            const unsigned long long four = two | ((unsigned long long)(unsigned)__shfl_down((int)two, 2, 64) << 32);   // x % 4 == 0
            const unsigned hi_lo = (unsigned)__shfl_down((int)(unsigned)four, 4, 64);
            const unsigned hi_hi = (unsigned)__shfl_down((int)(unsigned)(four >> 32), 4, 64);
            q0[pl] = four;                                                                                    // x % 8 == 0: dims x .. x+3
            q1[pl] = hi_lo | ((unsigned long long)hi_hi << 32);                                               // ... and x+4 .. x+7
        }
        if ((x & 7) != 0) return;
        const int kk = h * HD + 16 * n + x;
        const size_t tl = ((size_t)(t >> 4) * a.Kb_out + (kk >> 5)) * 2;
        const size_t off = (size_t)((t & 15) + 16 * ((kk >> 3) & 3)) * 8;
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out_a16 + (tl + pl) * TILE_HALVES + off);
            if constexpr (COH) {   // one 16-byte write-through store (aux 16 = sc1) through a descriptor of the whole buffer
                typedef unsigned int au4_t __attribute__((ext_vector_type(4)));
                const au4_t v = {(unsigned)q0[pl], (unsigned)(q0[pl] >> 32), (unsigned)q1[pl], (unsigned)(q1[pl] >> 32)};
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.out_a16, 0, 0x7ffffff0, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(((tl + pl) * TILE_HALVES + off) * 2), 0, 16);
            } else {
                *reinterpret_cast<ulonglong2*>(dst) = make_ulonglong2(q0[pl], q1[pl]);
            }
        }
    }

    // ---- one wave owns the item: lane (dim 16n + x, g) holds columns 4g..4g+3, whose (max, sum)
    // live in the lanes with x = column
    __device__ __forceinline__ void finish_solo() const {
        float Mc[4], Lc[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            Mc[rr] = __shfl(m, 4 * g + rr, 64);
            Lc[rr] = __shfl(lsum, 4 * g + rr, 64);
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int col = 4 * g + rr;
            const int ct = col >> a.gt_shift, ch = col & (a.GT - 1);
            if (ct >= ntok) continue;
            const int t = t0 + ct, h = h0 + ch;
            if (a.nsplit > 1 && x == 0) {
                float* pm = a.part_ml + (((size_t)t * a.H + h) * a.nsplit + split) * 2;
                pm[0] = Mc[rr] * 0.69314718055994530942f;
                pm[1] = Lc[rr];
            }
            const float invL = 1.0f / Lc[rr];
#pragma unroll
            for (int n = 0; n < NC; n++) {
                const float o = oacc[n][rr];
                const int d = 16 * n + x;
                if (a.nsplit > 1) {
                    a.part_acc[(((size_t)t * a.H + h) * a.nsplit + split) * HD + d] = o;
                    continue;
                }
                put(t, h, d, o * invL);
            }
        }
    }

    // ---- four waves share the item: deposit (max, sum, accumulators) in LDS ...
    __device__ __forceinline__ void deposit(unsigned char* lds, int wave_in_item) const {
        float (*s_m)[16] = reinterpret_cast<float (*)[16]>(lds);
        float (*s_l)[16] = reinterpret_cast<float (*)[16]>(lds + 4 * 16 * 4);
        af4_t (*s_o)[NC][64] = reinterpret_cast<af4_t (*)[NC][64]>(lds + 2 * 4 * 16 * 4);
        if (g == 0) {
            s_m[wave_in_item][x] = m;
            s_l[wave_in_item][x] = lsum;
        }
#pragma unroll
        for (int n = 0; n < NC; n++) s_o[wave_in_item][n][lane] = oacc[n];
    }
    // ... and, after a workgroup barrier, wave w finishes dim blocks n = w, w+4, ...; lane (dim 16n + x, g)
    // owns columns 4g..4g+3
    __device__ __forceinline__ void merge(const unsigned char* lds, int wave_in_item) const {
        const float (*s_m)[16] = reinterpret_cast<const float (*)[16]>(lds);
        const float (*s_l)[16] = reinterpret_cast<const float (*)[16]>(lds + 4 * 16 * 4);
        const af4_t (*s_o)[NC][64] = reinterpret_cast<const af4_t (*)[NC][64]>(lds + 2 * 4 * 16 * 4);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int col = 4 * g + rr;
            const int ct = col >> a.gt_shift, ch = col & (a.GT - 1);
            if (ct >= ntok) continue;
            const int t = t0 + ct, h = h0 + ch;
            const float M = fmaxf(fmaxf(s_m[0][col], s_m[1][col]), fmaxf(s_m[2][col], s_m[3][col]));
            const float Ms = fmaxf(M, -1.0e30f);
            float f[4], L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                f[w] = __builtin_amdgcn_exp2f(s_m[w][col] - Ms);   // a wave without visible keys: exp2(-inf) = 0
                L = fmaf(s_l[w][col], f[w], L);
            }
            if (a.nsplit > 1 && wave_in_item == 0 && x == 0) {
                float* pm = a.part_ml + (((size_t)t * a.H + h) * a.nsplit + split) * 2;
                pm[0] = M * 0.69314718055994530942f;   // natural-log domain, as k_attn_combine expects
                pm[1] = L;
            }
            const float invL = 1.0f / L;
            for (int n = wave_in_item; n < NC; n += 4) {
                float o = 0.f;
#pragma unroll
                for (int w = 0; w < 4; w++) o = fmaf(s_o[w][n][lane][rr], f[w], o);
                if (a.nsplit > 1) {
                    a.part_acc[(((size_t)t * a.H + h) * a.nsplit + split) * HD + 16 * n + x] = o;
                    continue;
                }
                put16(t, h, n, o * invL);
            }
        }
    }
};

}  // namespace nvllm
