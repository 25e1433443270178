// attention.hip -- causal attention of a ragged batch over the block pool
// (reference src/kernels/attention.c:256-336 and its single-sequence / contiguous
// siblings), fp32 arithmetic throughout.
//
// One WAVE owns one work item = (token, group of GT query heads that share a KV
// head, slice of the key range).  It walks its keys 64 at a time:
//   1. the 64 key rows of the KV head (one page = 16 rows) are staged into the
//      wave's private LDS tile with 16-byte loads, rows padded by 16 bytes so that
//      the per-key reads of step 2 are bank-conflict free;
//   2. lane = key: each lane forms q.k over the head dimension for all GT heads
//      (q comes in through scalar loads), scales after the dot like the reference;
//   3. online softmax per head with wave-wide max / sum reductions;
//   4. the value rows replace the keys in the same LDS tile; lane = output
//      dimension accumulates sum_j p_j v_j, p_j broadcast with v_readlane.
// Items of one token that cover different key slices leave (acc, max, sum)
// partials which k_attn_combine merges.  Output is either fp32 rows or the
// fp16 hi/lo tile order the output projection consumes.
#include "nvllm_internal.h"

namespace nvllm {

struct AttnKArgs {
    const float* q;
    const void* pool_k;
    const void* pool_v;
    long layer_off;  // elements
    int bs;
    const int* pos;
    int pos0;
    const int* seq;
    const int* tbl_off;
    const int* tables;
    int T, H, KVH;
    int nsplit;
    float scale;
    float* out_f32;
    __half* out_a16;
    int Kb_out;
    float* part_acc;  // [item][GT][HD]
    float* part_ml;   // [item][GT][2]
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int j) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j));
}

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const __half* p) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);
    const __half2 a = *reinterpret_cast<const __half2*>(&raw.x);
    const __half2 b = *reinterpret_cast<const __half2*>(&raw.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}

struct alignas(16) H8 {
    __half v[8];
};

template <int HD, int GT, int WAVES, typename KVT>
__global__ __launch_bounds__(WAVES * 64) void k_paged_attn(AttnKArgs a) {
    constexpr int CPR = HD / 4;     // 16-byte chunks per row
    constexpr int STRIDE = HD + 4;  // floats; odd number of chunks => conflict-free row reads
    constexpr int DPL = (HD + 63) / 64;
    __shared__ float smem[WAVES][64 * STRIDE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NG = a.H / GT;
    const long item = (long)blockIdx.x * WAVES + wave;
    if (item >= (long)a.T * NG * a.nsplit) return;
    const int split = (int)(item % a.nsplit);
    const int hg = (int)((item / a.nsplit) % NG);
    const int t = (int)(item / ((long)a.nsplit * NG));
    const int h0 = hg * GT;
    const int kvh = h0 / (a.H / a.KVH);
    const int p = a.pos ? a.pos[t] : a.pos0 + t;
    const int nkeys = p + 1;
    const int per = ((nkeys + 63) / 64 + a.nsplit - 1) / a.nsplit * 64;  // keys per slice, whole tiles
    const int kbegin = split * per;
    const int kend = min(nkeys, kbegin + per);
    const int* table = a.tables ? a.tables + (a.seq ? a.tbl_off[a.seq[t]] : 0) : nullptr;
    const KVT* pk = reinterpret_cast<const KVT*>(a.pool_k) + a.layer_off + (size_t)kvh * HD;
    const KVT* pv = reinterpret_cast<const KVT*>(a.pool_v) + a.layer_off + (size_t)kvh * HD;
    const long rowstride = (long)a.KVH * HD;
    float* tile = smem[wave];
    const float* __restrict__ qrow = a.q + ((size_t)t * a.H + h0) * HD;

    float m[GT], l[GT], acc[GT][DPL];
#pragma unroll
    for (int g = 0; g < GT; g++) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; e++) acc[g][e] = 0.f;
    }

    for (int k0 = kbegin; k0 < kend; k0 += 64) {
        // ---- 1. stage 64 key rows
#pragma unroll
        for (int i = 0; i < CPR; i++) {
            const int idx = i * 64 + lane;
            const int key = idx / CPR, c = idx % CPR;
            const int ta = k0 + key;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ta < kend) {
                const long row = table ? (long)table[ta / a.bs] * a.bs + (ta % a.bs) : (long)ta;
                v = load4(pk + row * rowstride + c * 4);
            }
            *reinterpret_cast<float4*>(tile + key * STRIDE + c * 4) = v;
        }
        // ---- 2. scores, lane = key
        float s[GT];
#pragma unroll
        for (int g = 0; g < GT; g++) s[g] = 0.f;
#pragma unroll
        for (int c = 0; c < CPR; c++) {
            const float4 kv = *reinterpret_cast<const float4*>(tile + lane * STRIDE + c * 4);
#pragma unroll
            for (int g = 0; g < GT; g++) {
                const float* qg = qrow + g * HD + c * 4;
                s[g] = fmaf(qg[0], kv.x, s[g]);
                s[g] = fmaf(qg[1], kv.y, s[g]);
                s[g] = fmaf(qg[2], kv.z, s[g]);
                s[g] = fmaf(qg[3], kv.w, s[g]);
            }
        }
        const bool valid = k0 + lane < kend;
        // ---- 3. online softmax
        float pr[GT];
#pragma unroll
        for (int g = 0; g < GT; g++) {
            const float sc = valid ? s[g] * a.scale : -INFINITY;
            const float mn = fmaxf(m[g], wmax(sc));
            const float e = valid ? expf(sc - mn) : 0.f;
            const float alpha = expf(m[g] - mn);
            l[g] = l[g] * alpha + wsum(e);
#pragma unroll
            for (int d = 0; d < DPL; d++) acc[g][d] *= alpha;
            m[g] = mn;
            pr[g] = e;
        }
        // ---- 4. stage the value rows over the keys, lane = output dimension
#pragma unroll
        for (int i = 0; i < CPR; i++) {
            const int idx = i * 64 + lane;
            const int key = idx / CPR, c = idx % CPR;
            const int ta = k0 + key;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ta < kend) {
                const long row = table ? (long)table[ta / a.bs] * a.bs + (ta % a.bs) : (long)ta;
                v = load4(pv + row * rowstride + c * 4);
            }
            *reinterpret_cast<float4*>(tile + key * STRIDE + c * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < 64; j++) {
            float vj[DPL];
#pragma unroll
            for (int d = 0; d < DPL; d++) {
                const int dim = lane + 64 * d;
                vj[d] = dim < HD ? tile[j * STRIDE + dim] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < GT; g++) {
                const float pj = lane_bcast(pr[g], j);
#pragma unroll
                for (int d = 0; d < DPL; d++) acc[g][d] = fmaf(pj, vj[d], acc[g][d]);
            }
        }
    }

    // ---- results
    if (a.nsplit > 1) {
        float* pa = a.part_acc + (size_t)item * GT * HD;
        float* pm = a.part_ml + (size_t)item * GT * 2;
#pragma unroll
        for (int g = 0; g < GT; g++) {
#pragma unroll
            for (int d = 0; d < DPL; d++) {
                const int dim = lane + 64 * d;
                if (dim < HD) pa[g * HD + dim] = acc[g][d];
            }
            if (lane == 0) {
                pm[2 * g] = m[g];
                pm[2 * g + 1] = l[g];
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < GT; g++) {
        const float inv = 1.0f / l[g];
#pragma unroll
        for (int d = 0; d < DPL; d++) {
            const int dim = lane + 64 * d;
            const float o = acc[g][d] * inv;
            if (dim < HD) {
                if (a.out_f32) a.out_f32[((size_t)t * a.H + h0 + g) * HD + dim] = o;
                tile[g * HD + dim] = o;
            }
        }
    }
    if (a.out_a16) {
        // re-read 8 consecutive dims per lane and emit the hi/lo tile chunks
        for (int c = lane; c < GT * HD / 8; c += 64) {
            const int kk = h0 * HD + c * 8;
            H8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = fminf(fmaxf(tile[c * 8 + j], -65504.f), 65504.f);
                hi.v[j] = __float2half_rn(x);
                lo.v[j] = __float2half_rn(x - __half2float(hi.v[j]));
            }
            const size_t tl = ((size_t)(t >> 4) * a.Kb_out + (kk >> 5)) * 2;
            const size_t off = (size_t)((t & 15) + 16 * ((kk >> 3) & 3)) * 8;
            *reinterpret_cast<H8*>(a.out_a16 + tl * TILE_HALVES + off) = hi;
            *reinterpret_cast<H8*>(a.out_a16 + (tl + 1) * TILE_HALVES + off) = lo;
        }
    }
}

// Merge the key-slice partials of each (token, head): thread = 8 consecutive dims.
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ part_acc,
                                                      const float* __restrict__ part_ml, int T, int H, int HD,
                                                      int GT, int nsplit, float* out_f32, __half* out_a16,
                                                      int Kb_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = HD / 8;
    if (idx >= (long)T * H * c8) return;
    const int c = (int)(idx % c8);
    const int h = (int)((idx / c8) % H);
    const int t = (int)(idx / ((long)c8 * H));
    const int NG = H / GT, hg = h / GT, g = h % GT;
    const long item0 = ((long)t * NG + hg) * nsplit;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; s++) M = fmaxf(M, part_ml[((item0 + s) * GT + g) * 2]);
    float L = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = 0.f;
    for (int s = 0; s < nsplit; s++) {
        const float* ml = part_ml + ((item0 + s) * GT + g) * 2;
        const float w = expf(ml[0] - M);
        L += ml[1] * w;
        const float* pa = part_acc + ((item0 + s) * GT + g) * HD + c * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(pa);
        const float4 a1 = *reinterpret_cast<const float4*>(pa + 4);
        o[0] += a0.x * w; o[1] += a0.y * w; o[2] += a0.z * w; o[3] += a0.w * w;
        o[4] += a1.x * w; o[5] += a1.y * w; o[6] += a1.z * w; o[7] += a1.w * w;
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] *= inv;
    if (out_f32) {
        float* dst = out_f32 + ((size_t)t * H + h) * HD + c * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (out_a16) {
        const int kk = h * HD + c * 8;
        H8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = fminf(fmaxf(o[j], -65504.f), 65504.f);
            hi.v[j] = __float2half_rn(x);
            lo.v[j] = __float2half_rn(x - __half2float(hi.v[j]));
        }
        const size_t tl = ((size_t)(t >> 4) * Kb_out + (kk >> 5)) * 2;
        const size_t off = (size_t)((t & 15) + 16 * ((kk >> 3) & 3)) * 8;
        *reinterpret_cast<H8*>(out_a16 + tl * TILE_HALVES + off) = hi;
        *reinterpret_cast<H8*>(out_a16 + (tl + 1) * TILE_HALVES + off) = lo;
    }
}

// ------------------------------------------------------------------ launch
template <int HD, int GT, int WAVES>
static void launch_cfg(const AttnKArgs& k, long items, int elem_bytes) {
    const dim3 grid((unsigned)((items + WAVES - 1) / WAVES)), block(WAVES * 64);
    if (elem_bytes == 4)
        hipLaunchKernelGGL((k_paged_attn<HD, GT, WAVES, float>), grid, block, 0, rt().stream, k);
    else
        hipLaunchKernelGGL((k_paged_attn<HD, GT, WAVES, __half>), grid, block, 0, rt().stream, k);
}

template <int HD, int WAVES>
static void launch_hd(const AttnKArgs& k, int GT, long items, int elem_bytes) {
    switch (GT) {
        case 8: launch_cfg<HD, 8, WAVES>(k, items, elem_bytes); break;
        case 4: launch_cfg<HD, 4, WAVES>(k, items, elem_bytes); break;
        case 2: launch_cfg<HD, 2, WAVES>(k, items, elem_bytes); break;
        default: launch_cfg<HD, 1, WAVES>(k, items, elem_bytes); break;
    }
}

extern "C" float nvllm_host_attn_scale(int head_dim);

void launch_paged_attention(const AttnArgs& a) {
    if (a.T <= 0) return;
    const int group = a.H / a.KVH;
    const int GT = (group == 8 || group == 4 || group == 2) ? group : 1;
    const int NG = a.H / GT;
    const long base_items = (long)a.T * NG;
    const int tiles = cdiv(a.max_pos + 1, 64);
    int nsplit = (int)((4096 + base_items - 1) / base_items);
    if (nsplit > cdiv(tiles, 2)) nsplit = cdiv(tiles, 2);
    if (nsplit < 1) nsplit = 1;
    const long items = base_items * nsplit;

    AttnKArgs k;
    k.q = a.q;
    k.pool_k = a.pool.k;
    k.pool_v = a.pool.v;
    k.layer_off = (long)a.layer * a.pool.layer_stride;
    k.bs = a.pool.block_size;
    k.pos = a.pos;
    k.pos0 = a.pos0;
    k.seq = a.seq;
    k.tbl_off = a.tbl_off;
    k.tables = a.tables;
    k.T = a.T;
    k.H = a.H;
    k.KVH = a.KVH;
    k.nsplit = nsplit;
    k.scale = nvllm_host_attn_scale(a.hd);
    k.out_f32 = a.out_f32;
    k.out_a16 = a.out_a16;
    k.Kb_out = cdiv(a.H * a.hd, 32);
    k.part_acc = nullptr;
    k.part_ml = nullptr;
    if (nsplit > 1) {
        float* part = (float*)ws_get(WS_ATT_PART, (size_t)items * GT * (a.hd + 2) * sizeof(float));
        k.part_acc = part;
        k.part_ml = part + (size_t)items * GT * a.hd;
    }
    {
        // algorithmic bytes: each (token, kv head) reads its keys and values once
        const double kvb = (double)a.T * (a.max_pos + 1) * 0.5 * 2.0 * a.KVH * a.hd * a.pool.elem_bytes;
        ProfScope ps(PC_ATTN, kvb + (double)a.T * a.H * a.hd * 8.0, 4.0 * a.T * (double)a.H * a.hd * (a.max_pos + 1) * 0.5);
        switch (a.hd) {
            case 16: launch_hd<16, 4>(k, GT, items, a.pool.elem_bytes); break;
            case 32: launch_hd<32, 4>(k, GT, items, a.pool.elem_bytes); break;
            case 48: launch_hd<48, 4>(k, GT, items, a.pool.elem_bytes); break;
            case 64: launch_hd<64, 4>(k, GT, items, a.pool.elem_bytes); break;
            case 96: launch_hd<96, 2>(k, GT, items, a.pool.elem_bytes); break;
            case 128: launch_hd<128, 2>(k, GT, items, a.pool.elem_bytes); break;
            default: NVLLM_FATAL("head_dim %d is not supported (16, 32, 48, 64, 96, 128)", a.hd);
        }
    }
    if (nsplit > 1) {
        ProfScope ps(PC_ATTN_COMBINE, (double)items * GT * a.hd * 4.0, 0);
        const long threads = (long)a.T * a.H * (a.hd / 8);
        hipLaunchKernelGGL(k_attn_combine, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, rt().stream,
                           k.part_acc, k.part_ml, a.T, a.H, a.hd, GT, nsplit, a.out_f32, a.out_a16, k.Kb_out);
    }
}

}  // namespace nvllm
