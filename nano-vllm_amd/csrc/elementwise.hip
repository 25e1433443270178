// elementwise.hip -- the memory-bound kernels around the projections: embedding
// gather, RMSNorm (fp32 and tile-order fp16 output, with the split-K / residual
// reduction of the preceding projection folded in), RoPE + KV scatter, SwiGLU,
// residual add, softmax, arg-max and the layout converters.
#include "nvllm_internal.h"
#include "resid_tail.h"

namespace nvllm {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for blocks of up to 1024 threads; every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red /*[17]*/) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; i++) s += red[i];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = red[0];
        for (int i = 1; i < nw; i++) s = fmaxf(s, red[i]);
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}

// ---------------------------------------------------------------- embedding
// Embedding gather fused with the first layer's input preparation: fp32 residual
// row, the row times the first RMSNorm gain as fp16 hi/lo tiles, and the sum of
// squares per 32-feature group (the consumer projection turns those into 1/rms).
__global__ __launch_bounds__(256) void k_embed_prescale(float* __restrict__ x, __half* __restrict__ a16,
                                                        float* __restrict__ ss, const __half* __restrict__ table,
                                                        const int* __restrict__ tokens,
                                                        const float* __restrict__ gain, int D, int ss_tiles,
                                                        const int* __restrict__ tok_src, const int* __restrict__ prev,
                                                        unsigned* __restrict__ zero, int n_zero, unsigned* __restrict__ zero_word) {
    const int t = blockIdx.x;
    // words the next launch expects to find zero (arrival counters of the decode chain: the first word of each 128-byte line)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) zero[(size_t)i * 32] = 0u;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
    // a token the host has not seen yet: the arg-max the previous pass left in HBM
    const int tok = (tok_src && tok_src[t] >= 0) ? prev[tok_src[t]] : tokens[t];
    const __half* row = table + (size_t)tok * D;
    float* dst = x + (size_t)t * D;
    const int Kb = (D + 31) >> 5;
    for (int i = threadIdx.x * 8; i < ((D + 63) & ~63); i += blockDim.x * 8) {
        float v[8], y[8], part = 0.f;
        if (i < D) {   // D % 8 == 0
            const Half8 h = *reinterpret_cast<const Half8*>(row + i);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                v[j] = __half2float(h.v[j]);
                y[j] = v[j] * gain[i + j];
                part += v[j] * v[j];
            }
            *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
            store_a16_8(a16, t, i, Kb, y);
        }
        // 4 consecutive threads cover one 32-feature group
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if ((threadIdx.x & 3) == 0 && (i >> 5) < ss_tiles) ss[(size_t)t * ss_tiles + (i >> 5)] = part;
    }
}

void launch_embed_prescale(float* x, __half* a16, float* ss, const __half* table, const int* tokens,
                           const float* gain, int T, int D, const int* tok_src, const int* prev, unsigned* zero, int n_zero,
                           unsigned* zero_word) {
    if (T <= 0) return;
    if (D % 32) NVLLM_FATAL("embedding width %d is not a multiple of 32", D);
    ProfScope ps(PC_EMBED, (double)T * D * 10, 0);
    NVLLM_LAUNCH(k_embed_prescale, dim3(T), dim3(256), 0, x, a16, ss, table, tokens, gain, D,
                       cdiv(D, 32), tok_src, prev, zero, n_zero, zero_word);
}

// Residual add + preparation of the next projection's input, for a projection whose
// output y was completed outside its own kernel (summed over tensor-parallel ranks):
// the same arithmetic as the fused epilogue of the projection kernel.
__global__ __launch_bounds__(256) void k_resid_prescale(float* __restrict__ x, const float* __restrict__ y,
                                                        __half* __restrict__ a16, float* __restrict__ ss,
                                                        const float* __restrict__ gain, int D, int ss_tiles) {
    const int t = blockIdx.x;
    float* xr = x + (size_t)t * D;
    const float* yr = y + (size_t)t * D;
    const int Kb = (D + 31) >> 5;
    for (int i = threadIdx.x * 8; i < ((D + 63) & ~63); i += blockDim.x * 8) {
        float part = 0.f;
        if (i < D)   // D % 8 == 0
            part = resid_tail_8(xr, *reinterpret_cast<const float4*>(yr + i), *reinterpret_cast<const float4*>(yr + i + 4), a16, gain, t, i, Kb);
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if ((threadIdx.x & 3) == 0 && (i >> 5) < ss_tiles) ss[(size_t)t * ss_tiles + (i >> 5)] = part;
    }
}

void launch_resid_prescale(float* x, const float* y, __half* a16, float* ss, const float* gain, int T, int D) {
    if (T <= 0) return;
    if (D % 32) NVLLM_FATAL("residual width %d is not a multiple of 32", D);
    ProfScope ps(PC_ELEMWISE, (double)T * D * 16, 0);
    NVLLM_LAUNCH(k_resid_prescale, dim3(T), dim3(256), 0, x, y, a16, ss, gain, D, cdiv(D, 32));
}

// Compact the rows that produce logits: tile-order rows idx[r] -> rows r.
__global__ __launch_bounds__(256) void k_gather_a16(__half* __restrict__ dst, float* __restrict__ dst_ss,
                                                    const __half* __restrict__ src,
                                                    const float* __restrict__ src_ss,
                                                    const int* __restrict__ idx, int Kb, int ss_tiles) {
    const int r = blockIdx.x;
    const int t = idx[r];
    for (int c = threadIdx.x; c < Kb * 4; c += blockDim.x) {   // 8-element chunks of the row
        const int k = c * 8;
        *reinterpret_cast<Half8*>(dst + a16_chunk(r, k, Kb, 0)) = *reinterpret_cast<const Half8*>(src + a16_chunk(t, k, Kb, 0));
        *reinterpret_cast<Half8*>(dst + a16_chunk(r, k, Kb, 1)) = *reinterpret_cast<const Half8*>(src + a16_chunk(t, k, Kb, 1));
    }
    for (int j = threadIdx.x; j < ss_tiles; j += blockDim.x) dst_ss[(size_t)r * ss_tiles + j] = src_ss[(size_t)t * ss_tiles + j];
}

void launch_gather_a16(__half* dst, float* dst_ss, const __half* src, const float* src_ss, const int* idx, int B,
                       int K, int ss_tiles) {
    if (B <= 0) return;
    ProfScope ps(PC_PACK, (double)B * K * 8.0, 0);
    NVLLM_LAUNCH(k_gather_a16, dim3(B), dim3(256), 0, dst, dst_ss, src, src_ss, idx, cdiv(K, 32),
                       ss_tiles);
}

// The batch description of one pass (a few thousand ints) from page-locked host memory into HBM:
// a kernel that reads the host buffer directly starts with the stream's other kernels, where a copy
// operation adds its own engine's start-up (about 18 us against 3 in the decode loop).
__global__ __launch_bounds__(256) void k_copy_ints(int4* __restrict__ dst, const int4* __restrict__ src, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}
void launch_copy_ints(int* dst, const int* host_src, size_t n) {
    const int n4 = (int)((n + 3) / 4);
    if (n4 <= 0) return;
    ProfScope ps(PC_EMBED, (double)n * 8.0, 0);
    NVLLM_LAUNCH(k_copy_ints, dim3(cdiv(n4, 256)), dim3(256), 0, (int4*)dst, (const int4*)host_src, n4);
}

// Token ids that the host does not know yet: tokens[t] = prev[src[t]] where src[t] >= 0 (prev = the
// arg-max indices the previous pass left in HBM).
__global__ __launch_bounds__(256) void k_patch_tokens(int* __restrict__ tokens, const int* __restrict__ src,
                                                      const int* __restrict__ prev, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && src[t] >= 0) tokens[t] = prev[src[t]];
}
void launch_patch_tokens(int* tokens, const int* src, const int* prev, int T) {
    if (T <= 0) return;
    ProfScope ps(PC_EMBED, (double)T * 12.0, 0);
    NVLLM_LAUNCH(k_patch_tokens, dim3(cdiv(T, 256)), dim3(256), 0, tokens, src, prev, T);
}

// Rows idx[r] of an fp32 [*, D] table -> rows r of dst (the embedding lookup and the output
// gather of the forward pass on fp32 weights: reference model.c:352-355, 411-414).
__global__ __launch_bounds__(256) void k_gather_rows_f32(float* __restrict__ dst, const float* __restrict__ src,
                                                         const int* __restrict__ idx, int D) {
    const float4* s = reinterpret_cast<const float4*>(src + (size_t)idx[blockIdx.x] * D);
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.x * D);
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}

void launch_gather_rows_f32(float* dst, const float* src, const int* idx, int B, int D) {
    if (B <= 0) return;
    if (D % 4) NVLLM_FATAL("row width %d is not a multiple of 4", D);
    ProfScope ps(PC_EMBED, (double)B * D * 8.0, 0);
    NVLLM_LAUNCH(k_gather_rows_f32, dim3(B), dim3(256), 0, dst, src, idx, D);
}

// ------------------------------------------------------------------ RMSNorm
// ops.h rms_norm (reference layers.c:23-38).  One workgroup per row; arithmetic per
// element as layers.c:32-35: r = 1/sqrtf(ss/D + eps), out = (x * r) * g.  (Inside the
// forward pass RMSNorm is folded into the projections, see gemm.hip.)
__global__ __launch_bounds__(256) void k_rmsnorm(float* out, const float* x, const float* __restrict__ gain, int D,
                                                 float eps) {
    __shared__ float red[17];
    const float* xr = x + (size_t)blockIdx.x * D;
    float ss = 0.f;
    for (int i = threadIdx.x * 4; i < D; i += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = block_sum(ss, red);
    const float rinv = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss, (float)D), eps)));
    float* o = out + (size_t)blockIdx.x * D;
    for (int i = threadIdx.x * 4; i < D; i += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);   // out may alias x: each thread rewrites what it read
        const float4 g = *reinterpret_cast<const float4*>(gain + i);
        float4 y;
        y.x = __fmul_rn(__fmul_rn(v.x, rinv), g.x);
        y.y = __fmul_rn(__fmul_rn(v.y, rinv), g.y);
        y.z = __fmul_rn(__fmul_rn(v.z, rinv), g.z);
        y.w = __fmul_rn(__fmul_rn(v.w, rinv), g.w);
        *reinterpret_cast<float4*>(o + i) = y;
    }
}

// generic width (not a multiple of 4): scalar loops, fp32 output only
__global__ __launch_bounds__(256) void k_rmsnorm_any(float* out, const float* x,
                                                     const float* __restrict__ gain, int D, float eps) {
    __shared__ float red[17];
    const float* xr = x + (size_t)blockIdx.x * D;
    float* o = out + (size_t)blockIdx.x * D;
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) ss += xr[i] * xr[i];
    ss = block_sum(ss, red);
    const float rinv = __fdiv_rn(1.0f, sqrtf(__fadd_rn(__fdiv_rn(ss, (float)D), eps)));
    for (int i = threadIdx.x; i < D; i += blockDim.x) o[i] = __fmul_rn(__fmul_rn(xr[i], rinv), gain[i]);
}

void launch_rmsnorm_f32(float* o, const float* x, const float* gain, int D, int T, float eps) {
    if (T <= 0) return;
    ProfScope ps(PC_RMSNORM, (double)T * D * 8.0, 0);
    if (D % 4 == 0 && ((uintptr_t)o % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)gain % 16 == 0))
        NVLLM_LAUNCH(k_rmsnorm, dim3(T), dim3(256), 0, o, x, gain, D, eps);
    else
        NVLLM_LAUNCH(k_rmsnorm_any, dim3(T), dim3(256), 0, o, x, gain, D, eps);
}

// -------------------------------------------------------------- pack to a16
__global__ __launch_bounds__(256) void k_pack_a16(__half* __restrict__ a16, const float* __restrict__ in,
                                                  int T, int K, int Kb, int Tb) {
    // one thread per (tile, lane): 8 consecutive k of one token
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)Tb * Kb * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long tile = idx >> 6;
    const int kb = (int)(tile % Kb), tb = (int)(tile / Kb);
    const int t = tb * 16 + (lane & 15);
    const int k = kb * 32 + 8 * (lane >> 4);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) y[j] = (t < T && k + j < K) ? in[(size_t)t * K + k + j] : 0.f;
    store_a16_8(a16, t, k, Kb, y);
}

void launch_pack_a16(__half* a16, const float* in, int T, int K) {
    if (T <= 0) return;
    const int Kb = cdiv(K, 32), Tb = cdiv(T, 16);
    const long total = (long)Tb * Kb * 64;
    ProfScope ps(PC_PACK, (double)T * K * 8.0, 0);
    NVLLM_LAUNCH(k_pack_a16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a16, in,
                       T, K, Kb, Tb);
}

// --------------------------------------------------------------------- RoPE
// Rotation of one interleaved pair with the reference's rounding
// (layers.c:129-130): two products, one subtraction/addition, no fused multiply-add.
__device__ __forceinline__ float2 rot(float a, float b, float2 cs) {
    float2 r;
    r.x = __fsub_rn(__fmul_rn(a, cs.x), __fmul_rn(b, cs.y));
    r.y = __fadd_rn(__fmul_rn(a, cs.y), __fmul_rn(b, cs.x));
    return r;
}

__global__ __launch_bounds__(256) void k_rope(float* q, float* k, const int* __restrict__ pos, int pos0,
                                              const float2* __restrict__ rope, int H, int KVH, int hd) {
    const int t = blockIdx.x;
    const int p = pos ? pos[t] : pos0 + t;
    const int half = hd >> 1;
    const float2* tab = rope + (size_t)p * half;
    float* qr = q + (size_t)t * H * hd;
    float* kr = k + (size_t)t * KVH * hd;
    for (int i = threadIdx.x; i < H * half; i += blockDim.x) {
        float2* pr = reinterpret_cast<float2*>(qr) + i;
        const float2 v = *pr;
        *pr = rot(v.x, v.y, tab[i % half]);
    }
    for (int i = threadIdx.x; i < KVH * half; i += blockDim.x) {
        float2* pr = reinterpret_cast<float2*>(kr) + i;
        const float2 v = *pr;
        *pr = rot(v.x, v.y, tab[i % half]);
    }
}

void launch_rope(float* q, float* k, const int* pos, int pos0, const float2* rope, int T, int H, int KVH,
                 int hd) {
    if (T <= 0) return;
    ProfScope ps(PC_ROPE_KV, (double)T * (H + KVH) * hd * 8.0, 0);
    NVLLM_LAUNCH(k_rope, dim3(T), dim3(256), 0, q, k, pos, pos0, rope, H, KVH, hd);
}

// Physical row (in units of token rows) of position p of the sequence that owns
// `table`; a null table means a contiguous cache.
__device__ __forceinline__ long kv_row(const int* table, int p, int bs) {
    return table ? (long)table[p / bs] * bs + (p % bs) : (long)p;
}

template <typename KVT>
__device__ __forceinline__ void kv_store2(KVT* dst, float a, float b);
template <>
__device__ __forceinline__ void kv_store2<float>(float* dst, float a, float b) {
    *reinterpret_cast<float2*>(dst) = make_float2(a, b);
}
template <>
__device__ __forceinline__ void kv_store2<__half>(__half* dst, float a, float b) {
    *reinterpret_cast<__half2*>(dst) = __halves2half2(__float2half_rn(a), __float2half_rn(b));
}
template <typename KVT>
__device__ __forceinline__ void kv_store1(KVT* dst, float a);
template <>
__device__ __forceinline__ void kv_store1<float>(float* dst, float a) { *dst = a; }
template <>
__device__ __forceinline__ void kv_store1<__half>(__half* dst, float a) {   // finite fp16 only
    *dst = __float2half_rn(fminf(fmaxf(a, -65504.f), 65504.f));
}

// Plain scatter of K/V rows (reference attention.c:11-30, :109-175).  fp32 pools
// keep the reference's rows; fp16 pools are in page order (see kv16_page).
template <typename KVT>
__global__ __launch_bounds__(256) void k_kv_write(const float* __restrict__ k, const float* __restrict__ v,
                                                  const int* __restrict__ pos, int pos0,
                                                  const int* __restrict__ seq,
                                                  const int* __restrict__ tbl_off,
                                                  const int* __restrict__ tables, KVT* pool_k, KVT* pool_v,
                                                  int bs, long layer_off, int KVH, int hd) {
    const int t = blockIdx.x;
    const int p = pos ? pos[t] : pos0 + t;
    const int* table = tables ? tables + (seq ? tbl_off[seq[t]] : 0) : nullptr;
    const int KD = KVH * hd;
    const float* ks = k + (size_t)t * KD;
    const float* vs = v + (size_t)t * KD;
    if (sizeof(KVT) == 4) {
        const long row = kv_row(table, p, bs);
        KVT* kd = pool_k + layer_off + row * KD;
        KVT* vd = pool_v + layer_off + row * KD;
        for (int i = threadIdx.x; i < KD / 2; i += blockDim.x) {
            const float2 a = reinterpret_cast<const float2*>(ks)[i];
            const float2 b = reinterpret_cast<const float2*>(vs)[i];
            kv_store2<KVT>(kd + 2 * i, a.x, a.y);
            kv_store2<KVT>(vd + 2 * i, b.x, b.y);
        }
    } else {
        const long block = table[p / KV16_PAGE_KEYS];
        const int key = p % KV16_PAGE_KEYS;
        for (int i = threadIdx.x; i < KD; i += blockDim.x) {
            const int kvh = i / hd, d = i % hd;
            const long page = layer_off + kv16_page(block, kvh, KVH, hd);
            kv_store1<KVT>(pool_k + page + kv16_k_off(key, d), ks[i]);
            kv_store1<KVT>(pool_v + page + kv16_v_off(key, d), vs[i]);
        }
    }
}

void launch_kv_write(const float* k, const float* v, const int* pos, int pos0, const int* seq,
                     const int* tbl_off, const int* tables, const KvPoolView& pool, int layer, int T, int KVH,
                     int hd) {
    if (T <= 0) return;
    const int KD = KVH * hd;
    if (KD % 2) NVLLM_FATAL("KV row width %d must be even", KD);
    ProfScope ps(PC_ROPE_KV, (double)T * KD * 2 * (4.0 + pool.elem_bytes), 0);
    const long layer_off = (long)layer * pool.layer_stride;
    if (pool.elem_bytes == 4) {
        NVLLM_LAUNCH(k_kv_write<float>, dim3(T), dim3(256), 0, k, v, pos, pos0, seq, tbl_off,
                           tables, (float*)pool.k, (float*)pool.v, pool.block_size, layer_off, KVH, hd);
    } else {
        if (!tables || pool.block_size != KV16_PAGE_KEYS || hd % 16)
            NVLLM_FATAL("an fp16 KV pool needs block tables, block_size 16 and head_dim %% 16 == 0");
        NVLLM_LAUNCH(k_kv_write<__half>, dim3(T), dim3(256), 0, k, v, pos, pos0, seq, tbl_off,
                           tables, (__half*)pool.k, (__half*)pool.v, pool.block_size, layer_off, KVH, hd);
    }
}

// ------------------------------------------------------------- small ops
// silu(g) * u with the reference's operation order (layers.c:156-158).
__device__ __forceinline__ float swiglu1(float g, float u) {
    const float act = __fdiv_rn(g, __fadd_rn(1.0f, expf(-g)));
    return __fmul_rn(act, u);
}

__global__ void k_swiglu(float* out, const float* g, const float* u, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = swiglu1(g[i], u[i]);
}
void launch_swiglu(float* out, const float* g, const float* u, long n) {
    if (n <= 0) return;
    ProfScope ps(PC_ELEMWISE, (double)n * 12.0, 0);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    NVLLM_LAUNCH(k_swiglu, dim3(blocks), dim3(256), 0, out, g, u, n);
}

__global__ void k_accum(float* a, const float* b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        a[i] = __fadd_rn(a[i], b[i]);
}
void launch_accum(float* a, const float* b, long n) {
    if (n <= 0) return;
    ProfScope ps(PC_ELEMWISE, (double)n * 12.0, 0);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    NVLLM_LAUNCH(k_accum, dim3(blocks), dim3(256), 0, a, b, n);
}

// softmax of one vector (reference layers.c:170-186): max, exp+sum, divide
__global__ __launch_bounds__(1024) void k_softmax(float* x, int n) {
    __shared__ float red[17];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, x[i]);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = expf(x[i] - m);
        x[i] = e;
        s += e;
    }
    s = block_sum(s, red);
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = __fdiv_rn(x[i], s);
}
void launch_softmax(float* x, int n) {
    if (n <= 0) return;
    ProfScope ps(PC_ELEMWISE, (double)n * 16.0, 0);
    NVLLM_LAUNCH(k_softmax, dim3(1), dim3(1024), 0, x, n);
}

// First index of the row maximum (reference sampler.c:86-97: strict '>' scan, so
// ties resolve to the lowest index).
// `pairs` (optional): instead of the index, (maximum, index_base + index as a float) per row --
// what a tensor-parallel rank contributes for its slice of the vocabulary.
// `out_host` (optional): the index also goes to that page-locked host array, so that no copy follows.
__global__ __launch_bounds__(1024) void k_argmax(int* __restrict__ out, const float* __restrict__ logits, int V, int ld,
                                                 float2* __restrict__ pairs, int index_base, int* __restrict__ out_host = nullptr) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* row = logits + (size_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // (the reference's scan starts from logits[0] and nothing displaces a NaN there: sampler.c:86-97)
    const bool nan_at_zero = index_base == 0 && V > 0 && row[0] != row[0];
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    };
    if ((V & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        // 16-byte loads, eight in flight per thread before the first compare
        const float4* r4 = reinterpret_cast<const float4*>(row);
        const int n4 = V >> 2;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * blockDim.x) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * blockDim.x;
                v[u] = r4[i < n4 ? i : n4 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u * blockDim.x;
                if (i < n4) {
                    take(v[u].x, 4 * i);
                    take(v[u].y, 4 * i + 1);
                    take(v[u].z, 4 * i + 2);
                    take(v[u].w, 4 * i + 3);
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < V; i += blockDim.x) take(row[i], i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[wid] = best;
        si[wid] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        // a row of -inf (or NaN in slot 0) leaves the scan at index 0 in the reference
        int idx = bi == 0x7fffffff ? 0 : bi;
        if (nan_at_zero) {   // wins every merge: +inf at index 0
            best = INFINITY;
            idx = 0;
        }
        if (pairs) pairs[blockIdx.x] = make_float2(V > 0 ? best : -INFINITY, (float)(index_base + idx));
        else {
            out[blockIdx.x] = idx;
            if (out_host) __hip_atomic_store(out_host + blockIdx.x, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
void launch_argmax(int* out, const float* logits, int B, int V) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)B * V * 4.0, 0);
    NVLLM_LAUNCH(k_argmax, dim3(B), dim3(1024), 0, out, logits, V, V, (float2*)nullptr, 0, (int*)nullptr);
}
void launch_argmax_ld(int* out, const float* logits, int B, int V, int ld, int* out_host) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)B * V * 4.0, 0);
    NVLLM_LAUNCH(k_argmax, dim3(B), dim3(1024), 0, out, logits, V, ld, (float2*)nullptr, 0, out_host);
}
void launch_argmax_pairs(float2* pairs, const float* logits, int B, int n_valid, int ld, int index_base) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)B * n_valid * 4.0, 0);
    NVLLM_LAUNCH(k_argmax, dim3(B), dim3(1024), 0, (int*)nullptr, logits, n_valid, ld, pairs, index_base, (int*)nullptr);
}

// The ranks' (maximum, index) pairs, gathered as [rank][row]: the overall arg-max, lowest index on
// ties (ranks own ascending index ranges, so the first maximum in rank order is it).
__global__ __launch_bounds__(256) void k_argmax_merge(int* __restrict__ out, const float2* __restrict__ pairs, int W, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float2 best = pairs[b];
    for (int r = 1; r < W; r++) {
        const float2 c = pairs[(size_t)r * B + b];
        if (c.x > best.x) best = c;
    }
    out[b] = (int)best.y;
}
// The classifier's per-tile pairs [tile][ld] (gemm_tile.h, F32 epilogue): tiles own ascending index
// ranges; an index below zero marks a tile that saw nothing but NaN for the row.  One wave per row.
__global__ __launch_bounds__(256) void k_argmax_tiles(int* __restrict__ out, const float2* __restrict__ pairs, int n_tiles, int B, int ld,
                                                      int* __restrict__ out_host) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    float best = -INFINITY, bi = -1.0f;
    for (int i = lane; i < n_tiles; i += 64) {
        const float2 c = pairs[(size_t)i * ld + b];
        if (c.y >= 0.f && (c.x > best || bi < 0.f || (c.x == best && c.y < bi))) {
            best = c.x;
            bi = c.y;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64), oi = __shfl_xor(bi, o, 64);
        if (oi >= 0.f && (ov > best || bi < 0.f || (ov == best && oi < bi))) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        const int idx = bi < 0.f ? 0 : (int)bi;   // (nothing but NaN: the reference's scan stays at index 0)
        out[b] = idx;
        if (out_host) __hip_atomic_store(out_host + b, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // page-locked host array
    }
}
void launch_argmax_tiles(int* out, const float2* pairs, int n_tiles, int B, int ld, int* out_host) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)n_tiles * B * 8.0, 0);
    NVLLM_LAUNCH(k_argmax_tiles, dim3(cdiv(B, 4)), dim3(256), 0, out, pairs, n_tiles, B, ld, out_host);
}

void launch_argmax_merge(int* out, const float2* pairs, int W, int B) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)W * B * 8.0, 0);
    NVLLM_LAUNCH(k_argmax_merge, dim3(cdiv(B, 256)), dim3(256), 0, out, pairs, W, B);
}

// Vocabulary slices gathered as [rank][row][Vl] -> logits [row][V]
__global__ __launch_bounds__(256) void k_logits_unshard(float* __restrict__ full, const float* __restrict__ parts, int W, int B,
                                                        int Vl, int V) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        const int r = i / Vl, c = i - r * Vl;
        full[(size_t)b * V + i] = parts[((size_t)r * B + b) * Vl + c];
    }
}
void launch_logits_unshard(float* full, const float* parts, int W, int B, int Vl, int V) {
    if (B <= 0) return;
    ProfScope ps(PC_ELEMWISE, (double)B * V * 8.0, 0);
    NVLLM_LAUNCH(k_logits_unshard, dim3(cdiv(V, 256 * 4), B), dim3(256), 0, full, parts, W, B, Vl, V);
}

// Temperature softmax of each row plus the candidate filter of nucleus sampling
// (reference sampler.c:100-121 and :41-52): probs[r, i] = softmax(logits[r, :] / T)[i];
// when cutoff >= 0 the entries with prob >= cutoff are compacted, IN INDEX ORDER (the
// order in which the reference's loop collects them, i.e. the input order of its
// qsort), into cand[r, 0 .. min(count, cap)), and count[r] is their number.
// One workgroup per row; each thread owns a contiguous chunk of the row.
__global__ __launch_bounds__(1024) void k_sample_prep(float* __restrict__ probs, const float* __restrict__ logits,
                                                      int V, float temperature, float cutoff,
                                                      ProbIndex* __restrict__ cand, int cap, int* __restrict__ count) {
    __shared__ float s_red[16];
    __shared__ int s_cnt[16];
    __shared__ float s_bcast;
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * V;
    float* prow = probs + (size_t)r * V;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // max of logits / T
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, __fdiv_rn(row[i], temperature));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) s_red[wid] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_red[0];
        for (int w = 1; w < nw; w++) m = fmaxf(m, s_red[w]);
        s_bcast = m;
    }
    __syncthreads();
    mx = s_bcast;
    __syncthreads();
    // exp and denominator
    float sum = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float e = expf(__fsub_rn(__fdiv_rn(row[i], temperature), mx));
        prow[i] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_red[wid] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = 0.f;
        for (int w = 0; w < nw; w++) d += s_red[w];
        s_bcast = d;
    }
    __syncthreads();
    const float denom = s_bcast;
    // normalise; count the candidates of this thread's contiguous chunk
    const int chunk = (V + blockDim.x - 1) / blockDim.x;
    const int i0 = min(V, (int)threadIdx.x * chunk), i1 = min(V, i0 + chunk);
    int mine = 0;
    for (int i = i0; i < i1; i++) {
        const float pr = __fdiv_rn(prow[i], denom);
        prow[i] = pr;
        mine += (cutoff >= 0.f && pr >= cutoff) ? 1 : 0;
    }
    if (cutoff < 0.f) return;
    // exclusive scan of the per-thread counts: within the wave, then across waves
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_cnt[wid] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wid; w++) base += s_cnt[w];
    int at = base + incl - mine;
    if (threadIdx.x == blockDim.x - 1) count[r] = base + incl;
    ProbIndex* crow = cand + (size_t)r * cap;
    for (int i = i0; i < i1; i++) {
        const float pr = prow[i];
        if (pr >= cutoff) {
            if (at < cap) {
                crow[at].prob = pr;
                crow[at].index = i;
            }
            at++;
        }
    }
}
void launch_sample_prep(float* probs, const float* logits, int B, int V, float temperature, float cutoff,
                        ProbIndex* cand, int cap, int* count) {
    if (B <= 0) return;
    ProfScope ps(PC_ARGMAX, (double)B * V * 8.0, 0);
    NVLLM_LAUNCH(k_sample_prep, dim3(B), dim3(1024), 0, probs, logits, V, temperature, cutoff, cand,
                       cap, count);
}

// ------------------------------------------------------ layout converters
// row-major fp16 [N, K] -> tile order, zero padded to whole tiles
__global__ __launch_bounds__(256) void k_tile_weights(__half* __restrict__ tiles,
                                                      const __half* __restrict__ rm, int N, int K, int Kb,
                                                      int Nb, long nb_stride) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)Nb * Kb * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long tile = idx >> 6;
    const int kb = (int)(tile % Kb), nb = (int)(tile / Kb);
    const int n = nb * 16 + (lane & 15);
    const int k = kb * 32 + 8 * (lane >> 4);
    Half8 h;
#pragma unroll
    for (int j = 0; j < 8; j++) h.v[j] = (n < N && k + j < K) ? rm[(size_t)n * K + k + j] : __float2half(0.f);
    *reinterpret_cast<Half8*>(tiles + (size_t)nb * nb_stride + (size_t)kb * TILE_HALVES + lane * 8) = h;
}
void launch_tile_weights(__half* tiles, const __half* rowmajor, int N, int K, long nb_stride) {
    const int Kb = cdiv(K, 32), Nb = cdiv(N, 16);
    const long total = (long)Nb * Kb * 64;
    NVLLM_LAUNCH(k_tile_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, tiles,
                       rowmajor, N, K, Kb, Nb, nb_stride);
}

__global__ void k_f32_to_f16(__half* dst, const float* src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __float2half_rn(src[i]);
}
void launch_f32_to_f16(__half* dst, const float* src, size_t n) {
    if (!n) return;
    NVLLM_LAUNCH(k_f32_to_f16, dim3(2048), dim3(256), 0, dst, src, n);
}
__global__ void k_f16_to_f32(float* dst, const __half* src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __half2float(src[i]);
}
void launch_f16_to_f32(float* dst, const __half* src, size_t n) {
    if (!n) return;
    NVLLM_LAUNCH(k_f16_to_f32, dim3(2048), dim3(256), 0, dst, src, n);
}

}  // namespace nvllm
