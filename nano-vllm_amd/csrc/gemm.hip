// gemm.hip -- projections: out[T, N] = act[T, K] x W[N, K]^T with fp16 weights
// streamed once from HBM in tile order, fp32 activations carried as fp16 hi/lo
// tile pairs, fp32 accumulation on the matrix cores.
//
// Shape of the problem at decode: T <= 64 tokens, so every weight byte is used
// for only 2*T flops and the kernel is bound by how fast the weights stream.
// Work decomposition (one launch per projection):
//   grid.x  groups of NBW 16-row blocks of W (NBW*16 output features)
//   grid.y  split of K across workgroups (partial sums go to slabs, summed by
//           the consumer kernel) -- only where the grid would otherwise be small
//   grid.z  groups of TB*16 tokens (prefill: weights re-read from L2/MALL)
//   inside a workgroup the NW waves take the K-steps round-robin, each with all
//   NBW row blocks, and their accumulators are summed through LDS at the end.
// Per K-step (32 k) a wave loads NBW weight tiles (1 KiB each, one fully
// coalesced dwordx4 per lane, HBM) and 2*TB activation tiles (L2), and issues
// 2*NBW*TB v_mfma_f32_16x16x32_f16.  Weight loads run RW steps ahead, activation
// loads RA steps ahead, in register rings.
#include "nvllm_internal.h"

namespace nvllm {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const __half* W;
    long nb_stride;   // halves between 16-row blocks
    int NB;           // 16-row blocks
    int Kb;           // k tiles
    const __half* A;  // activation tiles [Tb][Kb][2][512]
    int Tb;
    int N, T;
    int kb_per_split;
    float* out;       // F32: [T, ldo]; SLAB: slabs[ks][Tpad][N] (ldo == N)
    int ldo;
    long slab_stride;
    __half* a16_out;  // SWIGLU: tiles of [T, N/2]
    int Kb_out;
};

__device__ __forceinline__ float silu_mul(float g, float u) {
    // reference layers.c:156-158: g / (1 + expf(-g)) * u
    return __fmul_rn(__fdiv_rn(g, __fadd_rn(1.0f, expf(-g))), u);
}

__device__ __forceinline__ f4_t f4_add(f4_t a, f4_t b) { return a + b; }

template <int NBW, int TB, int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmArgs p) {
    static_assert(NBW % 2 == 0, "row blocks are reduced in pairs");
    constexpr int RW = 4;  // weight ring depth (K-steps in flight)
    constexpr int RA = 2;  // activation ring depth
    __shared__ f4_t red[NW][2][TB][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb0 = blockIdx.x * NBW;
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

    // K-tile of this wave's step s
    auto kb_of = [&](int s) { return first + NW * s; };

    // Loads are issued only for steps that exist (wave-uniform branches): a clamped
    // "harmless" re-load of the last tile would be real HBM traffic, since streamed
    // weights are loaded non-temporally and do not stay in L2.
#pragma unroll
    for (int s = 0; s < RW; s++)
        if (s < nsteps) {
            const int kb = kb_of(s);
#pragma unroll
            for (int i = 0; i < NBW; i++) w[s][i] = __builtin_nontemporal_load(wp[i] + (size_t)kb * 64);
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
                if (s + RW < nsteps) {
                    const int kbw = kb_of(s + RW);
#pragma unroll
                    for (int i = 0; i < NBW; i++) w[r][i] = __builtin_nontemporal_load(wp[i] + (size_t)kbw * 64);
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

    // ---- sum the waves' partial tiles through LDS, two row blocks per round ----
    const int Tlim = p.T;
#pragma unroll
    for (int h = 0; h < NBW / 2; h++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < TB; j++) red[wave][i][j][lane] = acc[2 * h + i][j];
        __syncthreads();
        if (EPI == GemmEpilogue::SWIGLU_A16) {
            // even row block = gate, odd = up (interleaved storage): one hidden block
            const int hid0 = ((nb0 + 2 * h) >> 1) * 16;
            for (int e = threadIdx.x; e < TB * 64; e += NW * 64) {
                const int j = e >> 6, l = e & 63;
                f4_t g = red[0][0][j][l], u = red[0][1][j][l];
#pragma unroll
                for (int wv = 1; wv < NW; wv++) {
                    g = f4_add(g, red[wv][0][j][l]);
                    u = f4_add(u, red[wv][1][j][l]);
                }
                const int t = (tb0 + j) * 16 + (l & 15);
                const int n = hid0 + 4 * (l >> 4);
                if (t < Tlim && n < (p.N >> 1) && tb0 + j < p.Tb) {
                    alignas(8) __half hi[4], lo[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float y = silu_mul(g[r], u[r]);
                        const float c = fminf(fmaxf(y, -65504.f), 65504.f);
                        hi[r] = __float2half_rn(c);
                        lo[r] = __float2half_rn(c - __half2float(hi[r]));
                    }
                    const size_t tile = ((size_t)(t >> 4) * p.Kb_out + (n >> 5)) * 2;
                    const size_t off = (size_t)((t & 15) + 16 * ((n >> 3) & 3)) * 8 + (n & 7);
                    *reinterpret_cast<uint2*>(p.a16_out + tile * TILE_HALVES + off) = *reinterpret_cast<uint2*>(hi);
                    *reinterpret_cast<uint2*>(p.a16_out + (tile + 1) * TILE_HALVES + off) = *reinterpret_cast<uint2*>(lo);
                }
            }
        } else {
            float* dst = p.out + (EPI == GemmEpilogue::SLAB ? (size_t)blockIdx.y * p.slab_stride : 0);
            for (int e = threadIdx.x; e < 2 * TB * 64; e += NW * 64) {
                const int l = e & 63, j = (e >> 6) % TB, i = (e >> 6) / TB;
                f4_t v = red[0][i][j][l];
#pragma unroll
                for (int wv = 1; wv < NW; wv++) v = f4_add(v, red[wv][i][j][l]);
                const int t = (tb0 + j) * 16 + (l & 15);
                const int n = (nb0 + 2 * h + i) * 16 + 4 * (l >> 4);
                if (t < Tlim && tb0 + j < p.Tb && nb0 + 2 * h + i < p.NB) {
                    float* o = dst + (size_t)t * p.ldo + n;
                    if (n + 3 < p.N && (p.ldo & 3) == 0) {
                        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (n + r < p.N) o[r] = v[r];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ launch
constexpr int GEMM_NBW = 4;
constexpr int GEMM_NW = 8;

int gemm_pick_ksplit(int N, int K, int T) {
    const int NB = cdiv(N, 16), Kb = cdiv(K, 32);
    const int groups = cdiv(NB, GEMM_NBW) * cdiv(cdiv(T, 16), 4);
    static const int target_wgs = getenv("NVLLM_GEMM_WGS") ? atoi(getenv("NVLLM_GEMM_WGS")) : 384;
    int ks = (target_wgs + groups / 2) / groups;      // aim for a few hundred workgroups
    const int ks_max = Kb / GEMM_NW > 0 ? Kb / GEMM_NW : 1;  // at least one K-step per wave
    if (ks > ks_max) ks = ks_max;
    if (ks < 1) ks = 1;
    const int per = cdiv(Kb, ks);
    return cdiv(Kb, per);
}

template <int TB>
static void launch_tb(const GemmArgs& a, int epi, dim3 grid) {
    const dim3 block(GEMM_NW * 64);
    switch (epi) {
        case GemmEpilogue::F32:
            hipLaunchKernelGGL((k_gemm<GEMM_NBW, TB, GEMM_NW, GemmEpilogue::F32>), grid, block, 0, rt().stream, a);
            break;
        case GemmEpilogue::SLAB:
            hipLaunchKernelGGL((k_gemm<GEMM_NBW, TB, GEMM_NW, GemmEpilogue::SLAB>), grid, block, 0, rt().stream, a);
            break;
        default:
            hipLaunchKernelGGL((k_gemm<GEMM_NBW, TB, GEMM_NW, GemmEpilogue::SWIGLU_A16>), grid, block, 0, rt().stream, a);
            break;
    }
}

int launch_gemm(const WeightEntry& w, const __half* tiles, int N, const __half* a16, int T,
                const GemmEpilogue& epi, int force_ksplit, int prof_cls) {
    if (T <= 0 || N <= 0) return 1;
    GemmArgs a;
    a.W = tiles;
    a.nb_stride = w.nb_stride;
    a.NB = cdiv(N, 16);
    a.Kb = w.Kb;
    a.A = a16;
    a.Tb = cdiv(T, 16);
    a.N = N;
    a.T = T;
    int ks = 1;
    if (epi.kind == GemmEpilogue::SLAB) ks = force_ksplit > 0 ? force_ksplit : gemm_pick_ksplit(N, w.K, T);
    a.kb_per_split = cdiv(a.Kb, ks);
    ks = cdiv(a.Kb, a.kb_per_split);
    a.out = epi.out;
    a.ldo = epi.kind == GemmEpilogue::SLAB ? N : epi.ldo;
    a.slab_stride = (long)cdiv(T, 16) * 16 * N;
    a.a16_out = epi.a16_out;
    a.Kb_out = cdiv(N / 2, 32);
    if (epi.kind == GemmEpilogue::SWIGLU_A16 && ((N / 2) % 32 != 0 || a.NB % 2 != 0))
        NVLLM_FATAL("fused gate/up projection needs hidden_dim %% 32 == 0, got %d", N / 2);

    const int TB = T <= 16 ? 1 : (T <= 32 ? 2 : 4);
    const dim3 grid(cdiv(a.NB, GEMM_NBW), ks, cdiv(a.Tb, TB));
    const double bytes = (double)N * w.K * 2.0 + (double)T * w.K * 4.0 +
                         (double)T * N * (epi.kind == GemmEpilogue::SWIGLU_A16 ? 2.0 : 4.0 * ks);
    ProfScope ps(prof_cls, bytes, 2.0 * T * (double)N * w.K);
    if (TB == 1) launch_tb<1>(a, epi.kind, grid);
    else if (TB == 2) launch_tb<2>(a, epi.kind, grid);
    else launch_tb<4>(a, epi.kind, grid);
    return ks;
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
    hipLaunchKernelGGL(k_matmul_f32w, dim3(cdiv(N, 4), cdiv(T, 8)), dim3(256), 0, rt().stream, out, in, w, T, K, N);
}

}  // namespace nvllm
