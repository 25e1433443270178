// resid_tail.h -- the tail of a projection that returns to the residual width, for eight consecutive
// features of one token: x += y (the reference's accum(), src/kernels/layers.c:163-168), the new x
// times the next norm's gain as fp16 hi + lo in tile order, and the lane's share of the sum of squares.
// One definition for the kernels that run it: the stand-alone residual kernel (elementwise.hip) and the
// last phase of the peer-memory all-reduce (tp_peer.hip), whose results must be the same bits.
#pragma once

#include "nvllm_internal.h"

namespace nvllm {

// fp32 -> (hi, lo) fp16 pair with hi + lo == x to ~22 bits.  |x| is clamped to
// the fp16 range first (activations entering a projection never get near it).
__device__ __forceinline__ void split16(float x, __half& hi, __half& lo) {
    const float c = fminf(fmaxf(x, -65504.f), 65504.f);
    hi = __float2half_rn(c);
    lo = __float2half_rn(c - __half2float(hi));
}

struct alignas(16) Half8 {
    __half v[8];
};

// offset (in halves) of the 8-element chunk that holds (t, k..k+7), k % 8 == 0
__device__ __forceinline__ size_t a16_chunk(int t, int k, int Kb, int plane) {
    const size_t tile = ((size_t)(t >> 4) * Kb + (k >> 5)) * 2 + plane;
    const int lane = (t & 15) + 16 * ((k >> 3) & 3);
    return tile * TILE_HALVES + (size_t)lane * 8;
}

__device__ __forceinline__ void store_a16_8(__half* a16, int t, int k, int Kb, const float* x8) {
    Half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; j++) split16(x8[j], hi.v[j], lo.v[j]);
    *reinterpret_cast<Half8*>(a16 + a16_chunk(t, k, Kb, 0)) = hi;
    *reinterpret_cast<Half8*>(a16 + a16_chunk(t, k, Kb, 1)) = lo;
}

// Features i .. i+7 (i % 8 == 0) of token t: xr = the token's residual row, (b0, b1) = the eight sums to add.
// Returns the sum of the eight new squares, added in feature order.
__device__ __forceinline__ float resid_tail_8(float* xr, float4 b0, float4 b1, __half* a16, const float* gain, int t, int i, int Kb) {
    float v[8], g[8], part = 0.f;
    const float4 a0 = *reinterpret_cast<const float4*>(xr + i), a1 = *reinterpret_cast<const float4*>(xr + i + 4);
    v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
    v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        g[j] = v[j] * gain[i + j];
        part += v[j] * v[j];
    }
    *reinterpret_cast<float4*>(xr + i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(xr + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    store_a16_8(a16, t, i, Kb, g);
    return part;
}

}  // namespace nvllm
