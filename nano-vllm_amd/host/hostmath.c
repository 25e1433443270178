/*
 * hostmath.c -- the few host-side floating-point computations whose results must
 * match the reference bit for bit.  Built with plain gcc (-O2, no FMA target, no
 * -ffast-math) so that the operations below round exactly as they do in the
 * reference's CPU build.
 */
#include <math.h>
#include <stddef.h>

/* RoPE angles.  table[(p * (head_dim/2) + i/2) * 2 + {0,1}] = {cos, sin} of
 * (float)p * (1 / powf(theta, i / (float)head_dim)) for even i: the expression
 * of reference src/kernels/layers.c:119-122, evaluated with the same libm. */
void nvllm_host_rope_table(float* table, float theta, int head_dim, int first_pos, int n_pos) {
    const int half = head_dim / 2;
    for (int p = first_pos; p < first_pos + n_pos; p++) {
        for (int i = 0; i < head_dim; i += 2) {
            float freq = 1.0f / powf(theta, i / (float)head_dim);
            float val = p * freq;
            float* dst = table + ((size_t)(p - first_pos) * half + i / 2) * 2;
            dst[0] = cosf(val);
            dst[1] = sinf(val);
        }
    }
}

/* 1/sqrtf(head_dim), the attention scale (reference src/kernels/attention.c:259). */
float nvllm_host_attn_scale(int head_dim) { return 1.0f / sqrtf(head_dim); }
