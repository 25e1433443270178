/*
 * sampler.h -- consumer of the forward pass's logits.
 *
 * The reference has no header for this: `Sampler` is declared twice, in
 * src/sampler.c:12-18 and again in src/main.c:39-45, and the two must agree.
 * This header states that layout once.  libnvllm_hip.so exports build_sampler and
 * sample with these signatures; a driver may equally link the reference's own
 * sampler.c, which works on the host-visible logits the library returns.
 */
#ifndef SAMPLER_H
#define SAMPLER_H

#include "structs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    unsigned long long state; /* seed; the reference draws from libc rand()    */
    float temperature;        /* 0 => greedy                                   */
    float topp;               /* nucleus mass, (0,1); otherwise plain sampling */
    ProbIndex* probindex;     /* host scratch, vocab_size entries              */
    int vocab_size;
} Sampler;

/* Reference: src/sampler.c:20-27 (also calls srand(seed)). */
void build_sampler(Sampler* s, int vocab_size, float temperature, float topp,
                   unsigned long long seed);

/* Next token for one row of logits (reference src/sampler.c:84-140).
 * temperature == 0: first index of the maximum (strict '>' scan, sampler.c:86-97).
 * temperature > 0 (sampler.c:100-139): softmax(logits / temperature), then plain or
 * nucleus sampling -- candidates with prob >= (1 - topp) / (n - 1), sorted with the
 * reference's comparator, drawn with the reference's coin, libc rand() seeded by
 * build_sampler.  In the HIP library the softmax and the candidate filter run on the GPU.
 * Which memory is read depends on the hand-off mode (nvllm.h, nvllm_set_logits_mode):
 *   NVLLM_LOGITS_HOST    `logits` is host memory and is read as the caller left it, so
 *                        edits made after transformer_batch (logit bias, banned tokens)
 *                        count; with a temperature the row is rewritten in place to the
 *                        probabilities, as the reference does (sampler.c:100-121).
 *   NVLLM_LOGITS_ARGMAX  RunState.logits is never filled: a pointer to its row k names
 *                        output k of the most recent transformer_batch call, which is
 *                        answered from the logits still in HBM (host edits cannot exist).
 *                        Any other pointer is treated as a host row, as above. */
int sample(Sampler* s, float* logits);

#ifdef __cplusplus
}
#endif

#endif /* SAMPLER_H */
