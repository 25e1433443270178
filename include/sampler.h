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

/* Next token for one row of logits.  temperature == 0: first index of the
 * maximum (strict '>' scan, reference src/sampler.c:86-97).  In the HIP library
 * a row that belongs to the most recent transformer_batch call is answered from
 * the arg-max computed on the GPU during that call; any other row is reduced by
 * the same kernel after an upload.  temperature > 0 (reference
 * src/sampler.c:100-139): the GPU turns the rows of the most recent pass into
 * softmax(logits / temperature) and compacts the nucleus candidates
 * (prob >= (1 - topp) / (n - 1), in index order); the host sorts them with the
 * reference's comparator and draws with the reference's coin, libc rand()
 * seeded by build_sampler.  The row is NOT rewritten in place (the reference
 * leaves the probabilities in it; its driver never reads them). */
int sample(Sampler* s, float* logits);

#ifdef __cplusplus
}
#endif

#endif /* SAMPLER_H */
