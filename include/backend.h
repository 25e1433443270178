/*
 * backend.h -- allocation / copy shim, HIP edition.
 *
 * Replaces the CPU shim of reference include/backend.h:13-41, where
 * device_malloc/device_free/device_memcpy/device_sync are `static inline`
 * wrappers over malloc/free/memcpy/no-op.  Here they are real entry points of
 * libnvllm_hip.so and move bytes between host memory and HBM on the library's
 * stream.  Names, argument order, the DeviceMemcpyKind values and the
 * `check_status(stmt)` pass-through macro are unchanged so that callers written
 * against the reference header compile as they are.
 *
 * Error behaviour: the reference discards every status (backend.h:20), so a
 * caller cannot observe a failure.  device_malloc still returns -1 when HBM is
 * exhausted (as the reference returns -1 on a NULL malloc, backend.h:24-28);
 * every other HIP failure inside the library prints file:line and aborts.
 */
#ifndef BACKEND_H
#define BACKEND_H

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int DeviceMemcpyKind;          /* reference backend.h:14-17 */
#define DEVICE_TO_HOST   0
#define DEVICE_TO_DEVICE 1
#define HOST_TO_DEVICE   2

#define check_status(stmt) stmt        /* reference backend.h:20 */

/* hipMalloc; *ptr is 256-byte aligned HBM.  0 on success, -1 on failure.
 * Replaces reference backend.h:24-28. */
int device_malloc(void** ptr, size_t size);

/* hipFree (also accepts pointers handed out by malloc_run_state for the
 * host-visible logits buffer).  NULL is ignored.  Replaces backend.h:30-32. */
void device_free(void* ptr);

/* Copy `size` bytes in the stated direction, ordered after everything already
 * enqueued on the library stream.  DEVICE_TO_HOST returns only when the bytes
 * are in `dst`; HOST_TO_DEVICE returns when `src` may be reused.  Returns 0.
 * Replaces reference backend.h:34-37. */
int device_memcpy(void* dst, const void* src, size_t size, DeviceMemcpyKind kind);

/* Block until the library stream is idle.  Replaces backend.h:39-41 (which the
 * reference never calls: transformer_batch is synchronous on return). */
void device_sync(void);

#ifdef __cplusplus
}
#endif

#endif /* BACKEND_H */
