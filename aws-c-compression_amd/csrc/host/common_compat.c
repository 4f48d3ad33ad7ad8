/*
 * The few aws-c-common entry points the Huffman path needs, exported only when
 * the real aws-c-common is not linked (see include/compat/aws/common/common.h).
 */
#ifndef AWS_HUFFMAN_AMD_USE_SYSTEM_AWS_C_COMMON

#include <aws/common/common.h>
#include <aws/compression/exports.h>

#include <stdlib.h>

static _Thread_local int tl_last_error;

AWS_COMPRESSION_API int aws_raise_error(int err) {
    tl_last_error = err;
    return AWS_OP_ERR;
}

AWS_COMPRESSION_API int aws_last_error(void) {
    return tl_last_error;
}

AWS_COMPRESSION_API void aws_reset_error(void) {
    tl_last_error = 0;
}

static void *heap_acquire(struct aws_allocator *a, size_t n) {
    (void)a;
    return malloc(n);
}
static void heap_release(struct aws_allocator *a, void *p) {
    (void)a;
    free(p);
}
static void *heap_realloc(struct aws_allocator *a, void *p, size_t o, size_t n) {
    (void)a;
    (void)o;
    return realloc(p, n);
}
static void *heap_calloc(struct aws_allocator *a, size_t k, size_t n) {
    (void)a;
    return calloc(k, n);
}

AWS_COMPRESSION_API struct aws_allocator *aws_default_allocator(void) {
    static struct aws_allocator heap = {heap_acquire, heap_release, heap_realloc, heap_calloc, NULL};
    return &heap;
}

#endif /* AWS_HUFFMAN_AMD_USE_SYSTEM_AWS_C_COMMON */
