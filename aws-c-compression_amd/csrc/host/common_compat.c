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

/* ---- error names: the registry packages add their lists to (one slot per package id, as the codes are banded) */

enum { ERROR_SLOTS = 32 };
static const struct aws_error_info_list *volatile s_error_lists[ERROR_SLOTS];

/* the aws-c-common codes this library can raise, so that they have names here too */
static const struct aws_error_info s_common_errors[] = {
    {AWS_ERROR_SUCCESS, "AWS_ERROR_SUCCESS", "Success.", "aws-c-common", "aws-c-common: AWS_ERROR_SUCCESS, Success."},
    {AWS_ERROR_OOM, "AWS_ERROR_OOM", "Out of memory.", "aws-c-common", "aws-c-common: AWS_ERROR_OOM, Out of memory."},
    {AWS_ERROR_UNKNOWN, "AWS_ERROR_UNKNOWN", "Unknown error.", "aws-c-common", "aws-c-common: AWS_ERROR_UNKNOWN, Unknown error."},
    {AWS_ERROR_SHORT_BUFFER, "AWS_ERROR_SHORT_BUFFER", "Buffer is not large enough to hold result.", "aws-c-common",
     "aws-c-common: AWS_ERROR_SHORT_BUFFER, Buffer is not large enough to hold result."},
    {AWS_ERROR_UNSUPPORTED_OPERATION, "AWS_ERROR_UNSUPPORTED_OPERATION", "Unsupported operation.", "aws-c-common",
     "aws-c-common: AWS_ERROR_UNSUPPORTED_OPERATION, Unsupported operation."},
    {AWS_ERROR_INVALID_ARGUMENT, "AWS_ERROR_INVALID_ARGUMENT", "An argument has an illegal value.", "aws-c-common",
     "aws-c-common: AWS_ERROR_INVALID_ARGUMENT, An argument has an illegal value."},
    {AWS_ERROR_INVALID_STATE, "AWS_ERROR_INVALID_STATE", "An object's state is not valid for the operation.", "aws-c-common",
     "aws-c-common: AWS_ERROR_INVALID_STATE, An object's state is not valid for the operation."},
};

static int slot_of(const struct aws_error_info_list *list) {
    if (!list || !list->error_list || list->count == 0) {
        return -1;
    }
    const unsigned slot = (unsigned)list->error_list[0].error_code >> AWS_ERROR_ENUM_STRIDE_BITS;
    return slot < ERROR_SLOTS ? (int)slot : -1;
}

AWS_COMPRESSION_API void aws_register_error_info(const struct aws_error_info_list *list) {
    const int slot = slot_of(list);
    if (slot >= 0) {
        s_error_lists[slot] = list;
    }
}

AWS_COMPRESSION_API void aws_unregister_error_info(const struct aws_error_info_list *list) {
    const int slot = slot_of(list);
    if (slot >= 0 && s_error_lists[slot] == list) {
        s_error_lists[slot] = NULL;
    }
}

static const struct aws_error_info *info_of(int err) {
    if (err < 0) {
        return NULL;
    }
    const unsigned slot = (unsigned)err >> AWS_ERROR_ENUM_STRIDE_BITS;
    if (slot == 0) {
        for (size_t i = 0; i < sizeof(s_common_errors) / sizeof(s_common_errors[0]); ++i) {
            if (s_common_errors[i].error_code == err) {
                return &s_common_errors[i];
            }
        }
        return NULL;
    }
    const struct aws_error_info_list *list = slot < ERROR_SLOTS ? s_error_lists[slot] : NULL;
    for (unsigned i = 0; list && i < list->count; ++i) {
        if (list->error_list[i].error_code == err) {
            return &list->error_list[i];
        }
    }
    return NULL;
}

AWS_COMPRESSION_API const char *aws_error_name(int err) {
    const struct aws_error_info *info = info_of(err);
    return info ? info->literal_name : "Unknown Error Code";
}

AWS_COMPRESSION_API const char *aws_error_str(int err) {
    const struct aws_error_info *info = info_of(err);
    return info ? info->error_str : "Unknown Error Code";
}

AWS_COMPRESSION_API void aws_common_library_init(struct aws_allocator *allocator) {
    (void)allocator; /* nothing of aws-c-common's own to bring up in this flavour */
}

AWS_COMPRESSION_API void aws_common_library_clean_up(void) {
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
