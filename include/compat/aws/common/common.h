#ifndef AWS_HUFFMAN_AMD_COMPAT_COMMON_H
#define AWS_HUFFMAN_AMD_COMPAT_COMMON_H
/*
 * Minimal stand-in for the slice of aws-c-common that the Huffman hot path touches.
 *
 * aws-c-common is an external, un-vendored dependency of the reference
 * (reference CMakeLists.txt:6, builder.json:3-5) and is not present on the build
 * or GPU boxes.  This directory (include/compat) is put on the include path ONLY
 * when the real library is absent; with the real aws-c-common installed, drop
 * -Iinclude/compat and define AWS_HUFFMAN_AMD_USE_SYSTEM_AWS_C_COMMON so that
 * libaws-c-compression-amd stops exporting its own aws_raise_error/aws_last_error.
 *
 * Written from the public API as the reference uses it (call sites listed in
 * SURVEY.md section 8c), not from aws-c-common's sources.  Numeric values of
 * the aws-c-common error codes are from memory of its public error.h and are
 * the one part of the boundary the reference's tests never pin (they compare
 * by name: reference tests/huffman_test.c:154,353).
 */

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
#    define AWS_EXTERN_C_BEGIN extern "C" {
#    define AWS_EXTERN_C_END }
#else
#    define AWS_EXTERN_C_BEGIN
#    define AWS_EXTERN_C_END
#endif

#define AWS_PUSH_SANE_WARNING_LEVEL
#define AWS_POP_SANE_WARNING_LEVEL

#define AWS_OP_SUCCESS (0)
#define AWS_OP_ERR (-1)

/* Each CRT package owns a 1024-wide band of error codes. */
#define AWS_ERROR_ENUM_STRIDE_BITS 10
#define AWS_ERROR_ENUM_STRIDE (1U << AWS_ERROR_ENUM_STRIDE_BITS)
#define AWS_ERROR_ENUM_BEGIN_RANGE(x) ((x)*AWS_ERROR_ENUM_STRIDE)
#define AWS_ERROR_ENUM_END_RANGE(x) (((x) + 1) * AWS_ERROR_ENUM_STRIDE - 1)

/* Only the codes the Huffman path can raise are named. */
enum aws_common_error {
    AWS_ERROR_SUCCESS = 0,
    AWS_ERROR_OOM = 1,
    AWS_ERROR_UNKNOWN = 3,
    AWS_ERROR_SHORT_BUFFER = 4,
    AWS_ERROR_UNSUPPORTED_OPERATION = 6,
    AWS_ERROR_INVALID_ARGUMENT = 34,
    AWS_ERROR_INVALID_STATE = 38
};

#ifndef AWS_ASSERT
#    if defined(DEBUG_BUILD)
#        include <assert.h>
#        define AWS_ASSERT(cond) assert(cond)
#    else
#        define AWS_ASSERT(cond) ((void)0)
#    endif
#endif
#define AWS_PRECONDITION(cond) AWS_ASSERT(cond)
#define AWS_FATAL_ASSERT(cond)                                                                                         \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            __builtin_trap();                                                                                          \
        }                                                                                                              \
    } while (0)

#define AWS_ZERO_STRUCT(object) memset(&(object), 0, sizeof(object))
#define AWS_ZERO_ARRAY(array) memset((void *)(array), 0, sizeof(array))
#define AWS_VARIABLE_LENGTH_ARRAY(type, name, length) type name[length]

struct aws_allocator {
    void *(*mem_acquire)(struct aws_allocator *allocator, size_t size);
    void (*mem_release)(struct aws_allocator *allocator, void *ptr);
    void *(*mem_realloc)(struct aws_allocator *allocator, void *oldptr, size_t oldsize, size_t newsize);
    void *(*mem_calloc)(struct aws_allocator *allocator, size_t num, size_t size);
    void *impl;
};

/* The error-string registry: a package hands a list of its codes' names to aws-c-common at library init
 * (reference source/compression.c:13-33) and anybody can ask for a code's name. */
struct aws_error_info {
    int error_code;
    const char *literal_name;
    const char *error_str;
    const char *lib_name;
    const char *formatted_name;
};
struct aws_error_info_list {
    const struct aws_error_info *error_list;
    uint16_t count;
};

AWS_EXTERN_C_BEGIN

/* Thread-local last-error and the error-name registry, exported by libaws-c-compression-amd when the
 * real aws-c-common is absent. */
int aws_raise_error(int err);
int aws_last_error(void);
void aws_reset_error(void);
struct aws_allocator *aws_default_allocator(void);
void aws_register_error_info(const struct aws_error_info_list *error_info);
void aws_unregister_error_info(const struct aws_error_info_list *error_info);
const char *aws_error_name(int err);  /* "Unknown Error Code" for a code nobody registered */
const char *aws_error_str(int err);
void aws_common_library_init(struct aws_allocator *allocator);
void aws_common_library_clean_up(void);

AWS_EXTERN_C_END

#endif /* AWS_HUFFMAN_AMD_COMPAT_COMMON_H */
