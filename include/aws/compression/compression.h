#ifndef AWS_COMPRESSION_COMPRESSION_H
#define AWS_COMPRESSION_COMPRESSION_H
/*
 * Package id and error codes of the compression package.
 * Replaces reference include/aws/compression/compression.h:13-37 with the same
 * names and values: package id 3, so AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL is
 * 3 * 0x400 = 0x0C00.  The value is part of the drop-in boundary because
 * aws_huffman_encode/decode raise it (reference source/huffman.c:63,246).
 */
#include <aws/common/common.h>
#include <aws/compression/exports.h>

/* aws-c-common hands every package a block of 1024 error codes; this package owns block 3 */
#define AWS_C_COMPRESSION_PACKAGE_ID 3

enum aws_compression_error {
    /* first code of the block: a byte (encode) or a bit pattern (decode) the symbol coder does not know */
    AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL = AWS_ERROR_ENUM_BEGIN_RANGE(AWS_C_COMPRESSION_PACKAGE_ID),
    /* last code of the block, never raised */
    AWS_ERROR_END_COMPRESSION_RANGE = AWS_ERROR_ENUM_END_RANGE(AWS_C_COMPRESSION_PACKAGE_ID)
};

AWS_EXTERN_C_BEGIN
/*
 * reference source/compression.c:26-44.  The reference registers its error
 * strings with aws-c-common here.  In this library the call is optional: the
 * Huffman entry points do not depend on it, and the GPU is not touched until
 * the first encode or decode.  Both calls are idempotent.
 */
AWS_COMPRESSION_API void aws_compression_library_init(struct aws_allocator *alloc);
AWS_COMPRESSION_API void aws_compression_library_clean_up(void);
AWS_EXTERN_C_END

#endif /* AWS_COMPRESSION_COMPRESSION_H */
