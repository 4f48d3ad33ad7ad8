/*
 * Library init/clean-up (reference source/compression.c:26-44).  The reference
 * registers its error strings with aws-c-common; the Huffman entry points of
 * this library do not need that, so these are bookkeeping only.
 */
#include <aws/compression/compression.h>

static int s_initialized;

void aws_compression_library_init(struct aws_allocator *alloc) {
    (void)alloc;
    s_initialized = 1;
}

void aws_compression_library_clean_up(void) {
    s_initialized = 0;
}

/* name of a compression error code, NULL for codes outside the package's range */
AWS_COMPRESSION_API const char *aws_compression_error_name(int err) {
    if (err == AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL) {
        return "AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL"; /* reference source/compression.c:13-17 */
    }
    return 0;
}
