/*
 * Library bring-up and tear-down (what reference source/compression.c:26-44 does for the package): the package's
 * one error code gets its name and text in aws-c-common's error registry, so that aws_error_name() /
 * aws_error_str() can name AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL (reference tests/library_test.c:9-22 checks
 * exactly that), and aws-c-common itself is brought up / torn down.  Both build flavours do the same: against the
 * real aws-c-common (make AWS_C_COMMON_PREFIX=...) the registry is aws-c-common's, stand-alone it is the small one
 * of csrc/host/common_compat.c behind the same functions.
 * The Huffman entry points work without either call; the GPU side is brought up lazily by the first engine
 * (engine.c), not here, so that init stays cheap and cannot fail.  Clean-up also drops the engines cached per coder.
 */
#include <aws/compression/compression.h>

#ifdef AWS_HUFFMAN_AMD_USE_SYSTEM_AWS_C_COMMON
#    include <aws/common/error.h>
#endif

static const char k_unknown_symbol_name[] = "AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL";

static const struct aws_error_info s_package_errors[1] = {
    {AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL, k_unknown_symbol_name, "Compression encountered an unknown symbol.", "aws-c-compression",
     "aws-c-compression: AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL, Compression encountered an unknown symbol."},
};
static struct aws_error_info_list s_package_error_list = {s_package_errors, 1};

static int s_up; /* 0 = down, 1 = up; init and clean-up are idempotent like the reference's */

static void set_up(int up, struct aws_allocator *alloc) {
    if (s_up == up) {
        return;
    }
    s_up = up;
    if (up) {
        aws_common_library_init(alloc);
        aws_register_error_info(&s_package_error_list);
    } else {
        aws_unregister_error_info(&s_package_error_list);
        aws_common_library_clean_up();
    }
}

void aws_compression_library_init(struct aws_allocator *alloc) {
    set_up(1, alloc);
}

void aws_huffman_amd_forget_all(void); /* huffman.c: the engines behind the eight reference entry points */

void aws_compression_library_clean_up(void) {
    aws_huffman_amd_forget_all(); /* the device tables and staging buffers cached per coder */
    set_up(0, 0);
}

/* name of a compression error code, NULL for codes outside the package's range */
AWS_COMPRESSION_API const char *aws_compression_error_name(int err) {
    return err == AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL ? k_unknown_symbol_name : 0;
}
