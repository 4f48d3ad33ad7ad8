/*
 * Library bring-up and tear-down (what reference source/compression.c:26-44 does for the package).
 *
 * Two build flavours share this file:
 *   - against the real aws-c-common (make AWS_C_COMMON_PREFIX=...): the package's single error
 *     string is handed to aws-c-common's registry, so aws_error_name()/aws_error_str() can name
 *     AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL, and aws-c-common itself is brought up / torn down;
 *   - stand-alone (include/compat): no registry exists; the calls only flip the flag.
 * The Huffman entry points work without either call; the GPU side is brought up lazily by the
 * first engine (engine.c), not here, so that init stays cheap and cannot fail.
 */
#include <aws/compression/compression.h>

#ifdef AWS_HUFFMAN_AMD_USE_SYSTEM_AWS_C_COMMON
#    include <aws/common/error.h>
#    define HUFFMAN_AMD_HAVE_ERROR_REGISTRY 1
#else
#    define HUFFMAN_AMD_HAVE_ERROR_REGISTRY 0
#endif

static const char k_unknown_symbol_name[] = "AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL";

#if HUFFMAN_AMD_HAVE_ERROR_REGISTRY
static struct aws_error_info s_package_errors[1];
static struct aws_error_info_list s_package_error_list = {s_package_errors, 1};
#endif

static int s_up; /* 0 = down, 1 = up; init and clean-up are idempotent like the reference's */

static void set_up(int up, struct aws_allocator *alloc) {
    if (s_up == up) {
        return;
    }
    s_up = up;
#if HUFFMAN_AMD_HAVE_ERROR_REGISTRY
    if (up) {
        s_package_errors[0].error_code = AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL;
        s_package_errors[0].literal_name = k_unknown_symbol_name;
        s_package_errors[0].error_str = "Compression encountered an unknown symbol.";
        s_package_errors[0].lib_name = "aws-c-compression";
        s_package_errors[0].formatted_name =
            "aws-c-compression: AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL, Compression encountered an unknown symbol.";
        aws_common_library_init(alloc);
        aws_register_error_info(&s_package_error_list);
    } else {
        aws_unregister_error_info(&s_package_error_list);
        aws_common_library_clean_up();
    }
#else
    (void)alloc;
#endif
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
