#ifndef AWS_COMPRESSION_HUFFMAN_TESTING_H
#define AWS_COMPRESSION_HUFFMAN_TESTING_H
/*
 * The coder-testing helpers the reference compiles into its library
 * (reference include/aws/compression/private/huffman_testing.h:35-97,
 * source/huffman_testing.c), here on top of the MI355X encode/decode entry points.
 * Users of the reference (aws-c-http's HPACK tests) call these to check their own
 * generated coders; same names, arguments and verdicts.
 *
 *   static struct huffman_test_code_point code_points[] = {
 *   #include "my_table.def"
 *   };
 */
#include <aws/compression/huffman.h>

/* one row of a table .def file (reference huffman_testing.h:35-38) */
struct huffman_test_code_point {
    uint8_t symbol;
    struct aws_huffman_code code;
};

/* expands a .def row into a huffman_test_code_point initialiser (reference huffman_testing.h:44-52) */
#define HUFFMAN_CODE(psymbol, pbit_string, pbit_pattern, pnum_bits)                                                    \
    {.symbol = (psymbol), .code = {.pattern = (pbit_pattern), .num_bits = (pnum_bits)}},

AWS_EXTERN_C_BEGIN

/*
 * input == decode(encode(input)) in one encode call and one decode call
 * (reference source/huffman_testing.c:15-73).  encoded_size 0 skips the length check.
 * AWS_OP_SUCCESS, or AWS_OP_ERR with *error_string saying which check failed.
 */
AWS_COMPRESSION_API
int huffman_test_transitive(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    const char **error_string);

/*
 * The same with the output offered output_chunk_size bytes at a time on both sides: every
 * call must write something and may only fail with AWS_ERROR_SHORT_BUFFER
 * (reference source/huffman_testing.c:75-173).
 */
AWS_COMPRESSION_API
int huffman_test_transitive_chunked(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    size_t output_chunk_size,
    const char **error_string);

AWS_EXTERN_C_END

#endif /* AWS_COMPRESSION_HUFFMAN_TESTING_H */
