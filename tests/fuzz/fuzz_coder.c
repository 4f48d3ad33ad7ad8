/*
 * TEST INFRASTRUCTURE.  The coder the fuzz properties run on: the reference's test table
 * (tests/golden/test_coder_table.json, turned into test_coder_rows.h by the Makefile), built at run time
 * with aws_huffman_amd_table_coder_new -- the counterpart of the generated test_get_coder() that the
 * reference's fuzz targets declare (reference tests/fuzz/decode.c:11).
 */
#include <aws/compression/huffman_amd.h>

#include <stdlib.h>

#include "test_coder_rows.h" /* static const uint32_t k_patterns[256]; static const uint8_t k_num_bits[256]; */

struct aws_huffman_symbol_coder *test_get_coder(void) {
    static struct aws_huffman_symbol_coder *coder;
    if (!coder) {
        coder = aws_huffman_amd_table_coder_new(k_patterns, k_num_bits);
        if (!coder) {
            abort();
        }
    }
    return coder;
}
