/*
 * TEST INFRASTRUCTURE.  Fuzz property F3 (reference tests/fuzz/transitive_chunked.c:13-30): the same round trip
 * with the output offered 1, 2, 4 ... 128 bytes at a time on both sides (huffman_test_transitive_chunked: every
 * call writes something, the only failure allowed on the way is AWS_ERROR_SHORT_BUFFER).
 */
#include <aws/compression/private/huffman_testing.h>

#include <stdio.h>
#include <stdlib.h>

struct aws_huffman_symbol_coder *test_get_coder(void);

int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) {
    if (size == 0) {
        return 0;
    }
    static const size_t steps[] = {1, 2, 4, 8, 16, 32, 64, 128};
    for (size_t k = 0; k < sizeof(steps) / sizeof(steps[0]); ++k) {
        const char *why = NULL;
        if (huffman_test_transitive_chunked(test_get_coder(), (const char *)data, size, 0, steps[k], &why) != AWS_OP_SUCCESS) {
            fprintf(stderr, "transitive_chunked, %zu bytes at a time: %s (input of %zu bytes)\n", steps[k], why ? why : "?", size);
            abort();
        }
    }
    return 0;
}
