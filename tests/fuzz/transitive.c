/*
 * TEST INFRASTRUCTURE.  Fuzz property F2 (reference tests/fuzz/transitive.c:13-24): every non-empty byte string
 * survives encode + decode (huffman_test_transitive, encoded size not checked).
 */
#include <aws/compression/private/huffman_testing.h>

#include <stdio.h>
#include <stdlib.h>

struct aws_huffman_symbol_coder *test_get_coder(void);

int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) {
    if (size == 0) {
        return 0;
    }
    const char *why = NULL;
    if (huffman_test_transitive(test_get_coder(), (const char *)data, size, 0, &why) != AWS_OP_SUCCESS) {
        fprintf(stderr, "transitive: %s (input of %zu bytes)\n", why ? why : "?", size);
        abort();
    }
    return 0;
}
