/*
 * TEST INFRASTRUCTURE.  Fuzz property F1 (reference tests/fuzz/decode.c:13-32): any bytes at all, handed to a
 * fresh decoder as if they were an encoded stream, with room for twice as many symbols -- the outcome is not
 * looked at, the call must come back without touching memory it does not own.
 * LLVMFuzzerTestOneInput is libFuzzer's entry point; corpus_driver.c calls it over a seeded corpus where
 * libFuzzer is not at hand.
 */
#include <aws/compression/huffman.h>

#include <stdlib.h>
#include <string.h>

struct aws_huffman_symbol_coder *test_get_coder(void);

int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) {
    if (size == 0) {
        return 0;
    }
    struct aws_huffman_decoder decoder;
    aws_huffman_decoder_init(&decoder, test_get_coder());

    /* guard bytes either side of the output: a decoder that writes past what it reports is caught here, too */
    const size_t room = size * 2, guard = 32;
    uint8_t *block = malloc(room + 2 * guard);
    if (!block) {
        abort();
    }
    memset(block, 0xA5, room + 2 * guard);
    struct aws_byte_cursor to_decode = {size, (uint8_t *)data};
    struct aws_byte_buf output = {0, block + guard, room, NULL};
    (void)aws_huffman_decode(&decoder, &to_decode, &output);
    if (output.len > room || to_decode.len > size) {
        abort();
    }
    for (size_t i = 0; i < guard; ++i) {
        if (block[i] != 0xA5 || block[guard + room + i] != 0xA5) {
            abort();
        }
    }
    for (size_t i = output.len; i < room; ++i) {
        if (block[guard + i] != 0xA5) {
            abort(); /* wrote behind the symbols it reports */
        }
    }
    free(block);
    return 0;
}
