/* TEST INFRASTRUCTURE.  Uses only what the reference's headers declare (no GPU needed: init and the error registry). */
#include <aws/compression/compression.h>
#include <aws/compression/huffman.h>

#include <stdio.h>
#include <string.h>

static struct aws_huffman_code no_code(uint8_t symbol, void *userdata) {
    (void)symbol;
    (void)userdata;
    struct aws_huffman_code c = {0, 0};
    return c;
}

int main(void) {
    struct aws_huffman_symbol_coder coder = {no_code, NULL, NULL};
    struct aws_huffman_encoder encoder;
    aws_huffman_encoder_init(&encoder, &coder);
    aws_compression_library_init(NULL);
    const int ok = encoder.eos_padding == 0xFF &&
                   strcmp(aws_error_name(AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL), "AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL") == 0;
    aws_compression_library_clean_up();
    printf("%s\n", ok ? "linked and initialised" : "unexpected state");
    return ok ? 0 : 1;
}
