#ifndef AWS_COMPRESSION_HUFFMAN_H
#define AWS_COMPRESSION_HUFFMAN_H
/*
 * Streaming byte-symbol Huffman codec -- the C ABI of libaws-c-compression-amd.
 *
 * Every declaration here replaces the declaration of the same name in the
 * reference's include/aws/compression/huffman.h (line numbers cited per item)
 * with identical type layout, argument meaning and error behaviour, so a caller
 * compiled against the reference header links against this library unchanged.
 * The symbol work behind aws_huffman_encode / aws_huffman_decode /
 * aws_huffman_get_encoded_length runs as HIP kernels on an MI355X; there is no
 * CPU implementation of it in this library (see huffman_amd.h for the
 * device-pointer and batched entry points that the same kernels serve).
 *
 * Bitstream (what "bit-exact" means, reference source/huffman.c:59-105,178-184):
 * codes are concatenated most-significant-bit first with no header; when an
 * encode call consumes all of its input the last partial byte is completed with
 * the LOW bits of encoder->eos_padding.
 */

#include <aws/compression/compression.h>

#include <aws/common/byte_buf.h>

AWS_PUSH_SANE_WARNING_LEVEL

/* reference huffman.h:18-26.  8 bytes: pattern@0 (code in the low num_bits bits), num_bits@4. */
struct aws_huffman_code {
    uint32_t pattern;
    uint8_t num_bits;
};

/*
 * reference huffman.h:37.  symbol -> code; num_bits == 0 means "no code for this symbol".
 * Must be a pure function of (symbol, userdata): the library tabulates it once
 * per coder and stages the table in device memory.
 */
typedef struct aws_huffman_code(aws_huffman_symbol_encoder_fn)(uint8_t symbol, void *userdata);

/*
 * reference huffman.h:48.  `bits` holds the next 32 stream bits, first bit in bit 31,
 * zero-filled past the end of data.  Returns the length of the code found and
 * stores its symbol, or returns 0 (and leaves *symbol alone) when no code matches.
 * Must be pure, like the encoder callback.
 */
typedef uint8_t(aws_huffman_symbol_decoder_fn)(uint32_t bits, uint8_t *symbol, void *userdata);

/* reference huffman.h:53-57.  24 bytes: encode@0 decode@8 userdata@16. */
struct aws_huffman_symbol_coder {
    aws_huffman_symbol_encoder_fn *encode;
    aws_huffman_symbol_decoder_fn *decode;
    void *userdata;
};

/*
 * reference huffman.h:63-70.  24 bytes: coder@0 eos_padding@8 overflow_bits@12.
 * eos_padding is a public knob (default 0xFF); overflow_bits carries the unwritten
 * tail of the last consumed symbol across a SHORT_BUFFER return.
 */
struct aws_huffman_encoder {
    struct aws_huffman_symbol_coder *coder;
    uint8_t eos_padding;

    struct aws_huffman_code overflow_bits;
};

/*
 * reference huffman.h:76-84.  32 bytes: coder@0 allow_growth@8 working_bits@16 num_bits@24.
 * working_bits holds the num_bits read-ahead bits, most significant first.
 */
struct aws_huffman_decoder {
    struct aws_huffman_symbol_coder *coder;
    bool allow_growth;

    uint64_t working_bits;
    uint8_t num_bits;
};

AWS_EXTERN_C_BEGIN

/* reference huffman.h:91-92, source/huffman.c:12-20.  Zeroes the state, eos_padding = 0xFF. */
AWS_COMPRESSION_API void aws_huffman_encoder_init(struct aws_huffman_encoder *encoder, struct aws_huffman_symbol_coder *coder);

/* reference huffman.h:97-98, source/huffman.c:22-27.  Drops pending overflow bits only. */
AWS_COMPRESSION_API void aws_huffman_encoder_reset(struct aws_huffman_encoder *encoder);

/* reference huffman.h:103-104, source/huffman.c:29-36.  Zeroes the state (allow_growth = false). */
AWS_COMPRESSION_API void aws_huffman_decoder_init(struct aws_huffman_decoder *decoder, struct aws_huffman_symbol_coder *coder);

/* reference huffman.h:109-110, source/huffman.c:38-42.  Drops the read-ahead bits only. */
AWS_COMPRESSION_API void aws_huffman_decoder_reset(struct aws_huffman_decoder *decoder);

/*
 * reference huffman.h:120-121, source/huffman.c:107-129.
 * ceil(sum of code lengths / 8) for the bytes under `to_encode`; pending
 * overflow bits are not counted.
 */
AWS_COMPRESSION_API size_t aws_huffman_get_encoded_length(struct aws_huffman_encoder *encoder,
                                                          struct aws_byte_cursor to_encode);

/*
 * reference huffman.h:132-136, source/huffman.c:131-187.
 * Appends the encoding of `to_encode` to `output`, advancing the cursor past
 * every symbol it consumed.  AWS_OP_SUCCESS when all input was consumed (the
 * last byte is then padded).  AWS_OP_ERR + AWS_ERROR_SHORT_BUFFER when the
 * output filled first: call again with more room, same encoder, same cursor.
 * AWS_OP_ERR + AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL for a symbol without a code.
 */
AWS_COMPRESSION_API int aws_huffman_encode(struct aws_huffman_encoder *encoder, struct aws_byte_cursor *to_encode,
                                           struct aws_byte_buf *output);

/*
 * reference huffman.h:148-152, source/huffman.c:213-286.
 * Appends decoded symbols to `output`, pulling bytes from the cursor up to 39
 * bits ahead of the decode point (they stay in decoder->working_bits).
 * AWS_OP_SUCCESS at end of data or when only an incomplete code / fewer than 32
 * undecodable bits remain.  AWS_OP_ERR + AWS_ERROR_SHORT_BUFFER when the output
 * is full and growth is off; AWS_OP_ERR + AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL
 * when 32 or more bits remain and they start with no valid code.
 */
AWS_COMPRESSION_API int aws_huffman_decode(struct aws_huffman_decoder *decoder, struct aws_byte_cursor *to_decode,
                                           struct aws_byte_buf *output);

/* reference huffman.h:158-159, source/huffman.c:44-46.  Off by default. */
AWS_COMPRESSION_API void aws_huffman_decoder_allow_growth(struct aws_huffman_decoder *decoder, bool allow_growth);

AWS_EXTERN_C_END
AWS_POP_SANE_WARNING_LEVEL

#endif /* AWS_COMPRESSION_HUFFMAN_H */
