/*
 * huffman_test_transitive / huffman_test_transitive_chunked: the round-trip checks the
 * reference ships inside its library (reference source/huffman_testing.c:15-173), run here
 * through this library's own aws_huffman_encode / aws_huffman_decode, i.e. on the GPU.
 */
#include <aws/compression/private/huffman_testing.h>

#include <stdlib.h>
#include <string.h>

/* the two scratch buffers of a round trip: encoded bytes (twice the input is taken to be enough,
 * huffman_testing.c:27) and the symbols decoded back */
struct round_trip {
    struct aws_huffman_encoder encoder;
    struct aws_huffman_decoder decoder;
    uint8_t *encoded;
    uint8_t *decoded;
    size_t encoded_room;
};

static bool round_trip_begin(struct round_trip *rt, struct aws_huffman_symbol_coder *coder, size_t size, size_t slack) {
    aws_huffman_encoder_init(&rt->encoder, coder);
    aws_huffman_decoder_init(&rt->decoder, coder);
    rt->encoded_room = size * 2;
    rt->encoded = calloc(rt->encoded_room + slack + 1, 1);
    rt->decoded = calloc(size + 1, 1);
    return rt->encoded && rt->decoded;
}

static int round_trip_end(struct round_trip *rt, const char *why, const char **error_string) {
    free(rt->encoded);
    free(rt->decoded);
    if (why) {
        *error_string = why;
        return AWS_OP_ERR;
    }
    return AWS_OP_SUCCESS;
}

int huffman_test_transitive(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    const char **error_string) {

    struct round_trip rt;
    if (!round_trip_begin(&rt, coder, size, 0)) {
        return round_trip_end(&rt, "out of memory", error_string);
    }
    struct aws_byte_cursor plain = aws_byte_cursor_from_array(input, size);
    struct aws_byte_buf packed = aws_byte_buf_from_empty_array(rt.encoded, rt.encoded_room);
    if (aws_huffman_encode(&rt.encoder, &plain, &packed) != AWS_OP_SUCCESS) {
        return round_trip_end(&rt, "aws_huffman_encode failed", error_string);
    }
    if (plain.len != 0) {
        return round_trip_end(&rt, "not all data encoded", error_string);
    }
    if (encoded_size && packed.len != encoded_size) {
        return round_trip_end(&rt, "encoded length is incorrect", error_string);
    }
    struct aws_byte_cursor stream = aws_byte_cursor_from_buf(&packed);
    struct aws_byte_buf symbols = aws_byte_buf_from_empty_array(rt.decoded, size);
    if (aws_huffman_decode(&rt.decoder, &stream, &symbols) != AWS_OP_SUCCESS) {
        return round_trip_end(&rt, "aws_huffman_decode failed", error_string);
    }
    if (stream.len != 0) {
        return round_trip_end(&rt, "not all encoded data was decoded", error_string);
    }
    if (symbols.len != size) {
        return round_trip_end(&rt, "decode output size incorrect", error_string);
    }
    if (memcmp(input, rt.decoded, size) != 0) {
        return round_trip_end(&rt, "decoded data does not match input data", error_string);
    }
    return round_trip_end(&rt, NULL, error_string);
}

int huffman_test_transitive_chunked(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    size_t output_chunk_size,
    const char **error_string) {

    struct round_trip rt;
    if (!round_trip_begin(&rt, coder, size, output_chunk_size)) {
        return round_trip_end(&rt, "out of memory", error_string);
    }
    struct aws_byte_cursor plain = aws_byte_cursor_from_array(input, size);
    struct aws_byte_buf packed = aws_byte_buf_from_empty_array(rt.encoded, 0);
    int rc = AWS_OP_ERR;
    while (rc != AWS_OP_SUCCESS) {
        /* one more chunk of room per call (huffman_testing.c:103-118) */
        const size_t had = packed.len;
        packed.capacity += output_chunk_size;
        if (packed.capacity > rt.encoded_room + output_chunk_size) {
            return round_trip_end(&rt, "too much data encoded", error_string);
        }
        rc = aws_huffman_encode(&rt.encoder, &plain, &packed);
        if (packed.len == had) {
            return round_trip_end(&rt, "encode didn't write any data", error_string);
        }
        if (rc != AWS_OP_SUCCESS && aws_last_error() != AWS_ERROR_SHORT_BUFFER) {
            return round_trip_end(&rt, "encode returned wrong error code", error_string);
        }
    }
    if (packed.len > rt.encoded_room) {
        return round_trip_end(&rt, "too much data encoded", error_string);
    }
    if (encoded_size && packed.len != encoded_size) {
        return round_trip_end(&rt, "encoded length is incorrect", error_string);
    }
    struct aws_byte_cursor stream = aws_byte_cursor_from_buf(&packed);
    struct aws_byte_buf symbols = aws_byte_buf_from_empty_array(rt.decoded, 0);
    rc = AWS_OP_ERR;
    while (rc != AWS_OP_SUCCESS) {
        /* the same on the way back, never more room than the input was long (huffman_testing.c:137-156) */
        const size_t had = symbols.len;
        symbols.capacity += output_chunk_size;
        if (symbols.capacity > size) {
            symbols.capacity = size;
        }
        rc = aws_huffman_decode(&rt.decoder, &stream, &symbols);
        if (symbols.len == had) {
            return round_trip_end(&rt, "decode didn't write any data", error_string);
        }
        if (rc != AWS_OP_SUCCESS && aws_last_error() != AWS_ERROR_SHORT_BUFFER) {
            return round_trip_end(&rt, "decode returned wrong error code", error_string);
        }
    }
    if (symbols.len != size) {
        return round_trip_end(&rt, "decode output size incorrect", error_string);
    }
    if (memcmp(input, rt.decoded, size) != 0) {
        return round_trip_end(&rt, "decoded data does not match input data", error_string);
    }
    return round_trip_end(&rt, NULL, error_string);
}
