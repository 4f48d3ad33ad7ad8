#ifndef HUFFMAN_ORACLE_H
#define HUFFMAN_ORACLE_H
/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's Huffman hot path (awslabs/aws-c-compression
 * 0.3.3, source/huffman.c + the coder the generator emits), used only as the
 * checker for the HIP path: by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py.  Nothing under aws-c-compression_amd/ may
 * include, link or call anything in this directory.
 *
 * Parity pin: the reference itself cannot be built in this image -- it needs
 * aws-c-common headers (reference CMakeLists.txt:6) that are absent and may not
 * be stubbed -- so this restatement is pinned against
 *   (1) every known-answer vector in the reference's own tests
 *       (tests/huffman_test.c:20-37,178-194,408; tests/test_huffman_static_table.def),
 *       extracted as data by tests/golden/make_golden.py,
 *   (2) the decision tree of the reference's generated coder
 *       (tests/test_huffman_static.c:276-2381), extracted as leaf/invalid-prefix
 *       data by the same script and cross-checked against the output of the
 *       reference's own generator tool built from source (oracle/Makefile: _ref),
 *   (3) the digests, partial-call records and decoder tail states that SURVEY.md
 *       section 8c records from a run of the real reference.
 * tests/test_oracle_pins.py holds those checks.
 *
 * All entry points carry an oracle_ prefix so the oracle and the product library
 * can be loaded into one test process without symbol clashes.  Types come from
 * the public headers under include/.
 */

#include <aws/compression/huffman.h>

#ifdef __cplusplus
extern "C" {
#endif

/* last-error slot of the oracle (thread-local), independent of the product's */
int oracle_raise_error(int err);
int oracle_last_error(void);
void oracle_reset_error(void);
struct aws_allocator *oracle_default_allocator(void);

/* restates reference source/huffman.c:12-46 */
void oracle_huffman_encoder_init(struct aws_huffman_encoder *encoder, struct aws_huffman_symbol_coder *coder);
void oracle_huffman_encoder_reset(struct aws_huffman_encoder *encoder);
void oracle_huffman_decoder_init(struct aws_huffman_decoder *decoder, struct aws_huffman_symbol_coder *coder);
void oracle_huffman_decoder_reset(struct aws_huffman_decoder *decoder);
void oracle_huffman_decoder_allow_growth(struct aws_huffman_decoder *decoder, bool allow_growth);

/* restates reference source/huffman.c:107-129 */
size_t oracle_huffman_get_encoded_length(struct aws_huffman_encoder *encoder, struct aws_byte_cursor to_encode);

/* restates reference source/huffman.c:131-187 with its helper :59-105 */
int oracle_huffman_encode(
    struct aws_huffman_encoder *encoder,
    struct aws_byte_cursor *to_encode,
    struct aws_byte_buf *output);

/* restates reference source/huffman.c:213-286 with its helper :196-211 */
int oracle_huffman_decode(
    struct aws_huffman_decoder *decoder,
    struct aws_byte_cursor *to_decode,
    struct aws_byte_buf *output);

/*
 * A symbol coder built from 256 (pattern, num_bits) rows, behaving like the C
 * file the reference's generator emits for the same rows
 * (source/huffman_generator/generator.c:239-278 trie build, :154-214 decode
 * walk: one bit test per level, first leaf wins, a missing child returns 0).
 * Rows with num_bits == 0 have no code.  NULL on a malformed table.
 */
struct aws_huffman_symbol_coder *oracle_table_coder_new(const uint32_t patterns[256], const uint8_t num_bits[256]);
void oracle_table_coder_destroy(struct aws_huffman_symbol_coder *coder);

/* a coder whose encode answers come from one coder and whose decode answers from another (either may be NULL: a coder
 * with one callback); the two must outlive it */
struct aws_huffman_symbol_coder *oracle_split_coder_new(
    struct aws_huffman_symbol_coder *encode_from,
    struct aws_huffman_symbol_coder *decode_from);
void oracle_split_coder_destroy(struct aws_huffman_symbol_coder *coder);

/* restates reference source/huffman_testing.c:15-73 and :75-173 (0 = pass) */
int oracle_huffman_test_transitive(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    const char **error_string);
int oracle_huffman_test_transitive_chunked(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    size_t output_chunk_size,
    const char **error_string);

/*
 * bench.py's all-core CPU baseline: `threads` threads, each encoding and decoding
 * buffers of `buffer_bytes` symbols of its own (splitmix64, seed 2 + thread) through oracle_huffman_encode /
 * oracle_huffman_decode for `seconds`; returns how many buffers made the round trip in all (0 on a mismatch).
 * Plain pthreads, no shared state but the coder's read-only tables.
 */
uint64_t oracle_batch_round_trips(
    struct aws_huffman_symbol_coder *coder, uint32_t threads, double seconds, uint32_t buffer_bytes, double *elapsed_seconds);

/* splitmix64 byte stream of SURVEY.md section 8c: draw i (0-based) mixes seed + (i+1)*0x9E3779B97F4A7C15,
 * 8 bytes little-endian per draw. */
void oracle_splitmix64_fill(uint8_t *dst, size_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif

#endif /* HUFFMAN_ORACLE_H */
