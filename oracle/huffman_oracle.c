/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See huffman_oracle.h.
 *
 * Plain C99 restatement of the reference CPU algorithm: one callback per symbol,
 * one bounds-checked byte write per output byte, one byte pulled per refill
 * step -- the same work profile as reference source/huffman.c, so it can stand
 * in for it as the CPU baseline ("port") where the reference cannot travel.
 */
#include "huffman_oracle.h"

#include <stdlib.h>

/* ---------------------------------------------------------------- error slot */

static _Thread_local int tl_last_error;

int oracle_raise_error(int err) {
    tl_last_error = err;
    return AWS_OP_ERR;
}
int oracle_last_error(void) {
    return tl_last_error;
}
void oracle_reset_error(void) {
    tl_last_error = 0;
}

static void *heap_acquire(struct aws_allocator *a, size_t n) {
    (void)a;
    return malloc(n);
}
static void heap_release(struct aws_allocator *a, void *p) {
    (void)a;
    free(p);
}
static void *heap_realloc(struct aws_allocator *a, void *p, size_t o, size_t n) {
    (void)a;
    (void)o;
    return realloc(p, n);
}
static void *heap_calloc(struct aws_allocator *a, size_t k, size_t n) {
    (void)a;
    return calloc(k, n);
}
struct aws_allocator *oracle_default_allocator(void) {
    static struct aws_allocator heap = {heap_acquire, heap_release, heap_realloc, heap_calloc, NULL};
    return &heap;
}

/* ------------------------------------------------------- init / reset (huffman.c:12-46) */

void oracle_huffman_encoder_init(struct aws_huffman_encoder *encoder, struct aws_huffman_symbol_coder *coder) {
    /* huffman.c:17-19: whole struct zeroed, then coder and the all-ones padding default */
    memset(encoder, 0, sizeof(*encoder));
    encoder->coder = coder;
    encoder->eos_padding = 0xFF;
}

void oracle_huffman_encoder_reset(struct aws_huffman_encoder *encoder) {
    /* huffman.c:26: only the carried overflow goes away */
    memset(&encoder->overflow_bits, 0, sizeof(encoder->overflow_bits));
}

void oracle_huffman_decoder_init(struct aws_huffman_decoder *decoder, struct aws_huffman_symbol_coder *coder) {
    /* huffman.c:34-35 */
    memset(decoder, 0, sizeof(*decoder));
    decoder->coder = coder;
}

void oracle_huffman_decoder_reset(struct aws_huffman_decoder *decoder) {
    /* huffman.c:40-41: allow_growth and coder survive */
    decoder->working_bits = 0;
    decoder->num_bits = 0;
}

void oracle_huffman_decoder_allow_growth(struct aws_huffman_decoder *decoder, bool allow_growth) {
    decoder->allow_growth = allow_growth; /* huffman.c:45 */
}

/* ------------------------------------------------------- encoded length (huffman.c:107-129) */

size_t oracle_huffman_get_encoded_length(struct aws_huffman_encoder *encoder, struct aws_byte_cursor to_encode) {
    struct aws_huffman_symbol_coder *coder = encoder->coder;
    size_t total_bits = 0;
    uint8_t sym = 0;
    while (aws_byte_cursor_read_u8(&to_encode, &sym)) { /* huffman.c:114-119 */
        total_bits += coder->encode(sym, coder->userdata).num_bits;
    }
    return (total_bits + 7) / 8; /* huffman.c:121-128 round up */
}

/* ------------------------------------------------------- encode (huffman.c:59-105, 131-187) */

/* One output byte under construction: `room` counts its still-free low bits (8 = empty). */
struct byte_packer {
    struct aws_huffman_encoder *encoder;
    struct aws_byte_buf *sink;
    uint8_t partial;
    uint8_t room;
};

/*
 * Places the low `nbits` bits of `pattern`, most significant first (huffman.c:59-105).
 * A byte that fills is written at once; if that write fills the sink, whatever is
 * left of this code becomes encoder->overflow_bits and SHORT_BUFFER is raised
 * (huffman.c:88-100).  Exactly-fitting codes return success with num_bits = 0.
 */
static int packer_put(struct byte_packer *pk, uint32_t pattern, uint8_t nbits) {
    if (nbits == 0) {
        return oracle_raise_error(AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL); /* huffman.c:62-64 */
    }

    uint8_t pending = nbits; /* low bits of `pattern` not yet placed */
    while (pending) {
        const uint8_t take = pending < pk->room ? pending : pk->room;

        /* huffman.c:70-76: left-align the pending bits (drops anything above them),
         * then slide them down so the first one lands on the byte's first free bit */
        const uint32_t left_aligned = pattern << (32u - pending);
        pk->partial |= (uint8_t)(left_aligned >> (32u - pk->room));

        pending = (uint8_t)(pending - take);
        pk->room = (uint8_t)(pk->room - take);

        if (pk->room == 0) {
            aws_byte_buf_write_u8(pk->sink, pk->partial); /* huffman.c:83 */
            pk->partial = 0;
            pk->room = 8;

            if (pk->sink->len == pk->sink->capacity) { /* huffman.c:88 */
                pk->encoder->overflow_bits.num_bits = pending;
                if (pending) {
                    pk->encoder->overflow_bits.pattern = (pattern << (32u - pending)) >> (32u - pending);
                    return oracle_raise_error(AWS_ERROR_SHORT_BUFFER);
                }
            }
        }
    }
    return AWS_OP_SUCCESS;
}

int oracle_huffman_encode(
    struct aws_huffman_encoder *encoder,
    struct aws_byte_cursor *to_encode,
    struct aws_byte_buf *output) {

    struct aws_huffman_symbol_coder *coder = encoder->coder;
    struct byte_packer pk = {encoder, output, 0, 8};

    /* huffman.c:149-159: first the tail left over from a SHORT_BUFFER return */
    if (encoder->overflow_bits.num_bits) {
        if (output->len == output->capacity) {
            return oracle_raise_error(AWS_ERROR_SHORT_BUFFER);
        }
        const struct aws_huffman_code carried = encoder->overflow_bits;
        if (packer_put(&pk, carried.pattern, carried.num_bits)) {
            return AWS_OP_ERR;
        }
        encoder->overflow_bits.num_bits = 0;
    }

    /* huffman.c:161-173: a symbol is only pulled while the sink has a free byte */
    while (to_encode->len) {
        if (output->len == output->capacity) {
            return oracle_raise_error(AWS_ERROR_SHORT_BUFFER);
        }
        uint8_t sym = 0;
        aws_byte_cursor_read_u8(to_encode, &sym);
        const struct aws_huffman_code code = coder->encode(sym, coder->userdata);
        if (packer_put(&pk, code.pattern, code.num_bits)) {
            return AWS_OP_ERR;
        }
    }

    /* huffman.c:178-184: complete the last byte with the low bits of eos_padding */
    if (pk.room != 8) {
        packer_put(&pk, encoder->eos_padding, pk.room);
    }
    return AWS_OP_SUCCESS;
}

/* ------------------------------------------------------- decode (huffman.c:196-211, 213-286) */

/* huffman.c:196-211: pull bytes until 32 bits are buffered or the cursor is dry */
static void window_refill(struct aws_huffman_decoder *decoder, struct aws_byte_cursor *src) {
    uint8_t byte = 0;
    while (decoder->num_bits < 32 && aws_byte_cursor_read_u8(src, &byte)) {
        decoder->working_bits |= (uint64_t)byte << (56u - decoder->num_bits);
        decoder->num_bits = (uint8_t)(decoder->num_bits + 8);
    }
}

int oracle_huffman_decode(
    struct aws_huffman_decoder *decoder,
    struct aws_byte_cursor *to_decode,
    struct aws_byte_buf *output) {

    struct aws_huffman_symbol_coder *coder = decoder->coder;

    /* huffman.c:228: carried bits plus everything the cursor still holds */
    size_t undecoded_bits = decoder->num_bits + to_decode->len * 8;

    for (;;) {
        window_refill(decoder, to_decode);

        uint8_t sym = 0;
        const uint8_t used = coder->decode((uint32_t)(decoder->working_bits >> 32), &sym, coder->userdata);

        if (used == 0) { /* huffman.c:240-247 */
            if (undecoded_bits < 32) {
                return AWS_OP_SUCCESS;
            }
            return oracle_raise_error(AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL);
        }
        if (used > undecoded_bits) { /* huffman.c:248-255: the match leaned on zero fill */
            return AWS_OP_SUCCESS;
        }

        if (output->len == output->capacity) { /* huffman.c:257-268 */
            if (!decoder->allow_growth) {
                return oracle_raise_error(AWS_ERROR_SHORT_BUFFER);
            }
            if (aws_byte_buf_reserve_relative(output, output->capacity)) {
                return AWS_OP_ERR;
            }
        }

        /* huffman.c:270-275 */
        undecoded_bits -= used;
        decoder->working_bits <<= used;
        decoder->num_bits = (uint8_t)(decoder->num_bits - used);
        aws_byte_buf_write_u8(output, sym);

        if (undecoded_bits == 0) { /* huffman.c:278-280 */
            return AWS_OP_SUCCESS;
        }
    }
}

/* ------------------------------------------------------- table coder (generator.c) */

struct trie_node {
    int16_t child[2]; /* index into nodes[], -1 = absent */
    int16_t symbol;   /* >= 0 on a leaf */
    uint8_t depth;
};

struct table_coder {
    struct aws_huffman_symbol_coder vtable;
    struct aws_huffman_code rows[256];
    struct trie_node *nodes;
    int node_count;
};

static struct aws_huffman_code table_encode(uint8_t symbol, void *userdata) {
    /* generated encode_symbol is a bare table read (tests/test_huffman_static.c:269-273) */
    return ((struct table_coder *)userdata)->rows[symbol];
}

static uint8_t table_decode(uint32_t bits, uint8_t *symbol, void *userdata) {
    /* generator.c:154-214: test one bit per level from bit 31 down; a leaf child
     * answers at once, an absent child is "return 0; invalid node" */
    const struct table_coder *tc = (const struct table_coder *)userdata;
    int at = 0;
    for (uint32_t probe = 0x80000000u; probe; probe >>= 1) {
        const int next = tc->nodes[at].child[(bits & probe) ? 1 : 0];
        if (next < 0) {
            return 0;
        }
        if (tc->nodes[next].symbol >= 0) {
            *symbol = (uint8_t)tc->nodes[next].symbol;
            return tc->nodes[next].depth;
        }
        at = next;
    }
    return 0;
}

struct aws_huffman_symbol_coder *oracle_table_coder_new(const uint32_t patterns[256], const uint8_t num_bits[256]) {
    struct table_coder *tc = calloc(1, sizeof(*tc));
    if (!tc) {
        return NULL;
    }
    /* worst case one fresh node per code bit */
    size_t budget = 1;
    for (int s = 0; s < 256; ++s) {
        if (num_bits[s] > 32) {
            free(tc);
            return NULL;
        }
        budget += num_bits[s];
    }
    tc->nodes = malloc(budget * sizeof(*tc->nodes));
    if (!tc->nodes) {
        free(tc);
        return NULL;
    }
    tc->nodes[0].child[0] = tc->nodes[0].child[1] = -1;
    tc->nodes[0].symbol = -1;
    tc->nodes[0].depth = 0;
    tc->node_count = 1;

    /* generator.c:240-278: walk each code from its first bit, creating interior
     * nodes on the way and hanging the symbol off the last bit */
    for (int s = 0; s < 256; ++s) {
        tc->rows[s].pattern = patterns[s];
        tc->rows[s].num_bits = num_bits[s];
        if (num_bits[s] == 0) {
            continue;
        }
        int at = 0;
        for (int level = 1; level <= num_bits[s]; ++level) {
            const int bit = (int)((patterns[s] >> (num_bits[s] - level)) & 1u);
            int next = tc->nodes[at].child[bit];
            const bool last = level == num_bits[s];
            if (next >= 0 && (last || tc->nodes[next].symbol >= 0)) {
                /* duplicate code or a code that runs through another code's leaf:
                 * the generator asserts on these (generator.c:259) */
                oracle_table_coder_destroy(&tc->vtable);
                return NULL;
            }
            if (next < 0) {
                next = tc->node_count++;
                tc->nodes[next].child[0] = tc->nodes[next].child[1] = -1;
                tc->nodes[next].symbol = last ? (int16_t)s : (int16_t)-1;
                tc->nodes[next].depth = (uint8_t)level;
                tc->nodes[at].child[bit] = (int16_t)next;
            }
            at = next;
        }
    }

    tc->vtable.encode = table_encode;
    tc->vtable.decode = table_decode;
    tc->vtable.userdata = tc;
    return &tc->vtable;
}

void oracle_table_coder_destroy(struct aws_huffman_symbol_coder *coder) {
    if (coder) {
        struct table_coder *tc = (struct table_coder *)coder->userdata;
        free(tc->nodes);
        free(tc);
    }
}

/* ------------------------------------------------------- round-trip helpers (huffman_testing.c) */

#define FAIL_WITH(msg)                                                                                                 \
    do {                                                                                                               \
        *error_string = (msg);                                                                                         \
        verdict = AWS_OP_ERR;                                                                                          \
        goto done;                                                                                                     \
    } while (0)

int oracle_huffman_test_transitive(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    const char **error_string) {

    int verdict = AWS_OP_SUCCESS;
    struct aws_huffman_encoder enc;
    struct aws_huffman_decoder dec;
    oracle_huffman_encoder_init(&enc, coder);
    oracle_huffman_decoder_init(&dec, coder);

    /* huffman_testing.c:27-31: twice the input is taken to be room enough */
    const size_t mid_cap = size * 2;
    uint8_t *mid = calloc(mid_cap ? mid_cap : 1, 1);
    char *back = calloc(size ? size : 1, 1);

    struct aws_byte_cursor src = aws_byte_cursor_from_array(input, size);
    struct aws_byte_buf mid_buf = aws_byte_buf_from_empty_array(mid, mid_cap);
    struct aws_byte_buf back_buf = aws_byte_buf_from_empty_array(back, size);

    if (oracle_huffman_encode(&enc, &src, &mid_buf) != AWS_OP_SUCCESS) {
        FAIL_WITH("aws_huffman_encode failed");
    }
    if (src.len != 0) {
        FAIL_WITH("not all data encoded");
    }
    if (encoded_size && mid_buf.len != encoded_size) {
        FAIL_WITH("encoded length is incorrect");
    }

    struct aws_byte_cursor mid_cur = aws_byte_cursor_from_buf(&mid_buf);
    if (oracle_huffman_decode(&dec, &mid_cur, &back_buf) != AWS_OP_SUCCESS) {
        FAIL_WITH("aws_huffman_decode failed");
    }
    if (mid_cur.len != 0) {
        FAIL_WITH("not all encoded data was decoded");
    }
    if (back_buf.len != size) {
        FAIL_WITH("decode output size incorrect");
    }
    if (memcmp(input, back, size) != 0) {
        FAIL_WITH("decoded data does not match input data");
    }

done:
    free(mid);
    free(back);
    return verdict;
}

int oracle_huffman_test_transitive_chunked(
    struct aws_huffman_symbol_coder *coder,
    const char *input,
    size_t size,
    size_t encoded_size,
    size_t output_chunk_size,
    const char **error_string) {

    int verdict = AWS_OP_SUCCESS;
    struct aws_huffman_encoder enc;
    struct aws_huffman_decoder dec;
    oracle_huffman_encoder_init(&enc, coder);
    oracle_huffman_decoder_init(&dec, coder);

    /* round the storage up to a whole chunk so a generous last capacity stays in bounds */
    const size_t mid_limit = size * 2;
    uint8_t *mid = calloc(mid_limit + output_chunk_size + 1, 1);
    char *back = calloc(size ? size : 1, 1);

    struct aws_byte_cursor src = aws_byte_cursor_from_array(input, size);
    struct aws_byte_buf mid_buf = aws_byte_buf_from_empty_array(mid, 0);
    struct aws_byte_buf back_buf = aws_byte_buf_from_empty_array(back, 0);
    int rc;

    /* huffman_testing.c:103-118: widen the output by one chunk per call; every call
     * must make progress and may only fail with SHORT_BUFFER */
    do {
        const size_t before = mid_buf.len;
        mid_buf.capacity += output_chunk_size;
        if (mid_buf.capacity > mid_limit + output_chunk_size) {
            FAIL_WITH("too much data encoded");
        }
        rc = oracle_huffman_encode(&enc, &src, &mid_buf);
        if (mid_buf.len == before) {
            FAIL_WITH("encode didn't write any data");
        }
        if (rc != AWS_OP_SUCCESS && oracle_last_error() != AWS_ERROR_SHORT_BUFFER) {
            FAIL_WITH("encode returned wrong error code");
        }
    } while (rc != AWS_OP_SUCCESS);

    if (mid_buf.len > mid_limit) {
        FAIL_WITH("too much data encoded");
    }
    if (encoded_size && mid_buf.len != encoded_size) {
        FAIL_WITH("encoded length is incorrect");
    }

    struct aws_byte_cursor mid_cur = aws_byte_cursor_from_buf(&mid_buf);

    /* huffman_testing.c:137-156: same on the decode side, capacity clamped to `size` */
    do {
        const size_t before = back_buf.len;
        back_buf.capacity += output_chunk_size;
        if (back_buf.capacity > size) {
            back_buf.capacity = size;
        }
        rc = oracle_huffman_decode(&dec, &mid_cur, &back_buf);
        if (back_buf.len == before) {
            FAIL_WITH("decode didn't write any data");
        }
        if (rc != AWS_OP_SUCCESS && oracle_last_error() != AWS_ERROR_SHORT_BUFFER) {
            FAIL_WITH("decode returned wrong error code");
        }
    } while (rc != AWS_OP_SUCCESS);

    if (back_buf.len != size) {
        FAIL_WITH("decode output size incorrect");
    }
    if (memcmp(input, back, size) != 0) {
        FAIL_WITH("decoded data does not match input data");
    }

done:
    free(mid);
    free(back);
    return verdict;
}

/* ------------------------------------------------------- synthetic input */

void oracle_splitmix64_fill(uint8_t *dst, size_t len, uint64_t seed) {
    uint64_t state = seed;
    size_t at = 0;
    while (at < len) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        for (int b = 0; b < 8 && at < len; ++b, ++at) {
            dst[at] = (uint8_t)(z >> (8 * b));
        }
    }
}

/* ------------------------------------------------------------------ all-core baseline (bench.py) */

#include <pthread.h>
#include <time.h>

struct batch_worker {
    struct aws_huffman_symbol_coder *coder;
    uint32_t buffer_bytes;
    uint64_t seed;
    double deadline;
    uint64_t done;
    int bad;
};

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *batch_run(void *arg) {
    struct batch_worker *w = arg;
    const size_t n = w->buffer_bytes;
    uint8_t *in = malloc(n + 8), *enc = malloc(2 * n + 64), *back = malloc(n + 8);
    if (!in || !enc || !back) {
        w->bad = 1;
        return NULL;
    }
    oracle_splitmix64_fill(in, n, w->seed);
    while (now_seconds() < w->deadline) {
        struct aws_huffman_encoder e;
        struct aws_huffman_decoder d;
        oracle_huffman_encoder_init(&e, w->coder);
        oracle_huffman_decoder_init(&d, w->coder);
        struct aws_byte_cursor src = {n, in};
        struct aws_byte_buf out = {0, enc, 2 * n + 64, NULL};
        if (oracle_huffman_encode(&e, &src, &out) != AWS_OP_SUCCESS) {
            w->bad = 1;
            break;
        }
        struct aws_byte_cursor stream = {out.len, enc};
        struct aws_byte_buf plain = {0, back, n, NULL};
        if (oracle_huffman_decode(&d, &stream, &plain) != AWS_OP_SUCCESS || plain.len != n || memcmp(back, in, n) != 0) {
            w->bad = 1;
            break;
        }
        ++w->done;
    }
    free(in);
    free(enc);
    free(back);
    return NULL;
}

uint64_t oracle_batch_round_trips(
    struct aws_huffman_symbol_coder *coder, uint32_t threads, double seconds, uint32_t buffer_bytes, double *elapsed_seconds) {
    struct batch_worker *w = calloc(threads ? threads : 1, sizeof(*w));
    pthread_t *t = calloc(threads ? threads : 1, sizeof(*t));
    if (!w || !t || threads == 0) {
        free(w);
        free(t);
        return 0;
    }
    const double start = now_seconds();
    for (uint32_t k = 0; k < threads; ++k) {
        w[k].coder = coder;
        w[k].buffer_bytes = buffer_bytes;
        w[k].seed = 2 + k;
        w[k].deadline = start + seconds;
        if (pthread_create(&t[k], NULL, batch_run, &w[k]) != 0) {
            w[k].bad = 1;
            t[k] = 0;
        }
    }
    uint64_t total = 0;
    int bad = 0;
    for (uint32_t k = 0; k < threads; ++k) {
        if (t[k]) {
            pthread_join(t[k], NULL);
        }
        total += w[k].done;
        bad |= w[k].bad;
    }
    if (elapsed_seconds) {
        *elapsed_seconds = now_seconds() - start;
    }
    free(w);
    free(t);
    return bad ? 0 : total;
}

/* ------------------------------------------------------------------ test helper: a coder made of two
 *
 * encode answers from one coder, decode from another (reference huffman.h:53-57 allows any pair of callbacks): what
 * tests/parity_cases.py::one_sided_coders uses for "a decoder that knows more codes than the encoder". */
struct oracle_split_coder {
    struct aws_huffman_symbol_coder vtable;
    struct aws_huffman_symbol_coder *enc, *dec;
};

static struct aws_huffman_code split_encode(uint8_t symbol, void *userdata) {
    struct oracle_split_coder *sc = userdata;
    return sc->enc->encode(symbol, sc->enc->userdata);
}

static uint8_t split_decode(uint32_t bits, uint8_t *symbol, void *userdata) {
    struct oracle_split_coder *sc = userdata;
    return sc->dec->decode(bits, symbol, sc->dec->userdata);
}

struct aws_huffman_symbol_coder *oracle_split_coder_new(
    struct aws_huffman_symbol_coder *encode_from,
    struct aws_huffman_symbol_coder *decode_from) {
    struct oracle_split_coder *sc = calloc(1, sizeof(*sc));
    if (!sc) {
        return NULL;
    }
    sc->enc = encode_from;
    sc->dec = decode_from;
    sc->vtable.encode = encode_from ? split_encode : NULL;
    sc->vtable.decode = decode_from ? split_decode : NULL;
    sc->vtable.userdata = sc;
    return &sc->vtable;
}

void oracle_split_coder_destroy(struct aws_huffman_symbol_coder *coder) {
    free(coder); /* (the vtable is the object's first member) */
}

