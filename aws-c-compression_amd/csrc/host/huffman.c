/*
 * The eight entry points of the reference's include/aws/compression/huffman.h,
 * host-pointer flavour, on top of the HIP engine.
 *
 * Replaces reference source/huffman.c.  Same names, same argument meaning, same
 * return codes and raised errors, same post-call contents of the caller-owned
 * encoder/decoder structs, cursors and buffers.  What differs is where the work
 * happens: the caller's bytes are staged into device memory, one encode or decode
 * item runs through the kernels, and the outcome record is translated back into
 * the reference's streaming state.  No symbol is ever coded on the host.
 */
#include "engine.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ engine cache */

/*
 * The structs of the reference have no room for a handle, so the staged tables
 * are found again through the coder: one engine per (coder pointer, callbacks,
 * userdata, current device), created on first use.
 *
 * A pointer says nothing about what it points to: a coder may be freed and another one
 * allocated at the same address, or its table changed behind the same userdata.  So an
 * engine also remembers a fingerprint of the 256 answers of the encode callback it was
 * tabulated from, and every look-up asks the callback again (256 calls, well under a
 * microsecond -- a call through this library costs tens) and retires an engine whose
 * fingerprint no longer matches.  aws_huffman_amd_table_coder_destroy and
 * aws_compression_library_clean_up drop what they can (aws_huffman_amd_forget_coder / _all).
 */
enum { ENGINE_SLOTS = 16 };
static struct aws_huffman_amd_engine *s_engines[ENGINE_SLOTS];
static unsigned s_engine_clock[ENGINE_SLOTS];
static unsigned s_clock;
static pthread_mutex_t s_engine_lock = PTHREAD_MUTEX_INITIALIZER;

uint64_t aws_huffman_amd_coder_fingerprint(struct aws_huffman_symbol_coder *coder) {
    uint64_t h = 0xCBF29CE484222325ull; /* FNV-1a over (pattern masked to its length, length) of every symbol ... */
    if (coder->decode) {
        /* ... and over what the decode callback makes of 128 windows spread over the 32-bit range (a coder may have only
         * this callback, and two coders may share an encode table and differ here) */
        for (uint32_t k = 0; k < 128; ++k) {
            uint8_t sym = 0;
            const uint32_t window = k * 0x02040811u + (k << 25);
            const uint8_t n = coder->decode(window, &sym, coder->userdata);
            h = (h ^ (uint64_t)(n ? sym : 0)) * 0x100000001B3ull;
            h = (h ^ n) * 0x100000001B3ull;
        }
    }
    if (!coder->encode) {
        return h;
    }
    for (int sym = 0; sym < 256; ++sym) {
        const struct aws_huffman_code c = coder->encode((uint8_t)sym, coder->userdata);
        const uint32_t n = c.num_bits;
        const uint32_t pat = n == 0 ? 0 : (n >= 32 ? c.pattern : c.pattern & ((1u << n) - 1u));
        h = (h ^ pat) * 0x100000001B3ull;
        h = (h ^ n) * 0x100000001B3ull;
    }
    return h;
}

/* retires slot i if nobody is inside its engine; otherwise makes sure it is never handed out again
 * (its turn comes when a slot is needed and its last user has left) */
static void engine_retire_locked(int i) {
    struct aws_huffman_amd_engine *e = s_engines[i];
    if (!e) {
        return;
    }
    if (e->users == 0) {
        aws_huffman_amd_engine_destroy(e);
        s_engines[i] = NULL;
    } else {
        e->coder = NULL;
        s_engine_clock[i] = 0;
    }
}

void aws_huffman_amd_forget_coder(struct aws_huffman_symbol_coder *coder) {
    pthread_mutex_lock(&s_engine_lock);
    for (int i = 0; i < ENGINE_SLOTS; ++i) {
        if (s_engines[i] && s_engines[i]->coder == coder) {
            engine_retire_locked(i);
        }
    }
    pthread_mutex_unlock(&s_engine_lock);
}

void aws_huffman_amd_forget_all(void) {
    pthread_mutex_lock(&s_engine_lock);
    for (int i = 0; i < ENGINE_SLOTS; ++i) {
        engine_retire_locked(i);
    }
    pthread_mutex_unlock(&s_engine_lock);
}

/* the engine of a coder, held for one call: counted (so that it is not retired under the caller) and locked (its
 * staging buffers and one-item plans serve one call at a time; different threads may share a coder, as the
 * reference allows for its function-static generated coders) */
static struct aws_huffman_amd_engine *engine_acquire(struct aws_huffman_symbol_coder *coder) {
    int device = 0;
    if (hufs_device_count() <= 0 || hufs_get_device(&device)) {
        aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION); /* no GPU: fail loudly, there is no CPU path */
        return NULL;
    }
    const uint64_t fingerprint = aws_huffman_amd_coder_fingerprint(coder);
    pthread_mutex_lock(&s_engine_lock);
    struct aws_huffman_amd_engine *found = NULL;
    int slot = -1;
    for (int i = 0; i < ENGINE_SLOTS && !found; ++i) {
        struct aws_huffman_amd_engine *e = s_engines[i];
        if (e && e->coder == coder && e->key_encode == (void *)coder->encode &&
            e->key_decode == (void *)coder->decode && e->key_userdata == coder->userdata && e->device == device) {
            if (e->fingerprint == fingerprint) {
                found = e;
                s_engine_clock[i] = ++s_clock;
            } else {
                engine_retire_locked(i); /* same address, another table */
            }
        }
    }
    if (!found) {
        /* a free slot, else the least recently used engine nobody is inside of */
        for (int i = 0; i < ENGINE_SLOTS; ++i) {
            if (!s_engines[i]) {
                slot = i;
                break;
            }
            if (s_engines[i]->users == 0 && (slot < 0 || s_engine_clock[i] < s_engine_clock[slot])) {
                slot = i;
            }
        }
        struct aws_huffman_amd_engine *fresh = NULL;
        if (slot < 0) {
            aws_raise_error(AWS_ERROR_INVALID_STATE); /* ENGINE_SLOTS different coders in use at this very moment */
        } else if (aws_huffman_amd_engine_new(&fresh, coder, device) == AWS_OP_SUCCESS) {
            aws_huffman_amd_engine_destroy(s_engines[slot]);
            s_engines[slot] = fresh;
            s_engine_clock[slot] = ++s_clock;
            found = fresh;
        }
    }
    if (found) {
        ++found->users;
    }
    pthread_mutex_unlock(&s_engine_lock);
    if (found) {
        pthread_mutex_lock(&found->one_lock);
    }
    return found;
}

static void engine_release(struct aws_huffman_amd_engine *eng) {
    pthread_mutex_unlock(&eng->one_lock);
    pthread_mutex_lock(&s_engine_lock);
    if (--eng->users == 0 && eng->coder == NULL) {
        /* retired while somebody was inside (its coder destroyed or forgotten, the library cleaned up): the last one out
         * frees its tables, buffers and streams -- nobody can be handed this engine again */
        for (int i = 0; i < ENGINE_SLOTS; ++i) {
            if (s_engines[i] == eng) {
                s_engines[i] = NULL;
            }
        }
        pthread_mutex_unlock(&s_engine_lock);
        aws_huffman_amd_engine_destroy(eng);
        return;
    }
    pthread_mutex_unlock(&s_engine_lock);
}

/* ------------------------------------------------------------------ init / reset */

void aws_huffman_encoder_init(struct aws_huffman_encoder *encoder, struct aws_huffman_symbol_coder *coder) {
    AWS_ASSERT(encoder);
    AWS_ASSERT(coder);
    memset(encoder, 0, sizeof(*encoder));
    encoder->coder = coder;
    encoder->eos_padding = UINT8_MAX;
}

void aws_huffman_encoder_reset(struct aws_huffman_encoder *encoder) {
    AWS_ASSERT(encoder);
    memset(&encoder->overflow_bits, 0, sizeof(encoder->overflow_bits));
}

void aws_huffman_decoder_init(struct aws_huffman_decoder *decoder, struct aws_huffman_symbol_coder *coder) {
    AWS_ASSERT(decoder);
    AWS_ASSERT(coder);
    memset(decoder, 0, sizeof(*decoder));
    decoder->coder = coder;
}

void aws_huffman_decoder_reset(struct aws_huffman_decoder *decoder) {
    decoder->working_bits = 0;
    decoder->num_bits = 0;
}

void aws_huffman_decoder_allow_growth(struct aws_huffman_decoder *decoder, bool allow_growth) {
    decoder->allow_growth = allow_growth;
}

/* ------------------------------------------------------------------ encoded length */

size_t aws_huffman_get_encoded_length(struct aws_huffman_encoder *encoder, struct aws_byte_cursor to_encode) {
    AWS_PRECONDITION(encoder);
    AWS_PRECONDITION(aws_byte_cursor_is_valid(&to_encode));
    if (to_encode.len == 0) {
        return 0;
    }
    struct aws_huffman_amd_engine *eng = engine_acquire(encoder->coder);
    if (!eng) {
        return 0;
    }
    struct aws_huffman_amd_encode_item item;
    memset(&item, 0, sizeof(item));
    item.in_len = to_encode.len;
    item.out_capacity = UINT64_MAX; /* pending overflow bits are not part of the answer (huffman.c:107-129) */
    struct hufd_enc_result raw;
    const int failed = aws_huffman_amd_engine_encode_host(eng, &item, to_encode.ptr, NULL, true, &raw);
    engine_release(eng);
    if (failed) {
        return 0;
    }
    return (size_t)((raw.total_bits + 7) / 8);
}

/* ------------------------------------------------------------------ encode */

int aws_huffman_encode(
    struct aws_huffman_encoder *encoder,
    struct aws_byte_cursor *to_encode,
    struct aws_byte_buf *output) {

    AWS_ASSERT(encoder);
    AWS_ASSERT(encoder->coder);
    AWS_ASSERT(to_encode);
    AWS_ASSERT(output);

    const size_t room = output->capacity - output->len;
    const uint8_t carried = encoder->overflow_bits.num_bits;

    /* the three outcomes that touch neither symbols nor output (huffman.c:149-152,161-164) */
    if (carried && room == 0) {
        return aws_raise_error(AWS_ERROR_SHORT_BUFFER);
    }
    if (!carried && to_encode->len == 0) {
        return AWS_OP_SUCCESS;
    }
    if (!carried && room == 0) {
        return aws_raise_error(AWS_ERROR_SHORT_BUFFER);
    }

    struct aws_huffman_amd_engine *eng = engine_acquire(encoder->coder);
    if (!eng) {
        return AWS_OP_ERR;
    }

    struct aws_huffman_amd_encode_item item;
    memset(&item, 0, sizeof(item));
    item.in_len = to_encode->len;
    item.out_capacity = room;
    item.overflow_in = encoder->overflow_bits;
    item.eos_padding = encoder->eos_padding;

    struct hufd_enc_result raw;
    const int failed = aws_huffman_amd_engine_encode_host(eng, &item, to_encode->ptr, output->buffer + output->len, false, &raw);
    engine_release(eng);
    if (failed) {
        return AWS_OP_ERR;
    }
    struct aws_huffman_amd_encode_result res;
    aws_huffman_amd_encode_result_from_raw(&raw, &res);

    to_encode->ptr += res.consumed;
    to_encode->len -= res.consumed;
    output->len += res.produced;
    /* only num_bits is ever reset by the reference; the pattern matters while num_bits > 0 */
    encoder->overflow_bits.num_bits = res.overflow_out.num_bits;
    if (res.overflow_out.num_bits) {
        encoder->overflow_bits.pattern = res.overflow_out.pattern;
    }
    return res.rc == AWS_OP_SUCCESS ? AWS_OP_SUCCESS : aws_raise_error(res.error);
}

/* ------------------------------------------------------------------ decode */

static int decode_piece(struct aws_huffman_decoder *decoder, struct aws_byte_cursor *to_decode, struct aws_byte_buf *output);

/* A device item holds less than 4 GiB of encoded bytes (32-bit chunk arithmetic); the reference's length is a size_t
 * (source/huffman.c:228).  A longer input is taken in pieces, each a call of its own with the decoder's window carried
 * from one to the next -- exactly what a caller who streams the input in pieces gets from the reference, ending at
 * the first piece that does not run to its end (error, or no room left). */
static size_t s_decode_piece_bytes = (size_t)1 << 31;

void aws_huffman_amd_testing_set_decode_piece_bytes(size_t bytes) {
    s_decode_piece_bytes = bytes ? bytes : (size_t)1 << 31;
}

int aws_huffman_decode(
    struct aws_huffman_decoder *decoder,
    struct aws_byte_cursor *to_decode,
    struct aws_byte_buf *output) {

    AWS_ASSERT(decoder);
    AWS_ASSERT(decoder->coder);
    AWS_ASSERT(to_decode);
    AWS_ASSERT(output);
    while (to_decode->len > s_decode_piece_bytes) {
        struct aws_byte_cursor piece = {s_decode_piece_bytes, to_decode->ptr};
        const uint64_t bits_before = decoder->working_bits;
        const uint8_t held_before = decoder->num_bits;
        const size_t len_before = output->len;
        const int rc = decode_piece(decoder, &piece, output);
        const int error = rc ? aws_last_error() : 0; /* (read at once: nothing below may replace it) */
        const size_t taken = s_decode_piece_bytes - piece.len;
        to_decode->ptr += taken;
        to_decode->len -= taken;
        if (rc != AWS_OP_SUCCESS && error != AWS_ERROR_SHORT_BUFFER && error != AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL && taken == 0 &&
            decoder->working_bits == bits_before && decoder->num_bits == held_before && output->len == len_before) {
            /* not an outcome of the stream -- no engine, no device, no memory for the staging buffers -- and nothing was
             * decoded: the reference consumes no input on such an error, the cursor and the decoder stay as they were */
            return aws_raise_error(error);
        }
        if (rc != AWS_OP_SUCCESS || piece.len != 0) {
            /* stopped inside the piece: the reference would have topped its window up from ALL the input left before
             * the symbol it stopped at (source/huffman.c:196-211), not only from what the piece still held */
            while (decoder->num_bits < 32 && to_decode->len > 0) {
                decoder->working_bits |= (uint64_t)*to_decode->ptr << (56 - decoder->num_bits);
                decoder->num_bits = (uint8_t)(decoder->num_bits + 8);
                ++to_decode->ptr;
                --to_decode->len;
            }
            return rc ? aws_raise_error(error) : rc;
        }
    }
    return decode_piece(decoder, to_decode, output);
}

static int decode_piece(struct aws_huffman_decoder *decoder, struct aws_byte_cursor *to_decode, struct aws_byte_buf *output) {
    const uint32_t held = decoder->num_bits;                 /* read-ahead bits from earlier calls */
    const uint64_t stream_bits = held + (uint64_t)to_decode->len * 8; /* huffman.c:228 */

    if (stream_bits == 0) {
        /* nothing to look at: the reference returns success whatever the coder says (huffman.c:240-255) */
        return AWS_OP_SUCCESS;
    }
    struct aws_huffman_amd_engine *eng = engine_acquire(decoder->coder);
    if (!eng) {
        return AWS_OP_ERR;
    }
    if (!eng->can_decode) {
        engine_release(eng);
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }

    /* the held bits become up to five whole bytes in front of the new input; their first
     * (8 * bytes - held) bits are stale and skipped via first_bit */
    uint8_t carry[8];
    const uint32_t carry_bytes = (held + 7) / 8;
    const uint32_t first_bit = carry_bytes * 8 - held;
    {
        /* working_bits holds the held bits at its top; shift them so they END on a byte edge */
        const uint64_t packed = held ? decoder->working_bits >> (64 - held) : 0; /* right-aligned */
        for (uint32_t i = 0; i < carry_bytes; ++i) {
            carry[i] = (uint8_t)(packed >> (8 * (carry_bytes - 1 - i)));
        }
    }

    /* with growth the output is as large as it needs to be; the quirk of capacity 0 (nothing can
     * ever be stored, huffman.c:262,275 and SURVEY.md A.3) also counts every symbol */
    const size_t room = output->capacity - output->len;
    const bool grow = decoder->allow_growth;
    const uint64_t device_cap = grow ? UINT64_MAX : room;

    struct aws_huffman_amd_decode_result res;
    if (aws_huffman_amd_engine_decode_host(
            eng, carry, carry_bytes, first_bit, to_decode->ptr, to_decode->len, device_cap, &res)) {
        engine_release(eng);
        return AWS_OP_ERR;
    }

    /* storage for the symbols */
    uint64_t to_store = res.produced;
    int reserve_error = 0; /* != 0: growing failed; the call ends as the reference's would at that symbol */
    if (grow && to_store > room) {
        /* double until it fits, exactly as the per-symbol loop would have (huffman.c:260-264) */
        while (output->capacity - output->len < to_store) {
            if (output->capacity == 0) {
                to_store = 0; /* reserve_relative(0) adds nothing and the writes are dropped */
                break;
            }
            /* the reference doubles when len == capacity; replay the same sequence of sizes */
            const size_t before = output->capacity;
            const size_t keep_len = output->len;
            output->len = before; /* doubling is relative to a full buffer */
            const int rc = aws_byte_buf_reserve_relative(output, before);
            output->len = keep_len;
            if (rc) {
                /* The reference meets the failed reserve with the buffer full (len == capacity, huffman.c:257-264): the
                 * symbols before it are stored, the decoder has moved past them, the call returns the reserve's error.
                 * The same here: the stream once more, with room for exactly what fits -- where that symbol count ends
                 * is what the state below is made from.  (A failure path: the second decode is not a cost anyone times.) */
                reserve_error = aws_last_error();
                if (aws_huffman_amd_engine_decode_host(
                        eng, carry, carry_bytes, first_bit, to_decode->ptr, to_decode->len, before - keep_len, &res)) {
                    engine_release(eng);
                    return AWS_OP_ERR;
                }
                to_store = res.produced;
                break;
            }
        }
    }
    if (to_store) {
        if (aws_huffman_amd_engine_fetch_output(eng, output->buffer + output->len, to_store)) {
            engine_release(eng);
            return AWS_OP_ERR;
        }
        output->len += to_store;
    }
    engine_release(eng); /* the symbols are out of its staging buffer */

    /*
     * Streaming state after the call: with C stream bits consumed by
     * the emitted symbols, the refill loop (huffman.c:196-211) has pulled the fewest bytes that
     * leave at least 32 bits in the window, or everything.
     */
    const uint64_t consumed_bits = res.bits_consumed;
    uint64_t pulled = 0;
    if (consumed_bits + 32 > held) {
        pulled = (consumed_bits + 32 - held + 7) / 8;
    }
    if (pulled > to_decode->len) {
        pulled = to_decode->len;
    }
    const uint64_t left = held + pulled * 8 - consumed_bits; /* 0..39 */
    uint64_t window = 0;
    uint32_t filled = 0;
    if (consumed_bits < held) {
        window = decoder->working_bits << consumed_bits;
        filled = held - (uint32_t)consumed_bits;
        for (uint64_t i = 0; i < pulled; ++i) {
            window |= (uint64_t)to_decode->ptr[i] << (56 - filled);
            filled += 8;
        }
    } else {
        const uint64_t into_new = consumed_bits - held; /* bits of the new bytes already used */
        uint64_t i = into_new / 8;
        const uint32_t used = (uint32_t)(into_new % 8);
        if (i < pulled) {
            window = (uint64_t)(uint8_t)(to_decode->ptr[i] << used) << 56;
            filled = 8 - used;
            for (++i; i < pulled; ++i) {
                window |= (uint64_t)to_decode->ptr[i] << (56 - filled);
                filled += 8;
            }
        }
    }
    AWS_ASSERT(filled == left);
    (void)left;
    decoder->working_bits = window;
    decoder->num_bits = (uint8_t)filled;
    to_decode->ptr += pulled;
    to_decode->len -= pulled;

    if (reserve_error) {
        return aws_raise_error(reserve_error);
    }
    return res.rc == AWS_OP_SUCCESS ? AWS_OP_SUCCESS : aws_raise_error(res.error);
}
