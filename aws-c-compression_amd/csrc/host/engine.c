/*
 * Host layer (C99) of libaws-c-compression-amd: engines, plans and the
 * translation between the reference's call semantics and the device records.
 * All symbol work happens in the HIP kernels (the *_kernels.hip files of csrc/hip);
 * this file tabulates coders, lays out items, launches, and reads results back.
 * There is no CPU implementation of encode or decode here.
 */
#include "engine.h"

#include <stdlib.h>
#include <string.h>

/* the tests' switches (huffman_amd.h "testing hooks"): process-wide words, read with the plans and launches that follow */
static uint32_t s_testing_encode_road = 0, s_testing_decode_road = 0;
static uint64_t s_testing_per_byte[2] = {0, 0}; /* [0] encode, [1] decode */

void aws_huffman_amd_testing_set_encode_road(uint32_t flags) {
    __atomic_store_n(&s_testing_encode_road, flags, __ATOMIC_RELAXED);
}
void aws_huffman_amd_testing_set_decode_road(uint32_t flags) {
    __atomic_store_n(&s_testing_decode_road, flags, __ATOMIC_RELAXED);
}
void aws_huffman_amd_testing_set_items_per_byte(uint64_t encode, uint64_t decode) {
    __atomic_store_n(&s_testing_per_byte[0], encode, __ATOMIC_RELAXED);
    __atomic_store_n(&s_testing_per_byte[1], decode, __ATOMIC_RELAXED);
}
static uint32_t testing_encode_road(void) {
    return __atomic_load_n(&s_testing_encode_road, __ATOMIC_RELAXED);
}
static uint32_t testing_decode_road(void) {
    return __atomic_load_n(&s_testing_decode_road, __ATOMIC_RELAXED);
}


/* ------------------------------------------------------------------ small helpers */

/* behind a plan's launch: what a later owner of the plan's device arrays waits for.  On a caller's stream an event (the
 * stream may be gone by then); on the engine's own stream nothing -- that stream lives as long as the engine and is
 * waited for itself (an event record is a packet of its own on the queue: ~4 us behind every launch) */
static int plan_mark_done(void **event, bool *on_engine_stream, const struct aws_huffman_amd_engine *eng, void *stream) {
    if (!stream || stream == eng->stream) {
        *on_engine_stream = true;
        return 0;
    }
    if (!*event) {
        *event = hufs_event_create_untimed();
        if (!*event) {
            return 2;
        }
    }
    return hufs_event_record(*event, stream);
}

static int plan_wait_done(void *event, bool on_engine_stream, const struct aws_huffman_amd_engine *eng) {
    int err = on_engine_stream ? hufs_stream_sync(eng->stream) : 0;
    if (!err && event) {
        err = hufs_event_sync(event);
    }
    return err;
}

/*
 * The calling thread's current HIP device is the caller's business: every entry point that works on an engine's device
 * switches to it for its own duration and leaves the thread where it found it (a library call that moves a thread to
 * another GPU makes the caller's next kernel launch, or the next engine picked by the current device, land there).
 */
struct device_scope {
    int previous; /* < 0: nothing to go back to */
};
static struct device_scope device_scope_enter(int device) {
    struct device_scope sc = {-1};
    int now = -1;
    if (hufs_get_device(&now) == 0 && now != device) {
        sc.previous = now;
    }
    hufs_set_device(device);
    return sc;
}
static void device_scope_leave(struct device_scope *sc) {
    if (sc->previous >= 0) {
        hufs_set_device(sc->previous);
    }
}
#define HUFS_PASTE2(a, b) a##b
#define HUFS_PASTE(a, b) HUFS_PASTE2(a, b)
#define ON_DEVICE(device)                                                                                              \
    struct device_scope HUFS_PASTE(device_scope_, __LINE__) __attribute__((cleanup(device_scope_leave), unused)) =     \
        device_scope_enter(device)

static int raise_hip(int hip_error) {
    (void)hip_error;
    return aws_raise_error(AWS_ERROR_UNKNOWN);
}

static void *device_upload(const void *host, size_t size, void *stream, int *err) {
    void *d = hufs_malloc(size);
    if (!d) {
        *err = 2; /* hipErrorOutOfMemory */
        return NULL;
    }
    *err = hufs_copy_h2d(d, host, size, stream);
    if (!*err) {
        *err = hufs_stream_sync(stream);
    }
    if (*err) {
        hufs_free(d);
        return NULL;
    }
    return d;
}

/* ------------------------------------------------------------------ engine */


/*
 * Decode tables for a coder with codes of more than HUFD_DEC_MAX_LUT_BITS bits: a root table indexed by the
 * first HUFD_DEEP_ROOT_BITS bits of the window, whose entries either answer (symbol, length / no code) or
 * link to a table indexed by the next few bits, and so on down to bit 32.  Filled by asking the decode
 * callback (reference huffman.h:48) about every index with the bits below it all zero and all one: the same
 * answer, no longer than the bits looked at, is an entry; anything else depends on later bits and gets a
 * table of its own.  Every code of the encode table must then come back out of the tables.
 */
struct deep_builder {
    struct aws_huffman_symbol_coder *coder;
    uint32_t *tab;
    uint32_t used;
    bool ok;
};

static void deep_table_fill(struct deep_builder *b, uint32_t base, uint32_t prefix, uint32_t depth, uint32_t width) {
    const uint32_t low = 32 - depth - width; /* window bits below this table's index */
    const uint32_t fill = low ? (uint32_t)((1ull << low) - 1) : 0;
    for (uint32_t w = 0; w < (1u << width) && b->ok; ++w) {
        const uint32_t bits = prefix | (uint32_t)((uint64_t)w << low);
        uint8_t s0 = 0, s1 = 0;
        const uint8_t n0 = b->coder->decode(bits, &s0, b->coder->userdata);
        const uint8_t n1 = b->coder->decode(bits | fill, &s1, b->coder->userdata);
        if (n0 == n1 && n0 <= depth + width && (n0 == 0 || s0 == s1)) {
            b->tab[base + w] = n0 ? ((uint32_t)s0 << 8) | n0 : 0;
        } else if (low == 0) {
            b->ok = false; /* an answer longer than the window */
        } else {
            const uint32_t sub = low < HUFD_DEEP_SUB_BITS ? low : HUFD_DEEP_SUB_BITS;
            if (b->used + (1u << sub) > HUFD_DEEP_MAX_ENTRIES) {
                b->ok = false;
                return;
            }
            const uint32_t at = b->used;
            b->used += 1u << sub;
            b->tab[base + w] = HUFD_DEEP_LINK | (sub << 16) | at;
            deep_table_fill(b, at, bits, depth + width, sub);
        }
    }
}

static uint32_t deep_table_lookup(const uint32_t *tab, uint32_t window) {
    uint32_t e = tab[window >> (32 - HUFD_DEEP_ROOT_BITS)], used = HUFD_DEEP_ROOT_BITS;
    while (e & HUFD_DEEP_LINK) {
        const uint32_t width = (e >> 16) & 0xFFu;
        e = tab[(e & 0xFFFFu) + ((window << used) >> (32 - width))];
        used += width;
    }
    return e;
}

static int deep_table_build(struct aws_huffman_amd_engine *eng, struct aws_huffman_symbol_coder *coder) {
    struct deep_builder b = {coder, malloc(HUFD_DEEP_MAX_ENTRIES * sizeof(uint32_t)), 1u << HUFD_DEEP_ROOT_BITS, true};
    if (!b.tab) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    deep_table_fill(&b, 0, 0, 0, HUFD_DEEP_ROOT_BITS);
    /* the longest and the shortest code the decoder knows (an encoder may know fewer, or none) */
    for (uint32_t i = 0; i < b.used && b.ok; ++i) {
        const uint32_t e = b.tab[i];
        if (!(e & HUFD_DEEP_LINK) && (e & 0xFFu)) {
            const uint32_t len = e & 0xFFu;
            eng->tables.max_bits = len > eng->tables.max_bits ? len : eng->tables.max_bits;
            eng->tables.min_bits = len < eng->tables.min_bits ? len : eng->tables.min_bits;
        }
    }
    eng->tables.n_states = eng->tables.max_bits > 8 ? eng->tables.max_bits : 8;
    for (int sym = 0; sym < 256 && b.ok && coder->encode; ++sym) {
        const uint32_t len = (uint32_t)(eng->enc_table[sym] >> 32);
        if (len) {
            const uint32_t code = (uint32_t)eng->enc_table[sym] << (32 - len);
            const uint32_t fill = len < 32 ? (1u << (32 - len)) - 1u : 0u;
            const uint32_t want = ((uint32_t)sym << 8) | len;
            b.ok = deep_table_lookup(b.tab, code) == want && deep_table_lookup(b.tab, code | fill) == want;
        }
    }
    if (!b.ok) {
        free(b.tab); /* not a table-shaped decoder: decode stays unavailable */
        return AWS_OP_SUCCESS;
    }
    eng->deep_lut_host = b.tab;
    eng->tables.deep_entries = b.used;
    eng->tables.lut_bits = 0;
    eng->can_decode = true;
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_engine_new(
    struct aws_huffman_amd_engine **out_engine,
    struct aws_huffman_symbol_coder *coder,
    int device) {

    *out_engine = NULL;
    /* (a coder may lack one of its two callbacks: the reference's encoder only ever calls `encode`, its decoder only
     * `decode` -- source/huffman.c:60,235 -- and an engine is then good for that half) */
    if (!coder || (!coder->encode && !coder->decode)) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    if (hufs_device_count() <= 0) {
        /* fail loudly: there is no CPU path to fall back to */
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    if (device < 0) {
        if (hufs_get_device(&device)) {
            return aws_raise_error(AWS_ERROR_UNKNOWN);
        }
    }
    if (device >= hufs_device_count()) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    ON_DEVICE(device);
    {
        const int e = hufk_init(); /* per device, once; safe from several threads */
        if (e) {
            return raise_hip(e);
        }
    }

    struct aws_huffman_amd_engine *eng = calloc(1, sizeof(*eng));
    if (!eng) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    pthread_mutex_init(&eng->one_lock, NULL);
    pthread_mutex_init(&eng->spare_lock, NULL);
    eng->device = device;
    eng->coder = coder;
    eng->key_encode = (void *)coder->encode;
    eng->key_decode = (void *)coder->decode;
    eng->key_userdata = coder->userdata;
    eng->fingerprint = aws_huffman_amd_coder_fingerprint(coder);

    /* encode table: one callback per symbol (reference huffman.h:37) */
    uint32_t max_bits = 0, min_bits = 33;
    eng->can_encode = coder->encode != NULL;
    for (int sym = 0; sym < 256 && coder->encode; ++sym) {
        const struct aws_huffman_code code = coder->encode((uint8_t)sym, coder->userdata);
        if (code.num_bits > 32) {
            free(eng);
            return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
        }
        const uint32_t masked =
            code.num_bits == 0 ? 0 : (code.num_bits == 32 ? code.pattern : code.pattern & ((1u << code.num_bits) - 1u));
        eng->enc_table[sym] = ((uint64_t)code.num_bits << 32) | masked;
        if (code.num_bits) {
            max_bits = code.num_bits > max_bits ? code.num_bits : max_bits;
            min_bits = code.num_bits < min_bits ? code.num_bits : min_bits;
        }
    }
    eng->tables.all_coded = 1;
    for (int sym = 0; sym < 256; ++sym) {
        if ((eng->enc_table[sym] >> 32) == 0) {
            eng->tables.all_coded = 0;
        }
    }
    {
        const uint32_t road = testing_encode_road();
        /* one pass over the input where the coder allows it (hufk_encode_one_pass_applies); the tests can ask for count /
         * scan / pack, the road every other coder takes */
        eng->single_pass = !(road & AWS_HUFFMAN_AMD_TEST_ENCODE_THREE_KERNEL);
        /* ... or for a wave of the one-pass kernel that gives up, as if a look-back wait had run out: the three-kernel
         * road queued behind it on the stream does the launch over */
        eng->encode_fails = (road & AWS_HUFFMAN_AMD_TEST_ENCODE_ONE_PASS_FAILS) != 0;
    }
    /* decode tables from the DECODE callback alone (the reference's decoder knows nothing else: source/huffman.c:235-238;
     * reference huffman.h:48).  Every window of HUFD_DEC_MAX_LUT_BITS bits is asked twice, the bits behind it all zero
     * and all one: the same answer both times, no longer than the window, for every window = a coder whose codes have at
     * most that many bits, and its table (exhaustive: nothing is inferred).  Otherwise codes are longer: the linked
     * tables (deep_table_build), checked against the encode table where there is one. */
    eng->can_decode = false;
    uint32_t dec_max = 0, dec_min = 33;
    if (coder->decode) {
        const uint32_t wide = HUFD_DEC_MAX_LUT_BITS, windows = 1u << wide;
        const uint32_t fill = (1u << (32 - wide)) - 1u;
        uint16_t *probe = malloc(windows * sizeof(uint16_t));
        if (!probe) {
            free(eng);
            return aws_raise_error(AWS_ERROR_OOM);
        }
        bool tabular = true;
        for (uint32_t w = 0; w < windows && tabular; ++w) {
            uint8_t s0 = 0, s1 = 0;
            const uint32_t bits = w << (32 - wide);
            const uint8_t n0 = coder->decode(bits, &s0, coder->userdata);
            const uint8_t n1 = coder->decode(bits | fill, &s1, coder->userdata);
            if (n0 != n1 || n0 > wide || (n0 && s0 != s1)) {
                tabular = false;
            }
            probe[w] = n0 ? (uint16_t)(((uint16_t)s0 << 8) | n0) : 0;
            if (n0) {
                dec_max = n0 > dec_max ? n0 : dec_max;
                dec_min = n0 < dec_min ? n0 : dec_min;
            }
        }
        /* (a code of n bits must own all windows that start with it: a callback that answers from more bits than it
         * says is not a prefix code's) */
        for (uint32_t w = 0; w < windows && tabular; ++w) {
            const uint32_t n = probe[w] & 0xFFu;
            if (n && n < wide) {
                const uint32_t first = w >> (wide - n) << (wide - n);
                tabular = probe[first] == probe[w] && probe[first + (1u << (wide - n)) - 1u] == probe[w];
            }
        }
        if (tabular && dec_max >= 1) {
            uint16_t *lut = malloc(((size_t)1 << dec_max) * sizeof(uint16_t));
            if (!lut) {
                free(probe);
                free(eng);
                return aws_raise_error(AWS_ERROR_OOM);
            }
            for (uint32_t w = 0; w < (1u << dec_max); ++w) {
                lut[w] = probe[w << (wide - dec_max)];
            }
            eng->dec_lut_host = lut;
            eng->tables.lut_bits = dec_max;
            eng->can_decode = true;
        } else if (tabular) {
            dec_min = 33; /* a decoder without a single code: every stream stops at its first window, as the reference's would */
            uint16_t *lut = calloc(2, sizeof(uint16_t));
            if (!lut) {
                free(probe);
                free(eng);
                return aws_raise_error(AWS_ERROR_OOM);
            }
            eng->dec_lut_host = lut;
            eng->tables.lut_bits = 1;
            eng->can_decode = true;
        }
        free(probe);
    }
    /* two sets of code-length bounds: the encode table's own size images and stages and pick the packer; the decode
     * kernels' entry states, walk tables and certain steps follow what the DECODE table knows -- an encoder with codes of
     * 13..32 bits beside a decoder whose table has 12 or fewer must not give the chunk kernels more entry states than
     * they have registers for (they hold HUFD_DEC_MAX_LUT_BITS, or 16 on the long way) */
    eng->tables.enc_max_bits = max_bits;
    eng->tables.enc_min_bits = min_bits > 32 ? 1 : min_bits;
    if (eng->can_decode && dec_max) {
        max_bits = dec_max;
        min_bits = dec_min;
    }
    eng->tables.max_bits = max_bits;
    eng->tables.min_bits = min_bits; /* (33: no code known yet) */
    eng->tables.n_states = max_bits > 8 ? max_bits : 8;
    /* every code the DECODE table knows has one length: symbol k starts at bit k * length (dec_fixed_*) */
    eng->tables.fixed_bits = eng->can_decode && dec_max && dec_min == dec_max && dec_max <= HUFD_DEC_MAX_LUT_BITS ? dec_max : 0;
    eng->tables.fixed_complete = 0;
    if (eng->tables.fixed_bits && eng->dec_lut_host) {
        /* ... and every window is one (256 codes of 8 bits): no stream has a symbol without a code to look for */
        eng->tables.fixed_complete = 1;
        for (uint32_t i = 0; i < (1u << eng->tables.lut_bits); ++i) {
            if ((eng->dec_lut_host[i] & 0xFFu) == 0) {
                eng->tables.fixed_complete = 0;
                break;
            }
        }
    }

    if (!eng->can_decode && coder->decode) {
        if (deep_table_build(eng, coder)) {
            free(eng);
            return AWS_OP_ERR;
        }
    }
    if (eng->tables.min_bits > 32) {
        eng->tables.min_bits = 1;
    }

    int err = hufs_stream_create(&eng->stream);
    if (!err && hufs_stream_create(&eng->side_stream) == 0) {
        eng->fork_event = hufs_event_create_untimed();
        eng->join_event = hufs_event_create_untimed();
        if (!eng->fork_event || !eng->join_event) {
            /* (no overlap then: the kernels run one after the other on the one stream) */
            hufs_event_destroy(eng->fork_event);
            hufs_event_destroy(eng->join_event);
            hufs_stream_destroy(eng->side_stream);
            eng->fork_event = eng->join_event = eng->side_stream = NULL;
        }
    } else {
        eng->side_stream = NULL;
    }
    if (!err && eng->deep_lut_host) {
        eng->d_deep_lut =
            device_upload(eng->deep_lut_host, (size_t)eng->tables.deep_entries * sizeof(uint32_t), eng->stream, &err);
    }
    if (!err) {
        eng->d_enc_table = device_upload(eng->enc_table, sizeof(eng->enc_table), eng->stream, &err);
    }
    if (!err && eng->dec_lut_host) {
        eng->d_dec_lut =
            device_upload(eng->dec_lut_host, (size_t)sizeof(uint16_t) << eng->tables.lut_bits, eng->stream, &err);
    }
    if (err) {
        aws_huffman_amd_engine_destroy(eng);
        return raise_hip(err);
    }
    eng->tables.enc_table = eng->d_enc_table;
    eng->tables.dec_lut = eng->d_dec_lut;
    eng->tables.deep_lut = eng->d_deep_lut;
    *out_engine = eng;
    return AWS_OP_SUCCESS;
}

void aws_huffman_amd_engine_destroy(struct aws_huffman_amd_engine *eng) {
    if (!eng) {
        return;
    }
    ON_DEVICE(eng->device);
    if (eng->one_enc) {
        aws_huffman_amd_encode_plan_destroy(eng->one_enc);
    }
    if (eng->one_dec) {
        aws_huffman_amd_decode_plan_destroy(eng->one_dec);
    }
    /* (the plans destroyed last, kept for the next *_plan_new: gone for good now) */
    {
        struct aws_huffman_amd_encode_plan *se = eng->spare_enc;
        struct aws_huffman_amd_decode_plan *sd = eng->spare_dec;
        eng->spare_enc = NULL;
        eng->spare_dec = NULL;
        eng->retiring = true;
        if (se) {
            aws_huffman_amd_encode_plan_destroy(se);
        }
        if (sd) {
            aws_huffman_amd_decode_plan_destroy(sd);
        }
    }
    hufs_free(eng->one_in);
    hufs_free(eng->one_out);
    hufs_free(eng->mini_dev);
    hufs_host_free(eng->mini_host);
    hufs_free(eng->d_enc_table);
    hufs_free(eng->d_dec_lut);
    hufs_free(eng->d_deep_lut);
    hufs_event_destroy(eng->fork_event);
    hufs_event_destroy(eng->join_event);
    if (eng->side_stream) {
        hufs_stream_destroy(eng->side_stream);
    }
    hufs_stream_destroy(eng->stream);
    free(eng->dec_lut_host);
    free(eng->deep_lut_host);
    pthread_mutex_destroy(&eng->one_lock);
    pthread_mutex_destroy(&eng->spare_lock);
    free(eng);
}

int aws_huffman_amd_engine_device(const struct aws_huffman_amd_engine *eng) {
    return eng->device;
}

int aws_huffman_amd_current_device(void) {
    int device = -1;
    return hufs_get_device(&device) ? -1 : device;
}

uint32_t aws_huffman_amd_engine_max_code_bits(const struct aws_huffman_amd_engine *eng) {
    return eng->tables.max_bits > eng->tables.enc_max_bits ? eng->tables.max_bits : eng->tables.enc_max_bits;
}

bool aws_huffman_amd_engine_can_decode(const struct aws_huffman_amd_engine *eng) {
    return eng->can_decode;
}

void *aws_huffman_amd_engine_stream(struct aws_huffman_amd_engine *eng) {
    return eng->stream;
}

/* ------------------------------------------------------------------ encode plans */

/* A plan's device arrays are cuts of ONE allocation (a new plan paid some twenty-five hipMallocs: 1 .. 15 ms for
 * BASELINE configs[3], where filling it takes 0.5 ms): every array starts at a multiple of 256 bytes. */
static size_t arena_cut(size_t *total, size_t bytes) {
    const size_t at = (*total + 255u) & ~(size_t)255u;
    *total = at + bytes;
    return at;
}

#define ENC_PLAN_ARRAYS(X, ci, cs, cl, ct)                                                                                 \
    X(d_items, (ci) * sizeof(struct hufd_enc_item))                                                                    \
    X(d_segs, (cs) * sizeof(struct hufd_enc_seg))                                                                      \
    X(d_large, (cl) * sizeof(uint32_t))                                                                                \
    X(d_tiny, (ct) * sizeof(uint32_t))                                                                                 \
    X(d_solo, (ci) * sizeof(uint32_t))                                                                                 \
    X(d_seg_bits, (cs) * sizeof(uint32_t))                                                                             \
    X(d_wave_bits, (cs) * 4 * sizeof(uint32_t))                                                                        \
    X(d_seg_unk, (cs) * sizeof(uint32_t))                                                                              \
    X(d_seg_bitoff, (cs) * sizeof(uint64_t))                                                                           \
    /* the scan lists up to two segments an item, the wave packer any segment it leaves */                              \
    X(d_careful, (2 * (ci) + (cs) + 4) * sizeof(uint32_t))                                                             \
    X(d_zero, hufk_encode_zero_bytes((uint32_t)(cs), (uint32_t)(ci)))                                                  \
    X(d_unk_seen, (cs))                                                                                                \
    X(d_item_total, (ci) * sizeof(uint64_t))                                                                           \
    X(d_states, (ci) * sizeof(struct hufd_enc_item_state))                                                             \
    X(d_results, (ci) * sizeof(struct hufd_enc_result))

static void enc_plan_release_device(struct aws_huffman_amd_encode_plan *p) {
    hufs_free(p->d_arena);
    p->d_arena = NULL;
#define ENC_FORGET(name, bytes) p->name = NULL;
    ENC_PLAN_ARRAYS(ENC_FORGET, 0, 0, 0, 0)
#undef ENC_FORGET
    p->cap_items = p->cap_segs = p->cap_large = p->cap_tiny = 0;
}

/* short items with anything to write (symbols or carried bits) are one thread's work, without segments */
/* fewer items than this: the host's loop over them costs less than an allocation and a launch */
#define PLAN_ON_DEVICE_MIN_ITEMS 4096u

/* items per byte of the longest item from which a length class goes to a thread per item (HUFD_*_TINY_PER_BYTE; the tests
 * set other numbers: aws_huffman_amd_testing_set_items_per_byte) */
static uint64_t tiny_per_byte(bool decode) {
    const uint64_t asked = __atomic_load_n(&s_testing_per_byte[decode], __ATOMIC_RELAXED);
    return asked ? asked : (decode ? HUFD_DEC_TINY_PER_BYTE : HUFD_ENC_TINY_PER_BYTE);
}

/* ... and for an encoder that packs ragged tiles in one pass (a wave a tile: ~1.6 ns an item whatever its length, where a lone
 * thread needs ~0.7 us a symbol of the LONGEST item): the thread road pays from ~430 items per byte on -- measured,
 * profiles/tools/mid_items.py: 447 K items of 300 B 1.02 ms by threads against 2.82 by tiles, but BASELINE configs[3]'s
 * 16 384 resume items of ~320 symbols 251 us by threads where tiles take a tenth */
static uint64_t enc_tiny_per_byte(const struct aws_huffman_amd_engine *eng) {
    const uint64_t asked = __atomic_load_n(&s_testing_per_byte[0], __ATOMIC_RELAXED);
    if (asked || !aws_huffman_amd_engine_encodes_in_one_pass(eng)) {
        return tiny_per_byte(false);
    }
    return HUFD_ENC_TINY_PER_BYTE_ONE_PASS;
}

struct item_stats { /* of a plan's items, from the pass that finds the thread-per-item limit */
    uint64_t shortest, longest;
    uint32_t worst_bits; /* largest first_bit (decode) / overflow_in.num_bits (encode) */
    uint64_t largest_out_cap; /* (encode) */
};

/* the caller's records go to the device as they are when every item of a plan is one thread's work: the kernels that
 * read them there (hufk_*_plan_tiny_items) see these layouts */
_Static_assert(sizeof(struct aws_huffman_amd_decode_item) == sizeof(struct hufd_raw_dec_item), "decode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_decode_item, first_bit) == offsetof(struct hufd_raw_dec_item, first_bit), "decode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_decode_item, out_offset) == offsetof(struct hufd_raw_dec_item, out_offset), "decode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_decode_item, out_capacity) == offsetof(struct hufd_raw_dec_item, out_capacity), "decode item layout");
_Static_assert(sizeof(struct aws_huffman_amd_encode_item) == sizeof(struct hufd_raw_enc_item), "encode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_encode_item, out_capacity) == offsetof(struct hufd_raw_enc_item, out_capacity), "encode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_encode_item, overflow_in) == offsetof(struct hufd_raw_enc_item, ovf_pattern), "encode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_encode_item, overflow_in) + offsetof(struct aws_huffman_code, num_bits) ==
                   offsetof(struct hufd_raw_enc_item, ovf_bits), "encode item layout");
_Static_assert(offsetof(struct aws_huffman_amd_encode_item, eos_padding) == offsetof(struct hufd_raw_enc_item, eos_padding), "encode item layout");

static bool enc_item_is_tiny(const struct aws_huffman_amd_encode_item *it, uint64_t limit) {
    return it->in_len <= limit && (it->in_len > 0 || it->overflow_in.num_bits);
}

/* the longest item a lone thread takes in this plan: see HUFD_ENC_TINY_PER_BYTE */
static uint64_t enc_tiny_limit(
    const struct aws_huffman_amd_engine *eng, const struct aws_huffman_amd_encode_item *items, size_t n_items, struct item_stats *st) {
    /* (the one-pass kernel packs ragged tiles at full speed: measured, a wave beats a thread from about 1 KiB an item) */
    const uint64_t classes[2] = {
        aws_huffman_amd_engine_encodes_in_one_pass(eng) ? HUFD_ENC_TINY_WAVE_BYTES : HUFD_TINY_MANY_BYTES, HUFD_ENC_TINY_BYTES};
    uint64_t count[2] = {0, 0}, longest[2] = {0, 0}; /* (one pass over the items for both classes) */
    st->shortest = UINT64_MAX;
    st->longest = 0;
    st->worst_bits = 0;
    st->largest_out_cap = 0;
    for (size_t i = 0; i < n_items; ++i) {
        const uint64_t len = items[i].in_len;
        st->shortest = len < st->shortest ? len : st->shortest;
        st->longest = len > st->longest ? len : st->longest;
        st->worst_bits = items[i].overflow_in.num_bits > st->worst_bits ? items[i].overflow_in.num_bits : st->worst_bits;
        st->largest_out_cap = items[i].out_capacity > st->largest_out_cap ? items[i].out_capacity : st->largest_out_cap;
        for (int c = 0; c < 2; ++c) {
            if (len <= classes[c]) {
                ++count[c];
                longest[c] = len > longest[c] ? len : longest[c];
            }
        }
    }
    for (int c = 0; c < 2; ++c) {
        if (longest[c] > HUFD_TINY_FEW_BYTES && count[c] >= enc_tiny_per_byte(eng) * longest[c]) {
            return classes[c];
        }
    }
    return HUFD_TINY_FEW_BYTES;
}

/* the longest item that is one wave's work without segments (enc_onepass<.., SOLO>: the one-pass packer, waiting for nobody) */
static uint64_t enc_solo_limit(const struct aws_huffman_amd_engine *eng, size_t n_items) {
    if (!aws_huffman_amd_engine_encodes_in_one_pass(eng)) {
        return 0;
    }
    return n_items >= HUFD_ENC_SOLO_MANY_ITEMS ? HUFD_ENC_SOLO_MANY_BYTES : HUFD_ENC_SOLO_BYTES;
}

static bool enc_item_is_solo(const struct aws_huffman_amd_encode_item *it, uint64_t tiny_limit, uint64_t solo_limit) {
    return !enc_item_is_tiny(it, tiny_limit) && it->in_len > 0 && it->in_len <= solo_limit;
}

static uint64_t enc_item_segments(const struct aws_huffman_amd_encode_item *it, uint64_t tiny_limit, uint64_t solo_limit) {
    if (enc_item_is_tiny(it, tiny_limit) || enc_item_is_solo(it, tiny_limit, solo_limit)) {
        return 0; /* one thread encodes it (enc_tiny), or one wave */
    }
    return (it->in_len + HUFD_ENC_SEG_BYTES - 1) / HUFD_ENC_SEG_BYTES;
}

/* (Re)fills a plan from host items, growing its device arrays when needed. */
/* the plan's device arrays for this many items, segments, large and thread-per-item items (grown, never shrunk); 0 or a HIP error */
static int enc_plan_reserve(struct aws_huffman_amd_encode_plan *p, size_t n_items, size_t n_segs, size_t n_large, size_t n_tiny) {
    if (n_items > p->cap_items || n_segs > p->cap_segs || n_large > p->cap_large || n_tiny > p->cap_tiny) {
        enc_plan_release_device(p);
        const size_t ci = n_items ? n_items : 1, cs = n_segs ? n_segs : 1, cl = n_large ? n_large : 1, ct = n_tiny ? n_tiny : 1;
        size_t total = 0;
#define ENC_SIZE(name, bytes) (void)arena_cut(&total, (bytes));
        ENC_PLAN_ARRAYS(ENC_SIZE, ci, cs, cl, ct)
#undef ENC_SIZE
        p->d_arena = hufs_malloc(total);
        if (!p->d_arena) {
            return 2;
        }
        total = 0;
#define ENC_PLACE(name, bytes) p->name = (void *)((uint8_t *)p->d_arena + arena_cut(&total, (bytes)));
        ENC_PLAN_ARRAYS(ENC_PLACE, ci, cs, cl, ct)
#undef ENC_PLACE
        p->cap_items = ci;
        p->cap_segs = cs;
        p->cap_large = cl;
        p->cap_tiny = ct;
        /* the one-pass encoder's look-back block starts out clear; from then on every launch clears it behind itself */
        int e = hufs_memset(p->d_zero, 0, (size_t)hufk_encode_zero_bytes((uint32_t)cs, (uint32_t)ci), p->engine->stream);
        if (!e) {
            e = hufs_stream_sync(p->engine->stream);
        }
        if (e) {
            return e;
        }
        p->zero_is_clear = true;
    }
    return 0;
}

static int enc_plan_fill(
    struct aws_huffman_amd_encode_plan *p,
    const struct aws_huffman_amd_encode_item *items,
    size_t n_items) {

    struct aws_huffman_amd_engine *eng = p->engine;
    if (!eng->can_encode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION); /* a coder without an encode callback: good for decoding only */
    }
    struct item_stats stats;
    const uint64_t tiny_limit = enc_tiny_limit(eng, items, n_items, &stats);
    /* a failed refill must not leave counts of the fill before behind (the device arrays may be gone or too small) */
    p->n_items = p->n_segs = p->n_large = p->n_tiny = p->n_solo = 0;
    p->launched = false; /* (the records of a launch of other items say nothing about these) */
    memset(&p->stats, 0, sizeof(p->stats));
    p->largest_out_cap = stats.largest_out_cap;
    p->most_overflow_bits = stats.worst_bits;
    if (n_items >= PLAN_ON_DEVICE_MIN_ITEMS && n_items < 0xFFFFFFFFull && stats.shortest >= 1 && stats.longest <= tiny_limit &&
        stats.worst_bits <= 32) {
        /* every item is one thread's work (enc_item_is_tiny): no segments, no lists to make -- the caller's records go to the
         * device as they are and become the kernels' there (hufk_encode_plan_tiny_items) */
        ON_DEVICE(eng->device);
        int e = enc_plan_reserve(p, n_items, 0, 0, n_items);
        void *d_raw = e ? NULL : hufs_malloc(n_items * sizeof(*items));
        if (!e && !d_raw) {
            e = 2;
        }
        if (!e) {
            e = hufs_copy_h2d(d_raw, items, n_items * sizeof(*items), eng->stream);
        }
        if (!e) {
            e = hufk_encode_plan_tiny_items(d_raw, (uint32_t)n_items, p->d_items, p->d_tiny, eng->stream);
        }
        if (!e) {
            e = hufs_stream_sync(eng->stream);
        }
        hufs_free(d_raw);
        if (e) {
            return raise_hip(e);
        }
        p->n_items = (uint32_t)n_items;
        p->n_tiny = (uint32_t)n_items;
        p->stats.items = p->stats.by_thread = n_items;
        p->stats.thread_limit = tiny_limit;
        return AWS_OP_SUCCESS;
    }
    const uint64_t solo_limit = enc_solo_limit(eng, n_items);
    uint64_t n_segs = 0, n_large = 0, n_tiny = 0, n_cut = 0, n_solo = 0;
    for (size_t i = 0; i < n_items; ++i) {
        if (items[i].overflow_in.num_bits > 32) {
            return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
        }
        const uint64_t segs = enc_item_segments(&items[i], tiny_limit, solo_limit);
        n_segs += segs;
        n_large += segs > HUFD_SCAN_SMALL_MAX;
        n_tiny += enc_item_is_tiny(&items[i], tiny_limit);
        n_solo += enc_item_is_solo(&items[i], tiny_limit, solo_limit);
        n_cut += segs != 0;
    }
    if (n_segs >= 0xFFFFFFFFull || n_items >= 0xFFFFFFFFull || n_solo >= (1ull << 30)) { /* (tile numbers are 32 bits: four tiles a wave's item) */
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }

    struct hufd_enc_item *h_items = malloc((n_items ? n_items : 1) * sizeof(*h_items));
    struct hufd_enc_seg *h_segs = malloc((n_segs ? n_segs : 1) * sizeof(*h_segs));
    uint32_t *h_large = malloc((n_large ? n_large : 1) * sizeof(uint32_t));
    uint32_t *h_tiny = malloc((n_tiny ? n_tiny : 1) * sizeof(uint32_t));
    uint32_t *h_solo = malloc((n_solo ? n_solo : 1) * sizeof(uint32_t));
    if (!h_items || !h_segs || !h_large || !h_tiny || !h_solo) {
        free(h_items);
        free(h_segs);
        free(h_large);
        free(h_tiny);
        free(h_solo);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    uint32_t seg = 0, large = 0, tiny = 0, solo = 0;
    for (size_t i = 0; i < n_items; ++i) {
        const struct aws_huffman_amd_encode_item *src = &items[i];
        struct hufd_enc_item *dst = &h_items[i];
        const uint32_t segs = (uint32_t)enc_item_segments(src, tiny_limit, solo_limit);
        const uint32_t ob = src->overflow_in.num_bits;
        dst->in_off = src->in_offset;
        dst->in_len = src->in_len;
        dst->out_off = src->out_offset;
        dst->out_cap = src->out_capacity;
        dst->ovf_bits = ob;
        dst->ovf_pattern = ob == 0 ? 0 : (ob == 32 ? src->overflow_in.pattern : src->overflow_in.pattern & ((1u << ob) - 1u));
        dst->eos_padding = src->eos_padding;
        dst->first_seg = seg;
        dst->n_segs = segs;
        dst->tiny = enc_item_is_tiny(src, tiny_limit) ? 1u : (enc_item_is_solo(src, tiny_limit, solo_limit) ? 2u : 0u);
        if (dst->tiny == 1) {
            h_tiny[tiny++] = (uint32_t)i;
        } else if (dst->tiny == 2) {
            h_solo[solo++] = (uint32_t)i;
        }
        for (uint32_t k = 0; k < segs; ++k) {
            struct hufd_enc_seg *sd = &h_segs[seg++];
            const uint64_t off = (uint64_t)k * HUFD_ENC_SEG_BYTES;
            const uint64_t left = src->in_len > off ? src->in_len - off : 0;
            const uint64_t after = left > HUFD_ENC_SEG_BYTES ? left - HUFD_ENC_SEG_BYTES : 0;
            sd->in_off = src->in_offset + off;
            sd->len = (uint32_t)(left < HUFD_ENC_SEG_BYTES ? left : HUFD_ENC_SEG_BYTES);
            sd->item = (uint32_t)i;
            sd->index = k;
            sd->flags = (k == 0 ? 1u : 0u) | (k + 1 == segs ? 2u : 0u);
            sd->next_len = (uint32_t)(after < HUFD_ENC_SEG_BYTES ? after : HUFD_ENC_SEG_BYTES);
            sd->reserved = 0;
        }
        if (segs > HUFD_SCAN_SMALL_MAX) {
            h_large[large++] = (uint32_t)i;
        }
    }

    int err = 0;
    ON_DEVICE(eng->device);
    err = enc_plan_reserve(p, n_items, n_segs, n_large, n_tiny);
    if (!err) {
        err = hufs_copy_h2d(p->d_items, h_items, n_items * sizeof(*h_items), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_segs, h_segs, n_segs * sizeof(*h_segs), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_large, h_large, n_large * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_tiny, h_tiny, n_tiny * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_solo, h_solo, n_solo * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    free(h_items);
    free(h_segs);
    free(h_large);
    free(h_tiny);
    free(h_solo);
    if (err) {
        return raise_hip(err);
    }
    p->n_items = (uint32_t)n_items;
    p->n_segs = (uint32_t)n_segs;
    p->n_large = (uint32_t)n_large;
    p->n_tiny = (uint32_t)n_tiny;
    p->n_solo = (uint32_t)n_solo;
    p->stats.items = n_items;
    p->stats.thread_limit = tiny_limit;
    p->stats.by_thread = n_tiny;
    p->stats.by_wave = n_solo;
    p->stats.by_pieces = n_cut;
    p->stats.pieces = n_segs;
    p->stats.empty = n_items - n_tiny - n_solo - n_cut;
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_encode_plan_stats(const struct aws_huffman_amd_encode_plan *plan, struct aws_huffman_amd_plan_stats *stats) {
    *stats = plan->stats;
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_decode_plan_stats(const struct aws_huffman_amd_decode_plan *plan, struct aws_huffman_amd_plan_stats *stats) {
    *stats = plan->stats;
    return AWS_OP_SUCCESS;
}

bool aws_huffman_amd_engine_encodes_in_one_pass(const struct aws_huffman_amd_engine *eng) {
    return eng->single_pass && hufk_encode_one_pass_applies(&eng->tables);
}

int aws_huffman_amd_encode_plan_new(
    struct aws_huffman_amd_encode_plan **out_plan,
    struct aws_huffman_amd_engine *eng,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count) {

    *out_plan = NULL;
    /* the plan destroyed last, if the engine kept it: its device arrays serve again where they are large enough */
    pthread_mutex_lock(&eng->spare_lock);
    struct aws_huffman_amd_encode_plan *p = eng->spare_enc;
    eng->spare_enc = NULL;
    pthread_mutex_unlock(&eng->spare_lock);
    if (p) {
        /* its last launch may still run on the caller's stream (freeing the arrays used to wait for it): waited for by the
         * event that launch left behind -- not for every stream of the device, on which other threads' batches may run */
        ON_DEVICE(eng->device);
        if (plan_wait_done(p->done_event, p->done_on_engine_stream, eng)) {
            p->unkeepable = true; /* (freed, not parked again for the next caller to fail on) */
            aws_huffman_amd_encode_plan_destroy(p);
            return aws_raise_error(AWS_ERROR_UNKNOWN);
        }
        p->done_on_engine_stream = false;
        p->last_input = NULL;
        p->last_output = NULL;
        p->launched = false;
        p->last_single_pass = false;
        p->last_timed_out = false;
        p->look_back_timed_out = false;
    } else {
        p = calloc(1, sizeof(*p));
        if (!p) {
            return aws_raise_error(AWS_ERROR_OOM);
        }
        p->engine = eng;
    }
    if (enc_plan_fill(p, items, item_count)) {
        aws_huffman_amd_encode_plan_destroy(p);
        return AWS_OP_ERR;
    }
    *out_plan = p;
    return AWS_OP_SUCCESS;
}

/* the scratch a planning pass on the device wants, grown as needed */
static int plan_scratch_reserve(void **scratch, size_t *cap, size_t n_items) {
    const size_t want = (size_t)hufk_plan_scratch_bytes(n_items);
    if (want > *cap) {
        hufs_free(*scratch);
        *scratch = hufs_malloc(want);
        *cap = *scratch ? want : 0;
    }
    return *scratch ? 0 : 2;
}

/* A plan from items the host does not look at (plan_kernels.hip): described by a stride, or lying in device memory.  Three
 * small launches count, one copy brings a few totals back (the wait of the call), then the records are written on `stream`:
 * the plan is good for launches on that stream, or on another once this one has been waited for. */
static int enc_plan_fill_on_device(struct aws_huffman_amd_encode_plan *p, const struct hufd_item_source *src, size_t n_items, void *stream) {
    struct aws_huffman_amd_engine *eng = p->engine;
    if (!eng->can_encode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    p->n_items = p->n_segs = p->n_large = p->n_tiny = p->n_solo = 0;
    p->launched = false;
    p->look_back_timed_out = false;
    memset(&p->stats, 0, sizeof(p->stats));
    p->largest_out_cap = 0;
    p->most_overflow_bits = 0;
    if (n_items == 0) {
        return AWS_OP_SUCCESS;
    }
    if (n_items >= 0xFFFFFFFFull) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    void *st = stream ? stream : eng->stream;
    ON_DEVICE(eng->device);
    struct hufk_plan_totals t;
    int e = plan_scratch_reserve(&p->d_plan_scratch, &p->cap_plan_scratch, n_items);
    if (!e) {
        const uint64_t class0 = aws_huffman_amd_engine_encodes_in_one_pass(eng) ? HUFD_ENC_TINY_WAVE_BYTES : HUFD_TINY_MANY_BYTES;
        e = hufk_encode_plan_count(
            src, (uint32_t)n_items, class0, HUFD_ENC_TINY_BYTES, enc_tiny_per_byte(eng), enc_solo_limit(eng, n_items), p->d_plan_scratch, &t, st);
    }
    if (e) {
        return raise_hip(e);
    }
    if (t.invalid || t.totals[0] >= 0xFFFFFFFFull || t.totals[2] >= (1ull << 30)) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    e = enc_plan_reserve(p, n_items, (size_t)t.totals[0], (size_t)t.totals[3], (size_t)t.totals[1]);
    if (!e) {
        e = hufk_encode_plan_fill(
            src, (uint32_t)n_items, (uint32_t)t.totals[0], enc_solo_limit(eng, n_items), p->d_plan_scratch, p->d_items, p->d_segs, p->d_tiny,
            p->d_large, p->d_solo, st);
    }
    if (e) {
        return raise_hip(e);
    }
    p->n_items = (uint32_t)n_items;
    p->n_segs = (uint32_t)t.totals[0];
    p->n_large = (uint32_t)t.totals[3];
    p->n_tiny = (uint32_t)t.totals[1];
    p->n_solo = (uint32_t)t.totals[2];
    p->largest_out_cap = t.largest_out_cap;
    p->most_overflow_bits = t.worst_bits;
    p->stats.items = n_items;
    p->stats.thread_limit = t.tiny_limit;
    p->stats.by_thread = t.totals[1];
    p->stats.by_wave = t.totals[2];
    p->stats.by_pieces = t.totals[7];
    p->stats.pieces = t.totals[0];
    p->stats.empty = n_items - t.totals[1] - t.totals[2] - t.totals[7];
    /* the record-writing kernels run on `st` behind the one wait above: whoever gets this plan's arrays next (a plan
     * destroyed before any launch is the engine's spare) waits for them as for a launch */
    e = plan_mark_done(&p->done_event, &p->done_on_engine_stream, eng, stream);
    return e ? raise_hip(e) : AWS_OP_SUCCESS;
}

static void strided_source(struct hufd_item_source *src, const struct aws_huffman_amd_strided_items *items) {
    memset(src, 0, sizeof(*src));
    src->kind = HUFD_ITEMS_STRIDED;
    src->first_bit = items->first_bit;
    src->eos_padding = items->eos_padding;
    src->in_offset = items->in_offset;
    src->in_stride = items->in_stride;
    src->in_len = items->in_len;
    src->out_offset = items->out_offset;
    src->out_stride = items->out_stride;
    src->out_capacity = items->out_capacity;
}

int aws_huffman_amd_encode_plan_reset_strided(
    struct aws_huffman_amd_encode_plan *p, const struct aws_huffman_amd_strided_items *items, void *stream) {
    struct hufd_item_source src;
    strided_source(&src, items);
    return enc_plan_fill_on_device(p, &src, (size_t)items->count, stream);
}

int aws_huffman_amd_encode_plan_reset_device_items(
    struct aws_huffman_amd_encode_plan *p, const struct aws_huffman_amd_encode_item *device_items, size_t item_count, void *stream) {
    struct hufd_item_source src;
    memset(&src, 0, sizeof(src));
    src.kind = HUFD_ITEMS_DEVICE_ARRAY;
    src.raw = device_items;
    return enc_plan_fill_on_device(p, &src, item_count, stream);
}

int aws_huffman_amd_encode_plan_reset(
    struct aws_huffman_amd_encode_plan *p,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count) {
    p->look_back_timed_out = false;
    return enc_plan_fill(p, items, item_count); /* (device arrays are kept where they are large enough) */
}

void aws_huffman_amd_encode_plan_destroy(struct aws_huffman_amd_encode_plan *p) {
    if (p) {
        struct aws_huffman_amd_engine *eng = p->engine;
        /* the engine keeps ONE destroyed plan with its device arrays for the next aws_huffman_amd_encode_plan_new */
        pthread_mutex_lock(&eng->spare_lock);
        const bool keep = !eng->retiring && !eng->spare_enc && p->cap_items && p != eng->one_enc && !p->unkeepable;
        if (keep) {
            eng->spare_enc = p;
        }
        pthread_mutex_unlock(&eng->spare_lock);
        if (keep) {
            return;
        }
        ON_DEVICE(eng->device);
        enc_plan_release_device(p);
        hufs_free(p->d_plan_scratch);
        hufs_event_destroy(p->done_event);
        free(p);
    }
}

int aws_huffman_amd_encode_plan_launch(
    struct aws_huffman_amd_encode_plan *p,
    const void *device_input,
    void *device_output,
    bool length_only,
    void *stream) {
    return aws_huffman_amd_encode_plan_launch_staged(p, device_input, device_output, length_only, stream, NULL);
}

int aws_huffman_amd_encode_plan_launch_staged(
    struct aws_huffman_amd_encode_plan *p,
    const void *device_input,
    void *device_output,
    bool length_only,
    void *stream,
    void **stage_events) {

    struct hufk_encode_args a;
    memset(&a, 0, sizeof(a));
    a.tables = p->engine->tables;
    a.items = p->d_items;
    a.n_items = p->n_items;
    a.segs = p->d_segs;
    a.n_segs = p->n_segs;
    a.large_items = p->d_large;
    a.tiny_items = p->d_tiny;
    a.n_tiny = p->n_tiny;
    a.solo_items = p->d_solo;
    a.n_solo = p->n_solo;
    a.n_large = p->n_large;
    a.length_only = length_only;
    a.d_in = device_input;
    a.d_out = device_output;
    a.seg_bits = p->d_seg_bits;
    a.wave_bits = p->d_wave_bits;
    a.seg_unk = p->d_seg_unk;
    a.seg_bitoff = p->d_seg_bitoff;
    a.careful_list = p->d_careful;
    a.zero_block = p->d_zero;
    a.zero_is_clear = p->zero_is_clear;
    a.zero_bytes = hufk_encode_zero_bytes((uint32_t)p->cap_segs, (uint32_t)p->cap_items);
    a.careful_count = (uint32_t *)p->d_zero + 2;
    a.seg_unk_seen = p->d_unk_seen;
    a.item_total = p->d_item_total;
    a.single_pass = p->engine->single_pass && !p->look_back_timed_out;
    a.fail_tile = p->engine->encode_fails;
    p->last_input = device_input;
    p->last_output = device_output;
    p->launched = !length_only; /* (a length query leaves lengths in the records and no output: nothing to chain a decode to) */
    p->last_single_pass =
        a.single_pass && (p->n_segs || p->n_solo) && !length_only && hufk_encode_one_pass_applies(&p->engine->tables);
    a.states = p->d_states;
    a.results = p->d_results;
    a.stage_events = stage_events;
    ON_DEVICE(p->engine->device);
    int err = hufk_encode_launch(&a, stream ? stream : p->engine->stream);
    if (p->last_single_pass) {
        p->zero_is_clear = !err; /* (the launch leaves the block clear behind itself -- unless it could not be queued whole) */
    }
    if (!err) {
        err = plan_mark_done(&p->done_event, &p->done_on_engine_stream, p->engine, stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_encode_plan_raw_results(
    struct aws_huffman_amd_encode_plan *p,
    struct hufd_enc_result *raw,
    void *stream) {
    void *st = stream ? stream : p->engine->stream;
    ON_DEVICE(p->engine->device);
    int err = 0;
    /* The one-pass kernel's waves wait for each other's totals; every wait is bounded, and one that ran out raises a
     * flag.  The kernels of the three-kernel road (no waits between workgroups) are queued behind it on the same
     * stream and look at that flag first: the output and these records are whole when the stream gets here, whichever
     * road made them.  The flag comes back with the results only so that the plan stays on the three-kernel road
     * from then on (a wait that ran out once -- the grid was not resident as a whole: another kernel on the device --
     * is likely to run out again, and costs milliseconds). */
    uint32_t timed_out = 0;
    if (p->last_single_pass) {
        err = hufs_copy_d2h(&timed_out, p->d_zero + 4 * sizeof(uint32_t), sizeof(timed_out), st); /* (enc_finish's copy of the word) */
    }
    if (!err) {
        err = hufs_copy_d2h(raw, p->d_results, (size_t)p->n_items * sizeof(*raw), st);
    }
    if (!err) {
        err = hufs_stream_sync(st);
    }
    if (!err && timed_out && !p->engine->encode_fails) {
        p->look_back_timed_out = true;
    }
    p->last_timed_out = !err && timed_out;
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_encode_plan_road(const struct aws_huffman_amd_encode_plan *p, uint32_t *road) {
    *road = !p->last_single_pass ? AWS_HUFFMAN_AMD_ROAD_TWO_PASS
                                 : (p->last_timed_out ? AWS_HUFFMAN_AMD_ROAD_ONE_PASS_GAVE_UP : AWS_HUFFMAN_AMD_ROAD_ONE_PASS);
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_encode_plan_encoded_lengths(struct aws_huffman_amd_encode_plan *p, uint64_t *lengths, void *stream) {
    struct hufd_enc_result *raw = malloc((p->n_items ? p->n_items : 1) * sizeof(*raw));
    if (!raw) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    if (aws_huffman_amd_encode_plan_raw_results(p, raw, stream)) {
        free(raw);
        return AWS_OP_ERR;
    }
    for (uint32_t i = 0; i < p->n_items; ++i) {
        lengths[i] = (raw[i].total_bits + 7) / 8; /* source/huffman.c:121-128 */
    }
    free(raw);
    return AWS_OP_SUCCESS;
}

void aws_huffman_amd_encode_result_from_raw(
    const struct hufd_enc_result *raw,
    struct aws_huffman_amd_encode_result *out) {
    out->consumed = raw->consumed;
    out->produced = raw->produced;
    out->overflow_out.pattern = raw->ovf_pattern;
    out->overflow_out.num_bits = (uint8_t)raw->ovf_bits;
    switch (raw->status) {
        case HUFD_ENC_OK:
            out->rc = AWS_OP_SUCCESS;
            out->error = 0;
            break;
        case HUFD_ENC_SHORT:
            out->rc = AWS_OP_ERR;
            out->error = AWS_ERROR_SHORT_BUFFER;
            break;
        case HUFD_ENC_UNKNOWN:
            out->rc = AWS_OP_ERR;
            out->error = AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL;
            break;
        default:
            out->rc = AWS_OP_ERR;
            out->error = AWS_ERROR_INVALID_STATE;
            break;
    }
}

int aws_huffman_amd_encode_plan_results(
    struct aws_huffman_amd_encode_plan *p,
    struct aws_huffman_amd_encode_result *results,
    void *stream) {

    struct hufd_enc_result *raw = malloc((p->n_items ? p->n_items : 1) * sizeof(*raw));
    if (!raw) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    if (aws_huffman_amd_encode_plan_raw_results(p, raw, stream)) {
        free(raw);
        return AWS_OP_ERR;
    }
    for (uint32_t i = 0; i < p->n_items; ++i) {
        aws_huffman_amd_encode_result_from_raw(&raw[i], &results[i]);
    }
    free(raw);
    return AWS_OP_SUCCESS;
}

/* ------------------------------------------------------------------ decode plans */

#define DEC_SUMMARY_BYTES 256u
#define DEC_PLAN_ARRAYS(X, ci, cc, cl, cr, ns)                                                                            \
    X(d_items, (ci) * sizeof(struct hufd_dec_item))                                                                    \
    X(d_chunk_item, (cc) * sizeof(uint32_t))                                                                           \
    X(d_tail, (ci) * 2 * sizeof(uint32_t))                                                                             \
    X(d_tiny, (ci) * sizeof(uint32_t))                                                                                 \
    X(d_large, (cl) * 2 * sizeof(uint32_t))                                                                            \
    X(d_runs, (cr) * 2 * sizeof(uint32_t))                                                                             \
    X(d_run_fn, (cr) * (ns) * sizeof(uint32_t))                                                                        \
    X(d_fn_tab, (cc) * (ns) * HUFD_DEC_LANES * sizeof(uint16_t))                                                       \
    X(d_cp_tab, (cc) * HUFD_DEC_CP_ROWS * HUFD_DEC_LANES * sizeof(uint16_t))                                           \
    X(d_chunk_fn, (cc) * (ns) * sizeof(uint32_t))                                                                      \
    X(d_slow_list, ((cc) + 1) * sizeof(uint32_t)) /* [0] count, [1..] chunks */                                        \
    X(d_emit_list, ((cc) + 1) * sizeof(uint32_t))                                                                      \
    X(d_dense_list, ((cc) + 1) * sizeof(uint32_t))                                                                     \
    X(d_counters, HUFK_DEC_COUNTERS * sizeof(uint32_t))                                                                \
    X(d_lane_count, (cc) * HUFD_DEC_LANES * sizeof(uint16_t))                                                          \
    X(d_chunk_regular, (cc))                                                                                           \
    X(d_tail_entry, (cc) * sizeof(uint32_t))                                                                           \
    X(d_chunk_entry, (cc) * sizeof(uint32_t))                                                                          \
    X(d_chunk_base, (cc) * sizeof(uint64_t))                                                                           \
    X(d_chunk_rec, (cc) * sizeof(struct hufd_chunk_rec))                                                               \
    X(d_states, (ci) * sizeof(struct hufd_dec_item_state))                                                             \
    /* the list counters as the launch's last kernel found them, one arena cut (256 bytes) in front of the records:    \
     * ONE copy brings both to the host */                                                                             \
    X(d_summary, DEC_SUMMARY_BYTES)                                                                                    \
    X(d_results, (ci) * sizeof(struct hufd_dec_result))

static void dec_plan_release_device(struct aws_huffman_amd_decode_plan *p) {
    hufs_free(p->d_arena);
    p->d_arena = NULL;
#define DEC_FORGET(name, bytes) p->name = NULL;
    DEC_PLAN_ARRAYS(DEC_FORGET, 0, 0, 0, 0, 0)
#undef DEC_FORGET
    p->cap_items = p->cap_chunks = p->cap_large = p->cap_runs = 0;
}

/* short items are one thread's work, without chunks */
/* short items are one thread's work, without chunks; with codes too long for the chunked decoder's tables the
 * longer items are one workgroup's, without chunks either */
static bool dec_item_is_tiny(const struct aws_huffman_amd_decode_item *it, uint64_t limit) {
    return it->in_len > 0 && it->in_len <= limit;
}

/* the longest item a lone thread takes in this plan: see HUFD_DEC_TINY_PER_BYTE */
static uint64_t dec_tiny_limit(const struct aws_huffman_amd_decode_item *items, size_t n_items, struct item_stats *st) {
    /* (round 4: the upper class ended at 3 KiB of encoded bytes; since several short chunks share a workgroup -- dec_sync_pack,
     * dec_emit_pack -- items above what one wave takes decode 1.8 .. 2.8 times faster through the chunk kernels than a thread each:
     * 1 KiB items 0.82 -> 0.45 ms, 1.5 KiB 1.13 -> 0.41 ms per 128 MiB) */
    static const uint64_t classes[2] = {HUFD_DEC_COOP_BYTES, HUFD_DEC_TINY_BYTES};
    /* (one pass over the items for both classes: a plan of a million header-sized items is read from memory once here) */
    uint64_t count[2] = {0, 0}, longest[2] = {0, 0};
    st->shortest = UINT64_MAX;
    st->longest = 0;
    st->worst_bits = 0;
    for (size_t i = 0; i < n_items; ++i) {
        const uint64_t len = items[i].in_len;
        st->shortest = len < st->shortest ? len : st->shortest;
        st->longest = len > st->longest ? len : st->longest;
        st->worst_bits = items[i].first_bit > st->worst_bits ? items[i].first_bit : st->worst_bits;
        for (int c = 0; c < 2; ++c) {
            if (len <= classes[c]) {
                ++count[c];
                longest[c] = len > longest[c] ? len : longest[c];
            }
        }
    }
    for (int c = 0; c < 2; ++c) {
        if (longest[c] > HUFD_TINY_FEW_BYTES && count[c] >= tiny_per_byte(true) * longest[c]) {
            return classes[c];
        }
    }
    return HUFD_TINY_FEW_BYTES;
}

/* Items of a coder with long codes of at least this many encoded bytes go across the chip, seven launches each
 * (dec_wide_*): from HUFD_WIDE_MIN_BYTES on when the batch's items do not fill the chip a workgroup each, from
 * HUFD_WIDE_MANY_MIN_BYTES on when they do.  Tests set one limit for both. */
static uint64_t s_wide_min_bytes = 0;

void aws_huffman_amd_testing_set_wide_min_bytes(uint64_t bytes) {
    s_wide_min_bytes = bytes;
}

static uint64_t wide_min_bytes(size_t deep_items) {
    if (s_wide_min_bytes) {
        return s_wide_min_bytes;
    }
    return deep_items < HUFD_WIDE_FEW_ITEMS ? HUFD_WIDE_MIN_BYTES : HUFD_WIDE_MANY_MIN_BYTES;
}

static bool dec_item_is_deep(
    const struct aws_huffman_amd_engine *eng, const struct aws_huffman_amd_decode_item *it, uint64_t tiny_limit) {
    /* (the same kernel, as one wave, takes every coder's items that are too short to be worth a chunk's workgroup) */
    return it->in_len > tiny_limit && (eng->tables.deep_entries || it->in_len <= HUFD_DEC_COOP_BYTES);
}

/* (codes of one length: no chunks, no walks to bring into step) */
static bool dec_item_is_fixed(
    const struct aws_huffman_amd_engine *eng, const struct aws_huffman_amd_decode_item *it, uint64_t tiny_limit) {
    return eng->tables.fixed_bits && it->in_len > tiny_limit;
}

static uint64_t dec_item_chunks(
    const struct aws_huffman_amd_engine *eng, const struct aws_huffman_amd_decode_item *it, uint64_t tiny_limit) {
    if (dec_item_is_tiny(it, tiny_limit) || dec_item_is_fixed(eng, it, tiny_limit) || dec_item_is_deep(eng, it, tiny_limit)) {
        return 0;
    }
    return (it->in_len + HUFD_DEC_CHUNK_BYTES - 1) / HUFD_DEC_CHUNK_BYTES;
}

/* the plan's device arrays for this many items, chunks, large items and runs (grown, never shrunk); 0 or a HIP error */
/* whole lanes of a chunk with `left` bytes of its item from its first byte on: sub-chunks with 8 more bytes behind them */
static uint64_t whole_lanes_of(uint64_t left) {
    const uint64_t in_chunk = left < HUFD_DEC_CHUNK_BYTES + 8u ? left : HUFD_DEC_CHUNK_BYTES + 8u;
    return in_chunk >= 8 ? (in_chunk - 8) / HUFD_DEC_SUB_BYTES : 0;
}

static int dec_plan_reserve(struct aws_huffman_amd_decode_plan *p, size_t n_items, size_t n_chunks, size_t n_large, size_t n_runs) {
    const uint32_t ns = p->engine->tables.n_states;
    if (n_items > p->cap_items || n_chunks > p->cap_chunks || n_large > p->cap_large || n_runs > p->cap_runs) {
        dec_plan_release_device(p);
        const size_t ci = n_items ? n_items : 1, cc = n_chunks ? n_chunks : 1, cl = n_large ? n_large : 1;
        const size_t cr = n_runs ? n_runs : 1;
        size_t total = 0;
#define DEC_SIZE(name, bytes) (void)arena_cut(&total, (bytes));
        DEC_PLAN_ARRAYS(DEC_SIZE, ci, cc, cl, cr, ns)
#undef DEC_SIZE
        p->d_arena = hufs_malloc(total);
        if (!p->d_arena) {
            return 2;
        }
        total = 0;
#define DEC_PLACE(name, bytes) p->name = (void *)((uint8_t *)p->d_arena + arena_cut(&total, (bytes)));
        DEC_PLAN_ARRAYS(DEC_PLACE, ci, cc, cl, cr, ns)
#undef DEC_PLACE
        p->cap_items = ci;
        p->cap_chunks = cc;
        p->cap_large = cl;
        p->cap_runs = cr;
        /* the list counters start out clear; from then on every launch clears them for itself and for the one behind it (hufk_decode_args.counters) */
        int e = hufs_memset(p->d_counters, 0, HUFK_DEC_COUNTERS * sizeof(uint32_t), p->engine->stream);
        if (!e) {
            e = hufs_stream_sync(p->engine->stream);
        }
        if (e) {
            return e;
        }
        p->launches_with_chunks = 0;
        if ((uint8_t *)p->d_results != (uint8_t *)p->d_summary + DEC_SUMMARY_BYTES) {
            return 1; /* (the one copy of aws_huffman_amd_decode_plan_results counts on it: cuts are 256 bytes apart) */
        }
    }
    return 0;
}

static int dec_plan_fill(
    struct aws_huffman_amd_decode_plan *p,
    const struct aws_huffman_amd_decode_item *items,
    size_t n_items) {

    struct aws_huffman_amd_engine *eng = p->engine;
    if (!eng->can_decode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    struct item_stats stats;
    const uint64_t tiny_limit = dec_tiny_limit(items, n_items, &stats);
    /* a failed refill must not leave counts of the fill before behind (the device arrays may be gone or too small) */
    p->n_items = p->n_chunks = p->n_large = p->n_runs = p->n_tail = p->n_fixed = p->n_wide = 0;
    p->quiet = false; /* (other items: nothing is known of what their launches list) */
    p->n_tiny = p->n_deep = 0;
    memset(&p->stats, 0, sizeof(p->stats));
    p->chained = false;
    if (n_items >= PLAN_ON_DEVICE_MIN_ITEMS && n_items < 0xFFFFFFFFull && stats.shortest >= 1 && stats.longest <= tiny_limit &&
        stats.worst_bits <= 7) {
        /* every item is one thread's work (dec_item_is_tiny): no chunks, no lists to make -- the caller's records go to the
         * device as they are and become the kernels' there (hufk_decode_plan_tiny_items) */
        ON_DEVICE(eng->device);
        int e = dec_plan_reserve(p, n_items, 0, 0, 0);
        void *d_raw = e ? NULL : hufs_malloc(n_items * sizeof(*items));
        if (!e && !d_raw) {
            e = 2;
        }
        if (!e) {
            e = hufs_copy_h2d(d_raw, items, n_items * sizeof(*items), eng->stream);
        }
        if (!e) {
            e = hufk_decode_plan_tiny_items(d_raw, (uint32_t)n_items, p->d_items, p->d_tiny, eng->stream);
        }
        struct aws_huffman_amd_decode_item *keep = e ? NULL : realloc(p->h_items, n_items * sizeof(*keep));
        if (!e && !keep) {
            e = 2;
        }
        if (!e) {
            memcpy(keep, items, n_items * sizeof(*keep)); /* (while the copy and the kernel run) */
            p->h_items = keep;
            e = hufs_stream_sync(eng->stream);
        }
        hufs_free(d_raw);
        if (e) {
            return raise_hip(e);
        }
        p->wide_from = wide_min_bytes(0);
        p->tail_stage_bytes = 0;
        p->tail_lanes = 0;
    p->tail_wide_lanes = 0;
        p->tail_wide_lanes = 0;
        p->n_tail_narrow = 0;
        p->n_items = (uint32_t)n_items;
        p->n_tiny = (uint32_t)n_items;
        p->stats.items = p->stats.by_thread = n_items;
        p->stats.thread_limit = tiny_limit;
        return AWS_OP_SUCCESS;
    }
    uint64_t n_chunks = 0, n_large = 0, n_runs = 0, n_cut = 0;
    for (size_t i = 0; i < n_items; ++i) {
        /* symbol counts in the scan's function entries are 26-bit; 4 GiB of encoded bytes per item is the limit */
        if (items[i].first_bit > 7 || items[i].in_len > 0xFFFFFFFFull) {
            return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
        }
        const uint64_t chunks = dec_item_chunks(eng, &items[i], tiny_limit);
        n_chunks += chunks;
        n_cut += chunks != 0;
        n_large += chunks > HUFD_SCAN_SMALL_MAX;
        n_runs += chunks > HUFD_SCAN_SMALL_MAX ? (chunks + HUFD_SCAN_RUN_CHUNKS - 1) / HUFD_SCAN_RUN_CHUNKS : 0;
    }
    if (n_chunks >= 0xFFFFFFFFull || n_items >= 0xFFFFFFFFull) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }

    struct hufd_dec_item *h_items = malloc((n_items ? n_items : 1) * sizeof(*h_items));
    uint32_t *h_large = malloc((n_large ? n_large : 1) * 2 * sizeof(uint32_t));
    uint32_t *h_runs = malloc((n_runs ? n_runs : 1) * 2 * sizeof(uint32_t));
    uint32_t *h_tail = malloc((n_items ? n_items : 1) * 2 * sizeof(uint32_t));
    uint32_t *h_tiny = malloc((n_items ? n_items : 1) * sizeof(uint32_t));
    if (!h_items || !h_large || !h_runs || !h_tail || !h_tiny) {
        free(h_tail);
        free(h_tiny);
        free(h_runs);
        free(h_items);
        free(h_large);
        return aws_raise_error(AWS_ERROR_OOM);
    }
    uint32_t chunk = 0, large = 0, run = 0, tail = 0, narrow = 0, wide = 0, tiny = 0, deep = 0;
    size_t deep_items = 0;
    for (size_t i = 0; i < n_items && eng->tables.deep_entries; ++i) {
        deep_items += !dec_item_is_tiny(&items[i], tiny_limit) && dec_item_is_deep(eng, &items[i], tiny_limit);
    }
    const uint64_t wide_from = wide_min_bytes(deep_items);
    uint32_t *h_fixed = NULL;
    uint64_t n_fixed = 0, fixed_items = 0;
    struct hufk_wide_item *h_wide = NULL;
    uint32_t n_wide = 0;
    uint64_t wide_bytes = 0;
    bool wide_oom = false;
    uint64_t tail_stage = 0; /* the most symbols a chunk that holds the end of a stream can decode to */
    uint64_t tail_lanes = 0; /* ... and the most whole lanes it has */
    uint64_t wide_lanes = 0; /* ... and a wide one has (dec_emit_fast<TAIL>'s workgroup size) */
    for (size_t i = 0; i < n_items; ++i) {
        const struct aws_huffman_amd_decode_item *src = &items[i];
        struct hufd_dec_item *dst = &h_items[i];
        const uint32_t chunks = (uint32_t)dec_item_chunks(eng, src, tiny_limit);
        dst->in_off = src->in_offset;
        dst->in_len = src->in_len;
        dst->out_off = src->out_offset;
        dst->out_cap = src->out_capacity;
        dst->first_bit = src->first_bit;
        dst->first_chunk = chunk;
        dst->n_chunks = chunks;
        dst->tiny = 0;
        if (dec_item_is_tiny(src, tiny_limit)) {
            dst->tiny = 1;
            h_tiny[tiny++] = (uint32_t)i;
        } else if (dec_item_is_fixed(eng, src, tiny_limit)) {
            dst->tiny = 2;
            ++fixed_items;
            const uint64_t blocks = (src->in_len + HUFD_FIXED_BLOCK_BYTES - 1) / HUFD_FIXED_BLOCK_BYTES;
            uint32_t *more = realloc(h_fixed, (n_fixed + blocks) * 2 * sizeof(uint32_t));
            if (!more) {
                wide_oom = true;
            } else {
                h_fixed = more;
                for (uint64_t k = 0; k < blocks; ++k) {
                    h_fixed[2 * (n_fixed + k)] = (uint32_t)i;
                    h_fixed[2 * (n_fixed + k) + 1] = (uint32_t)k;
                }
                n_fixed += blocks;
            }
        } else if (dec_item_is_deep(eng, src, tiny_limit)) {
            dst->tiny = 1;
            h_tiny[n_items - ++deep] = (uint32_t)i;
            if (eng->tables.deep_entries && src->in_len >= wide_from) {
                struct hufk_wide_item *more = realloc(h_wide, (n_wide + 1) * sizeof(*more));
                if (!more) {
                    wide_oom = true;
                } else {
                    h_wide = more;
                    h_wide[n_wide].slot = deep; /* from the back of the list, for now */
                    h_wide[n_wide].n_blocks = (uint32_t)((src->in_len + HUFD_WIDE_BLOCK_BYTES - 1) / HUFD_WIDE_BLOCK_BYTES);
                    h_wide[n_wide].block_offset = wide_bytes;
                    wide_bytes += hufk_decode_wide_bytes(h_wide[n_wide].n_blocks);
                    ++n_wide;
                }
            }
        }
        /* the chunks that hold the end of the stream, or lie just behind it: fewer than a chunk + 8 bytes left from their
         * first byte on -- the item's last chunk, and the one in front of it when the last one holds less than 8 bytes.
         * (chunk -> item and the records per chunk are made on the device: hufk_decode_plan_chunks) */
        for (uint32_t k = chunks > 2 ? chunks - 2 : 0; k < chunks; ++k) {
            if (src->in_len - (uint64_t)k * HUFD_DEC_CHUNK_BYTES < (uint64_t)HUFD_DEC_CHUNK_BYTES + 8u) {
                const uint64_t left = src->in_len - (uint64_t)k * HUFD_DEC_CHUNK_BYTES;
                /* (the chunks with few whole lanes first: several of those share a workgroup, dec_sync_pack) */
                if (whole_lanes_of(left) <= HUFD_DEC_PACK_LANES) {
                    h_tail[narrow++] = chunk + k;
                } else {
                    h_tail[2 * (n_items ? n_items : 1) - ++wide] = chunk + k;
                }
                ++tail;
                const uint32_t shortest = eng->tables.min_bits ? eng->tables.min_bits : 1;
                uint64_t holds = left * 8 / shortest + 1;
                holds = holds < src->out_capacity ? holds : src->out_capacity;
                tail_stage = holds > tail_stage ? holds : tail_stage;
                const uint64_t whole = whole_lanes_of(left);
                if (whole <= HUFD_DEC_PACK_LANES) {
                    tail_lanes = whole > tail_lanes ? whole : tail_lanes;
                } else {
                    wide_lanes = whole > wide_lanes ? whole : wide_lanes;
                }
            }
        }
        chunk += chunks;
        if (chunks > HUFD_SCAN_SMALL_MAX) {
            h_large[2 * large] = (uint32_t)i;
            h_large[2 * large + 1] = run;
            ++large;
            for (uint32_t k = 0; k * HUFD_SCAN_RUN_CHUNKS < chunks; ++k) {
                h_runs[2 * run] = (uint32_t)i;
                h_runs[2 * run + 1] = k;
                ++run;
            }
        }
    }

    int err = wide_oom ? 2 : 0;
    ON_DEVICE(eng->device);
    for (uint32_t k = 0; k < n_wide; ++k) {
        h_wide[k].slot = deep - h_wide[k].slot; /* the deep items are the last `deep` of d_tiny, filled from the back */
    }
    if (!err && n_fixed > 0xFFFFFFFFull) {
        err = 2;
    }
    if (!err && n_fixed > p->cap_fixed) {
        hufs_free(p->d_fixed);
        p->d_fixed = hufs_malloc(n_fixed * 2 * sizeof(uint32_t));
        p->cap_fixed = p->d_fixed ? n_fixed : 0;
        err = p->d_fixed ? 0 : 2;
    }
    if (!err && n_fixed) {
        err = hufs_copy_h2d(p->d_fixed, h_fixed, n_fixed * 2 * sizeof(uint32_t), eng->stream);
        if (!err) {
            err = hufs_stream_sync(eng->stream); /* (h_fixed is freed below) */
        }
    }
    free(h_fixed);
    if (!err && wide_bytes > p->cap_wide_block) {
        hufs_free(p->d_wide_block);
        p->d_wide_block = hufs_malloc(wide_bytes);
        p->cap_wide_block = p->d_wide_block ? wide_bytes : 0;
        err = p->d_wide_block ? 0 : 2;
    }
    if (!err) {
        err = dec_plan_reserve(p, n_items, n_chunks, n_large, n_runs);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_items, h_items, n_items * sizeof(*h_items), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_runs, h_runs, n_runs * 2 * sizeof(uint32_t), eng->stream);
    }
    /* (the wide chunks were listed from the back, last one first: behind the narrow ones, in their order) */
    if (wide) {
        /* (the two blocks may overlap when nearly every slot of the list is taken: turn the back block round where it
         * lies, then move it down as a whole -- a copy entry by entry overwrote wide chunks it had not read yet) */
        uint32_t *back = h_tail + (2 * (n_items ? n_items : 1) - wide);
        for (uint32_t lo = 0, hi = wide - 1; lo < hi; ++lo, --hi) {
            const uint32_t t = back[lo];
            back[lo] = back[hi];
            back[hi] = t;
        }
        memmove(h_tail + narrow, back, (size_t)wide * sizeof(uint32_t));
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_tail, h_tail, tail * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_tiny, h_tiny, (tiny || deep ? n_items : 0) * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufk_decode_plan_chunks(p->d_items, (uint32_t)n_items, (uint32_t)n_chunks, p->d_chunk_item, p->d_chunk_rec, eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d(p->d_large, h_large, n_large * 2 * sizeof(uint32_t), eng->stream);
    }
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    /* keep what the result translation needs */
    if (!err) {
        struct aws_huffman_amd_decode_item *keep = realloc(p->h_items, (n_items ? n_items : 1) * sizeof(*keep));
        if (!keep) {
            err = 2;
        } else {
            if (n_items) { /* (a plan of no items may be given no array: memcpy wants a pointer even for nothing) */
                memcpy(keep, items, n_items * sizeof(*keep));
            }
            p->h_items = keep;
        }
    }
    free(h_items);
    free(h_large);
    free(h_runs);
    free(h_tail);
    free(h_tiny);
    if (err) {
        free(h_wide);
        return raise_hip(err);
    }
    free(p->h_wide);
    p->h_wide = h_wide;
    p->n_wide = n_wide;
    p->wide_from = wide_from;
    p->n_fixed = (uint32_t)n_fixed;
    p->n_items = (uint32_t)n_items;
    p->n_chunks = (uint32_t)n_chunks;
    p->n_large = (uint32_t)n_large;
    p->n_runs = (uint32_t)n_runs;
    p->n_tail = tail;
    p->tail_stage_bytes = tail_stage + 32 < 0xFFFFFFFFu ? (uint32_t)tail_stage + 32u : 0u;
    p->tail_lanes = tail_lanes < HUFD_DEC_LANES ? (uint32_t)tail_lanes : HUFD_DEC_LANES;
    p->tail_wide_lanes = wide_lanes < HUFD_DEC_LANES ? (uint32_t)wide_lanes : HUFD_DEC_LANES;
    p->n_tail_narrow = narrow;
    p->n_tiny = tiny;
    p->n_deep = deep;
    {
        const bool packs = narrow >= HUFD_DEC_PACK_MIN_CHUNKS && p->tail_lanes + 2u <= HUFD_DEC_LANES / 2;
        p->stats.items = n_items;
        p->stats.thread_limit = tiny_limit;
        p->stats.by_thread = tiny;
        p->stats.by_blocks = n_wide + fixed_items;
        p->stats.by_wave = eng->tables.deep_entries ? 0 : deep;
        p->stats.by_workgroup = eng->tables.deep_entries ? deep - n_wide : 0;
        p->stats.by_pieces = n_cut;
        p->stats.pieces = n_chunks;
        /* (a few among many chunks inside streams: workgroups of the big kernels' own grids -- hufk_host::tails_are_folded) */
        const bool folded = tail && tail < n_chunks && (uint64_t)tail * 8 <= n_chunks;
        p->stats.end_pieces_folded = folded ? tail : 0;
        p->stats.end_pieces_packed = packs && !folded ? narrow : 0;
        p->stats.end_pieces_single = tail - p->stats.end_pieces_packed - p->stats.end_pieces_folded;
        p->stats.empty = n_items - tiny - deep - fixed_items - n_cut;
    }
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_decode_plan_new(
    struct aws_huffman_amd_decode_plan **out_plan,
    struct aws_huffman_amd_engine *eng,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count) {

    *out_plan = NULL;
    pthread_mutex_lock(&eng->spare_lock);
    struct aws_huffman_amd_decode_plan *p = eng->spare_dec;
    eng->spare_dec = NULL;
    pthread_mutex_unlock(&eng->spare_lock);
    if (p) {
        /* (as for an encode plan: the spare's last launch is waited for by its own event) */
        ON_DEVICE(eng->device);
        if (plan_wait_done(p->done_event, p->done_on_engine_stream, eng)) {
            p->unkeepable = true;
            aws_huffman_amd_decode_plan_destroy(p);
            return aws_raise_error(AWS_ERROR_UNKNOWN);
        }
        p->done_on_engine_stream = false;
    } else {
        p = calloc(1, sizeof(*p));
        if (!p) {
            return aws_raise_error(AWS_ERROR_OOM);
        }
        p->engine = eng;
    }
    if (dec_plan_fill(p, items, item_count)) {
        aws_huffman_amd_decode_plan_destroy(p);
        return AWS_OP_ERR;
    }
    *out_plan = p;
    return AWS_OP_SUCCESS;
}

/* the items of a source the device planner does not take (a coder with codes of more than 12 bits or of one length: their plans
 * hold what only the host lays out), brought to the host: the loop over them makes the plan */
static int dec_items_to_host(
    struct aws_huffman_amd_decode_plan *p, const struct hufd_item_source *src, size_t n_items, void *st, struct aws_huffman_amd_decode_item **out) {
    struct aws_huffman_amd_decode_item *items = calloc(n_items ? n_items : 1, sizeof(*items));
    if (!items) {
        return 2;
    }
    int e = 0;
    if (src->kind == HUFD_ITEMS_STRIDED) {
        for (size_t i = 0; i < n_items; ++i) {
            items[i].in_offset = src->in_offset + i * src->in_stride;
            items[i].in_len = src->in_len;
            items[i].first_bit = (uint8_t)src->first_bit;
            items[i].out_offset = src->out_offset + i * src->out_stride;
            items[i].out_capacity = src->out_capacity;
        }
    } else if (src->kind == HUFD_ITEMS_DEVICE_ARRAY) {
        e = hufs_copy_d2h(items, src->raw, n_items * sizeof(*items), st);
        if (!e) {
            e = hufs_stream_sync(st);
        }
    } else {
        struct hufd_enc_item *ei = malloc((n_items ? n_items : 1) * sizeof(*ei));
        struct hufd_enc_result *er = malloc((n_items ? n_items : 1) * sizeof(*er));
        e = ei && er ? 0 : 2;
        if (!e) {
            e = hufs_copy_d2h(ei, src->enc_items, n_items * sizeof(*ei), st);
        }
        if (!e) {
            e = hufs_copy_d2h(er, src->enc_results, n_items * sizeof(*er), st);
        }
        if (!e) {
            e = hufs_stream_sync(st);
        }
        for (size_t i = 0; i < n_items && !e; ++i) {
            items[i].in_offset = ei[i].out_off;
            items[i].in_len = er[i].produced < ei[i].out_cap ? er[i].produced : ei[i].out_cap;
            items[i].out_offset = ei[i].in_off;
            items[i].out_capacity = ei[i].in_len;
        }
        free(ei);
        free(er);
    }
    (void)p;
    if (e) {
        free(items);
        return e;
    }
    *out = items;
    return 0;
}

/* as enc_plan_fill_on_device: chunks, lists and per-chunk records of the items of `src`, made on the device */
static int dec_plan_fill_on_device(struct aws_huffman_amd_decode_plan *p, const struct hufd_item_source *src, size_t n_items, void *stream) {
    struct aws_huffman_amd_engine *eng = p->engine;
    if (!eng->can_decode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    void *st = stream ? stream : eng->stream;
    if (eng->tables.deep_entries || eng->tables.fixed_bits) {
        struct aws_huffman_amd_decode_item *items = NULL;
        ON_DEVICE(eng->device);
        const int e = dec_items_to_host(p, src, n_items, st, &items);
        if (e) {
            return raise_hip(e);
        }
        const int rc = dec_plan_fill(p, items, n_items);
        free(items);
        return rc;
    }
    p->n_items = p->n_chunks = p->n_large = p->n_runs = p->n_tail = p->n_fixed = p->n_wide = 0;
    p->quiet = false; /* (other items: nothing is known of what their launches list) */
    p->n_tiny = p->n_deep = 0;
    memset(&p->stats, 0, sizeof(p->stats));
    p->chained = false;
    p->wide_from = wide_min_bytes(0);
    p->tail_stage_bytes = 0;
    p->tail_lanes = 0;
    p->tail_wide_lanes = 0;
    p->n_tail_narrow = 0;
    if (n_items == 0) {
        return AWS_OP_SUCCESS;
    }
    if (n_items >= 0xFFFFFFFFull) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    ON_DEVICE(eng->device);
    struct hufk_plan_totals t;
    int e = plan_scratch_reserve(&p->d_plan_scratch, &p->cap_plan_scratch, n_items);
    if (!e) {
        e = hufk_decode_plan_count(src, (uint32_t)n_items, tiny_per_byte(true), eng->tables.min_bits, p->d_plan_scratch, &t, st);
    }
    if (e) {
        return raise_hip(e);
    }
    if (t.invalid || t.totals[0] >= 0xFFFFFFFFull) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT);
    }
    const uint64_t n_chunks = t.totals[0], tiny = t.totals[1], coop = t.totals[2], n_large = t.totals[3], n_runs = t.totals[4];
    const uint64_t narrow = t.totals[5], wide = t.totals[6];
    e = dec_plan_reserve(p, n_items, (size_t)n_chunks, (size_t)n_large, (size_t)n_runs);
    if (!e) {
        e = hufk_decode_plan_fill(
            src, (uint32_t)n_items, eng->tables.min_bits, p->d_plan_scratch, p->d_items, p->d_tiny, p->d_tail, p->d_large, p->d_runs, st);
    }
    if (!e) {
        e = hufk_decode_plan_chunks(p->d_items, (uint32_t)n_items, (uint32_t)n_chunks, p->d_chunk_item, p->d_chunk_rec, st);
    }
    if (e) {
        return raise_hip(e);
    }
    p->n_items = (uint32_t)n_items;
    p->n_chunks = (uint32_t)n_chunks;
    p->n_large = (uint32_t)n_large;
    p->n_runs = (uint32_t)n_runs;
    p->n_tail = (uint32_t)(narrow + wide);
    p->n_tail_narrow = (uint32_t)narrow;
    p->n_tiny = (uint32_t)tiny;
    p->n_deep = (uint32_t)coop;
    p->tail_stage_bytes = t.tail_stage + 32 < 0xFFFFFFFFu ? (uint32_t)t.tail_stage + 32u : 0u;
    p->tail_lanes = t.tail_lanes < HUFD_DEC_LANES ? (uint32_t)t.tail_lanes : HUFD_DEC_LANES;
    p->tail_wide_lanes = t.wide_lanes < HUFD_DEC_LANES ? (uint32_t)t.wide_lanes : HUFD_DEC_LANES;
    p->chained = true; /* (the items are known on the device only: the results are translated from its records) */
    {
        const bool packs = narrow >= HUFD_DEC_PACK_MIN_CHUNKS && p->tail_lanes + 2u <= HUFD_DEC_LANES / 2;
        p->stats.items = n_items;
        p->stats.thread_limit = t.tiny_limit;
        p->stats.by_thread = tiny;
        p->stats.by_wave = coop;
        p->stats.by_pieces = t.totals[7];
        p->stats.pieces = n_chunks;
        const bool folded = narrow + wide && narrow + wide < n_chunks && (narrow + wide) * 8 <= n_chunks;
        p->stats.end_pieces_folded = folded ? narrow + wide : 0;
        p->stats.end_pieces_packed = packs && !folded ? narrow : 0;
        p->stats.end_pieces_single = narrow + wide - p->stats.end_pieces_packed - p->stats.end_pieces_folded;
        p->stats.empty = n_items - tiny - coop - t.totals[7];
    }
    e = plan_mark_done(&p->done_event, &p->done_on_engine_stream, eng, stream); /* (as enc_plan_fill_on_device) */
    return e ? raise_hip(e) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_decode_plan_reset_strided(
    struct aws_huffman_amd_decode_plan *p, const struct aws_huffman_amd_strided_items *items, void *stream) {
    struct hufd_item_source src;
    strided_source(&src, items);
    return dec_plan_fill_on_device(p, &src, (size_t)items->count, stream);
}

int aws_huffman_amd_decode_plan_reset_device_items(
    struct aws_huffman_amd_decode_plan *p, const struct aws_huffman_amd_decode_item *device_items, size_t item_count, void *stream) {
    struct hufd_item_source src;
    memset(&src, 0, sizeof(src));
    src.kind = HUFD_ITEMS_DEVICE_ARRAY;
    src.raw = device_items;
    return dec_plan_fill_on_device(p, &src, item_count, stream);
}

int aws_huffman_amd_decode_plan_reset(
    struct aws_huffman_amd_decode_plan *p,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count) {
    return dec_plan_fill(p, items, item_count);
}

void aws_huffman_amd_decode_plan_destroy(struct aws_huffman_amd_decode_plan *p) {
    if (p) {
        struct aws_huffman_amd_engine *eng = p->engine;
        pthread_mutex_lock(&eng->spare_lock);
        const bool keep = !eng->retiring && !eng->spare_dec && p->cap_items && p != eng->one_dec && !p->unkeepable;
        if (keep) {
            eng->spare_dec = p;
        }
        pthread_mutex_unlock(&eng->spare_lock);
        if (keep) {
            return;
        }
        ON_DEVICE(p->engine->device);
        dec_plan_release_device(p);
        hufs_free(p->d_plan_scratch);
        hufs_event_destroy(p->done_event);
        hufs_free(p->d_wide_block);
        hufs_free(p->d_fixed);
        free(p->h_wide);
        free(p->h_items);
        free(p);
    }
}

int aws_huffman_amd_decode_plan_launch(
    struct aws_huffman_amd_decode_plan *p,
    const void *device_input,
    void *device_output,
    void *stream) {
    return aws_huffman_amd_decode_plan_launch_staged(p, device_input, device_output, stream, NULL);
}

int aws_huffman_amd_decode_plan_launch_staged(
    struct aws_huffman_amd_decode_plan *p,
    const void *device_input,
    void *device_output,
    void *stream,
    void **stage_events) {

    struct hufk_decode_args a;
    memset(&a, 0, sizeof(a));
    a.tables = p->engine->tables;
    a.items = p->d_items;
    a.n_items = p->n_items;
    a.chunk_item = p->d_chunk_item;
    a.n_chunks = p->n_chunks;
    a.tail_chunks = p->d_tail;
    a.n_tail = p->n_tail;
    a.tail_stage_bytes = p->tail_stage_bytes;
    a.tail_lanes = p->tail_lanes;
    a.tail_wide_lanes = p->tail_wide_lanes;
    a.n_tail_narrow = p->n_tail_narrow;
    a.deep_items = p->d_tiny + (p->n_items - p->n_deep);
    a.n_deep = p->n_deep;
    a.wide = p->h_wide;
    a.n_wide = p->n_wide;
    a.wide_from = p->wide_from;
    a.wide_block = p->d_wide_block;
    a.fixed_blocks = p->d_fixed;
    a.n_fixed_blocks = p->n_fixed;
    a.tiny_items = p->d_tiny;
    a.n_tiny = p->n_tiny;
    a.large_items = p->d_large;
    a.n_large = p->n_large;
    a.runs = p->d_runs;
    a.n_runs = p->n_runs;
    a.run_fn = p->d_run_fn;
    a.d_in = device_input;
    a.d_out = device_output;
    a.fn_tab = p->d_fn_tab;
    a.cp_tab = p->d_cp_tab;
    a.chunk_fn = p->d_chunk_fn;
    a.slow_list = p->d_slow_list + 1;
    a.emit_list = p->d_emit_list + 1;
    a.dense_list = p->d_dense_list + 1;
    a.counters = p->d_counters;
    a.counters_self_cleared = 1;
    if (p->n_chunks) {
        ++p->launches_with_chunks;
    }
    a.summary = p->d_summary;
    a.quiet = p->quiet && !(testing_decode_road() & AWS_HUFFMAN_AMD_TEST_DECODE_ALL_KERNELS);
    a.lane_count = p->d_lane_count;
    a.chunk_regular = p->d_chunk_regular;
    a.tail_entry = p->d_tail_entry;
    a.chunk_entry = p->d_chunk_entry;
    a.chunk_base = p->d_chunk_base;
    a.chunk_rec = p->d_chunk_rec;
    a.side_stream = p->engine->side_stream;
    a.fork_event = p->engine->fork_event;
    a.join_event = p->engine->join_event;
    a.states = p->d_states;
    a.results = p->d_results;
    {
        const uint32_t road = testing_decode_road();
        a.one_chunk_a_workgroup = (road & AWS_HUFFMAN_AMD_TEST_DECODE_ONE_CHUNK_A_WORKGROUP) != 0;
        a.tails_apart = (road & AWS_HUFFMAN_AMD_TEST_DECODE_TAILS_APART) != 0;
        /* dec_wide_* give every long item of a coder with long codes up: dec_wide_fn_*, the road for such an item by
         * transfer functions, takes it -- or gives it up as well, and dec_deep takes it (the ways back) */
        a.wide_fails = road & AWS_HUFFMAN_AMD_TEST_DECODE_WIDE_FN_FAILS ? 2u : (road & AWS_HUFFMAN_AMD_TEST_DECODE_WIDE_FAILS ? 1u : 0u);
        /* the chunks whose walks do not fall into step through dec_sync and dec_emit (not dec_sync_few / _true) */
        a.few_walks = road & AWS_HUFFMAN_AMD_TEST_DECODE_LONG_WAY ? 0u : 1u;
    }
    a.stage_events = stage_events;
    ON_DEVICE(p->engine->device);
    int err = hufk_decode_launch(&a, stream ? stream : p->engine->stream);
    if (!err) {
        err = plan_mark_done(&p->done_event, &p->done_on_engine_stream, p->engine, stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

/*
 * Device record -> what aws_huffman_decode would have returned
 * (reference source/huffman.c:240-268; DESIGN.md "Decode outcome").
 */
void aws_huffman_amd_decode_result_from_raw(
    const struct hufd_dec_result *raw,
    const struct aws_huffman_amd_decode_item *item,
    struct aws_huffman_amd_decode_result *out) {

    if (raw->total_symbols > item->out_capacity) {
        /* symbol number out_capacity was recognised with nowhere to go: huffman.c:257-268 */
        out->rc = AWS_OP_ERR;
        out->error = AWS_ERROR_SHORT_BUFFER;
        out->produced = item->out_capacity;
        out->bits_consumed = raw->cap_bit - item->first_bit;
        return;
    }
    out->produced = raw->total_symbols;
    out->bits_consumed = raw->stop_bit - item->first_bit;
    out->rc = AWS_OP_SUCCESS;
    out->error = 0;
    if (raw->stop_kind == HUFD_STOP_INVALID) {
        const uint64_t left = item->in_len * 8 - raw->stop_bit;
        if (left >= 32) { /* huffman.c:240-247 */
            out->rc = AWS_OP_ERR;
            out->error = AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL;
        }
    } else if (raw->stop_kind == HUFD_STOP_NONE) {
        out->rc = AWS_OP_ERR;
        out->error = AWS_ERROR_INVALID_STATE;
    }
}

bool aws_huffman_amd_decode_plan_is_quiet(const struct aws_huffman_amd_decode_plan *p) {
    return p->quiet;
}

int aws_huffman_amd_decode_plan_road(struct aws_huffman_amd_decode_plan *p, void *stream, uint32_t *road, uint32_t *detail) {
    *road = AWS_HUFFMAN_AMD_ROAD_TWO_PASS;
    if (detail) {
        detail[0] = detail[1] = 0;
    }
    (void)p;
    (void)stream;
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_decode_plan_results(
    struct aws_huffman_amd_decode_plan *p,
    struct aws_huffman_amd_decode_result *results,
    void *stream) {

    void *st = stream ? stream : p->engine->stream;
    uint8_t *fetched = malloc(DEC_SUMMARY_BYTES + (p->n_items ? p->n_items : 1) * sizeof(struct hufd_dec_result));
    if (!fetched) {
        return aws_raise_error(AWS_ERROR_OOM);
    }
    struct hufd_dec_result *raw = (struct hufd_dec_result *)(fetched + DEC_SUMMARY_BYTES);
    ON_DEVICE(p->engine->device);
    /* (the records and, in front of them, what the last launch's lists held: one copy) */
    int err = hufs_copy_d2h(fetched, p->d_summary, DEC_SUMMARY_BYTES + (size_t)p->n_items * sizeof(*raw), st);
    if (!err) {
        err = hufs_stream_sync(st);
    }
    if (err) {
        free(fetched);
        return raise_hip(err);
    }
    if (p->n_chunks && p->launches_with_chunks) {
        /* A launch that listed no chunk for any kernel but the regular ones (an ordinary stream of a coder whose walks fall
         * into step): the plan's next launches go without the kernels that only make listed chunks faster -- four empty
         * launches of ~4 us each.  A launch of a quiet plan that does list chunks sends them the long way (exact for every
         * chunk, a fifth of the speed), says so here, and the kernels are back from the next launch on. */
        const uint32_t *listed = (const uint32_t *)fetched;
        p->quiet = !(listed[HUFK_DEC_COUNT_SLOW] | listed[HUFK_DEC_COUNT_LONG] | listed[HUFK_DEC_COUNT_FEW] |
                     listed[HUFK_DEC_COUNT_EMIT] | listed[HUFK_DEC_COUNT_DENSE]);
    }
    if (p->chained) {
        /* the items' lengths were never on the host: the device's records say what each result is a result of */
        struct hufd_dec_item *dev_items = malloc((p->n_items ? p->n_items : 1) * sizeof(*dev_items));
        err = dev_items ? hufs_copy_d2h(dev_items, p->d_items, (size_t)p->n_items * sizeof(*dev_items), st) : 2;
        if (!err) {
            err = hufs_stream_sync(st);
        }
        for (uint32_t i = 0; i < p->n_items && !err; ++i) {
            struct aws_huffman_amd_decode_item it;
            memset(&it, 0, sizeof(it));
            it.in_offset = dev_items[i].in_off;
            it.in_len = dev_items[i].in_len;
            it.first_bit = (uint8_t)dev_items[i].first_bit;
            it.out_offset = dev_items[i].out_off;
            it.out_capacity = dev_items[i].out_cap;
            aws_huffman_amd_decode_result_from_raw(&raw[i], &it, &results[i]);
        }
        free(dev_items);
        free(fetched);
        return err ? raise_hip(err) : AWS_OP_SUCCESS;
    }
    for (uint32_t i = 0; i < p->n_items; ++i) {
        aws_huffman_amd_decode_result_from_raw(&raw[i], &p->h_items[i], &results[i]);
    }
    free(fetched);
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_decode_plan_from_encode(
    struct aws_huffman_amd_decode_plan *p,
    const struct aws_huffman_amd_encode_plan *encoded,
    void *stream) {

    struct aws_huffman_amd_engine *eng = p->engine;
    if (!eng->can_decode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    if (!encoded || encoded->engine->device != eng->device || !encoded->launched) {
        return aws_raise_error(AWS_ERROR_INVALID_ARGUMENT); /* (a plan that was never launched has no records to read lengths from) */
    }
    /* Whatever the launch produced, every item must be ONE THREAD's work for the decoder (then the plan has no chunk
     * geometry, which only the host can lay out): the most bytes an item can have left is its output capacity.  The same
     * rule as for a plan from host records (dec_tiny_limit), with the capacities for the lengths. */
    const uint64_t n_items = encoded->n_items, longest = encoded->largest_out_cap;
    bool thread_each = n_items >= 1 && n_items < 0xFFFFFFFFull && longest <= HUFD_TINY_FEW_BYTES;
    {
        static const uint64_t classes[2] = {HUFD_DEC_COOP_BYTES, HUFD_DEC_TINY_BYTES};
        for (int c = 0; c < 2 && !thread_each && n_items >= 1 && n_items < 0xFFFFFFFFull; ++c) {
            thread_each = longest <= classes[c] && n_items >= tiny_per_byte(true) * longest;
        }
    }
    if (!thread_each) {
        /* items with chunks (or too few short ones for a thread each): the general plan, made on the device from the
         * launch's records -- the lengths stay there; what comes back are the few totals that size the plan */
        struct hufd_item_source src;
        memset(&src, 0, sizeof(src));
        src.kind = HUFD_ITEMS_FROM_ENCODE;
        src.enc_items = encoded->d_items;
        src.enc_results = encoded->d_results;
        return dec_plan_fill_on_device(p, &src, (size_t)n_items, stream);
    }
    p->n_items = p->n_chunks = p->n_large = p->n_runs = p->n_tail = p->n_fixed = p->n_wide = 0;
    p->quiet = false; /* (other items: nothing is known of what their launches list) */
    p->n_tiny = p->n_deep = 0;
    memset(&p->stats, 0, sizeof(p->stats));
    p->chained = false;
    ON_DEVICE(eng->device);
    int e = dec_plan_reserve(p, n_items, 0, 0, 0);
    if (!e) {
        e = hufk_decode_plan_from_encode(
            encoded->d_items, encoded->d_results, (uint32_t)n_items, p->d_items, p->d_tiny, stream ? stream : eng->stream);
    }
    if (e) {
        return raise_hip(e);
    }
    p->wide_from = wide_min_bytes(0);
    p->tail_stage_bytes = 0;
    p->tail_lanes = 0;
    p->tail_wide_lanes = 0;
    p->n_tail_narrow = 0;
    p->n_items = (uint32_t)n_items;
    p->n_tiny = (uint32_t)n_items;
    p->chained = true;
    p->stats.items = p->stats.by_thread = n_items;
    p->stats.thread_limit = longest;
    e = plan_mark_done(&p->done_event, &p->done_on_engine_stream, eng, stream); /* (hufk_decode_plan_from_encode is in flight) */
    return e ? raise_hip(e) : AWS_OP_SUCCESS;
}

/* ------------------------------------------------------------------ one-item helpers for the host-pointer API */

static int one_shot_reserve(struct aws_huffman_amd_engine *eng, size_t in_bytes, size_t out_bytes) {
    ON_DEVICE(eng->device);
    if (in_bytes > eng->one_in_cap) {
        hufs_free(eng->one_in);
        eng->one_in_cap = in_bytes + in_bytes / 4 + 4096;
        eng->one_in = hufs_malloc(eng->one_in_cap);
        if (!eng->one_in) {
            eng->one_in_cap = 0;
            return aws_raise_error(AWS_ERROR_OOM);
        }
    }
    if (out_bytes > eng->one_out_cap) {
        hufs_free(eng->one_out);
        eng->one_out_cap = out_bytes + out_bytes / 4 + 4096;
        eng->one_out = hufs_malloc(eng->one_out_cap);
        if (!eng->one_out) {
            eng->one_out_cap = 0;
            return aws_raise_error(AWS_ERROR_OOM);
        }
    }
    return AWS_OP_SUCCESS;
}

/*
 * Inputs of up to a few KiB (what the reference's HPACK consumer passes is tens of bytes, one call per header field):
 * the item record and the input go up in ONE copy from a page-locked block, ONE launch does the work -- one thread
 * (enc_tiny / dec_tiny) up to MINI_MAX_IN bytes, beyond that one workgroup (enc_block, dec_block; with long codes one
 * wave of dec_deep) --, the result record and the output come back in ONE copy.  The general road costs a plan upload, four
 * small pageable copies, four or five launches and two synchronisations for the same call.
 */
enum {
    MINI_MAX_IN = 128,     /* symbols to encode / encoded bytes (carried ones included) that are ONE THREAD's work */
    MINI_IN_AT = 64,       /* block layout: [0] item record, [64] input (decode: 16 bytes for the carried ones first), */
    MINI_ZERO_AT = 32896,  /* [32896] zero word, [32960] scratch, */
    MINI_SCRATCH_AT = 32960,
    MINI_RESULT_AT = 33024, /* [33024] result record, [33088] output */
    MINI_OUT_AT = 33088,
    MINI_BLOCK = 131072,
    MINI_MAX_OUT = MINI_BLOCK - MINI_OUT_AT
};

static bool mini_ready(struct aws_huffman_amd_engine *eng) {
    if (eng->mini_host && eng->mini_dev) {
        return true;
    }
    ON_DEVICE(eng->device);
    if (!eng->mini_host) {
        eng->mini_host = hufs_host_alloc(MINI_BLOCK);
    }
    if (!eng->mini_dev) {
        eng->mini_dev = hufs_malloc(MINI_BLOCK);
        if (eng->mini_dev && (hufs_memset(eng->mini_dev, 0, MINI_BLOCK, eng->stream) || hufs_stream_sync(eng->stream))) {
            hufs_free(eng->mini_dev);
            eng->mini_dev = NULL;
        }
    }
    return eng->mini_host && eng->mini_dev;
}

static int mini_encode(
    struct aws_huffman_amd_engine *eng,
    const struct aws_huffman_amd_encode_item *item,
    uint64_t dev_out,
    const uint8_t *host_in,
    uint8_t *host_out,
    bool length_only,
    struct hufd_enc_result *raw) {

    struct hufd_enc_item rec;
    memset(&rec, 0, sizeof(rec));
    rec.in_off = MINI_IN_AT;
    rec.in_len = item->in_len;
    rec.out_off = MINI_OUT_AT;
    rec.out_cap = dev_out;
    rec.ovf_bits = item->overflow_in.num_bits;
    rec.ovf_pattern = rec.ovf_bits >= 32 ? item->overflow_in.pattern
                                         : item->overflow_in.pattern & ((1u << rec.ovf_bits) - 1u);
    rec.eos_padding = item->eos_padding;
    rec.tiny = 1;
    rec.n_segs = 0;
    memcpy(eng->mini_host, &rec, sizeof(rec));
    if (item->in_len) { /* an empty cursor may carry a NULL pointer (source/huffman.c:161-167 never touches it) */
        memcpy(eng->mini_host + MINI_IN_AT, host_in, item->in_len);
    }
    ON_DEVICE(eng->device);
    int err = hufs_copy_h2d(eng->mini_dev, eng->mini_host, MINI_IN_AT + item->in_len, eng->stream);
    if (!err && item->in_len <= MINI_MAX_IN) {
        err = hufk_encode_one_tiny(
            &eng->tables, (const struct hufd_enc_item *)eng->mini_dev, (const uint32_t *)(eng->mini_dev + MINI_ZERO_AT),
            eng->mini_dev, eng->mini_dev, (struct hufd_enc_result *)(eng->mini_dev + MINI_RESULT_AT), length_only,
            eng->stream);
    } else if (!err) {
        err = hufk_encode_one_block(
            &eng->tables, (const struct hufd_enc_item *)eng->mini_dev, (uint32_t)item->in_len, eng->mini_dev, eng->mini_dev,
            (struct hufd_enc_result *)(eng->mini_dev + MINI_RESULT_AT), length_only, eng->stream);
    }
    if (!err) {
        const size_t back = (MINI_OUT_AT - MINI_RESULT_AT) + (length_only ? 0 : dev_out);
        err = hufs_copy_d2h(eng->mini_host + MINI_RESULT_AT, eng->mini_dev + MINI_RESULT_AT, back, eng->stream);
    }
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    if (err) {
        return raise_hip(err);
    }
    memcpy(raw, eng->mini_host + MINI_RESULT_AT, sizeof(*raw));
    if (!length_only && raw->produced) {
        memcpy(host_out, eng->mini_host + MINI_OUT_AT, raw->produced);
    }
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_engine_encode_host(
    struct aws_huffman_amd_engine *eng,
    const struct aws_huffman_amd_encode_item *item_in,
    const uint8_t *host_in,
    uint8_t *host_out,
    bool length_only,
    struct hufd_enc_result *raw) {

    struct aws_huffman_amd_encode_item item = *item_in;
    /*
     * Only the symbols this call can consume are staged.  The reference reads symbol k only while the output has a
     * free byte (source/huffman.c:162-164), i.e. while the bits in front of it -- carried ones + at least min_bits
     * per earlier symbol -- are fewer than 8 * room: at most ceil((8 * room - carried) / min_bits) symbols, one more
     * here so that a call that fills its room exactly still sees input behind it (and says SHORT_BUFFER, as the
     * reference would at the top of its next turn).  A caller that offers its output a few bytes at a time
     * (huffman_test_transitive_chunked, HPACK's SHORT_BUFFER resumption) otherwise pays for the whole rest of its
     * input in every call: O(n^2 / chunk) bytes over the bus and through the count.
     */
    if (!length_only && eng->tables.enc_min_bits && item.out_capacity < (UINT64_MAX >> 4)) {
        const uint64_t room_bits = item.out_capacity * 8;
        const uint64_t carried = item.overflow_in.num_bits;
        const uint64_t fit = room_bits > carried ? (room_bits - carried + eng->tables.enc_min_bits - 1) / eng->tables.enc_min_bits : 0;
        if (fit + 1 < item.in_len) {
            item.in_len = fit + 1;
        }
    }
    /* the device output never needs more than the worst-case encoding */
    const uint64_t worst = (item.in_len * eng->tables.enc_max_bits + item.overflow_in.num_bits + 7) / 8;
    const uint64_t dev_out = item.out_capacity < worst ? item.out_capacity : worst;
    item.in_offset = 0;
    item.out_offset = 0;
    if (hufk_encode_one_block_fits(&eng->tables, item.in_len) && (item.in_len || item.overflow_in.num_bits) && item.overflow_in.num_bits <= 32 &&
        (length_only || dev_out <= MINI_MAX_OUT) && mini_ready(eng)) {
        return mini_encode(eng, &item, dev_out, host_in, host_out, length_only, raw);
    }
    if (one_shot_reserve(eng, item.in_len + 16, length_only ? 0 : dev_out + 16)) {
        return AWS_OP_ERR;
    }
    if (!eng->one_enc) {
        eng->one_enc = calloc(1, sizeof(*eng->one_enc));
        if (!eng->one_enc) {
            return aws_raise_error(AWS_ERROR_OOM);
        }
        eng->one_enc->engine = eng;
    }
    if (enc_plan_fill(eng->one_enc, &item, 1)) {
        return AWS_OP_ERR;
    }
    int err = hufs_copy_h2d(eng->one_in, host_in, item.in_len, eng->stream);
    if (err) {
        return raise_hip(err);
    }
    if (aws_huffman_amd_encode_plan_launch(eng->one_enc, eng->one_in, eng->one_out, length_only, eng->stream)) {
        return AWS_OP_ERR;
    }
    if (aws_huffman_amd_encode_plan_raw_results(eng->one_enc, raw, eng->stream)) {
        return AWS_OP_ERR;
    }
    if (!length_only && raw->produced) {
        err = hufs_copy_d2h(host_out, eng->one_out, raw->produced, eng->stream);
        if (!err) {
            err = hufs_stream_sync(eng->stream);
        }
        if (err) {
            return raise_hip(err);
        }
    }
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_engine_decode_host(
    struct aws_huffman_amd_engine *eng,
    const uint8_t *carry,
    uint32_t carry_bytes,
    uint32_t first_bit,
    const uint8_t *host_in,
    uint64_t in_len,
    uint64_t out_capacity,
    struct aws_huffman_amd_decode_result *result) {

    if (!eng->can_decode) {
        return aws_raise_error(AWS_ERROR_UNSUPPORTED_OPERATION);
    }
    /*
     * Only the bytes this call can get through are staged: with room for r symbols the reference stops at the latest
     * when it has recognised symbol r + 1 (source/huffman.c:257-268), which lies within (r + 1) * max_bits bits of
     * the stream's first one; eight bytes more keep a code without a match "at least 32 bits from the end" exactly
     * when it is so in the whole stream (source/huffman.c:240-247).  (The caller's cursor and the decoder's window
     * are worked out from the bits consumed and the real length, in huffman.c.)
     */
    if (out_capacity < (UINT64_MAX >> 8) && eng->tables.max_bits) {
        const uint64_t need_bits = first_bit + (out_capacity + 1) * (eng->tables.max_bits ? eng->tables.max_bits : 32);
        const uint64_t need_bytes = (need_bits + 7) / 8 + 8;
        if (need_bytes < carry_bytes + in_len) {
            in_len = need_bytes > carry_bytes ? need_bytes - carry_bytes : 0;
        }
    }
    /* the carried bytes sit right before the new bytes, which start 16-byte aligned */
    const uint64_t stream_bits = (carry_bytes + in_len) * 8 - first_bit;
    const uint64_t most_symbols = stream_bits / eng->tables.min_bits;
    const uint64_t dev_out = out_capacity < most_symbols ? out_capacity : most_symbols;
    eng->mini_output = false;
    /* (decode: one thread, or one workgroup up to HUFD_DEC_BLOCK_BYTES -- with long codes one wave up to
     * HUFD_DEC_COOP_BYTES; the chunk kernels take what is longer) */
    /* (codes of one length: dec_block's lanes would never settle -- a thread, or the plan's dec_fixed kernels) */
    const uint64_t mini_in =
        eng->tables.deep_entries ? HUFD_DEC_COOP_BYTES : (eng->tables.fixed_bits ? MINI_MAX_IN : HUFD_DEC_BLOCK_MAX_BYTES);
    if (carry_bytes + in_len <= mini_in && carry_bytes + in_len > 0 && dev_out <= MINI_MAX_OUT && mini_ready(eng)) {
        struct hufd_dec_item rec;
        memset(&rec, 0, sizeof(rec));
        rec.in_off = MINI_IN_AT + 16 - carry_bytes;
        rec.in_len = carry_bytes + in_len;
        rec.out_off = MINI_OUT_AT;
        rec.out_cap = dev_out;
        rec.first_bit = first_bit;
        rec.tiny = 1;
        memcpy(eng->mini_host, &rec, sizeof(rec));
        memcpy(eng->mini_host + MINI_IN_AT + 16 - carry_bytes, carry, carry_bytes);
        if (in_len) { /* carried bits and an empty cursor, which may be {0, NULL} (source/huffman.c:196-211) */
            memcpy(eng->mini_host + MINI_IN_AT + 16, host_in, in_len);
        }
        ON_DEVICE(eng->device);
        int e = hufs_copy_h2d(eng->mini_dev, eng->mini_host, MINI_IN_AT + 16 + in_len, eng->stream);
        if (!e) {
            /* a lone thread up to MINI_MAX_IN bytes, a workgroup (a wave with long codes) above */
            if (carry_bytes + in_len <= MINI_MAX_IN) {
                e = hufk_decode_one_tiny(
                    &eng->tables, (const struct hufd_dec_item *)eng->mini_dev, (const uint32_t *)(eng->mini_dev + MINI_ZERO_AT),
                    eng->mini_dev, eng->mini_dev, (struct hufd_dec_item_state *)(eng->mini_dev + MINI_SCRATCH_AT),
                    (struct hufd_dec_result *)(eng->mini_dev + MINI_RESULT_AT), eng->stream);
            } else if (eng->tables.deep_entries) {
                e = hufk_decode_one_coop(
                    &eng->tables, (const struct hufd_dec_item *)eng->mini_dev, (const uint32_t *)(eng->mini_dev + MINI_ZERO_AT),
                    eng->mini_dev, eng->mini_dev, (struct hufd_dec_item_state *)(eng->mini_dev + MINI_SCRATCH_AT),
                    (struct hufd_dec_result *)(eng->mini_dev + MINI_RESULT_AT), eng->stream);
            } else {
                e = hufk_decode_one_block(
                    &eng->tables, &rec, eng->mini_dev, eng->mini_dev,
                    (struct hufd_dec_item_state *)(eng->mini_dev + MINI_SCRATCH_AT),
                    (struct hufd_dec_result *)(eng->mini_dev + MINI_RESULT_AT), eng->stream);
            }
        }
        if (!e) {
            e = hufs_copy_d2h(
                eng->mini_host + MINI_RESULT_AT, eng->mini_dev + MINI_RESULT_AT, (MINI_OUT_AT - MINI_RESULT_AT) + dev_out,
                eng->stream);
        }
        if (!e) {
            e = hufs_stream_sync(eng->stream);
        }
        if (e) {
            return raise_hip(e);
        }
        struct hufd_dec_result raw;
        memcpy(&raw, eng->mini_host + MINI_RESULT_AT, sizeof(raw));
        if (raw.stop_kind == HUFD_STOP_GAVE_UP) {
            goto chunked; /* a stream dec_block's lanes do not settle on: nothing is decoded yet */
        }
        struct aws_huffman_amd_decode_item as_item;
        memset(&as_item, 0, sizeof(as_item));
        as_item.in_len = carry_bytes + in_len;
        as_item.first_bit = first_bit;
        as_item.out_capacity = dev_out;
        aws_huffman_amd_decode_result_from_raw(&raw, &as_item, result);
        eng->mini_output = true;
        if (result->error == AWS_ERROR_SHORT_BUFFER && dev_out < out_capacity) {
            return aws_raise_error(AWS_ERROR_INVALID_STATE);
        }
        return AWS_OP_SUCCESS;
    }
chunked:
    if (one_shot_reserve(eng, 16 + in_len + 16, dev_out + 16)) {
        return AWS_OP_ERR;
    }
    struct aws_huffman_amd_decode_item item;
    memset(&item, 0, sizeof(item));
    item.in_offset = 16 - carry_bytes;
    item.in_len = carry_bytes + in_len;
    item.first_bit = first_bit;
    item.out_offset = 0;
    item.out_capacity = dev_out;
    if (!eng->one_dec) {
        eng->one_dec = calloc(1, sizeof(*eng->one_dec));
        if (!eng->one_dec) {
            return aws_raise_error(AWS_ERROR_OOM);
        }
        eng->one_dec->engine = eng;
    }
    if (dec_plan_fill(eng->one_dec, &item, 1)) {
        return AWS_OP_ERR;
    }
    int err = 0;
    if (carry_bytes) {
        err = hufs_copy_h2d((uint8_t *)eng->one_in + 16 - carry_bytes, carry, carry_bytes, eng->stream);
    }
    if (!err) {
        err = hufs_copy_h2d((uint8_t *)eng->one_in + 16, host_in, in_len, eng->stream);
    }
    if (err) {
        return raise_hip(err);
    }
    if (aws_huffman_amd_decode_plan_launch(eng->one_dec, eng->one_in, eng->one_out, eng->stream)) {
        return AWS_OP_ERR;
    }
    if (aws_huffman_amd_decode_plan_results(eng->one_dec, result, eng->stream)) {
        return AWS_OP_ERR;
    }
    /* the device capacity was clipped to what the stream can hold; undo that in the verdict */
    if (result->error == AWS_ERROR_SHORT_BUFFER && dev_out < out_capacity) {
        return aws_raise_error(AWS_ERROR_INVALID_STATE);
    }
    return AWS_OP_SUCCESS;
}

int aws_huffman_amd_engine_fetch_output(struct aws_huffman_amd_engine *eng, uint8_t *host_out, uint64_t size) {
    if (eng->mini_output) {
        if (size) {
            memcpy(host_out, eng->mini_host + MINI_OUT_AT, size); /* came back with the result record */
        }
        return AWS_OP_SUCCESS;
    }
    int err = hufs_copy_d2h(host_out, eng->one_out, size, eng->stream);
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

/* ------------------------------------------------------------------ device helpers */

int aws_huffman_amd_device_count(void) {
    return hufs_device_count();
}

void *aws_huffman_amd_device_alloc(struct aws_huffman_amd_engine *eng, size_t size) {
    ON_DEVICE(eng->device);
    void *p = hufs_malloc(size);
    if (!p) {
        aws_raise_error(AWS_ERROR_OOM);
    }
    return p;
}

void aws_huffman_amd_device_free(struct aws_huffman_amd_engine *eng, void *ptr) {
    ON_DEVICE(eng->device);
    hufs_free(ptr);
}

int aws_huffman_amd_copy_to_device(struct aws_huffman_amd_engine *eng, void *dst, const void *src, size_t size) {
    ON_DEVICE(eng->device);
    int err = hufs_copy_h2d(dst, src, size, eng->stream);
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_copy_to_host(struct aws_huffman_amd_engine *eng, void *dst, const void *src, size_t size) {
    ON_DEVICE(eng->device);
    int err = hufs_copy_d2h(dst, src, size, eng->stream);
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_device_fill(struct aws_huffman_amd_engine *eng, void *dst, int byte, size_t size) {
    ON_DEVICE(eng->device);
    int err = hufs_memset(dst, byte, size, eng->stream);
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_device_fill_splitmix64(struct aws_huffman_amd_engine *eng, void *dst, size_t size, uint64_t seed) {
    ON_DEVICE(eng->device);
    int err = hufk_fill_splitmix64(dst, size, seed, eng->stream);
    if (!err) {
        err = hufs_stream_sync(eng->stream);
    }
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_stream_synchronize(struct aws_huffman_amd_engine *eng, void *stream) {
    ON_DEVICE(eng->device);
    const int err = hufs_stream_sync(stream ? stream : eng->stream);
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

void *aws_huffman_amd_event_new(struct aws_huffman_amd_engine *eng) {
    ON_DEVICE(eng->device);
    return hufs_event_create();
}

void aws_huffman_amd_event_destroy(struct aws_huffman_amd_engine *eng, void *event) {
    ON_DEVICE(eng->device);
    hufs_event_destroy(event);
}

int aws_huffman_amd_event_record(struct aws_huffman_amd_engine *eng, void *event, void *stream) {
    ON_DEVICE(eng->device);
    const int err = hufs_event_record(event, stream ? stream : eng->stream);
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}

int aws_huffman_amd_event_elapsed_ms(struct aws_huffman_amd_engine *eng, void *start, void *stop, float *ms) {
    ON_DEVICE(eng->device);
    const int err = hufs_event_elapsed_ms(start, stop, ms);
    return err ? raise_hip(err) : AWS_OP_SUCCESS;
}
