#ifndef AWS_COMPRESSION_HUFFMAN_AMD_H
#define AWS_COMPRESSION_HUFFMAN_AMD_H
/*
 * MI355X extensions of the Huffman C ABI: device-pointer and batched entry
 * points.  The reference has no counterpart (its API is host pointers, one
 * stream per call); the meaning of every item processed here is defined as
 * "what the reference function would do for that item":
 *
 *   encode item  ==  aws_huffman_encode   (reference source/huffman.c:131-187)
 *                    on an encoder with the item's overflow_in / eos_padding,
 *                    a cursor over the item's input range and a byte_buf with
 *                    len 0 and capacity out_capacity over the item's output range;
 *   decode item  ==  aws_huffman_decode   (reference source/huffman.c:213-286)
 *                    on a freshly reset decoder whose stream starts `first_bit`
 *                    bits into the item's first input byte.
 *
 * The host-pointer functions in huffman.h are thin wrappers over these: they
 * stage the caller's bytes into device memory, run one item, and copy back.
 *
 * Plain C, plain pointers and sizes only (no HIP or torch types): `stream`
 * arguments are a hipStream_t passed as void *, device pointers are void *.
 * All functions return AWS_OP_SUCCESS / AWS_OP_ERR and raise through
 * aws_raise_error like the rest of the CRT.
 */

#include <aws/compression/huffman.h>

AWS_EXTERN_C_BEGIN

/* One device + the tables of one symbol coder staged in its memory + a stream. */
struct aws_huffman_amd_engine;

/* A batch of encode (or decode) items laid out against one input and one output
 * base pointer, with its segment map and scratch memory resident on the device. */
struct aws_huffman_amd_encode_plan;
struct aws_huffman_amd_decode_plan;

struct aws_huffman_amd_encode_item {
    uint64_t in_offset;    /* bytes from the input base pointer */
    uint64_t in_len;       /* symbols to encode */
    uint64_t out_offset;   /* bytes from the output base pointer */
    uint64_t out_capacity; /* bytes the item may write */
    struct aws_huffman_code overflow_in; /* encoder->overflow_bits on entry (num_bits 0 = none) */
    uint8_t eos_padding;   /* encoder->eos_padding */
};

struct aws_huffman_amd_encode_result {
    int32_t rc;        /* AWS_OP_SUCCESS or AWS_OP_ERR */
    int32_t error;     /* 0, AWS_ERROR_SHORT_BUFFER or AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL */
    uint64_t consumed; /* symbols the cursor would have advanced by */
    uint64_t produced; /* bytes appended to the output */
    struct aws_huffman_code overflow_out; /* encoder->overflow_bits on return */
};

struct aws_huffman_amd_decode_item {
    uint64_t in_offset;
    uint64_t in_len;       /* encoded bytes */
    uint32_t first_bit;    /* 0..7: bits of the first byte that are already consumed */
    uint64_t out_offset;
    uint64_t out_capacity; /* symbols the item may write */
};

struct aws_huffman_amd_decode_result {
    int32_t rc;
    int32_t error;          /* 0, AWS_ERROR_SHORT_BUFFER or AWS_ERROR_COMPRESSION_UNKNOWN_SYMBOL */
    uint64_t produced;      /* symbols written */
    uint64_t bits_consumed; /* stream bits used by those symbols, counted from first_bit */
};

/*
 * Tabulates `coder` (256 encode calls; decode calls for every index of the
 * decode tables with zero and one fill, cross-checked against the encode
 * table) and stages the tables on HIP device `device` (-1 = current device).
 * Codes of up to 32 bits either way; up to 12 bits decode through the chunked
 * kernels, longer ones one thread or one workgroup per item.  A decode callback that is not
 * table-shaped leaves the engine encode-only
 * (aws_huffman_amd_engine_can_decode() == false; decode entry points raise
 * AWS_ERROR_UNSUPPORTED_OPERATION).  There is no CPU path to fall back to.
 *
 * An engine (its plans included) is one host thread's at a time: launches of one
 * engine's plans are made one after the other -- they share the engine's second
 * stream and its events.  Threads that work side by side take an engine each
 * (aws_huffman_amd_shards does that per device).
 */
AWS_COMPRESSION_API
int aws_huffman_amd_engine_new(
    struct aws_huffman_amd_engine **engine,
    struct aws_huffman_symbol_coder *coder,
    int device);

AWS_COMPRESSION_API
void aws_huffman_amd_engine_destroy(struct aws_huffman_amd_engine *engine);

/* Longest code of the staged coder, and whether the engine can decode. */
AWS_COMPRESSION_API
uint32_t aws_huffman_amd_engine_max_code_bits(const struct aws_huffman_amd_engine *engine);
AWS_COMPRESSION_API
bool aws_huffman_amd_engine_can_decode(const struct aws_huffman_amd_engine *engine);

/* Whether encode plans of this engine run as one kernel that reads the symbols once (coders whose 256 symbols all have
 * codes of 4 .. 15 bits) or as count / scan / pack: which kernels the
 * stage events of aws_huffman_amd_encode_plan_launch_staged bracket depends on it. */
AWS_COMPRESSION_API
bool aws_huffman_amd_engine_encodes_in_one_pass(const struct aws_huffman_amd_engine *engine);

/* ---- batched encode ------------------------------------------------------ */

/* Uploads the items, builds the segment map, sizes the scratch memory. */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_new(
    struct aws_huffman_amd_encode_plan **plan,
    struct aws_huffman_amd_engine *engine,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count);

/* The plan for other items (its device arrays are kept where they are large enough: no allocation in the steady state).
 * The arrays are rewritten on the engine's stream: a launch of this plan that is still in flight on ANOTHER stream must
 * have finished (the caller waits for that stream, or fetched the launch's results, which does).  If the call fails the
 * plan holds no items: launching it is a no-op until a reset succeeds.  What making a plan costs the host:
 * profiles/tools/plan_time.py (BASELINE configs[3]'s 65 536 buffers: 0.4 ms encode / 0.5 ms decode a reset; a plan whose
 * items are ALL one thread's work -- header-sized strings -- and number at least 4096 is made on the device from the
 * caller's records as they are: 2..3 ms for a million of them). */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_reset(
    struct aws_huffman_amd_encode_plan *plan,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count);

/* The engine keeps the device arrays of ONE destroyed plan of each kind (encode, decode) for its next
 * aws_huffman_amd_*_plan_new, which waits for the device before it rewrites them (as freeing them did): in the steady
 * state of make / launch / destroy a new plan costs what a reset costs and allocates nothing.  The arrays go back to the
 * device with the engine; `plan` must not be used after this call either way. */
AWS_COMPRESSION_API
void aws_huffman_amd_encode_plan_destroy(struct aws_huffman_amd_encode_plan *plan);

/*
 * Enqueues the encode kernels on `stream` (NULL = the engine's stream) and
 * returns without waiting.  `length_only` stops after the length scan: nothing
 * is written and each result's `produced` is aws_huffman_get_encoded_length
 * (+ pending overflow bits) for the item.
 *
 * What the launch leaves behind it on the stream is the finished output: work queued on the same stream after it (a
 * decode launch of the plan's output, a copy, a graph node) may read device_output without fetching the results first.
 * (The one-pass encoder's waits are bounded; a wave that gives up raises a word that the count / scan / pack kernels,
 * queued behind it in the same launch, look at first: they do the launch over on the device.  The road a launch took is
 * reported by aws_huffman_amd_encode_plan_road once its results have been fetched.)
 */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_launch(
    struct aws_huffman_amd_encode_plan *plan,
    const void *device_input,
    void *device_output,
    bool length_only,
    void *stream);

/*
 * Same launch with HIP events recorded between its kernels, for per-kernel timing on the
 * stream the kernels run on.  stage_events: 4 events from aws_huffman_amd_event_new --
 * [0] before the length count, [1] after it, [2] after the offset scan, [3] after the pack;
 * for an engine that encodes in one pass: [0] before the one-pass kernel, [1] after it,
 * [2] after the per-item outcomes, [3] after the (gated, normally empty) kernels of the count / scan / pack road.
 */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_launch_staged(
    struct aws_huffman_amd_encode_plan *plan,
    const void *device_input,
    void *device_output,
    bool length_only,
    void *stream,
    void **stage_events);

/* Waits for the last launch and copies the per-item results to the host. */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_results(
    struct aws_huffman_amd_encode_plan *plan,
    struct aws_huffman_amd_encode_result *results,
    void *stream);

/*
 * Capacity planning (aws_huffman_get_encoded_length, reference source/huffman.c:107-129,
 * for every item of a plan): after a launch with length_only = true, lengths[i] = bytes item
 * i appends when it has room for all of them -- ceil((carried overflow bits + code bits) / 8);
 * symbols without a code count as 0 bits, as in the reference.
 */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_encoded_lengths(
    struct aws_huffman_amd_encode_plan *plan,
    uint64_t *lengths,
    void *stream);

/* ---- batched decode ------------------------------------------------------ */

AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_new(
    struct aws_huffman_amd_decode_plan **plan,
    struct aws_huffman_amd_engine *engine,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count);

/* (as aws_huffman_amd_encode_plan_reset: the plan's previous launch must have finished; a failed call leaves a plan
 * without items) */
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_reset(
    struct aws_huffman_amd_decode_plan *plan,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count);

/*
 * Plans without a loop over the items on the host: the items are DESCRIBED (a stride: a batch of equal buffers), or their
 * records already LIE in device memory (made there by the caller's own kernels, or uploaded once and used again).  The
 * plan's records, segments / chunks and lists are made by a few small launches on `stream` (NULL: the engine's); the
 * call waits once, for a handful of totals that size the plan's arrays and the launch's grids -- O(1) host work whatever
 * the number of items (BASELINE configs[3]'s 65 536 buffers: a tenth of a millisecond a plan where the host's loop, its
 * arrays and their copies took 0.4 - 1.9 ms).  The plan is then good for launches on the same stream, or on another once
 * that stream has been waited for.  Per item nothing changes: what aws_huffman_encode / aws_huffman_decode do for it.
 * As for the _reset calls: the plan's previous launch must have finished; a failed call leaves a plan without items.
 * (Coders with codes of more than 12 bits, or of one length: the decode plan needs what only the host lays out -- the
 * items are brought to the host and its loop makes the plan; same results, not the same speed.)
 */
struct aws_huffman_amd_strided_items {
    uint64_t count;
    uint64_t in_offset;    /* item i reads in_len bytes at in_offset + i * in_stride of the launch's input ... */
    uint64_t in_stride;
    uint64_t in_len;       /* (symbols to encode / encoded bytes to decode) */
    uint64_t out_offset;   /* ... and writes at out_offset + i * out_stride of its output, */
    uint64_t out_stride;
    uint64_t out_capacity; /* at most this many bytes */
    uint8_t first_bit;     /* decode: as aws_huffman_amd_decode_item.first_bit, the same for every item */
    uint8_t eos_padding;   /* encode: as aws_huffman_amd_encode_item.eos_padding (no carried overflow bits) */
};
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_reset_strided(
    struct aws_huffman_amd_encode_plan *plan, const struct aws_huffman_amd_strided_items *items, void *stream);
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_reset_strided(
    struct aws_huffman_amd_decode_plan *plan, const struct aws_huffman_amd_strided_items *items, void *stream);
/* device_items: item_count records of the public layout in DEVICE memory, readable on `stream` */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_reset_device_items(
    struct aws_huffman_amd_encode_plan *plan, const struct aws_huffman_amd_encode_item *device_items, size_t item_count, void *stream);
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_reset_device_items(
    struct aws_huffman_amd_decode_plan *plan, const struct aws_huffman_amd_decode_item *device_items, size_t item_count, void *stream);

/*
 * `plan` reset to decode what `encoded`'s LAST LAUNCH produced, made on the device from that launch's records: the
 * encoded lengths never come to the host (encode -> decode of a batch without a round trip).  Decode item i is encode
 * item i's output -- at its out_offset in the buffer that launch wrote, which is the decode launch's input; as many
 * bytes as its record says were produced, whatever the record's verdict; from bit 0 -- decoded to where its symbols came
 * from: out_offset = the encode item's in_offset, out_capacity = its in_len.  The records are written on `stream`
 * (NULL: the engine's): behind the encode launch if that is the launch's stream, and in front of a decode launch on it.
 * A batch of short items -- every item, whatever it produced, one thread's work for the decoder (the most an item can have
 * left is its out_capacity: up to 128 bytes each, or up to 512 / 768 when the batch has thousands of them: header
 * fields) -- costs the host a few microseconds and no wait.  Any other batch (items with chunks: BASELINE configs[3]'s
 * 16 KiB buffers) is planned on the device as for aws_huffman_amd_decode_plan_reset_device_items, from the launch's
 * records: the call waits for the encode launch and a handful of totals; the lengths stay on the device.  Both plans on
 * one device; `encoded`'s last launch since it was filled must have written output -- never launched, or a length
 * query last (its records hold lengths of bytes nobody wrote): AWS_ERROR_INVALID_ARGUMENT, nothing changed.  A HIP failure inside the call leaves `plan` without items, as a failed reset does.
 */
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_from_encode(
    struct aws_huffman_amd_decode_plan *plan,
    const struct aws_huffman_amd_encode_plan *encoded,
    void *stream);

AWS_COMPRESSION_API
void aws_huffman_amd_decode_plan_destroy(struct aws_huffman_amd_decode_plan *plan);

AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_launch(
    struct aws_huffman_amd_decode_plan *plan,
    const void *device_input,
    void *device_output,
    void *stream);

/* stage_events: [0] before the sync kernel, [1] after it, [2] after the entry scan, [3] after the emit */
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_launch_staged(
    struct aws_huffman_amd_decode_plan *plan,
    const void *device_input,
    void *device_output,
    void *stream,
    void **stage_events);

AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_results(
    struct aws_huffman_amd_decode_plan *plan,
    struct aws_huffman_amd_decode_result *results,
    void *stream);

/* A launch lists the chunks its regular kernels do not take (a chunk whose walks do not fall into step, that is damaged,
 * that holds more symbols than the emit stage) for the long way -- dec_sync / dec_emit, exact for any chunk -- and for a
 * handful of kernels in front of it that take most listed chunks several times faster (dec_sync_guess, _few, _true,
 * dec_emit_big).  An ordinary stream lists nothing, and those four are then empty launches of ~4 us each.  The fetch of
 * a launch's results also brings back how many chunks it listed: none, and the plan is QUIET -- its next launches go
 * without the four, whatever they list goes the long way (the same results, a fifth of the speed for those chunks),
 * the next fetch says so and the four are back.  A plan is not quiet until a fetch has said so, and not after a reset.
 * (Diagnostics and tests; AWS_HUFFMAN_AMD_TEST_DECODE_ALL_KERNELS makes every launch queue all of them.) */
AWS_COMPRESSION_API
bool aws_huffman_amd_decode_plan_is_quiet(const struct aws_huffman_amd_decode_plan *plan);

#define AWS_HUFFMAN_AMD_ROAD_TWO_PASS 0u
#define AWS_HUFFMAN_AMD_ROAD_ONE_PASS 1u
#define AWS_HUFFMAN_AMD_ROAD_ONE_PASS_GAVE_UP 2u
/* Which kernels encoded the plan's last launch whose results were fetched (diagnostics and tests: the results are the same
 * either way; no wait of its own):
 *   TWO_PASS           count + scan + pack, the three-kernel road (every coder)
 *   ONE_PASS           enc_onepass: every symbol read once (coders with codes of 4..15 bits for all 256 symbols; the default)
 *   ONE_PASS_GAVE_UP   a look-back wait of enc_onepass ran out (the grid was not resident as a whole) and the three-kernel
 *                      kernels queued behind it on the same stream did the launch over; the plan stays on that road
 * Output and records are whole when the stream reaches the end of the launch, whichever road made them: work queued on
 * the stream behind a launch may use them without the host in between. */
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_road(const struct aws_huffman_amd_encode_plan *plan, uint32_t *road);

/* The decoder has one road for the chunks inside streams since round 5 (sync + scan + emit: TWO_PASS; the one-pass decoder
 * of rounds 3-4 measured no faster and is retired, DESIGN.md "Tried"); kept so that callers of the query still link. */
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_road(
    struct aws_huffman_amd_decode_plan *plan,
    void *stream,
    uint32_t *road,
    uint32_t *detail /* NULL, or two words: zero */);

/*
 * How a plan's items are taken, for diagnostics and tests that have to know WHICH kernels an item went through (the
 * results do not depend on it).  Counted when the plan is made or reset; nothing is launched, nothing waited for.
 */
struct aws_huffman_amd_plan_stats {
    uint64_t items;
    uint64_t thread_limit;    /* the longest item (symbols to encode / encoded bytes to decode) a lone thread takes in this plan */
    uint64_t by_thread;       /* items one THREAD takes (enc_tiny / dec_tiny) */
    uint64_t by_wave;         /* decode: items one WAVE takes (dec_deep, short codes) */
    uint64_t by_workgroup;    /* decode: items one WORKGROUP takes (dec_deep, codes of more than 12 bits) */
    uint64_t by_blocks;       /* decode: items taken a workgroup per block (long codes across the chip, codes of one length) */
    uint64_t by_pieces;       /* items cut into segments of 16 Ki symbols (encode) / chunks of 32 KiB (decode) ... */
    uint64_t pieces;          /* ... and how many of those there are */
    uint64_t end_pieces_packed; /* decode: chunks a stream ends in that share a workgroup with others (dec_sync_pack) ... */
    uint64_t end_pieces_single; /* ... and that have one of their own */
    uint64_t empty;           /* items with nothing to do */
    uint64_t end_pieces_folded; /* decode: chunks a stream ends in that are workgroups of the big kernels' own grids (a few among
                                 * many chunks inside streams: one long stream's one) */
};
AWS_COMPRESSION_API
int aws_huffman_amd_encode_plan_stats(const struct aws_huffman_amd_encode_plan *plan, struct aws_huffman_amd_plan_stats *stats);
AWS_COMPRESSION_API
int aws_huffman_amd_decode_plan_stats(const struct aws_huffman_amd_decode_plan *plan, struct aws_huffman_amd_plan_stats *stats);

/* ---- testing hooks: process-wide, for the library's own tests (none selects a road that is faster or slower by
 *      design -- they force the ways BACK the launches carry, so that those can be compared with the oracle too) ---- */

/* read when an ENGINE is made */
#define AWS_HUFFMAN_AMD_TEST_ENCODE_THREE_KERNEL 1u   /* count + scan + pack for every coder */
#define AWS_HUFFMAN_AMD_TEST_ENCODE_ONE_PASS_FAILS 2u /* a wave of enc_onepass gives up half-way: the kernels behind it do the launch over */
AWS_COMPRESSION_API
void aws_huffman_amd_testing_set_encode_road(uint32_t flags /* 0: back to the default */);
/* read at every decode LAUNCH */
#define AWS_HUFFMAN_AMD_TEST_DECODE_LONG_WAY 1u              /* chunks whose walks never fall into step: dec_sync + dec_emit, not dec_sync_few / _true */
#define AWS_HUFFMAN_AMD_TEST_DECODE_WIDE_FAILS 2u            /* dec_wide_* give every long item of a long-code coder up (dec_wide_fn_* take it) */
#define AWS_HUFFMAN_AMD_TEST_DECODE_WIDE_FN_FAILS 4u         /* ... and dec_wide_fn_* as well (dec_deep takes it) */
#define AWS_HUFFMAN_AMD_TEST_DECODE_ONE_CHUNK_A_WORKGROUP 8u /* short end-of-stream chunks do not share workgroups */
#define AWS_HUFFMAN_AMD_TEST_DECODE_TAILS_APART 32u          /* a few end-of-stream chunks among many chunks inside streams: kernels of their own, not workgroups of the big kernels */
#define AWS_HUFFMAN_AMD_TEST_DECODE_ALL_KERNELS 16u          /* every launch queues the kernels for listed chunks, whatever the plan's last fetched launch listed */
AWS_COMPRESSION_API
void aws_huffman_amd_testing_set_decode_road(uint32_t flags /* 0: back to the default */);
/* read when a PLAN is made or reset: from how many items per byte of its longest item on a class of short items goes to a
 * thread per item (0: the built-in rule, HUFD_*_TINY_PER_BYTE) */
AWS_COMPRESSION_API
void aws_huffman_amd_testing_set_items_per_byte(uint64_t encode, uint64_t decode);

/*
 * aws_huffman_decode takes an input of any length (the reference's is a size_t, source/huffman.c:228); a device item holds
 * less than 4 GiB of encoded bytes, so a longer input is taken in pieces of 2 GiB inside the call, the decoder's window
 * carried from piece to piece -- what a caller who streams the input gets.  Tests make the pieces small:
 */
AWS_COMPRESSION_API
void aws_huffman_amd_testing_set_decode_piece_bytes(size_t bytes /* 0: back to 2 GiB */);
/* testing: from how many encoded bytes on an item of a coder with codes of more than 12 bits is decoded a workgroup
 * per 32 KiB block (dec_wide_*) instead of by one workgroup (dec_deep) */
AWS_COMPRESSION_API
void aws_huffman_amd_testing_set_wide_min_bytes(uint64_t bytes /* 0: back to 128 KiB (2 MiB in a batch of 128 such items or more) */);

/* ---- several GPUs: independent items sharded over the devices of one node ---- */

/*
 * Items are independent (each has its own encoder / decoder state and its own padded output), so they shard with no
 * exchange between devices: item i of a call goes to shard i mod G.  A shard = one device with the coder's tables
 * staged on it, one engine and stream, and -- inside a call -- one host thread that builds the plan of its items,
 * launches it and fetches their result records; the call returns when every shard has, with the records gathered in
 * item order.  Inputs and outputs stay on the shard's device: an item's offsets are relative to that shard's base
 * pointers (`io[i mod G]`).  No collective, no peer traffic.  (BASELINE.json configs[3] split over G GPUs is one
 * such call; configs[4], one stream per GPU, is G items.)  The same device may be listed more than once.
 */
struct aws_huffman_amd_shards;

struct aws_huffman_amd_shard_io {
    const void *device_input; /* base pointers on the shard's device */
    void *device_output;
};

AWS_COMPRESSION_API
int aws_huffman_amd_shards_new(
    struct aws_huffman_amd_shards **shards,
    struct aws_huffman_symbol_coder *coder,
    const int *devices,
    size_t device_count);

AWS_COMPRESSION_API
void aws_huffman_amd_shards_destroy(struct aws_huffman_amd_shards *shards);

AWS_COMPRESSION_API
size_t aws_huffman_amd_shards_count(const struct aws_huffman_amd_shards *shards);

/* the engine of shard g: for its device memory (aws_huffman_amd_device_alloc, copies, fills) */
AWS_COMPRESSION_API
struct aws_huffman_amd_engine *aws_huffman_amd_shards_engine(struct aws_huffman_amd_shards *shards, size_t g);

/* items[i] on shard i mod G against io[i mod G]; results[i] as aws_huffman_amd_encode_plan_results would give them.
 * AWS_OP_ERR if any shard failed (the error raised is the first failing shard's). */
AWS_COMPRESSION_API
int aws_huffman_amd_shards_encode(
    struct aws_huffman_amd_shards *shards,
    const struct aws_huffman_amd_encode_item *items,
    size_t item_count,
    const struct aws_huffman_amd_shard_io *io,
    struct aws_huffman_amd_encode_result *results);

AWS_COMPRESSION_API
int aws_huffman_amd_shards_decode(
    struct aws_huffman_amd_shards *shards,
    const struct aws_huffman_amd_decode_item *items,
    size_t item_count,
    const struct aws_huffman_amd_shard_io *io,
    struct aws_huffman_amd_decode_result *results);

/* ---- device memory and timing helpers (so a C caller needs no HIP headers) ---- */

AWS_COMPRESSION_API
int aws_huffman_amd_device_count(void);
/* the device an engine works on, and the calling thread's current device (which no call of this library changes: every
 * entry point switches to its engine's device for its own duration and back) */
AWS_COMPRESSION_API
int aws_huffman_amd_engine_device(const struct aws_huffman_amd_engine *engine);
AWS_COMPRESSION_API
int aws_huffman_amd_current_device(void);
AWS_COMPRESSION_API
void *aws_huffman_amd_device_alloc(struct aws_huffman_amd_engine *engine, size_t size);
AWS_COMPRESSION_API
void aws_huffman_amd_device_free(struct aws_huffman_amd_engine *engine, void *ptr);
AWS_COMPRESSION_API
int aws_huffman_amd_copy_to_device(struct aws_huffman_amd_engine *engine, void *dst, const void *src, size_t size);
AWS_COMPRESSION_API
int aws_huffman_amd_copy_to_host(struct aws_huffman_amd_engine *engine, void *dst, const void *src, size_t size);
AWS_COMPRESSION_API
int aws_huffman_amd_device_fill(struct aws_huffman_amd_engine *engine, void *dst, int byte, size_t size);
/* splitmix64 byte stream of BASELINE.md section 4, generated in place on the device */
AWS_COMPRESSION_API
int aws_huffman_amd_device_fill_splitmix64(struct aws_huffman_amd_engine *engine, void *dst, size_t size, uint64_t seed);
/* the engine's own hipStream_t */
AWS_COMPRESSION_API
void *aws_huffman_amd_engine_stream(struct aws_huffman_amd_engine *engine);
AWS_COMPRESSION_API
int aws_huffman_amd_stream_synchronize(struct aws_huffman_amd_engine *engine, void *stream);

/* HIP events on a stream: create / record / elapsed milliseconds between two recorded events */
AWS_COMPRESSION_API
void *aws_huffman_amd_event_new(struct aws_huffman_amd_engine *engine);
AWS_COMPRESSION_API
void aws_huffman_amd_event_destroy(struct aws_huffman_amd_engine *engine, void *event);
AWS_COMPRESSION_API
int aws_huffman_amd_event_record(struct aws_huffman_amd_engine *engine, void *event, void *stream);
AWS_COMPRESSION_API
int aws_huffman_amd_event_elapsed_ms(struct aws_huffman_amd_engine *engine, void *start, void *stop, float *ms);

/* ---- a coder from a table (runtime counterpart of the reference's offline generator) ---- */

/*
 * Builds an aws_huffman_symbol_coder from 256 (pattern, num_bits) rows: encode is
 * the table, decode walks the code tree one bit per level the way the C file
 * emitted by reference source/huffman_generator/generator.c:154-214 does.
 * Rows with num_bits 0 have no code.  NULL (+ AWS_ERROR_INVALID_ARGUMENT) when
 * the rows are not a prefix code.
 */
AWS_COMPRESSION_API
struct aws_huffman_symbol_coder *aws_huffman_amd_table_coder_new(
    const uint32_t patterns[256],
    const uint8_t num_bits[256]);

/*
 * The same from the text of a table .def file -- rows of
 *     HUFFMAN_CODE(symbol, "bit string", 0xpattern, num_bits)
 * between comments and preprocessor lines -- the input format of the reference's
 * generator (source/huffman_generator/generator.c:42-104; e.g.
 * tests/test_huffman_static_table.def, aws-c-http's HPACK table).  No C file is
 * generated and compiled: the coder exists at once, and an engine tabulates it for
 * the device.  Symbols missing from the file have no code; rows for symbols above
 * 255 (HPACK's EOS) are skipped.  NULL + AWS_ERROR_INVALID_ARGUMENT for a malformed
 * row, a symbol listed twice, or rows that are not a prefix code.
 */
AWS_COMPRESSION_API
struct aws_huffman_symbol_coder *aws_huffman_amd_table_coder_from_def(const char *text, size_t length);

AWS_COMPRESSION_API
void aws_huffman_amd_table_coder_destroy(struct aws_huffman_symbol_coder *coder);

AWS_EXTERN_C_END

#endif /* AWS_COMPRESSION_HUFFMAN_AMD_H */
