#ifndef HUFFMAN_AMD_KERNELS_H
#define HUFFMAN_AMD_KERNELS_H
/*
 * C interface of the kernel file: argument packs and launch wrappers.  The
 * pointers are device pointers; the launches are asynchronous on `stream`
 * (a hipStream_t carried as void *).  Return value: 0 or a hipError_t.
 */
#include <stdint.h>

#include "device_types.h"

#ifdef __cplusplus
extern "C" {
#endif

struct hufk_encode_args {
    struct hufd_tables tables;
    const struct hufd_enc_item *items;
    uint32_t n_items;
    const struct hufd_enc_seg *segs; /* [n_segs] */
    uint32_t n_segs;
    const uint32_t *large_items; /* [n_large] items with more than HUFD_SCAN_SMALL_MAX segments */
    uint32_t n_large;
    const uint32_t *tiny_items;  /* [n_tiny] items of at most HUFD_ENC_TINY_BYTES symbols: no segments, one thread each */
    uint32_t n_tiny;
    const uint32_t *solo_items;  /* [n_solo] items of one tile at most (hufd_enc_item.tiny == 2): no segments, one wave each */
    uint32_t n_solo;
    uint32_t length_only; /* stop after the scan */
    const void *d_in;
    void *d_out;
    uint32_t *seg_bits;   /* [n_segs] scratch */
    uint32_t *wave_bits;  /* [n_segs][4] scratch: bits of each quarter (4 KiB of symbols) of a segment */
    uint32_t *seg_unk;    /* [n_segs] scratch */
    uint64_t *seg_bitoff; /* [n_segs] scratch */
    uint32_t *careful_list;  /* [2 * n_items] scratch: segments for the per-symbol packer */
    uint32_t *careful_count; /* [1] scratch */
    /* one-pass path (every symbol coded, codes of 4 .. 15 bits): */
    void *zero_block;        /* hufk_encode_zero_bytes(n_segs, n_items) bytes, clear when the launch starts and when it is
                              * through: the control words (word 0 the way back's tickets, word 1 "a wait ran out", word 2
                              * careful_count, word 4 word 1 of the last launch), then the look-back tables, twice */
    uint32_t zero_is_clear;  /* 1: the block is known to be clear (the plan's last launch left it so); 0: the launch clears it first */
    uint64_t zero_bytes;     /* all of the block (what a clearing takes when the layout of the launch that dirtied it is not known) */
    uint8_t *seg_unk_seen;   /* [n_segs] scratch */
    uint64_t *item_total;    /* [n_items] scratch */
    uint32_t single_pass;    /* 1: one kernel reads the symbols once (enc_onepass) instead of count / scan / pack */
    uint32_t fail_tile;      /* 1: a wave of enc_onepass is made to give up (the way back, for tests: aws_huffman_amd_testing_set_encode_road) */
    struct hufd_enc_item_state *states; /* [n_items] scratch */
    struct hufd_enc_result *results;    /* [n_items] */
    void **stage_events; /* NULL, or 4 hipEvent_t: before count, after count, after scan, after pack */
};

struct hufk_wide_item {
    uint32_t slot;         /* of the item's index in deep_items */
    uint32_t n_blocks;     /* 32 KiB blocks (HUFD_WIDE_BLOCK_BYTES) */
    uint64_t block_offset; /* of its scratch in wide_block */
};

/* hufk_decode_args.counters: */
#define HUFK_DEC_COUNT_SLOW 0u      /* slow_list: the chunks dec_sync_one / dec_sync_pack leave to dec_sync_guess (or, without one, to dec_sync) */
#define HUFK_DEC_COUNT_LONG 1u      /* emit_list: the chunks for the long way (dec_sync_few's and dec_sync's list) */
#define HUFK_DEC_COUNT_FEW 2u       /* slow_list again: dec_sync_few's chunks, for dec_sync_true */
#define HUFK_DEC_COUNT_EMIT 3u      /* emit_list again: the chunks dec_emit_fast leaves to dec_emit */
#define HUFK_DEC_COUNT_DENSE 4u     /* dense_list */
#define HUFK_DEC_COUNTERS 8u

struct hufk_decode_args {
    struct hufd_tables tables;
    const struct hufd_dec_item *items;
    uint32_t n_items;
    const uint32_t *chunk_item; /* [n_chunks] */
    uint32_t n_chunks;
    const uint32_t *tail_chunks; /* [n_tail] chunks with fewer than HUFD_DEC_CHUNK_BYTES + 8 bytes of their item left: the end of a stream is in them */
    uint32_t n_tail;
    const uint32_t *tiny_items;  /* [n_tiny] items of at most HUFD_DEC_TINY_BYTES encoded bytes: no chunks, one thread each */
    uint32_t n_tiny;
    const uint32_t *deep_items;  /* [n_deep] longer items of a coder with codes of more than HUFD_DEC_MAX_LUT_BITS bits: no chunks, one workgroup each */
    uint32_t n_deep;
    /* the deep items of at least wide_from bytes: every 32 KiB block of one is a workgroup's (dec_wide_*) */
    const struct hufk_wide_item *wide; /* [n_wide], HOST memory */
    uint32_t n_wide;
    uint64_t wide_from;
    void *wide_block; /* scratch: hufk_decode_wide_bytes(blocks) each, at the offsets in `wide` */
    uint32_t wide_fails; /* 1: they give up (tests of the way back); 2: and dec_wide_fn_* behind them as well */
    uint32_t few_walks;  /* 1: the chunks inside streams whose walks do not fall into step go to dec_sync_few / dec_sync_true; 0: the long way */
    /* a coder with codes of one length (tables.fixed_bits): its items beyond a thread's work, 16 KiB blocks of them */
    const uint32_t *fixed_blocks; /* [n_fixed_blocks][2]: item, block of HUFD_FIXED_BLOCK_BYTES inside it */
    uint32_t n_fixed_blocks;
    const uint32_t *large_items; /* per item with more than HUFD_SCAN_SMALL_MAX chunks: item index, its first run */
    uint32_t n_large;
    const uint32_t *runs;        /* per run of HUFD_SCAN_RUN_CHUNKS chunks of a large item: item index, run number */
    uint32_t n_runs;
    uint32_t *run_fn;            /* [n_runs][n_states] scratch */
    const void *d_in;
    void *d_out;
    uint16_t *fn_tab;      /* [n_chunks][n_states][HUFD_DEC_LANES] scratch */
    uint16_t *cp_tab;      /* [n_chunks][HUFD_DEC_CP_ROWS][HUFD_DEC_LANES] scratch: walk checkpoints */
    uint32_t *chunk_fn;    /* [n_chunks][n_states] scratch */
    uint32_t *slow_list;   /* [n_chunks] scratch: chunks that take the long way through dec_sync */
    uint32_t *emit_list;   /* [n_chunks] scratch: chunks left to dec_emit by dec_emit_fast */
    uint32_t *dense_list;  /* [n_chunks] scratch: chunks dec_emit_fast leaves to dec_emit_dense */
    uint32_t *counters;    /* [HUFK_DEC_COUNTERS] scratch: how many entries the lists hold, one word for every use a launch
                            * makes of a list (the arrays take turns, the words do not).  Clear when the plan gets them, and
                            * every launch leaves them as it needs them itself (so that a launch captured in a graph can be
                            * replayed): the sync stage's words are cleared by the launch's LAST kernel (dec_emit, the long
                            * way: nobody reads them behind the sync stage), the emit stage's by the last kernel of the
                            * sync stage (dec_sync, the long way: nobody has touched them yet) -- both run in every launch
                            * with chunks.  A clearing command in front of every launch was ~4 us of its own. */
    uint32_t counters_self_cleared; /* 1: as above; 0: cleared by a command in front of the launch */
    uint32_t *summary;       /* NULL, or [HUFK_DEC_COUNTERS]: the launch's last kernel leaves its counters here (in front of the
                              * result records: one copy fetches both) -- what the host reads `quiet` from */
    uint32_t quiet;          /* 1: the plan's last fetched launch listed no chunk for any kernel but the regular ones: this
                              * launch goes without the kernels that only make listed chunks FASTER (dec_sync_guess, _few,
                              * _true, dec_emit_big -- an empty launch is ~4 us each); whatever is listed takes the long way
                              * (dec_sync, dec_emit: exact for every chunk), and the counters say so at the next fetch */
    uint16_t *lane_count;  /* [n_chunks][HUFD_DEC_LANES] scratch */
    uint8_t *chunk_regular; /* [n_chunks] scratch */
    uint32_t *tail_entry;   /* [n_chunks] scratch: state in which the last whole lane of an end-of-stream chunk leaves */
    uint32_t *chunk_entry; /* [n_chunks] scratch */
    uint64_t *chunk_base;  /* [n_chunks] scratch */
    const struct hufd_chunk_rec *chunk_rec; /* [n_chunks] built with the plan */
    void *side_stream;  /* NULL, or a stream of the engine's for the kernels of the chunks streams end in ... */
    void *fork_event;   /* ... and two events to fork it off the launch's stream and join it again */
    void *join_event;
    struct hufd_dec_item_state *states; /* [n_items] scratch */
    struct hufd_dec_result *results;    /* [n_items] */
    uint32_t tail_stage_bytes; /* the most symbols a chunk that holds the end of a stream can decode to, +32 (0: unknown) */
    uint32_t tail_lanes;       /* the most whole lanes (sub-chunks with 8 more bytes behind them) a NARROW such chunk has */
    uint32_t tail_wide_lanes;  /* ... and a wide one (0: not known -- as many as a chunk has) */
    uint32_t n_tail_narrow;    /* the first so many of tail_chunks have at most HUFD_DEC_PACK_LANES whole lanes: they may share workgroups */
    uint32_t one_chunk_a_workgroup; /* 1: the chunks streams end in get a workgroup each, however short and many (tests: the road a plan of
                                     * few such chunks takes, for a plan of many) */
    uint32_t tails_apart;           /* 1: ... and kernels of their own even where they are few among many chunks inside streams (tests:
                                     * such a launch folds them into the big kernels' grids) */
    void **stage_events; /* NULL, or 4 hipEvent_t: before sync, after sync, after scan, after emit */
};

/* one-time per-process kernel attribute setup (dynamic LDS above 64 KiB) */
int hufk_init(void);

uint32_t hufk_enc_image_words(uint32_t max_bits);
/* whether the one-pass encoder takes this coder, and the size of the block it wants zeroed per launch */
int hufk_encode_one_pass_applies(const struct hufd_tables *tables);
uint64_t hufk_encode_zero_bytes(uint32_t n_segs, uint32_t n_items);
int hufk_encode_launch(const struct hufk_encode_args *args, void *stream);
int hufk_decode_launch(const struct hufk_decode_args *args, void *stream);
/* a plan of items that are all one thread's work: the kernels' item records and the list of such items (= all of them) from
 * the caller's records, copied to the device as they are (struct hufd_raw_dec_item / hufd_raw_enc_item) */
int hufk_decode_plan_tiny_items(const void *raw_items, uint32_t n_items, struct hufd_dec_item *items, uint32_t *tiny_list, void *stream);
/* items[i] = the encoded output of encode item i as its result record on the device describes it, decoded back to where its symbols came from */
int hufk_decode_plan_from_encode(
    const struct hufd_enc_item *enc_items, const struct hufd_enc_result *enc_results, uint32_t n_items, struct hufd_dec_item *items,
    uint32_t *tiny_list, void *stream);
int hufk_encode_plan_tiny_items(const void *raw_items, uint32_t n_items, struct hufd_enc_item *items, uint32_t *tiny_list, void *stream);
/* fills chunk_item[n_chunks] and chunk_rec[n_chunks] of a decode plan from its item records, on the device */
int hufk_decode_plan_chunks(
    const struct hufd_dec_item *items, uint32_t n_items, uint32_t n_chunks, uint32_t *chunk_item, struct hufd_chunk_rec *chunk_rec,
    void *stream);
/* Plans made on the device from items the host never looks at (plan_kernels.hip).  _count: what the items come to -- statistics,
 * the thread-per-item rule, counts per item scanned into positions; copies the totals back and WAITS for the stream (the sizes
 * of the plan's arrays and of the launch's grids are the host's to know).  _fill: the records and lists, from the same scratch. */
struct hufk_plan_totals {
    uint64_t totals[8]; /* decode: chunks, thread items, wave items, large items, runs, narrow / wide end-of-stream chunks, items
                         * with chunks; encode: segments, thread items, one-tile items, large items, -, -, -, items with segments */
    uint64_t tiny_limit;
    uint64_t shortest, longest, largest_out_cap, tail_stage, tail_lanes, wide_lanes;
    uint32_t worst_bits, invalid;
};
uint64_t hufk_plan_scratch_bytes(uint64_t n_items);
int hufk_decode_plan_count(
    const struct hufd_item_source *src, uint32_t n_items, uint64_t per_byte, uint32_t shortest_code_bits, void *scratch,
    struct hufk_plan_totals *totals, void *stream);
int hufk_decode_plan_fill(
    const struct hufd_item_source *src, uint32_t n_items, uint32_t shortest_code_bits, const void *scratch, struct hufd_dec_item *items,
    uint32_t *tiny_list, uint32_t *tail_list, uint32_t *large_list, uint32_t *run_list, void *stream);
int hufk_encode_plan_count(
    const struct hufd_item_source *src, uint32_t n_items, uint64_t class0, uint64_t class1, uint64_t per_byte, uint64_t solo_limit,
    void *scratch, struct hufk_plan_totals *totals, void *stream);
int hufk_encode_plan_fill(
    const struct hufd_item_source *src, uint32_t n_items, uint32_t n_segs, uint64_t solo_limit, const void *scratch,
    struct hufd_enc_item *items, struct hufd_enc_seg *segs, uint32_t *tiny_list, uint32_t *large_list, uint32_t *solo_list, void *stream);
/* one short item whose record already sits in device memory (the host-pointer calls' small-input road): one launch */
int hufk_encode_one_tiny(
    const struct hufd_tables *tables, const struct hufd_enc_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_enc_result *result, uint32_t length_only, void *stream);
int hufk_decode_one_tiny(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream);
/* scratch bytes of one wide item of n_blocks blocks */
uint64_t hufk_decode_wide_bytes(uint64_t n_blocks);
/* the same for an item of up to HUFD_DEC_COOP_BYTES encoded bytes (any size with long codes): dec_deep, one launch */
int hufk_decode_one_coop(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream);
/* the same for an item of up to HUFD_DEC_BLOCK_MAX_BYTES encoded bytes of a coder without long codes: one workgroup, one
 * launch (dec_block_kernel); item = the record, in HOST memory (it travels with the launch) */
int hufk_decode_one_block(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream);
/* one item of up to HUFD_ENC_BLOCK_MAX_BYTES symbols, one workgroup, one launch (enc_block_kernel); `symbols` = the
 * item's in_len (the record itself is in device memory); _fits: whether such an item of this coder is taken */
int hufk_encode_one_block_fits(const struct hufd_tables *tables, uint64_t symbols);
int hufk_encode_one_block(
    const struct hufd_tables *tables, const struct hufd_enc_item *item, uint32_t symbols, const void *d_in, void *d_out,
    struct hufd_enc_result *result, uint32_t length_only, void *stream);
int hufk_fill_splitmix64(void *dst, uint64_t len, uint64_t seed, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* HUFFMAN_AMD_KERNELS_H */
