#ifndef HUFFMAN_AMD_ENGINE_H
#define HUFFMAN_AMD_ENGINE_H
/* Internal layout of the opaque types of huffman_amd.h and the helpers huffman.c uses. */
#include <aws/compression/huffman_amd.h>

#include <pthread.h>

#include "../hip/device_types.h"
#include "../hip/hip_shim.h"
#include "../hip/huffman_kernels.h"

struct aws_huffman_amd_engine {
    int device;
    void *stream;
    /* a second stream and two events: the few kernels for the chunks streams end in run beside the big ones for the
     * chunks inside the streams (hufk_decode_launch forks and joins; NULL: one after the other) */
    void *side_stream;
    void *fork_event, *join_event;

    /* identity of the tabulated coder: the callbacks are assumed pure (huffman.h) */
    struct aws_huffman_symbol_coder *coder;
    void *key_encode;
    void *key_decode;
    void *key_userdata;
    uint64_t fingerprint; /* aws_huffman_amd_coder_fingerprint(coder) when the tables were made */

    uint64_t enc_table[256]; /* host copy: length << 32 | masked code */
    uint16_t *dec_lut_host;
    uint32_t *deep_lut_host; /* codes longer than HUFD_DEC_MAX_LUT_BITS */
    void *d_deep_lut;
    bool can_encode; /* the coder has an encode callback */
    bool can_decode;

    void *d_enc_table;
    void *d_dec_lut;
    struct hufd_tables tables;
    bool single_pass; /* enc_onepass where the coder allows it (the tests can make an engine that keeps to count / scan / pack) */
    bool encode_fails; /* a wave of enc_onepass is made to give up (tests of the way back: aws_huffman_amd_testing_set_encode_road) */

    /* scratch of the host-pointer API: one item at a time, one caller at a time (`one_lock`); `users` keeps the
     * engine cache of huffman.c from retiring an engine somebody is inside of */
    pthread_mutex_t one_lock;
    int users;
    void *one_in;
    size_t one_in_cap;
    void *one_out;
    size_t one_out_cap;
    struct aws_huffman_amd_encode_plan *one_enc;
    struct aws_huffman_amd_decode_plan *one_dec;
    /* ... and of its road for header-sized inputs: one block up, one launch, one block back (MINI_* in engine.c) */
    uint8_t *mini_host; /* page-locked */
    uint8_t *mini_dev;
    bool mini_output; /* the last decode left its symbols in mini_host */
    /* one destroyed plan of each kind, kept with its device arrays for the next *_plan_new of this engine to take over
     * (a fresh plan otherwise pays some twenty-five device allocations: 1 .. 15 ms for BASELINE configs[3], where filling
     * it takes 0.5 ms); freed with the engine */
    pthread_mutex_t spare_lock;
    struct aws_huffman_amd_encode_plan *spare_enc;
    struct aws_huffman_amd_decode_plan *spare_dec;
    bool retiring; /* the engine is being destroyed: plans are freed, not kept */
};

struct aws_huffman_amd_encode_plan {
    struct aws_huffman_amd_engine *engine;
    uint32_t n_items, n_segs, n_large;
    size_t cap_items, cap_segs, cap_large;
    void *d_arena; /* the ONE device allocation the arrays below are cuts of */
    struct hufd_enc_item *d_items;
    struct hufd_enc_seg *d_segs;
    uint32_t *d_large;
    uint32_t *d_tiny; /* items of at most HUFD_ENC_TINY_BYTES symbols */
    uint32_t n_tiny;
    uint32_t *d_solo; /* items of at most HUFD_ENC_SOLO_BYTES symbols that no thread takes: a wave each, no segments */
    uint32_t n_solo;
    size_t cap_tiny;
    uint32_t *d_seg_bits;
    uint32_t *d_wave_bits; /* [n_segs][4]: bits of each quarter of a segment */
    uint32_t *d_seg_unk;
    uint64_t *d_seg_bitoff;
    uint32_t *d_careful; /* [2 * cap_items + 4]: segments for the per-symbol packer */
    uint8_t *d_zero;     /* control words (tickets, "a wait ran out", careful count, the last launch's "a wait ran out") | look-back words, twice: clear between launches */
    bool zero_is_clear;         /* d_zero is clear (the last launch left it so, or the reserve did) */
    uint8_t *d_unk_seen; /* [cap_segs] */
    uint64_t *d_item_total; /* [cap_items] */
    struct hufd_enc_item_state *d_states;
    struct hufd_enc_result *d_results;
    /* single-pass bookkeeping: what the last launch was given, and whether look-back ever timed out */
    const void *last_input;
    void *last_output;
    void *done_event; /* recorded behind every launch on a caller's stream: what a new plan on this one's arrays waits for */
    bool done_on_engine_stream; /* ... a launch on the engine's own stream: that stream is waited for */
    bool unkeepable;  /* not to be kept as the engine's spare (waiting for its last launch failed) */
    void *d_plan_scratch; /* of a plan made on the device (hufk_plan_scratch_bytes) */
    size_t cap_plan_scratch;
    bool launched; /* the plan's items have been launched at least once: their records exist (aws_huffman_amd_decode_plan_from_encode asks) */
    bool last_single_pass;
    bool last_timed_out; /* the last launch whose results were fetched was done over by the three-kernel road */
    bool look_back_timed_out;
    struct aws_huffman_amd_plan_stats stats; /* how the items are taken (aws_huffman_amd_encode_plan_stats) */
    /* of the items as they were given: what aws_huffman_amd_decode_plan_from_encode asks before it chains a decode plan */
    uint64_t largest_out_cap; /* the most encoded bytes an item can leave */
    uint32_t most_overflow_bits;
};

struct aws_huffman_amd_decode_plan {
    struct aws_huffman_amd_engine *engine;
    uint32_t n_items, n_chunks, n_large, n_runs, n_tail, n_tiny, n_deep;
    uint32_t tail_stage_bytes; /* symbols (+32) a chunk that holds the end of a stream can decode to: sizes dec_emit_fast<TAIL>'s LDS stage */
    uint32_t tail_lanes; /* the most whole lanes a chunk that holds the end of a stream has (dec_sync_pack's slot width) */
    uint32_t tail_wide_lanes; /* ... and one that is not narrow (dec_emit_fast<TAIL>'s workgroup size) */
    uint32_t n_tail_narrow; /* the first so many of d_tail have at most HUFD_DEC_PACK_LANES whole lanes */
    size_t cap_items, cap_chunks, cap_large, cap_runs;
    struct aws_huffman_amd_decode_item *h_items; /* host copy for result translation */
    void *d_arena; /* the ONE device allocation the arrays below are cuts of (but d_wide_block and d_fixed) */
    struct hufd_dec_item *d_items;
    uint32_t *d_chunk_item;
    uint32_t *d_tiny;      /* [n_items]: from the front the items of at most HUFD_DEC_TINY_BYTES encoded bytes, from the back the longer ones of a coder with long codes */
    uint32_t *d_tail;      /* chunks that may hold the end of their stream (room for 2 per item) */
    uint32_t *d_large;     /* per large item: item index, its first run */
    uint32_t *d_runs;      /* per run: item index, run number inside the item */
    uint32_t *d_run_fn;    /* [n_runs][n_states] */
    uint16_t *d_fn_tab;
    uint16_t *d_cp_tab;
    uint32_t *d_chunk_fn;
    uint32_t *d_slow_list; /* [0] how many, [1..] the chunks the regular chunks' kernels left to dec_sync */
    uint32_t *d_emit_list; /* the same for dec_emit_fast / dec_emit */
    uint32_t *d_dense_list; /* [0] how many, [1..] chunks with more symbols than one emit stage */
    uint32_t *d_counters;   /* [HUFK_DEC_COUNTERS] the lists' lengths (hufk_decode_args.counters): clear between launches */
    uint32_t launches_with_chunks; /* (a fetch of results says what the last of them listed: `quiet`) */
    uint16_t *d_lane_count;
    uint8_t *d_chunk_regular;
    uint32_t *d_tail_entry;
    uint32_t *d_chunk_entry;
    uint64_t *d_chunk_base;
    struct hufd_chunk_rec *d_chunk_rec;
    struct aws_huffman_amd_plan_stats stats; /* how the items are taken (aws_huffman_amd_decode_plan_stats) */
    void *d_plan_scratch; /* of a plan made on the device (hufk_plan_scratch_bytes) */
    size_t cap_plan_scratch;
    void *done_event; /* recorded behind every launch on a caller's stream: what a new plan on this one's arrays waits for */
    bool done_on_engine_stream; /* ... a launch on the engine's own stream: that stream is waited for */
    bool unkeepable;  /* not to be kept as the engine's spare (waiting for its last launch failed) */
    bool chained; /* made on the device (from an encode plan's records, a stride, or items in device memory): the items are known there only (h_items is not filled) */
    struct hufd_dec_item_state *d_states;
    uint32_t *d_summary; /* [HUFK_DEC_COUNTERS] of the last launch, as its last kernel left them: 256 bytes in front of d_results */
    struct hufd_dec_result *d_results;
    bool quiet; /* the last fetched launch listed no chunk for any kernel but the regular ones (see aws_huffman_amd_decode_plan_results) */
    /* the long items of a coder with long codes: a workgroup per 32 KiB block (dec_wide_*) */
    struct hufk_wide_item *h_wide; /* [n_wide] */
    uint32_t n_wide;
    uint64_t wide_from; /* encoded bytes from which an item was taken for one */
    void *d_wide_block;
    size_t cap_wide_block;
    /* a coder with codes of one length: (item, 16 KiB block) for every block of its items beyond a thread's work (dec_fixed_*) */
    uint32_t *d_fixed;
    uint32_t n_fixed;
    size_t cap_fixed;
};

/* the engine cache of huffman.c: what an engine is recognised by besides the coder's address, and the two ways out of it */
uint64_t aws_huffman_amd_coder_fingerprint(struct aws_huffman_symbol_coder *coder);
void aws_huffman_amd_forget_coder(struct aws_huffman_symbol_coder *coder);
void aws_huffman_amd_forget_all(void);

int aws_huffman_amd_encode_plan_raw_results(struct aws_huffman_amd_encode_plan *plan, struct hufd_enc_result *raw, void *stream);
void aws_huffman_amd_encode_result_from_raw(const struct hufd_enc_result *raw, struct aws_huffman_amd_encode_result *out);
void aws_huffman_amd_decode_result_from_raw(
    const struct hufd_dec_result *raw,
    const struct aws_huffman_amd_decode_item *item,
    struct aws_huffman_amd_decode_result *out);

/* one item through the engine's own staging buffers; host_out receives raw->produced bytes */
int aws_huffman_amd_engine_encode_host(
    struct aws_huffman_amd_engine *engine,
    const struct aws_huffman_amd_encode_item *item,
    const uint8_t *host_in,
    uint8_t *host_out,
    bool length_only,
    struct hufd_enc_result *raw);

/* decodes carry bytes + new bytes; the symbols stay in the engine's output buffer until fetched */
int aws_huffman_amd_engine_decode_host(
    struct aws_huffman_amd_engine *engine,
    const uint8_t *carry,
    uint32_t carry_bytes,
    uint32_t first_bit,
    const uint8_t *host_in,
    uint64_t in_len,
    uint64_t out_capacity,
    struct aws_huffman_amd_decode_result *result);
int aws_huffman_amd_engine_fetch_output(struct aws_huffman_amd_engine *engine, uint8_t *host_out, uint64_t size);

#endif /* HUFFMAN_AMD_ENGINE_H */
