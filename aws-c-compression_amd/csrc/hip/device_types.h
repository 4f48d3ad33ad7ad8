#ifndef HUFFMAN_AMD_DEVICE_TYPES_H
#define HUFFMAN_AMD_DEVICE_TYPES_H
/*
 * Plain-C records shared by the C99 host layer (csrc/host) and the HIP shim
 * (csrc/hip).  Everything here lives in device memory unless stated otherwise.
 *
 * Vocabulary
 *   item      one independent encode or decode job (one aws_huffman_encode /
 *             aws_huffman_decode call's worth of work)
 *   segment   HUFD_ENC_SEG_BYTES consecutive input symbols of one encode item;
 *             one workgroup packs one segment
 *   chunk     HUFD_DEC_CHUNK_BYTES consecutive encoded bytes of one decode item;
 *             one workgroup handles one chunk, one lane one sub-chunk of it
 *   entry state  for a sub-chunk starting at stream bit B: the offset e in
 *             [0, n_states) such that the first code that starts at or after B
 *             starts at B + e
 */
#include <stdint.h>

#define HUFD_ENC_SEG_BYTES 16384u
#define HUFD_ENC_THREADS 256u
#define HUFD_ENC_BLOCK_BYTES 4096u /* one host-pointer call of up to this many symbols is one workgroup's work (enc_block): one launch */
#define HUFD_ENC_BLOCK_MAX_BYTES 16384u /* ... and up to this many a workgroup of four times the lanes', if the bit image fits 64 KiB of LDS */
#define HUFD_ENC_TINY_BYTES 512u /* encode items up to this long are one thread's work (enc_tiny): no segments */

#define HUFD_DEC_SUB_BYTES 128u
#define HUFD_DEC_SUB_BITS (HUFD_DEC_SUB_BYTES * 8u)
#define HUFD_DEC_LANES 256u
#define HUFD_DEC_CHUNK_BYTES (HUFD_DEC_SUB_BYTES * HUFD_DEC_LANES)
/* One thread per item pays when the walk of the longest such item (a lone thread needs ~0.5 us a symbol) is
 * shorter than the segments / chunks of all of them (~33 / ~74 ns an item): the plan takes the largest of these
 * length classes that holds at least HUFD_*_TINY_PER_BYTE items per byte of its longest item. */
#define HUFD_TINY_MANY_BYTES 2048u /* symbols (encode); encoded bytes x 2 / 3 (decode) */
#define HUFD_ENC_SOLO_BYTES 4096u /* encode items up to one tile of symbols that no thread takes are one wave's work where the
                                   * one-pass encoder applies (enc_onepass<.., SOLO>): no segments, no look-back */
#define HUFD_ENC_SOLO_MANY_BYTES HUFD_ENC_SEG_BYTES /* ... and items of up to a segment (four tiles, one after the other by the
                                   * same wave) when the plan holds at least HUFD_ENC_SOLO_MANY_ITEMS items: the chip is full
                                   * of waves either way, and none of them waits (BASELINE configs[3]: 65 536 items of 16 KiB) */
#define HUFD_ENC_SOLO_MANY_ITEMS 256u
#define HUFD_ENC_TINY_WAVE_BYTES 1024u /* the same class for encode where the one-pass kernel is used */
#define HUFD_ENC_TINY_PER_BYTE 18u
#define HUFD_ENC_TINY_PER_BYTE_ONE_PASS 100u /* the same where the one-pass encoder packs ragged tiles (csrc/host/engine.c: enc_tiny_per_byte) */
#define HUFD_DEC_TINY_PER_BYTE 40u /* (round 2: a chunk of a short stream costs ~14 ns now, was ~74) */
#define HUFD_TINY_FEW_BYTES 128u   /* the class that always goes to a thread */
#define HUFD_DEC_MAX_STATES 16u
#define HUFD_DEC_CP_ROWS 4u /* per sub-chunk: three checkpoints of the walk + the merged-state mask */
#define HUFD_DEC_MAX_LUT_BITS 12u
/* coders with longer codes (HPACK: 30 bits) decode through a tree of tables, one thread per item */
#define HUFD_DEEP_ROOT_BITS 10u
#define HUFD_DEEP_SUB_BITS 8u
#define HUFD_DEEP_MAX_ENTRIES 16384u
#define HUFD_DEEP_LINK 0x80000000u /* entry is a link: [15:0] first entry of the next table, [23:16] its index width */
#define HUFD_DEC_TINY_BYTES 512u /* decode items up to this long are one thread's work (dec_tiny): no chunks */
#define HUFD_DEC_PACK_MIN_CHUNKS 64u /* fewer such chunks in a launch: a workgroup each (dec_sync_one<TAIL>) */
#define HUFD_DEC_PACK_LANES 83u /* end-of-stream chunks with at most this many whole lanes share workgroups (dec_sync_pack: three slots of 85 lanes or more of fewer; two chunks a workgroup measured no faster than one) */
#define HUFD_DEC_COOP_BYTES 768u /* ... and up to this long one wave's (dec_deep<false>): no chunks either */
#define HUFD_DEC_BLOCK_BYTES 8192u /* what dec_block's workgroup takes in one turn: a lane per 64 bits */
#define HUFD_DEC_BLOCK_MAX_BYTES (4u * HUFD_DEC_BLOCK_BYTES) /* one host-pointer call of up to this many encoded bytes (short codes) is that one workgroup's work: one launch */
#define HUFD_FIXED_BLOCK_BYTES 16384u /* a coder whose codes all have one length: its longer items are decoded this many bytes a workgroup (dec_fixed_*) */
#define HUFD_WIDE_BLOCK_BYTES 32768u /* a long item of a coder with long codes is decoded this many bytes a workgroup (dec_wide_*) */
#define HUFD_WIDE_MIN_BYTES (4u * HUFD_WIDE_BLOCK_BYTES) /* ... when it is at least this long and the batch has fewer than HUFD_WIDE_FEW_ITEMS such items, */
#define HUFD_WIDE_FEW_ITEMS 128u
#define HUFD_WIDE_MANY_MIN_BYTES (64u * HUFD_WIDE_BLOCK_BYTES) /* or this long whatever the batch */
#define HUFD_DEC_STAGE_BYTES 34304u /* LDS bytes for a chunk's decoded symbols (dec_emit_fast: four workgroups per CU) */

#define HUFD_SCAN_SMALL_MAX 64u /* items with at most this many segments/chunks are scanned by one thread */
#define HUFD_SCAN_LARGE_THREADS 1024u
#ifndef HUFD_SCAN_RUN_CHUNKS /* (tests/emu builds with short runs, so that its streams of a few MB have several) */
#define HUFD_SCAN_RUN_CHUNKS 256u /* a large decode item is scanned in runs of this many chunks, one workgroup per run */
#endif
#define HUFD_SCAN_SUB_CHUNKS 16u  /* ... each folded in sub-runs of this many */

/* look-back of the one-pass encoder: tiles per group, groups per round (the emulator build of tests/emu
 * makes them small so that a short input crosses many boundaries) */
#ifndef HUFD_OP_GROUP_TILES
#define HUFD_OP_GROUP_TILES 64u
#endif
#ifndef HUFD_OP_ROUND_GROUPS
#define HUFD_OP_ROUND_GROUPS 64u
#endif
#ifndef HUFD_OP_GROUP_STRIDE
#define HUFD_OP_GROUP_STRIDE 8u /* u64 words from one group's counter to the next */
#endif

#define HUFD_NONE32 0xFFFFFFFFu

/* encode item status */
#define HUFD_ENC_OK 0u
#define HUFD_ENC_SHORT 1u
#define HUFD_ENC_UNKNOWN 2u
#define HUFD_ENC_DECIDE 3u /* unknown symbol and capacity edge fall into one segment: its workgroup decides */

/* why a decode walk ended */
#define HUFD_STOP_NONE 0u
#define HUFD_STOP_END 1u        /* all stream bits consumed */
#define HUFD_STOP_INCOMPLETE 2u /* a code runs past the end of the stream */
#define HUFD_STOP_INVALID 3u    /* no code matches */
#define HUFD_STOP_GAVE_UP 4u    /* (dec_block only) the lanes did not settle in its rounds: nothing decoded, the chunk kernels take the call */

struct hufd_tables {
    const uint64_t *enc_table; /* [256] low 32 bits: code masked to its length, high 32 bits: length (0 = no code) */
    const uint16_t *dec_lut;   /* [1 << lut_bits] symbol << 8 | length, length 0 = no code; NULL when decode is unavailable */
    uint32_t max_bits; /* longest / shortest code the DECODE kernels can meet (the decode table's own bounds when there is
                        * one: an encoder with longer codes than its decoder knows must not widen the walk's tables) */
    uint32_t min_bits;
    uint32_t lut_bits;
    uint32_t n_states; /* max(max_bits, 8) */
    uint32_t enc_max_bits; /* longest / shortest code of the ENCODE table (images, stages, which packer): 0 / 1 without one */
    uint32_t enc_min_bits;
    uint32_t all_coded; /* every one of the 256 symbols has a code */
    uint32_t deep_entries; /* != 0: codes longer than HUFD_DEC_MAX_LUT_BITS, decode walks deep_lut instead of dec_lut */
    const uint32_t *deep_lut; /* [deep_entries] root table of 1 << HUFD_DEEP_ROOT_BITS entries, then the linked ones;
                               * an entry is symbol << 8 | length, 0 = no code, or a link */
    uint32_t fixed_bits; /* != 0: every code the decode table knows has this length: symbol k starts at bit k * fixed_bits,
                          * no walk has to find it (dec_fixed_*) */
    uint32_t fixed_complete; /* ... and every window of the decode table is a code: nothing to check before the symbols are written */
};

struct hufd_enc_item {
    uint64_t in_off;
    uint64_t in_len;
    uint64_t out_off;
    uint64_t out_cap;
    uint32_t ovf_pattern; /* masked to ovf_bits */
    uint32_t ovf_bits;
    uint32_t eos_padding;
    uint32_t first_seg; /* index of the item's first segment in the plan's segment numbering */
    uint32_t n_segs;
    uint32_t tiny; /* 1: the item has no segments, enc_tiny encodes it; 2: no segments either, a wave of enc_onepass<.., SOLO> does */
};

/* one per segment, built with the plan: where the segment's symbols are, without pointer chasing */
struct hufd_enc_seg {
    uint64_t in_off;   /* bytes from the input base pointer to the segment's first symbol */
    uint32_t len;      /* symbols in the segment (0 .. HUFD_ENC_SEG_BYTES) */
    uint32_t item;     /* index of the owning item */
    uint32_t index;    /* segment number inside the item */
    uint32_t flags;    /* bit 0: first segment of the item, bit 1: last */
    uint32_t next_len; /* symbols in the item's next segment (0 when this is the last) */
    uint32_t reserved;
};

/* written by the scan kernel, read by the pack kernel */
struct hufd_enc_item_state {
    uint64_t total_bits; /* overflow-in bits + every code bit of the item */
    uint32_t status;     /* HUFD_ENC_* */
    uint32_t unk_seg;    /* segment holding the first symbol without a code, or HUFD_NONE32 */
    uint32_t unk_idx;    /* its index inside that segment */
    uint32_t reserved;
};

struct hufd_enc_result {
    uint32_t status; /* HUFD_ENC_OK / _SHORT / _UNKNOWN */
    uint32_t ovf_pattern;
    uint32_t ovf_bits;
    uint32_t reserved;
    uint64_t consumed;
    uint64_t produced;
    uint64_t total_bits; /* overflow-in bits + every code bit of the item (length queries) */
};

struct hufd_dec_item {
    uint64_t in_off;
    uint64_t in_len;
    uint64_t out_off;
    uint64_t out_cap;
    uint32_t first_bit;
    uint32_t first_chunk;
    uint32_t n_chunks;
    uint32_t tiny; /* 1: the item has no chunks, dec_tiny decodes it */
};

/* The caller's item records as the public header lays them out (include/aws/compression/huffman_amd.h: struct
 * aws_huffman_amd_decode_item / _encode_item; csrc/host/engine.c holds the two pairs to the same sizes and offsets): a
 * plan whose items are all one thread's work is made from them ON THE DEVICE (hufk_*_plan_tiny_items). */
struct hufd_raw_dec_item {
    uint64_t in_offset;
    uint64_t in_len;
    uint32_t first_bit;
    uint32_t pad;
    uint64_t out_offset;
    uint64_t out_capacity;
};
struct hufd_raw_enc_item {
    uint64_t in_offset;
    uint64_t in_len;
    uint64_t out_offset;
    uint64_t out_capacity;
    uint32_t ovf_pattern;
    uint8_t ovf_bits;
    uint8_t pad0[3];
    uint8_t eos_padding;
    uint8_t pad1[7];
};

/* written by the scan kernel */
struct hufd_dec_item_state {
    uint64_t total_symbols; /* symbols on the true path before it stops */
};

/* raw record; the host layer turns it into rc / error / bits_consumed */
struct hufd_dec_result {
    uint64_t total_symbols; /* same as the state: how many symbols the stream holds */
    uint64_t stop_bit;      /* stream bit (from byte 0 of the item) where the true path stopped */
    uint64_t cap_bit;       /* start bit of symbol number out_cap, when total_symbols > out_cap */
    uint32_t stop_kind;     /* HUFD_STOP_* */
    uint32_t reserved;
};

/* one per chunk, built with the plan: what the chunk kernels need of the chunk's item, without following
 * chunk -> item -> record (a workgroup's time is mostly the latency of what it loads before its first walk) */
struct hufd_chunk_rec {
    uint64_t src_off; /* bytes from the input base pointer to the chunk */
    uint64_t out_off; /* bytes from the output base pointer to the item's first symbol */
    uint64_t out_cap; /* the item's output capacity */
    uint32_t valid;   /* bytes of the item from the chunk's first on (saturated) */
    uint32_t item;
    uint32_t entry_bit; /* the item's first chunk: the bit the item starts at -- the one state the chunk is ever entered in;
                         * HUFD_NONE32: a chunk inside its item, entered as the chunk in front of it is left */
    uint32_t reserved;
};

/* Where the items of a plan that is made on the device come from (plan_kernels.hip): the caller's records in DEVICE memory,
 * a stride, or what an encode launch left */
#define HUFD_ITEMS_DEVICE_ARRAY 0u
#define HUFD_ITEMS_STRIDED 1u
#define HUFD_ITEMS_FROM_ENCODE 2u
struct hufd_item_source {
    uint32_t kind;
    uint32_t first_bit;   /* strided, decode */
    uint32_t eos_padding; /* strided, encode */
    uint32_t pad;
    const void *raw; /* device array: struct hufd_raw_dec_item[] / hufd_raw_enc_item[] */
    /* strided: item i lies at in_offset + i * in_stride (in_len bytes) and goes to out_offset + i * out_stride (room: out_capacity) */
    uint64_t in_offset, in_stride, in_len, out_offset, out_stride, out_capacity;
    /* from an encode plan's last launch: its item records and result records */
    const struct hufd_enc_item *enc_items;
    const struct hufd_enc_result *enc_results;
};

#endif /* HUFFMAN_AMD_DEVICE_TYPES_H */
