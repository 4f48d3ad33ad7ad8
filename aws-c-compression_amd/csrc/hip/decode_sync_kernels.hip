/*
 * Decode, first pass: the transfer function of every sub-chunk (entry state -> exit state, symbols) and of every chunk,
 * replacing the window / walk loop of reference source/huffman.c:230-281 for counting.
 *   dec_sync                      the long way: all entry states, nothing assumed (any chunk, on a list)
 *   dec_sync_one                  regular chunks: the sub-chunk in registers, one guessed walk and one from the true entry
 *   dec_sync_pack                 several short end-of-stream chunks a workgroup
 *   dec_sync_guess                second chance for chunks inside a stream
 *   dec_sync_few, dec_sync_true   chunks whose walks never fall into step
 *   dec_sync_tail                 the last symbols of a stream, a thread each
 */
#include "decode_common.hpp"
#include "launch_common.hpp"

namespace {

/* ------------------------------------------------------------------ decode: sync */

/*
 * The transfer function of every sub-chunk: entry state s (the first code starts s bits in)
 * -> (exit state, symbols started, or STOP).  Two phases per lane:
 *
 *   U  all entry states at once.  The walks from the ns possible start bits are followed
 *      together, lowest head first, so every stream position is looked up once however many
 *      walks pass through it; heads never sit more than one code length apart, so the set of
 *      heads is a small bit mask M relative to the lowest head p.  A walk that meets an
 *      invalid or cut-off window dies (its function value is STOP).  The phase ends as soon
 *      as ONE head is left: every surviving walk stands on that bit P0, and all that differs
 *      between them is how many symbols they took to get there (cnt[s]).  This is the
 *      self-synchronisation of Huffman streams; for the test coder P0 is ~40 bits in.
 *   R  the single surviving walk from P0 to the end of the sub-chunk: count and exit state,
 *      shared by all survivors.
 *
 * If the heads never collapse (possible for degenerate streams) phase U simply runs to the end
 * of the sub-chunk and every walk keeps its own exit state.  (source/huffman.c:213-286 is the
 * walk being reproduced; one lane's 128 bytes are one sub-chunk.)
 */
template <u32 NS> /* compile-time bound of tb.n_states: the per-state registers are unrolled */
__device__ __forceinline__ void dec_sync_chunk(
    const hufd_tables &tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u8 *d_in,
    u16 *fn_tab,   /* [chunk][state][lane] */
    u16 *cp_tab,   /* [chunk][kCpRows][lane]: checkpoints of the reference walk + merged-state mask */
    u32 *chunk_fn, /* [chunk][state] */
    u8 *chunk_regular,      /* [chunk]: cleared here */
    u32 c) {

    const u32 ns = tb.n_states;
    u32 *timg = reinterpret_cast<u32 *>(dyn_lds);
    u16 *ftab = reinterpret_cast<u16 *>(timg);                       /* [ns][lanes], over the image once the walks are done */
    u32 *gtab = timg + kChunkWords;                                  /* [groups][ns] */
    u16 *lut = reinterpret_cast<u16 *>(gtab + kGroups * ns);

    const u32 lane = threadIdx.x;
    const hufd_dec_item it = items[chunk_item[c]];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len > chunk_off ? it.in_len - chunk_off : 0;
    if (lane == 0) {
        chunk_regular[c] = 0;
    }

    HUFD_STAMP(0, 0);
    chunk_load(timg, d_in + it.in_off + chunk_off, valid);
    lut_load(lut, tb);
    const u32 shift = 32 - tb.lut_bits;

    __syncthreads();
    HUFD_STAMP(0, 1);

    const u32 rem = clamp_remaining(valid, lane);
    constexpr u32 kDead = 0xFFFFFFFFu;  /* pos[] of a walk that has died */
    constexpr u32 kNobody = 0xFFFFFFFEu; /* a head position no walk is at */

    /* ---- phase U */
    u32 pos[NS], cnt[NS];
    u32 p = kDead; /* the lowest head */
    {
        /* the first code of every entry state at once: independent lookups in the first 64 bits */
        const u64 first = ((u64)chunk_word(timg, lane, 0) << 32) | chunk_word(timg, lane, 1);
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            const u32 len = lut[(u32)((first << s) >> 32) >> shift] & 0xFFu;
            const bool ok = s < ns && len != 0 && s + len <= rem;
            pos[s] = ok ? s + len : kDead;
            cnt[s] = 1; /* a walk that dies has counted the visit that killed it: taken off below */
            p = pos[s] < p ? pos[s] : p;
        }
    }
    u32 heads = 0; /* bit j: some walk stands at p + j; bit 0 is set while any walk lives */
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        heads |= pos[s] != kDead ? 1u << (pos[s] - p) : 0u; /* all within 9 + max_bits of each other */
    }
    p = heads ? p : 0;
    bool u_live = (heads & (heads - 1u)) != 0; /* several heads, the lowest inside the sub-chunk */
    lane_window br;
    br.start(timg, lane, p);
    if (__any(u_live)) do {
        const u32 len = lut[br.peek() >> shift] & 0xFFu;
        const bool ok = len != 0 && p + len <= rem; /* a whole code of the stream starts at p */
        const u32 np = ok ? p + len : kDead;
        const u32 at = u_live ? p : kNobody;
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            const bool hit = pos[s] == at;
            cnt[s] += hit ? 1u : 0u;
            pos[s] = hit ? np : pos[s];
        }
        u32 moved = (heads & ~1u) | (ok ? 1u << len : 0u);
        moved = u_live ? moved : heads;
        const u32 j = (u_live && moved) ? (u32)__builtin_ctz(moved | 0x80000000u) : 0u;
        p += j;
        heads = moved >> j;
        br.skip(timg, lane, j);
        u_live = u_live && (heads & (heads - 1u)) != 0 && p < HUFD_DEC_SUB_BITS;
    } while (__any(u_live));
    HUFD_STAMP(0, 2);

    /* ---- phase R */
    const u32 end = rem < HUFD_DEC_SUB_BITS ? rem : HUFD_DEC_SUB_BITS;
    const bool have_ref = heads == 1u && p < HUFD_DEC_SUB_BITS;
    u32 ref_pos = p, ref_steps = 0;
    bool ref_stop = false;
    bool r_live = have_ref && ref_pos < end;
    /*
     * One bounded loop per quarter of the sub-chunk.  Where the walk stands when it enters a
     * quarter is a checkpoint: dec_emit starts an extra thread there, so its walks are a
     * quarter as long.  (Recorded between the loops, so the loop body does not pay for it.)
     */
    u32 cp_pos[kQuarters - 1], cp_steps[kQuarters - 1];
    bool cp_ok[kQuarters - 1];
#pragma unroll
    for (u32 qq = 0; qq < kQuarters; ++qq) {
        const u32 bound = (qq + 1) * kQuarterBits;
        const u32 lim = bound < end ? bound : end;
        bool act = r_live && ref_pos < lim;
        if (__any(act)) do {
            const u32 len = lut[br.peek() >> shift] & 0xFFu;
            const bool bad = len == 0 || ref_pos + len > rem;
            ref_stop = ref_stop || (act && bad);
            act = act && !bad;
            const u32 step = act ? len : 0;
            ref_pos += step;
            ref_steps += act ? 1u : 0u;
            br.skip(timg, lane, step);
            act = act && ref_pos < lim;
        } while (__any(act));
        r_live = r_live && !ref_stop && ref_pos < end;
        if (qq + 1 < kQuarters) {
            cp_ok[qq] = r_live && ref_pos - bound < 16u; /* the walk goes on, from a code start just past the boundary */
            cp_pos[qq] = ref_pos;
            cp_steps[qq] = ref_steps;
        }
    }
    if (have_ref && ref_pos < HUFD_DEC_SUB_BITS) {
        ref_stop = true; /* it ended on the last stream bit, or stopped on a bad window */
    }
    const u32 ref_exit = ref_stop ? 0 : ref_pos - HUFD_DEC_SUB_BITS;
    u16 fn[NS];
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        if (pos[s] == kDead) {
            fn[s] = fn_pack(true, 0, cnt[s] - 1u);
        } else if (have_ref) {
            fn[s] = fn_pack(ref_stop, ref_exit, (cnt[s] + ref_steps) & 0x7FFu);
        } else {
            fn[s] = fn_pack(false, pos[s] - HUFD_DEC_SUB_BITS, cnt[s]); /* it left the sub-chunk on its own */
        }
    }
    HUFD_STAMP(0, 3);
    __syncthreads(); /* every lane is done with the image: its first rows become the function table */
    HUFD_STAMP(0, 4);
#pragma unroll
    for (u32 s = 0; s < NS; ++s) {
        if (s < ns) {
            ftab[s * HUFD_DEC_LANES + lane] = fn[s];
            fn_tab[((u64)c * ns + s) * HUFD_DEC_LANES + lane] = fn[s]; /* for dec_emit */
        }
    }
    {
        /* checkpoint: [15] usable, [14:11] bits past the quarter boundary, [10:0] symbols from it to the end of the walk */
        u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES + lane;
#pragma unroll
        for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
            const u32 tail = ref_steps - cp_steps[qq];
            cp[qq * HUFD_DEC_LANES] =
                (u16)(cp_ok[qq] ? 0x8000u | ((cp_pos[qq] - (qq + 1) * kQuarterBits) << 11) | tail : 0u);
        }
        u32 merged = 0; /* entry states whose walk runs into the reference walk */
#pragma unroll
        for (u32 s = 0; s < NS; ++s) {
            merged |= (have_ref && pos[s] != kDead) ? 1u << s : 0u;
        }
        /* [15:12] where every merged state comes out: exit state, kExitStop, or kExitNoRef without a reference walk */
        const u32 common = have_ref ? (ref_stop ? kExitStop : ref_exit) : kExitNoRef;
        cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)(merged | (common << 12));
    }
    __syncthreads();

    /* fold 16 lanes per group, then the 16 groups: the chunk's own transfer function */
    if (lane < kGroups * ns) {
        const u32 g = lane / ns, start = lane % ns;
        gtab[g * ns + start] = wide_pack(chain_fold(kGroupLanes, start, [&](u32 i, u32 stt) {
            return widen(ftab[stt * HUFD_DEC_LANES + g * kGroupLanes + i]);
        }));
    }
    __syncthreads();
    if (lane < ns) {
        chunk_fn[(u64)c * ns + lane] =
            wide_pack(chain_fold(kGroups, lane, [&](u32 g, u32 stt) { return gtab[g * ns + stt]; }));
    }
    HUFD_STAMP(0, 5);
}

/* the chunks list[0 .. *list_count), a few workgroups taking turns (list == NULL: every chunk) */
template <u32 NS>
__global__ __launch_bounds__(HUFD_DEC_LANES) void dec_sync_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    u32 n_chunks,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u8 *chunk_regular,
    const u32 *list,
    const u32 *list_count,
    u32 *clear /* NULL, or the launch's list counters: the EMIT stage's words are cleared here, by the sync stage's last kernel
                * (the launch before left them as they were when its last kernel read them; nothing of this launch has
                * looked at them yet) */) {
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) {
        clear[HUFK_DEC_COUNT_EMIT] = 0;
        clear[HUFK_DEC_COUNT_DENSE] = 0;
    }
    const u32 n = list ? *list_count : n_chunks;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u32 c = list ? list[i] : i;
        if (list && chunk_regular[c] == kRegularFew) {
            continue; /* dec_sync_few, in front of this kernel on the same list, took it */
        }
        dec_sync_chunk<NS>(tb, items, chunk_item, d_in, fn_tab, cp_tab, chunk_fn, chunk_regular, c);
        __syncthreads(); /* the image is loaded anew for the next chunk */
    }
}

template <u32 LB>
struct lean_shared {
    u32 wlut[1u << LB]; /* 0x10000 - length, length 48 = no code; at a multiple of its own size */
    u32 exit_state[HUFD_DEC_LANES];
    u32 sub0[kFastMaxMeet + 4]; /* the first rows of sub-chunk 0, for the threads that try its entry states */
    u32 wave_sum[HUFD_DEC_LANES / 64];
    u32 bad;
    u32 pad[3];
    u16 hops[1u << LB]; /* 1 << code length of a window (the head it sends on), 0 = no code */
};


/* ------------------------------------------------------------------ decode: sync, regular chunks, one guessed walk */

/*
 * Round 2-4's dec_sync_lean (retired: profiles/tools/micro/retired) spent a quarter of a workgroup's life in phase U -- every entry state of every sub-chunk followed as a
 * mask of heads until one is left, five rows that cost what twenty rows of one walk cost -- only to learn where the walks
 * meet.  Nobody needs to know that in advance.  Here every lane starts ONE walk at bit 0 of its sub-chunk, a guess, and
 * walks to the end, counting (phase R over all 32 rows; a window without a code moves the walk one bit on, so that a walk
 * on a wrong phase keeps looking for the right one: it falls into step with the true walk after two rows on average).
 * The walk's state is kept at a few row boundaries.  Lane i's true entry is how lane i - 1's walk LEAVES -- right
 * whenever that walk fell into step before its sub-chunk's end -- and phase H walks from there until it stands where the
 * guessed walk stood at one of the kept boundaries: from that row on the two are one walk, and the lane's symbols are
 * H's up to there and R's from there.  Per wave H runs as long as its slowest lane needs (8 rows on average for the test
 * coder, where U + H took five rows each): 32 + 8 rows of one walk instead of U + 27 + 5.
 *
 * Exactness: a sub-chunk's records are kept only if H met R (both walks then ARE the true path from that row), neither
 * walk stepped over a window without a code on the true path (counted in the walk's state: such a window is where the
 * reference stops, source/huffman.c:240-247 -- the long way finds the stop), and the exit is a state; the chain of
 * entries is true by induction from sub-chunk 0, whose candidates (threads 0 .. ns-1: every entry state the chunk may be
 * entered in -- one, the item's first bit, for an item's first chunk) must each die or meet lane 0's walk inside the
 * sub-chunk: at any row boundary, the wave's last thread walking lane 0's guessed walk again beside them.  A lane whose two
 * walks never meet (the true one has then covered the whole sub-chunk) keeps the true walk's records and leaves as it
 * does; the lane behind it walks again from that entry, a few rounds at most.  Anything else: the chunk is not regular,
 * as for that kernel, same lists.  Same records out.
 */
constexpr u32 kOneRecs = 9;
__device__ __host__ constexpr u32 one_rec_row(u32 j) { /* the row boundaries at which the guessed walk's state is kept; the last: the sub-chunk's end */
    return j == 0 ? 6u : j == 1 ? 8u : j == 2 ? 10u : j == 3 ? 12u : j == 4 ? 16u : j == 5 ? 20u : j == 6 ? 24u : j == 7 ? 28u : 32u;
}
/* (a walk that meets the guessed one only at the sub-chunk's END has walked all of it from the true entry: it needs nothing
 * of the guessed walk but that it leaves the same way -- one lane in ten thousand meets later than row 16, and a chunk
 * that is given up for it costs the launches behind this one their latency) */
constexpr u32 kOneMaxMerge = kSubWords;
constexpr u32 kOneMaxMerge0 = kSubWords;
constexpr u32 kOneRecs0 = kOneRecs;
constexpr u32 kOneGarbageRow = 4, kOneGarbageDead = 48; /* that many windows without a code in a lane's first rows: not a stream of this coder */

/* which kept boundary a row is (kOneRecs: none) */
__device__ __host__ constexpr u32 one_rec_index(u32 row) {
    u32 at = kOneRecs;
    for (u32 j = 0; j < kOneRecs; ++j) {
        at = one_rec_row(j) == row ? j : at;
    }
    return at;
}

/*
 * state = symbols << 19 | windows without a code << 8 | position, a byte: thr + 32 - bits into the row.
 *
 * The rows of this kernel are not taken from the sub-chunk's words as they lie but from words SHIFTED by 31 - LB bits
 * (`shifted`: row r's first word starts 31 - LB bits in front of word r): the window of a code that starts k bits into row
 * r then stands at bits 2 .. LB + 1 of (xh:xl) >> (31 - k), a shift below 32 for every k of the row -- ONE 32-bit funnel
 * shift (v_alignbit_b32) of two registers that need not be a pair, its amount the position's low five bits as they stand.
 * With the words as they lie the shift runs to 62 - LB: a 64-bit shift of an aligned register pair, which the compiler put
 * together anew for every row (two v_perm a row: the byte swap fused with the copy), and the position was ten bits with the
 * other counts on top of it -- an AND in front of every compare where a byte compares by itself (SDWA).  A row's ~20
 * vector instructions became ~16, and these kernels' vector units are busy 80 % of their time (`SQ_ACTIVE_INST_VALU`).
 */
struct one_walk {
    u32 thr, mask;
    static constexpr u32 kPos = 0xFFu;
    __device__ __forceinline__ one_walk(u32 lut_bits) {
        thr = 64 + 31;
        mask = ((1u << lut_bits) - 1u) << 2;
    }
    __device__ __forceinline__ u32 state_at(u32 k) const {
        return thr + 32 - k;
    }
    __device__ __forceinline__ u32 offset_of(u32 state) const {
        return thr + 32 - (state & kPos);
    }
    static __device__ __forceinline__ u32 count_of(u32 state) {
        return state >> 19;
    }
    static __device__ __forceinline__ u32 dead_of(u32 state) {
        return (state >> 8) & 0x7FFu;
    }
    static __device__ __forceinline__ u32 entry_of(u32 len) { /* a code: a symbol more, `len` bits on; none: a mark more, one bit on */
        return len ? (1u << 19) - len : (1u << 8) - 1u;
    }
    /* a row's first word, from the stream's word in front of the row's (big-endian, as bswap gives it) and the row's own */
    template <u32 LB>
    static __device__ __forceinline__ u32 shifted(u32 before, u32 word) {
        return funnel(before, word, 31u - LB);
    }
    /* the table entry of the code the walk stands at, in the row whose shifted words are xh:xl */
    __device__ __forceinline__ u32 look(u32 state, u32 xh, u32 xl, u32 table) const {
        return lds_word_at((funnel_by_low5(xh, xl, state) & mask) | table);
    }
    /* every code of the walk that starts in that row (SURE of them without asking) */
    template <u32 SURE>
    __device__ __forceinline__ u32 row(u32 state, u32 xh, u32 xl, u32 table) const {
#pragma unroll
        for (u32 i = 0; i < SURE; ++i) {
            state += look(state, xh, xl, table);
        }
        return byte_rows_uncertain(state, xh, xl, table, mask, thr);
    }
};

template <u32 LB>
struct one_shared {
    u32 wlut[1u << LB]; /* one_walk::entry_of(length); at a multiple of its own size */
    u32 exit_state[HUFD_DEC_LANES];
    u32 sub0[kSubWords + 4]; /* sub-chunk 0's rows, for the threads that try its entry states */
    /* (every lane's guessed walk at the kept boundaries lived here in round 5, 12 KiB: the states are in registers now) */
    u32 last0;             /* lane 0's guessed walk at the sub-chunk's end: what the candidates of sub-chunk 0 ask of it */
    u32 tiny[kTailWords];  /* FOLLOW, a chunk of fewer than 136 bytes: its words for the one thread that follows the stream */
    u32 wave_sum[HUFD_DEC_LANES / 64];
    u32 bad;
    u32 decided0; /* every entry state of sub-chunk 0 died or met lane 0's walk: what dec_sync_guess needs of a chunk */
    u32 moved;    /* a lane leaves its sub-chunk in another state than the lane behind it took for its entry */
    u32 bad_mine; /* a lane's walk from its entry went wrong (which may be the entry's fault: see `moved`) */

};

template <u32 LB, u32 SURE, bool TAIL = false> /* TAIL: the chunks listed in tail_chunks (a stream ends in them) */
__global__ __launch_bounds__(HUFD_DEC_LANES, 8) void dec_sync_one_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u32 *tail_chunks,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count,
    u8 *chunk_regular,
    u32 *tail_entry, /* [chunk] TAIL: the state in which the last whole lane leaves (dec_sync_tail picks it up) */
    u32 *slow_list,  /* chunks inside a stream that are not regular by this kernel's rules but whose first sub-chunk's
                      * walks are decided: dec_sync_guess tries them its way */
    u32 *slow_count,
    u32 *long_list,  /* the others that are not regular: dec_sync's (may be the same list as slow_list) */
    u32 *long_count) {
    constexpr bool FOLLOW = false;
    const u32 c = TAIL ? wave_uniform(tail_chunks[blockIdx.x]) : blockIdx.x;
#include "decode_sync_one_body.inc"
}

/*
 * A launch with a FEW chunks that streams end in among many inside streams (one long stream: one): the grid's first
 * `n_tail` workgroups take those -- each with the careful walk over its stream's last symbols at its end (FOLLOW) --, the
 * others the chunks inside streams.  Kernels of their own for the few (dec_sync_one<TAIL> and, behind it, dec_sync_tail) are
 * one workgroup's work each, one after the other a fifth of the big kernel's time: they ran beside it on a second stream,
 * forked off the launch's and joined again -- four commands on the launch's stream, ~4 us each, in every decode.
 */
template <u32 LB, u32 SURE>
__global__ __launch_bounds__(HUFD_DEC_LANES, 8) void dec_sync_one_mixed_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u32 *tail_chunks,
    u32 n_tail,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count,
    u8 *chunk_regular,
    u32 *tail_entry,
    u32 *slow_list,
    u32 *slow_count,
    u32 *long_list,
    u32 *long_count) {
    if (blockIdx.x < n_tail) {
        constexpr bool TAIL = true, FOLLOW = true;
        const u32 c = wave_uniform(tail_chunks[blockIdx.x]);
#include "decode_sync_one_body.inc"
    } else {
        constexpr bool TAIL = false, FOLLOW = false;
        const u32 c = blockIdx.x - n_tail;
#include "decode_sync_one_body.inc"
    }
}


/* ------------------------------------------------------------------ decode: sync, several short end-of-stream chunks a workgroup */

/*
 * A batch of items of a few KiB each is all chunks that streams END in, one per item, with a handful of whole lanes: 19
 * of 256 for a 2 KiB item.  dec_sync_one<TAIL> gives such a chunk a workgroup of its own -- one wave of 19 lanes, a table
 * of 4 KiB filled, three barriers -- and the batch decodes at a seventh of a stream's rate (bench.py, the mid_items leg).
 * Here a workgroup takes SEVERAL such chunks: its 256 threads are `slots` of `width` lanes (the most whole lanes any
 * end-of-stream chunk of the launch has, at least 16), a chunk a slot, so that the waves are full and the table and the
 * barriers are shared.  Same phases per lane, same records out as dec_sync_one<TAIL>; what is per chunk there (the
 * candidates' walks of sub-chunk 0, the sum of the lanes' symbols, the verdict) is per slot here, through LDS words
 * instead of wave votes, because a slot need not start on a wave.
 */

template <u32 LB>
struct pack_shared {
    u32 wlut[1u << LB]; /* 0x10000 - length, length 48 = no code; at a multiple of its own size */
    u32 exit_state[HUFD_DEC_LANES];
    u32 sub0[kPackMaxSlots][kFastMaxMeet + 4]; /* a slot's first rows of sub-chunk 0, for the threads that try its entry states */
    u32 sum[kPackMaxSlots];      /* symbols of the slot's lanes >= 1 */
    u32 bad[kPackMaxSlots];
    u32 alive[kPackMaxSlots];    /* entry states of the slot's chunk that reach the meeting bit */
    u32 meet0[kPackMaxSlots];    /* of the slot's sub-chunk 0: meeting row << 8 | meeting bit | its walks have met << 31 */
    u32 tail0[kPackMaxSlots];    /* ... its symbols from the meeting bit on */
    u16 hops[1u << LB]; /* 1 << code length of a window (the head it sends on), 0 = no code */
};

/* (six waves a SIMD: at the 64 registers that eight allow the kernel spills 22 -- and a spill inside these divergent walks
 * is what once came back wrong, DESIGN.md 5 "Tried"; 80 registers, none) */
template <u32 LB, u32 SURE>
__global__ __launch_bounds__(HUFD_DEC_LANES, 6) void dec_sync_pack_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u32 *tail_chunks,
    u32 n_tail,
    u32 width, /* lanes a slot: >= 16, >= the whole lanes of every chunk of the launch, <= 128 */
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count,
    u8 *chunk_regular,
    u32 *tail_entry,
    u32 *long_list,
    u32 *long_count) {

    pack_shared<LB> &sh = *reinterpret_cast<pack_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 t = threadIdx.x;
    const u32 slots = HUFD_DEC_LANES / width;
    const u32 slot = t / width, lane = t % width;
    const row_walk rw(LB, tb.max_bits);
    const u32 table = lds_offset_of(sh.wlut);
    /* the table: every thread its share, whatever becomes of its slot */
    {
        constexpr u32 kLutPerLane = (1u << LB) / HUFD_DEC_LANES;
        u32 lut_raw[kLutPerLane];
#pragma unroll
        for (u32 j = 0; j < kLutPerLane; ++j) {
            lut_raw[j] = tb.dec_lut[(t + j * HUFD_DEC_LANES) >> (LB - tb.lut_bits)];
        }
#pragma unroll
        for (u32 j = 0; j < kLutPerLane; ++j) {
            const u32 len = lut_raw[j] & 0xFFu;
            sh.wlut[t + j * HUFD_DEC_LANES] = 0x10000u - (len ? len : kWalkDeadLen);
            sh.hops[t + j * HUFD_DEC_LANES] = (u16)(len ? 1u << len : 0u);
        }
    }
    const u32 li = blockIdx.x * slots + slot;
    const bool have = slot < slots && li < n_tail;
    const u32 c = have ? tail_chunks[li] : 0u;
    const hufd_chunk_rec rec = chunk_rec[c];
    const u64 valid = rec.valid;
    const u8 *src = d_in + rec.src_off;
    const u32 n_full = valid >= 8u ? (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES) : 0u;
    const bool eligible = tb.lut_bits <= LB && tb.max_bits <= HUFD_DEC_MAX_LUT_BITS && rw.sure >= SURE && (table & ((4u << LB) - 1u)) == 0 &&
                          n_full <= width;
    /* (threads that leave here still count for the barriers below as long as their wave lives: `mine` keeps them out of
     * everything but the barriers) */
    bool mine = have;
    if (mine && n_full == 0 && tb.lut_bits <= HUFD_DEC_MAX_LUT_BITS) {
        if (lane == 0) {
            chunk_regular[c] = 3; /* fewer than 136 bytes: the whole chunk is one thread's work in dec_sync_tail / dec_emit_tail */
        }
        mine = false;
    }
    if (mine && !eligible) {
        if (lane == 0) {
            chunk_regular[c] = 0;
            long_list[atomicAdd(long_count, 1u)] = c;
        }
        mine = false;
    }
    const bool active = mine && lane < n_full;
    u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES;
    u32 w[kFastRows];
    {
        const u32 from = active ? lane : 0u;
        const u8 *at = mine ? src + (u64)from * HUFD_DEC_SUB_BYTES : d_in; /* (a thread without a chunk reads the input's first bytes: never looked at) */
        const unaligned_uint4 *line = reinterpret_cast<const unaligned_uint4 *>(at);
#pragma unroll
        for (u32 q = 0; q < kSubWords / 4; ++q) {
            const unaligned_uint4 v = mine ? line[q] : unaligned_uint4{0, 0, 0, 0};
            w[4 * q + 0] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        w[kSubWords] = mine ? reinterpret_cast<const unaligned_u32 *>(at + HUFD_DEC_SUB_BYTES)->x : 0u;
    }
#pragma unroll
    for (u32 r = 0; r < kFastRows; ++r) {
        w[r] = __builtin_bswap32(w[r]);
    }
    if (slot < kPackMaxSlots && lane == 0) {
        sh.bad[slot] = 0;
        sh.sum[slot] = 0;
        sh.alive[slot] = 0;
#pragma unroll
        for (u32 r = 0; r <= kFastMaxMeet; ++r) {
            sh.sub0[slot][r] = w[r];
        }
    }
    __syncthreads();

    /* U: all entry states as one mask of heads per row, until every lane of the wave is down to one */
    u64 heads = active ? (1ull << ns) - 1ull : 0ull;
    u32 meet_row = 0; /* the same for the whole wave */
    bool one = false, settled = false;
#pragma unroll
    for (u32 r = 0; r < kFastMaxMeet; ++r) {
        if (!settled) {
            heads = r == 0 ? union_first_row<LB>(ns, active, w[0], w[1], sh.hops) : union_row_fast<LB>(heads, w[r], w[r + 1], sh.hops);
            one = heads != 0 && (heads & (heads - 1)) == 0;
            meet_row = r + 1;
            settled = __all(one || heads == 0);
        }
    }
    const u32 meet_bit = one ? (u32)__builtin_ctzll(heads) : 0u; /* bits into row meet_row */
    bool ok = !active || (one && settled);

    /* R: the one walk from the meeting bit to the end of the sub-chunk */
    u32 state = rw.state_at(meet_bit, 0);
    u32 cp_state[kQuarters - 1] = {0, 0, 0};
    bool dead = false;
#pragma unroll
    for (u32 r = 1; r < kSubWords; ++r) {
        if (r >= meet_row) {
            if (r % (kSubWords / kQuarters) == 0) {
                cp_state[r / (kSubWords / kQuarters) - 1] = state;
            }
            state = lean_row<SURE>(state, w[r], w[r + 1], table, rw);
            dead = dead || rw.died(state);
            state = rw.next_row(state, dead); /* (lanes without data walk zeros: put back on a row start, their state stays in range) */
        }
    }
    const u32 ref_count = state >> 16; /* symbols from the meeting bit to the end of the sub-chunk */
    const u32 ref_exit = rw.offset_of(state);
    ok = ok && (!active || (!dead && ref_exit < ns));
    sh.exit_state[t] = ref_exit;
    if (mine && lane == 0) {
        sh.meet0[slot] = (meet_row << 8) | meet_bit | (one ? 0x80000000u : 0u);
        sh.tail0[slot] = ref_count;
    }
    __syncthreads();

    /* H: my own sub-chunk from my true entry state, to the meeting bit */
    const u32 entry = lane ? sh.exit_state[t - 1] : 0u;
    u32 count;
    u32 head_cp = 0; /* the head walk where it enters the second quarter, when the meeting row lies behind that */
    const bool late = meet_row > kSubWords / kQuarters;
    {
        u32 st = rw.state_at(entry < ns ? entry : 0u, 0);
        bool dd = false;
#pragma unroll
        for (u32 r = 0; r < kFastMaxMeet; ++r) {
            if (r < meet_row) {
                if (r == kSubWords / kQuarters) {
                    head_cp = st;
                }
                st = lean_row<SURE>(st, w[r], w[r + 1], table, rw);
                dd = dd || rw.died(st);
                st = rw.next_row(st, dd);
            }
        }
        const bool reached = !dd && rw.offset_of(st) == meet_bit;
        ok = ok && (lane == 0 || !active || reached);
        count = active ? (st >> 16) + ref_count : 0u; /* symbols of the true path that start in my sub-chunk (lanes >= 1) */
    }

    /* H: sub-chunk 0 of my slot's chunk from every entry state the chunk may be entered in (lanes 0 .. ns-1 of the slot),
     * step by step: the count of a walk that dies has to be right */
    u32 cand_count = 0, cand_dead = 0;
    bool cand_reached = false;
    if (mine && lane < ns) {
        const u32 m0 = sh.meet0[slot], rows0 = (m0 >> 8) & 0xFFu, target = m0 & 0xFFu;
        u32 st = rw.state_at(lane, 0);
        bool dd = false;
        u32 hi = sh.sub0[slot][0];
        for (u32 r = 0; r < rows0; ++r) {
            const u32 lo = sh.sub0[slot][r + 1];
            st = lean_row<SURE, true>(st, hi, lo, table, rw);
            const bool now = rw.died(st) && !dd;
            cand_dead = now ? (st >> 16) - 1u : cand_dead; /* the step that found no code is not a symbol */
            dd = dd || now;
            st = rw.next_row(st, dd);
            hi = lo;
        }
        cand_reached = !dd && rw.offset_of(st) == target;
        cand_count = (st >> 16) + sh.tail0[slot];
        if (cand_reached) {
            atomicOr(&sh.alive[slot], 1u << lane);
        }
    }
    if (active && lane) {
        atomicAdd(&sh.sum[slot], count);
    }
    if (mine && !ok) {
        sh.bad[slot] = 1;
    }
    __syncthreads();
    if (!mine) {
        return;
    }
    if (sh.bad[slot]) {
        if (lane == 0) {
            chunk_regular[c] = 0;
            long_list[atomicAdd(long_count, 1u)] = c;
        }
        return;
    }

    /* the tables dec_scan and dec_emit read (the regular chunks' format) */
    if (active) {
        u16 *mcp = cp + lane;
#pragma unroll
        for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
            const bool usable = (qq + 1) * (kSubWords / kQuarters) >= meet_row;
            u32 tail = ref_count - (cp_state[qq] >> 16), bits = rw.offset_of(cp_state[qq]);
            bool have_cp = usable;
            if (qq == 0 && late && lane != 0) {
                tail = count - (head_cp >> 16);
                bits = rw.offset_of(head_cp);
                have_cp = true;
            }
            mcp[qq * HUFD_DEC_LANES] = (u16)(have_cp ? 0x8000u | (bits << 11) | tail : 0u);
        }
        lane_count[(u64)c * HUFD_DEC_LANES + lane] = (u16)(lane ? count : ref_count);
        mcp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)((lane ? 1u << entry : sh.alive[slot]) | (ref_exit << 12));
    }
    /* the chunk's lanes behind the whole ones: never reached, as far as this kernel knows (dec_sync_tail follows the true
     * path through the one or two sub-chunks the stream ends in and rewrites their records) */
    if (lane < 2 && n_full + lane < HUFD_DEC_LANES) {
        /* (the two sub-chunks the stream can end in; dec_emit_fast<TAIL> takes the lanes behind them as empty without
         * looking: writing a record for each of the chunk's 256 lanes cost this kernel more than its walks) */
        const u32 l = n_full + lane;
#pragma unroll
        for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
            cp[qq * HUFD_DEC_LANES + l] = 0;
        }
        lane_count[(u64)c * HUFD_DEC_LANES + l] = 0;
        cp[(kQuarters - 1) * HUFD_DEC_LANES + l] = (u16)(kExitStop << 12);
    }
    if (lane + 1 == n_full) {
        tail_entry[c] = ref_exit;
    }
    if (lane == 0) {
        chunk_regular[c] = 2;
    }
    if (lane < ns) {
        const u32 rest = sh.sum[slot];
        const u32 first_exit = sh.exit_state[slot * width];
        fn_tab[((u64)c * ns + lane) * HUFD_DEC_LANES] =
            cand_reached ? fn_pack(false, first_exit, cand_count & 0x7FFu) : fn_pack(true, 0, cand_dead);
        /* (symbols of the whole lanes only, and no exit yet: dec_sync_tail adds the stream's last symbols and how it ends) */
        chunk_fn[(u64)c * ns + lane] = cand_reached ? wide_pack(false, 0u, cand_count + rest) : wide_pack(true, 0, cand_dead);
    }
}

/* ------------------------------------------------------------------ decode: sync, second chance for chunks inside a stream */

/*
 * Rounds 2-4's dec_sync_lean wanted ALL entry states of EVERY sub-chunk to fall into one walk within 16 rows; a coder that
 * synchronises slowly on its own kind of data (codes of 4 .. 12 bits on symbols drawn to match them: a quarter of the
 * sub-chunks still have several heads after 16 rows) had no regular chunk, and this kernel was written for those.
 * dec_sync_one (round 5) takes late meetings itself; what it gives up inside a stream are chunks with a wave most of
 * whose lanes stay apart after six rows, or with a window without a code on the true path.  This kernel takes them from
 * its list and asks less: only sub-chunk 0, whose entry state nobody in the chunk can know, goes through phase U (all
 * entry states as a mask of heads, union_row_fast)
 * and the candidates' walks as there.  Every other lane starts ONE walk kGuessRows rows in front of its sub-chunk
 * (a window without a code moves it one bit on), takes where that walk crosses into the sub-chunk as its entry state
 * -- a guess -- and walks on to the end, counting.  Then lane j's guess is checked against lane j - 1's exit state,
 * true by induction from lane 0; who guessed wrong walks again from the true state (its exit may change: the check
 * is repeated).  Exact: at the end every lane's walk starts where its neighbour's ends.  What is not settled after
 * kGuessRounds, or not regular for another reason, goes on the next list, for dec_sync.  Same tables out.  (As the
 * FIRST kernel for every chunk this was measured slower than dec_sync_lean on the test coder: 0.63 against 0.44 ms.)
 */
constexpr u32 kGuessRows = 8;
constexpr u32 kGuessRounds = 6;
/* a window without a code: one bit on, and a mark above the count that is looked at once a row (counts stay below 512) */
constexpr u32 kGuessDeadMark = 1u << 25;
constexpr u32 kGuessDeadEntry = kGuessDeadMark + 0x10000u - 1u;

template <u32 LB, u32 SURE>
__device__ __forceinline__ void dec_sync_guess_chunk(
    const u32 c,
    const hufd_tables &tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count,
    u8 *chunk_regular,
    u32 *slow_list,
    u32 *slow_count) {

    lean_shared<LB> &sh = *reinterpret_cast<lean_shared<LB> *>(dyn_lds);
    u32 *again = sh.pad; /* [2]: somebody walks again, one flag for the even rounds, one for the odd ones */
    const u32 ns = tb.n_states;
    const u32 lane = threadIdx.x;
    const hufd_chunk_rec rec = chunk_rec[c];
    const u8 *src = d_in + rec.src_off;
    const row_walk rw(LB, tb.max_bits);
    const u32 table = lds_offset_of(sh.wlut);
    const bool eligible = tb.lut_bits <= LB && tb.max_bits <= HUFD_DEC_MAX_LUT_BITS && tb.min_bits >= 3 && rw.sure >= SURE &&
                          (table & ((4u << LB) - 1u)) == 0;
    if (!eligible) {
        if (lane == 0) {
            slow_list[atomicAdd(slow_count, 1u)] = c;
        }
        return;
    }

    /* First what decides whether the chunk can be regular at all, and only that: U for sub-chunk 0 -- all entry states
     * as one mask of heads per row, until one is left -- by wave 0 on words it loads for this alone.  If its walks do not
     * meet, nothing the other lanes find out helps: the chunk goes on at once, having cost a table and sixteen rows (a
     * stream that never synchronises gets here with every chunk). */
    for (u32 i = lane; i < (1u << LB); i += HUFD_DEC_LANES) {
        const u32 len = tb.dec_lut[i >> (LB - tb.lut_bits)] & 0xFFu;
        sh.wlut[i] = len ? 0x10000u - len : kGuessDeadEntry;
        sh.hops[i] = (u16)(len ? 1u << len : 0u);
    }
    if (lane == 0) {
        sh.bad = 0;
        again[0] = 0;
    }
    __syncthreads();
    u32 meet_row = 0, meet_bit = 0; /* wave 0: where sub-chunk 0's walks meet */
    if (lane < kWave) {
        u32 w0[kFastMaxMeet + 1];
#pragma unroll
        for (u32 r = 0; r <= kFastMaxMeet; ++r) {
            w0[r] = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(src + 4 * r)->x);
        }
        u64 heads = lane == 0 ? (1ull << ns) - 1ull : 0ull;
        bool one = false, settled = false;
#pragma unroll
        for (u32 r = 0; r < kFastMaxMeet; ++r) {
            if (!settled) {
                heads = union_row_fast<LB>(heads, w0[r], w0[r + 1], sh.hops);
                one = heads != 0 && (heads & (heads - 1)) == 0;
                meet_row = r + 1;
                settled = __all(one || heads == 0);
            }
        }
        meet_bit = __shfl(one ? (u32)__builtin_ctzll(heads) : 0u, 0); /* bits into row meet_row */
        if (lane == 0) {
            if (!(one && settled)) {
                sh.bad = 1;
            }
#pragma unroll
            for (u32 r = 0; r <= kFastMaxMeet; ++r) {
                sh.sub0[r] = w0[r];
            }
        }
    }
    __syncthreads();
    if (sh.bad) {
        if (lane == 0) {
            slow_list[atomicAdd(slow_count, 1u)] = c;
        }
        return;
    }

    u32 w[kFastRows], pw[kGuessRows];
    {
        const unaligned_uint4 *line = reinterpret_cast<const unaligned_uint4 *>(src + (u64)lane * HUFD_DEC_SUB_BYTES);
        /* (lane 0 reads its own first rows here: never looked at, and inside the chunk) */
        const unaligned_uint4 *front =
            reinterpret_cast<const unaligned_uint4 *>(src + (u64)lane * HUFD_DEC_SUB_BYTES - (lane ? kGuessRows * 4 : 0u));
#pragma unroll
        for (u32 q = 0; q < kGuessRows / 4; ++q) {
            const unaligned_uint4 v = front[q];
            pw[4 * q + 0] = __builtin_bswap32(v.x);
            pw[4 * q + 1] = __builtin_bswap32(v.y);
            pw[4 * q + 2] = __builtin_bswap32(v.z);
            pw[4 * q + 3] = __builtin_bswap32(v.w);
        }
#pragma unroll
        for (u32 q = 0; q < kSubWords / 4; ++q) {
            const unaligned_uint4 v = line[q];
            w[4 * q + 0] = __builtin_bswap32(v.x);
            w[4 * q + 1] = __builtin_bswap32(v.y);
            w[4 * q + 2] = __builtin_bswap32(v.z);
            w[4 * q + 3] = __builtin_bswap32(v.w);
        }
        w[kSubWords] = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(src + (u64)(lane + 1) * HUFD_DEC_SUB_BYTES)->x);
    }

    /* the walk in front of the sub-chunk (lanes >= 1): where it crosses into the sub-chunk is the guess */
    u32 guess = 0;
    bool guess_ok = true;
    if (lane) {
        u32 st = rw.state_at(0, 0);
#pragma unroll
        for (u32 r = 0; r < kGuessRows; ++r) {
            st = lean_row<SURE>(st, pw[r], r + 1 < kGuessRows ? pw[r + 1] : w[0], table, rw);
            st += 32u;
        }
        guess = rw.offset_of(st);
        guess_ok = guess < ns;
        guess = guess_ok ? guess : 0u;
    }

    /* the one walk of a sub-chunk from its entry state (lane 0: from the meeting bit in row meet_row), then the guesses
     * against the exit states; whoever guessed wrong walks again, from the true entry state (the others stand by) */
    u32 cp_state[kQuarters - 1] = {0, 0, 0};
    u32 state = 0, exit_bit = 0, entry = 0;
    bool ok = true;
    bool walking = true;
    u32 from_bit = lane ? guess : meet_bit;
    const u32 first_row = lane ? 0u : meet_row; /* (meet_row >= 1) */
    for (u32 round = 0;; ++round) {
        if (__any(walking)) {
            /* (the words as values the compiler cannot trace through the rounds: it otherwise builds every row's 64-bit
             * window register pair once, in front of the loop -- twice the registers, and a value spilled inside this
             * divergent loop has come back wrong on this toolchain) */
#pragma unroll
            for (u32 r = 0; r < kFastRows; ++r) {
                opaque(w[r]);
            }
            u32 st = rw.state_at(from_bit, 0);
            bool dd = false;
#pragma unroll
            for (u32 r = 0; r < kSubWords; ++r) {
                if (walking && r >= first_row) {
                    if (r && r % (kSubWords / kQuarters) == 0) {
                        cp_state[r / (kSubWords / kQuarters) - 1] = st;
                    }
                    st = lean_row<SURE>(st, w[r], w[r + 1], table, rw);
                    dd = dd || st >= kGuessDeadMark; /* (it walks on, a bit at a time: nothing of it is kept) */
                    st += 32u;
                }
            }
            if (walking) {
                state = st & (kGuessDeadMark - 1u);
                exit_bit = rw.offset_of(st);
                ok = !dd && exit_bit < ns;
                sh.exit_state[lane] = ok ? exit_bit : 0xFFu;
            }
        }
        __syncthreads();
        entry = lane ? sh.exit_state[lane - 1] : 0u;
        walking = lane != 0 && entry < ns && (!guess_ok || entry != guess);
        if (walking) {
            again[round & 1u] = 1;
            from_bit = guess = entry;
            guess_ok = true;
        }
        if (lane == 0) {
            again[(round & 1u) ^ 1u] = 0; /* (the next round's: last read a round ago, in front of this round's barrier) */
        }
        __syncthreads();
        if (!again[round & 1u]) {
            break;
        }
        if (round == kGuessRounds) {
            ok = false; /* (every lane leaves the loop in the same round) */
            break;
        }
    }
    ok = ok && (lane == 0 || entry < ns);
    const u32 count = state >> 16; /* symbols of the true path that start in my sub-chunk (lane 0: from the meeting bit on) */

    /* sub-chunk 0 from every entry state the chunk may be entered in (threads 0 .. ns-1), step by step to the meeting
     * row: the count of a walk that dies has to be right */
    u32 cand_count = 0, cand_dead = 0;
    bool cand_reached = false;
    u64 cand_alive = 0;
    if (lane < kWave) {
        const u32 target = meet_bit, tail0 = __shfl(count, 0); /* sub-chunk 0 is lane 0's */
        u32 st = rw.state_at(lane < ns ? lane : 0u, 0);
        bool dd = false;
        u32 hi = sh.sub0[0];
        for (u32 r = 0; r < meet_row; ++r) {
            const u32 lo = sh.sub0[r + 1];
            const u64 pair = ((u64)hi << 32) | lo;
            while (!dd && (st & 0xFFFFu) > rw.thr) {
                const u32 e = lds_word_at(((u32)(pair >> (st & 63u)) & rw.mask) | table);
                if (e & kGuessDeadMark) {
                    dd = true;
                    cand_dead = st >> 16; /* the symbols in front of the window without a code */
                } else {
                    st += e;
                }
            }
            st = rw.next_row(st, dd);
            hi = lo;
        }
        cand_reached = !dd && lane < ns && rw.offset_of(st) == target;
        cand_alive = __ballot(cand_reached);
        cand_count = (st >> 16) + tail0;
    }

    const u32 wsum = wave_sum(lane ? count : 0u);
    if ((lane & (kWave - 1)) == 0) {
        sh.wave_sum[lane / kWave] = wsum;
    }
    if (!ok) {
        sh.bad = 1;
    }
    __syncthreads();
    if (sh.bad) {
        if (lane == 0) {
            slow_list[atomicAdd(slow_count, 1u)] = c; /* (chunk_regular[c] is 0 already: dec_sync_one's) */
        }
        return;
    }

    /* the tables dec_scan and dec_emit read (the regular chunks' format).  (The lane number as a value the compiler cannot trace:
     * where the records go is worked out here, not in front of the walks where the registers are needed.) */
    u32 lane_o = lane;
    opaque(lane_o);
    u16 *fn_out = fn_tab + (u64)c * ns * HUFD_DEC_LANES;
    u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES + lane_o;
#pragma unroll
    for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
        /* lane 0: a checkpoint in front of the meeting row is not on its walk (dec_emit_fast goes on from the chunk's entry) */
        const bool have = lane != 0 || (qq + 1) * (kSubWords / kQuarters) >= meet_row;
        const u32 tail = count - (cp_state[qq] >> 16), bits = rw.offset_of(cp_state[qq]);
        cp[qq * HUFD_DEC_LANES] = (u16)(have ? 0x8000u | (bits << 11) | tail : 0u);
    }
    lane_count[(u64)c * HUFD_DEC_LANES + lane_o] = (u16)count;
    cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)((lane ? 1u << entry : (u32)cand_alive) | (exit_bit << 12));
    if (lane == 0) {
        chunk_regular[c] = 1;
    }
    if (lane < ns) {
        u32 rest = 0;
#pragma unroll
        for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
            rest += sh.wave_sum[wv];
        }
        const u32 first_exit = sh.exit_state[0];
        const u32 last_exit = sh.exit_state[HUFD_DEC_LANES - 1];
        fn_out[(u64)lane_o * HUFD_DEC_LANES] =
            cand_reached ? fn_pack(false, first_exit, cand_count & 0x7FFu) : fn_pack(true, 0, cand_dead);
        chunk_fn[(u64)c * ns + lane_o] =
            cand_reached ? wide_pack(false, last_exit, cand_count + rest) : wide_pack(true, 0, cand_dead);
    }
}

template <u32 LB, u32 SURE>
__global__ __launch_bounds__(HUFD_DEC_LANES, 4) void dec_sync_guess_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count,
    u8 *chunk_regular,
    const u32 *given_up,       /* dec_sync_one's list ... */
    const u32 *given_up_count,
    u32 *slow_list,            /* ... and the one dec_sync works through */
    u32 *slow_count) {
    const u32 n = *given_up_count;
    for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
        const u32 c = given_up[k];
        if (chunk_rec[c].valid < (u64)HUFD_DEC_CHUNK_BYTES + 8u) {
            /* holds the end of its stream: not this kernel's */
            if (threadIdx.x == 0) {
                slow_list[atomicAdd(slow_count, 1u)] = c;
            }
            continue;
        }
        dec_sync_guess_chunk<LB, SURE>(c, tb, chunk_rec, d_in, fn_tab, cp_tab, chunk_fn, lane_count, chunk_regular, slow_list, slow_count);
        __syncthreads(); /* the tables in LDS are written again */
    }
}

/* ------------------------------------------------------------------ decode: sync, chunks whose walks do not fall into step */

/*
 * The chunks inside a stream that dec_sync_one and dec_sync_guess gave up: in some sub-chunk the walks from the
 * possible entry bits do not become one -- one symbol over and over (as many walks as its code has bits, each valid
 * for ever), two symbols of one length taking turns, any stream whose code lengths share a divisor.  An adversary picks
 * those; the long way (dec_sync) follows every entry's walk a bit of the stream at a time out of an LDS image, 1 ms for
 * 160 MB, and the emit kernel behind it has no checkpoints to start threads at.  Here, as for the long-code coders
 * (dec_wide_fn):
 *   dec_sync_few    a lane's sub-chunk in registers as in dec_sync_one; from every entry bit a walk over the first two
 *                   rows, and from every DISTINCT bit these land on ONE walk to the end of the sub-chunk (as many as the
 *                   stream has phases, at most kFewMaxWalks -- more, or the end of a stream in the chunk: the long way
 *                   after all).  Each entry's (exit, symbols, or where its walk stops) goes into the tables in the long
 *                   way's format, folded to the chunk's function for dec_scan as there.
 *   dec_sync_true   behind dec_scan, which says where each such chunk is truly entered: one thread follows the lanes'
 *                   functions to every lane's true entry, every lane walks its sub-chunk ONCE more from there and leaves
 *                   the records of a regular chunk (count, exit, a checkpoint a quarter, all on the true walk) -- so the
 *                   fast emit kernels take the chunk, a thread a quarter.  A true walk that stops in the chunk leaves it
 *                   to the long way's emit kernel with dec_sync_few's tables.
 * Between the two a chunk is marked kRegularFew in chunk_regular (nobody else looks at it then).
 */
constexpr u32 kFewMaxWalks = 8;
constexpr u32 kFewHeadRows = 2;

template <u32 LB>
struct few_shared {
    u32 wlut[1u << LB]; /* 0x10000 - length, length 48 = no code; at a multiple of its own size */
    u16 ftab[HUFD_DEC_MAX_STATES * HUFD_DEC_LANES];
    u32 gtab[kGroups * HUFD_DEC_MAX_STATES];
    u32 entry_of[HUFD_DEC_LANES];
    u32 bad;
    u32 pad[3];
};

/* one row of a walk whose count has to be right when it dies (dec_sync_one's walks of sub-chunk 0's entries) */
template <u32 LB>
__device__ __forceinline__ u32 few_row(u32 st, u32 hi, u32 lo, u32 table, const row_walk &rw, bool &dd, u32 &dead_count) {
    st = lean_row<0, true, true>(st, hi, lo, table, rw);
    const bool now = rw.died(st) && !dd;
    dead_count = now ? (st >> 16) - 1u : dead_count; /* the step that found no code is not a symbol */
    dd = dd || now;
    return rw.next_row(st, dd);
}

template <u32 LB>
__device__ __forceinline__ void few_load_words(u32 (&w)[kFastRows], const u8 *sub) {
    const unaligned_uint4 *line = reinterpret_cast<const unaligned_uint4 *>(sub);
#pragma unroll
    for (u32 q = 0; q < kSubWords / 4; ++q) {
        const unaligned_uint4 v = line[q];
        w[4 * q + 0] = __builtin_bswap32(v.x);
        w[4 * q + 1] = __builtin_bswap32(v.y);
        w[4 * q + 2] = __builtin_bswap32(v.z);
        w[4 * q + 3] = __builtin_bswap32(v.w);
    }
    w[kSubWords] = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(sub + HUFD_DEC_SUB_BYTES)->x);
}

template <u32 LB>
__device__ __forceinline__ void few_table(few_shared<LB> &sh, const hufd_tables &tb, u32 lane) {
    for (u32 i = lane; i < (1u << LB); i += HUFD_DEC_LANES) {
        const u32 len = tb.dec_lut[i >> (LB - tb.lut_bits)] & 0xFFu;
        sh.wlut[i] = 0x10000u - (len ? len : kWalkDeadLen);
    }
}

template <u32 LB>
__global__ __launch_bounds__(HUFD_DEC_LANES, 4) void dec_sync_few_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u8 *chunk_regular,
    const u32 *list, /* what the kernels in front gave up; dec_sync, behind this one, goes through it again and skips the chunks marked here */
    const u32 *list_count,
    u32 *done_list, /* the chunks taken here, for dec_sync_true */
    u32 *done_count) {

    few_shared<LB> &sh = *reinterpret_cast<few_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 lane = threadIdx.x;
    const u32 table = lds_offset_of(sh.wlut);
    const row_walk rw(LB, tb.max_bits);
    const u32 n = *list_count;
    if (n == 0 || tb.lut_bits > LB || tb.max_bits > HUFD_DEC_MAX_LUT_BITS || ns > HUFD_DEC_MAX_STATES || (table & ((4u << LB) - 1u)) != 0) {
        return; /* (nearly always: nothing was given up) */
    }
    few_table<LB>(sh, tb, lane);
    if (lane == 0) {
        sh.bad = 0;
    }
    __syncthreads();
    for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
        const u32 c = list[k];
        const hufd_chunk_rec rec = chunk_rec[c];
        if (rec.valid < (u64)HUFD_DEC_CHUNK_BYTES + 8u) {
            continue; /* holds the end of its stream: the long way's */
        }
        u32 w[kFastRows];
        few_load_words<LB>(w, d_in + rec.src_off + (u64)lane * HUFD_DEC_SUB_BYTES);

        /* every entry bit over the first rows: where it lands and what it counted (kept in the entry's place in the LDS
         * table, the landing bit where the exit will be: registers are for the sub-chunk), or where it died */
        u32 landed = 0, pending = 0;
#pragma unroll
        for (u32 s = 0; s < HUFD_DEC_MAX_STATES; ++s) {
            if (s < ns) {
                u32 st = rw.state_at(s, 0), dead_count = 0;
                bool dd = false;
#pragma unroll
                for (u32 r = 0; r < kFewHeadRows; ++r) {
                    st = few_row<LB>(st, w[r], w[r + 1], table, rw, dd, dead_count);
                }
                const u32 o = rw.offset_of(st);
                const bool on = !dd && o < 16u;
                sh.ftab[s * HUFD_DEC_LANES + lane] = on ? fn_pack(false, o, (st >> 16) & 0x7FFu) : fn_pack(true, 0, dead_count & 0x7FFu);
                pending |= on ? 1u << s : 0u;
                landed |= on ? 1u << o : 0u;
            }
        }
        bool ok = __builtin_popcount(landed) <= (int)kFewMaxWalks;
        /* one walk from every bit a walk landed on, to the end of the sub-chunk */
        u32 todo = ok ? landed : 0u;
        while (todo) {
            const u32 o = (u32)__builtin_ctz(todo);
            todo &= todo - 1;
            u32 st = rw.state_at(o, 0), dead_count = 0;
            bool dd = false;
#pragma unroll
            for (u32 r = kFewHeadRows; r < kSubWords; ++r) {
                st = few_row<LB>(st, w[r], w[r + 1], table, rw, dd, dead_count);
            }
            const u32 ex = rw.offset_of(st);
            ok = ok && (dd || ex < ns);
            const u32 more = dd ? dead_count : st >> 16;
            for (u32 s = 0; s < ns; ++s) {
                const u32 f = sh.ftab[s * HUFD_DEC_LANES + lane];
                if (((pending >> s) & 1u) && ((f >> 11) & 15u) == o) {
                    sh.ftab[s * HUFD_DEC_LANES + lane] = fn_pack(dd, dd ? 0u : ex & 15u, ((f & 0x7FFu) + more) & 0x7FFu);
                    pending &= ~(1u << s);
                }
            }
        }
        if (!ok) {
            sh.bad = 1;
        }
        __syncthreads();
        const bool bad = sh.bad != 0;
        __syncthreads();
        if (bad) {
            if (lane == 0) {
                sh.bad = 0;
            }
            __syncthreads();
            continue; /* (too many walks in some lane: the long way) */
        }
        /* the tables, as dec_sync leaves them for a chunk without a walk all entries run into: no checkpoints */
        for (u32 s = 0; s < ns; ++s) {
            fn_tab[((u64)c * ns + s) * HUFD_DEC_LANES + lane] = sh.ftab[s * HUFD_DEC_LANES + lane];
        }
        {
            u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES + lane;
#pragma unroll
            for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
                cp[qq * HUFD_DEC_LANES] = 0;
            }
            cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)(kExitNoRef << 12);
        }
        __syncthreads();
        if (lane < kGroups * ns) {
            const u32 g = lane / ns, start = lane % ns;
            sh.gtab[g * ns + start] = wide_pack(chain_fold(kGroupLanes, start, [&](u32 i, u32 stt) {
                return widen(sh.ftab[stt * HUFD_DEC_LANES + g * kGroupLanes + i]);
            }));
        }
        __syncthreads();
        if (lane < ns) {
            chunk_fn[(u64)c * ns + lane] =
                wide_pack(chain_fold(kGroups, lane, [&](u32 g, u32 stt) { return sh.gtab[g * ns + stt]; }));
        }
        if (lane == 0) {
            chunk_regular[c] = kRegularFew;
            done_list[atomicAdd(done_count, 1u)] = c;
        }
        __syncthreads(); /* the tables in LDS are written again */
    }
}

template <u32 LB>
__global__ __launch_bounds__(HUFD_DEC_LANES, 8) void dec_sync_true_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    const u16 *fn_tab,
    u16 *cp_tab,
    u16 *lane_count,
    u8 *chunk_regular,
    const u32 *chunk_entry,
    const u32 *list, /* dec_sync_few's chunks */
    const u32 *list_count) {

    few_shared<LB> &sh = *reinterpret_cast<few_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 lane = threadIdx.x;
    const u32 table = lds_offset_of(sh.wlut);
    const row_walk rw(LB, tb.max_bits);
    const u32 n = *list_count;
    if (n == 0) {
        return;
    }
    few_table<LB>(sh, tb, lane);
    if (lane == 0) {
        sh.bad = 0;
    }
    __syncthreads();
    for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
        const u32 c = list[k];
        const u32 centry = chunk_entry[c];
        if (!(centry & 0x100u)) {
            /* the stream ended before this chunk: nobody emits it, and it must not look regular to anybody */
            if (lane == 0) {
                chunk_regular[c] = 0;
            }
            continue;
        }
        const hufd_chunk_rec rec = chunk_rec[c];
        u32 w[kFastRows];
        few_load_words<LB>(w, d_in + rec.src_off + (u64)lane * HUFD_DEC_SUB_BYTES);
        for (u32 s = 0; s < ns; ++s) {
            sh.ftab[s * HUFD_DEC_LANES + lane] = fn_tab[((u64)c * ns + s) * HUFD_DEC_LANES + lane];
        }
        __syncthreads();
        if (lane == 0) {
            u32 at = centry & 0xFFu;
            bool stops = at >= ns;
            for (u32 l = 0; l < HUFD_DEC_LANES && !stops; ++l) {
                sh.entry_of[l] = at;
                const u32 f = sh.ftab[at * HUFD_DEC_LANES + l];
                stops = (f & 0x8000u) != 0;
                at = (f >> 11) & 15u;
            }
            sh.bad = stops ? 1u : 0u;
        }
        __syncthreads();
        bool ok = sh.bad == 0;
        const u32 entry = ok ? sh.entry_of[lane] : 0u;
        const u32 next_entry = ok && lane + 1 < HUFD_DEC_LANES ? sh.entry_of[lane + 1] : HUFD_NONE32;
        __syncthreads();
        /* the true walk: count, exit, where it enters the quarters */
        u32 st = rw.state_at(entry, 0), dead_count = 0;
        u32 cp_state[kQuarters - 1] = {0, 0, 0};
        bool dd = false;
#pragma unroll
        for (u32 r = 0; r < kSubWords; ++r) {
            if (r && r % (kSubWords / kQuarters) == 0) {
                cp_state[r / (kSubWords / kQuarters) - 1] = st;
            }
            st = few_row<LB>(st, w[r], w[r + 1], table, rw, dd, dead_count);
        }
        const u32 count = st >> 16, ex = rw.offset_of(st);
        /* (what dec_sync_few said of this walk holds: anything else is a chunk for the long way) */
        if (ok && (dd || ex >= ns || (next_entry != HUFD_NONE32 && ex != next_entry))) {
            sh.bad = 1;
        }
        __syncthreads();
        const bool good = sh.bad == 0;
        __syncthreads();
        if (good) {
            u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES + lane;
#pragma unroll
            for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
                const u32 tail = count - (cp_state[qq] >> 16), bits = rw.offset_of(cp_state[qq]);
                cp[qq * HUFD_DEC_LANES] = (u16)(bits < 16u ? 0x8000u | (bits << 11) | tail : 0u);
            }
            lane_count[(u64)c * HUFD_DEC_LANES + lane] = (u16)count;
            cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)((1u << entry) | (ex << 12));
        }
        if (lane == 0) {
            chunk_regular[c] = good ? 1 : 0;
            sh.bad = 0;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kTailThreads) void dec_sync_tail_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u32 *tail_chunks,
    u32 n_tail,
    const u8 *d_in,
    const u8 *chunk_regular,
    const u32 *tail_entry,
    u16 *fn_tab,
    u16 *cp_tab,
    u32 *chunk_fn,
    u16 *lane_count) {

    tail_lds &sh = *reinterpret_cast<tail_lds *>(dyn_lds);
    u16 *lut = reinterpret_cast<u16 *>(dyn_lds + sizeof(tail_lds));
    for (u32 i = threadIdx.x; i < (1u << tb.lut_bits); i += kTailThreads) {
        lut[i] = tb.dec_lut[i];
    }
    __syncthreads();
    const u32 i = blockIdx.x * kTailThreads + threadIdx.x;
    if (i >= n_tail) {
        return;
    }
    const u32 c = tail_chunks[i];
    const u32 kind = chunk_regular[c];
    if (kind != 2 && kind != 3) {
        return; /* not taken by the regular chunks' kernel: the long way does all of it */
    }
    const u32 ns = tb.n_states;
    const hufd_dec_item it = items[chunk_item[c]];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len - chunk_off;
    if (kind == 3) {
        /* a chunk of fewer than 136 bytes (a short item, or the last bytes of a long one): its whole transfer function */
        u32 *tiny = sh.words[threadIdx.x];
        load_be32_run(tiny, d_in + it.in_off + chunk_off, valid, kTailWords);
        /* (an item's first chunk is only ever entered at the item's first bit) */
        const bool only = c == it.first_chunk;
        for (u32 st = only ? it.first_bit : 0u; st < (only ? it.first_bit + 1u : ns); ++st) {
            u32 stop_pos = 0, stop_why = 0;
            const tail_walk tw = tail_follow(tiny, lut, tb.lut_bits, st, (u32)(valid * 8), 2 * HUFD_DEC_SUB_BITS, nullptr, &stop_pos, &stop_why);
            chunk_fn[(u64)c * ns + st] = wide_pack(true, 0, tw.count[0] + tw.count[1]); /* the stream ends here whatever the entry */
        }
        return;
    }
    const u32 n_full = (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES); /* >= 1 and < HUFD_DEC_LANES here */
    const u8 *tsrc = d_in + it.in_off + chunk_off + (u64)n_full * HUFD_DEC_SUB_BYTES;
    const u64 tail_bytes = valid - (u64)n_full * HUFD_DEC_SUB_BYTES; /* 8 .. 135 */
    u32 *words = sh.words[threadIdx.x];
    load_be32_run(words, tsrc, tail_bytes, kTailWords);
    const u32 entry = tail_entry[c];
    const u32 limit = (n_full + 1 < HUFD_DEC_LANES ? 2u : 1u) * HUFD_DEC_SUB_BITS; /* the last lane's walk ends with the chunk */
    u32 stop_pos = 0, stop_why = 0;
    const tail_walk tw = tail_follow(words, lut, tb.lut_bits, entry, (u32)(tail_bytes * 8), limit, nullptr, &stop_pos, &stop_why);

    /* the records of the one or two lanes the true path gets to */
    u16 *fn_out = fn_tab + (u64)c * ns * HUFD_DEC_LANES;
    u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES;
    for (u32 k = 0; k < 2; ++k) {
        const u32 lane = n_full + k;
        const bool reached = k == 0 || tw.stop != 0;
        if (lane >= HUFD_DEC_LANES || !reached) {
            break;
        }
        const u32 my_entry = k == 0 ? entry : tw.exit;
        const bool stops_here = tw.stop == k;
        lane_count[(u64)c * HUFD_DEC_LANES + lane] = (u16)tw.count[k];
        fn_out[(u64)my_entry * HUFD_DEC_LANES + lane] =
            stops_here ? fn_pack(true, 0, tw.count[k] & 0x7FFu) : fn_pack(false, tw.exit, tw.count[k] & 0x7FFu);
        cp[(kQuarters - 1) * HUFD_DEC_LANES + lane] = (u16)((1u << my_entry) | ((stops_here ? kExitStop : tw.exit) << 12));
    }
    /* the chunk function: every walk that gets through sub-chunk 0 goes on to the end of the stream */
    const bool stops = tw.stop != 2u;
    for (u32 st = 0; st < ns; ++st) {
        const u32 f = chunk_fn[(u64)c * ns + st];
        if (!wide_stop(f)) {
            chunk_fn[(u64)c * ns + st] = wide_pack(stops, stops ? 0u : tw.exit, wide_count(f) + tw.count[0] + tw.count[1]);
        }
    }
}


static uint32_t dec_sync_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + kGroups * tb->n_states * 4 + (2u << tb->lut_bits);
}

} /* namespace */

using hufk_host::persistent_grid;
using hufk_host::stage_mark;
using hufk_host::current_compute_units;
using hufk_host::kBesideMinChunks;
using hufk_host::decode_launch_state;

hipError_t hufk_host::init_decode_sync(int lds_max) {
    hipError_t e = hipSuccess;
    const void *kernels[] = {
        reinterpret_cast<const void *>(&dec_sync_kernel<8>), reinterpret_cast<const void *>(&dec_sync_kernel<10>),
        reinterpret_cast<const void *>(&dec_sync_kernel<12>)};
    for (const void *k : kernels) {
        if (e == hipSuccess) {
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        }
    }
    return e;
}

void hufk_host::decode_sync_stage(const struct hufk_decode_args *a, hipStream_t st, decode_launch_state &s) {
    const uint32_t ns = a->tables.n_states, lb_of_launch = s.lb, sure = s.sure;
    bool few = false;
    /* chunks inside the stream the short way; the rest, and those that turn out irregular, through the list */
    const auto sync = ns <= 8 ? dec_sync_kernel<8> : (ns <= 10 ? dec_sync_kernel<10> : dec_sync_kernel<12>);
    u32 *const slow_count = a->counters + HUFK_DEC_COUNT_SLOW, *const few_count = a->counters + HUFK_DEC_COUNT_FEW;
    const bool some_inside = a->n_tail < a->n_chunks; /* chunks with a whole chunk + 8 bytes of stream left */
    /* two lists of chunks that are not regular by dec_sync_one's rules: the ones dec_sync_guess may still take
     * (inside a stream, first sub-chunk's walks meet) and the ones for the long way.  The second is the emit stage's
     * list, free until then; one list where there is no dec_sync_guess for the launch. */
    const bool guessing = some_inside && !a->quiet; /* (quiet: dec_sync_one's two lists are one, the long way's) */
    u32 *lean_long_list = guessing ? a->emit_list : a->slow_list;
    u32 *lean_long_count = guessing ? a->counters + HUFK_DEC_COUNT_LONG : slow_count;
    /* A few chunks that streams end in beside many inside streams (one long stream: ONE): their kernels are tiny
     * and, one after the other behind the big ones, cost a tenth of the decode time in launch and drain.  They run on
     * a second stream of the engine's, beside the big kernels, forked off and joined with events. */
    /* (not for a launch of a few chunks: the fork and the join are four commands, ~30 us of a small call) */
    /* Since round 6 the few go INTO the big kernel's grid, as its first workgroups, each with its stream's last symbols
     * behind it (dec_sync_one_mixed_kernel): no second stream, no fork, no join -- four commands of ~4 us less on the
     * launch's stream.  (`tails_apart`, the tests': the kernels of their own, as for a launch of many such chunks.) */
    const bool folded = hufk_host::tails_are_folded(a);
    const bool beside = !folded && some_inside && a->n_tail && a->side_stream && a->fork_event && a->join_event &&
                        (uint64_t)a->n_tail * 8 <= a->n_chunks && a->n_chunks >= kBesideMinChunks;
    hipStream_t tst = beside ? (hipStream_t)a->side_stream : st;
    if (beside) {
        (void)hipEventRecord((hipEvent_t)a->fork_event, st);
        (void)hipStreamWaitEvent(tst, (hipEvent_t)a->fork_event, 0);
    }
    /* the chunks streams end in: several to a workgroup where they are short and many (dec_sync_pack) */
    const uint32_t pack_width = a->tail_lanes + 2u < 16u ? 16u : a->tail_lanes + 2u; /* (+ the two sub-chunks a stream can end in: dec_emit_pack's scan) */
    const bool pack = a->one_chunk_a_workgroup == 0 && a->n_tail_narrow >= kPackMinChunks && pack_width <= HUFD_DEC_LANES / 2;
    const uint32_t pack_slots = HUFD_DEC_LANES / pack_width;
    /* (the plan lists the chunks with few whole lanes first: those go several to a workgroup, the others one each) */
    const uint32_t n_packed = pack ? a->n_tail_narrow : 0u, n_single = a->n_tail - n_packed;
    const u32 *single_chunks = a->tail_chunks + n_packed;
#define HUFK_LAUNCH_SYNC_LEAN(LBV, SUREV)                                                                               \
if (folded) {                                                                                                      \
    hipLaunchKernelGGL(                                                                                            \
        (dec_sync_one_mixed_kernel<LBV, SUREV>), dim3(a->n_chunks + a->n_tail), dim3(HUFD_DEC_LANES),              \
        (uint32_t)sizeof(one_shared<LBV>), st, a->tables, a->chunk_rec, a->tail_chunks, a->n_tail,                  \
        (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
        a->slow_list, slow_count, lean_long_list, lean_long_count);                                                \
} else if (n_packed) {                                                                                                    \
    hipLaunchKernelGGL(                                                                                            \
        (dec_sync_pack_kernel<LBV, SUREV>), dim3((n_packed + pack_slots - 1) / pack_slots), dim3(HUFD_DEC_LANES),   \
        (uint32_t)sizeof(pack_shared<LBV>), tst, a->tables, a->chunk_rec, a->tail_chunks, n_packed, pack_width,     \
        (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
        lean_long_list, lean_long_count);                                                                          \
}                                                                                                                  \
if (!folded && n_single) {                                                                                         \
    hipLaunchKernelGGL(                                                                                            \
        (dec_sync_one_kernel<LBV, SUREV, true>), dim3(n_single), dim3(HUFD_DEC_LANES),                              \
        (uint32_t)sizeof(one_shared<LBV>), tst, a->tables, a->chunk_rec, single_chunks,                             \
        (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
        a->slow_list, slow_count, lean_long_list, lean_long_count);                                                \
}                                                                                                                  \
if (!folded && some_inside) {                                                                                      \
    hipLaunchKernelGGL(                                                                                            \
        (dec_sync_one_kernel<LBV, SUREV, false>), dim3(a->n_chunks), dim3(HUFD_DEC_LANES),                          \
        (uint32_t)sizeof(one_shared<LBV>), st, a->tables, a->chunk_rec, a->tail_chunks,                             \
        (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
        a->slow_list, slow_count, lean_long_list, lean_long_count);                                                \
}
    if (lb_of_launch == 10) {
        switch (sure) {
            case 3: HUFK_LAUNCH_SYNC_LEAN(10, 3); break;
            default: HUFK_LAUNCH_SYNC_LEAN(10, 4); break;
        }
    } else {
        HUFK_LAUNCH_SYNC_LEAN(12, 2);
    }
#undef HUFK_LAUNCH_SYNC_LEAN
    if (a->n_tail && !folded) {
        /* the last symbols of every stream, a thread each; then the chunk functions are complete */
        const uint32_t lds = (uint32_t)sizeof(tail_lds) + (2u << a->tables.lut_bits);
        hipLaunchKernelGGL(
            dec_sync_tail_kernel, dim3((a->n_tail + kTailThreads - 1) / kTailThreads), dim3(kTailThreads), lds, tst,
            a->tables, a->items, a->chunk_item, a->tail_chunks, a->n_tail, (const u8 *)a->d_in,
            (const u8 *)a->chunk_regular, (const u32 *)a->tail_entry, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count);
    }
    if (beside) {
        (void)hipEventRecord((hipEvent_t)a->join_event, tst);
        (void)hipStreamWaitEvent(st, (hipEvent_t)a->join_event, 0);
    }
    /* the chunks inside streams that dec_sync_one gave up on: a second chance that asks less of the coder
     * (dec_sync_guess); what that gives up on goes on a second list (the emit stage's, free until then) */
    const u32 *long_list = a->slow_list, *long_count = slow_count;
    if (guessing) {
#define HUFK_LAUNCH_SYNC_GUESS(LBV, SUREV)                                                                              \
hipLaunchKernelGGL(                                                                                                \
    (dec_sync_guess_kernel<LBV, SUREV>),                                                                           \
    dim3(persistent_grid(dec_sync_guess_kernel<LBV, SUREV>, HUFD_DEC_LANES, (uint32_t)sizeof(lean_shared<LBV>),     \
                         a->n_chunks)),                                                                            \
    dim3(HUFD_DEC_LANES), (uint32_t)sizeof(lean_shared<LBV>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,     \
    a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, (const u32 *)a->slow_list,                 \
    (const u32 *)slow_count, a->emit_list, a->counters + HUFK_DEC_COUNT_LONG)
        if (lb_of_launch == 10) {
            switch (sure) {
                case 3: HUFK_LAUNCH_SYNC_GUESS(10, 3); break;
                default: HUFK_LAUNCH_SYNC_GUESS(10, 4); break;
            }
        } else {
            HUFK_LAUNCH_SYNC_GUESS(12, 2);
        }
#undef HUFK_LAUNCH_SYNC_GUESS
        long_list = a->emit_list;
        long_count = a->counters + HUFK_DEC_COUNT_LONG;
        /* of those, the chunks inside streams whose walks do not fall into step: a few walks a lane, not the long
         * way's every bit (dec_sync_few; its list -- dec_sync_one's, used up by now -- is for dec_sync_true below) */
        if (a->few_walks) {
            few = true; /* (its list: dec_sync_one's array, used up by now, with a counter of its own) */
            if (a->tables.lut_bits <= 10) {
                hipLaunchKernelGGL(
                    (dec_sync_few_kernel<10>),
                    dim3(persistent_grid(dec_sync_few_kernel<10>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<10>), a->n_chunks)),
                    dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<10>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                    a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count, a->slow_list, few_count);
            } else {
                hipLaunchKernelGGL(
                    (dec_sync_few_kernel<12>),
                    dim3(persistent_grid(dec_sync_few_kernel<12>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<12>), a->n_chunks)),
                    dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<12>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                    a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count, a->slow_list, few_count);
            }
        }
    }
    hipLaunchKernelGGL(
        sync, dim3(persistent_grid(sync, HUFD_DEC_LANES, dec_sync_lds_bytes(&a->tables), a->n_chunks)),
        dim3(HUFD_DEC_LANES), dec_sync_lds_bytes(&a->tables), st, a->tables, a->items, a->chunk_item, a->n_chunks,
        (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count,
        a->counters_self_cleared ? a->counters : (u32 *)nullptr);
    s.few = few;
}

void hufk_host::decode_sync_true_stage(const struct hufk_decode_args *a, hipStream_t st, const decode_launch_state &s) {
    if (!s.few) {
        return;
    }
    /* dec_sync_few's chunks, now that dec_scan has said where each is entered: the true walk's records (dec_sync_true) */
    if (a->tables.lut_bits <= 10) {
        hipLaunchKernelGGL(
            (dec_sync_true_kernel<10>),
            dim3(persistent_grid(dec_sync_true_kernel<10>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<10>), a->n_chunks)),
            dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<10>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
            (const u16 *)a->fn_tab, a->cp_tab, a->lane_count, a->chunk_regular, (const u32 *)a->chunk_entry,
            (const u32 *)a->slow_list, (const u32 *)(a->counters + HUFK_DEC_COUNT_FEW));
    } else {
        hipLaunchKernelGGL(
            (dec_sync_true_kernel<12>),
            dim3(persistent_grid(dec_sync_true_kernel<12>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<12>), a->n_chunks)),
            dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<12>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
            (const u16 *)a->fn_tab, a->cp_tab, a->lane_count, a->chunk_regular, (const u32 *)a->chunk_entry,
            (const u32 *)a->slow_list, (const u32 *)(a->counters + HUFK_DEC_COUNT_FEW));
    }
}
