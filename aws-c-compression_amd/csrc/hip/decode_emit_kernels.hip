/*
 * Decode, second pass: the symbols.  Every lane's true entry is known (dec_scan); the walk is the one of reference
 * source/huffman.c:230-281 with the symbols kept.
 *   dec_emit                       the long way (any chunk, on a list): entry states from the lane functions, stops found
 *   dec_emit_fast, dec_emit_big    regular chunks: two quarters a thread from checkpoints, symbols staged in LDS
 *   dec_emit_pack                  several short end-of-stream chunks a workgroup
 *   dec_emit_tail                  the last symbols of a stream, a thread each
 */
#include "decode_common.hpp"
#include "launch_common.hpp"

namespace {

/* ------------------------------------------------------------------ decode: emit */

/*
 * kEmitThreads threads per chunk: thread (lane, q) walks the part of lane's sub-chunk between
 * checkpoint q and the next usable one (q = 0: from the true entry state).  The waves of one
 * q run the same number of steps, a quarter of what one thread per sub-chunk would.
 */
__device__ __forceinline__ void dec_emit_chunk(
    const hufd_tables &tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u8 *d_in,
    u8 *d_out,
    const u16 *fn_tab,
    const u16 *cp_tab,
    const u16 *lane_count_tab, /* regular chunks: the symbol counts of lanes >= 1 are here, not in fn_tab */
    const u8 *chunk_regular,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    u32 c) {

    const u32 ns = tb.n_states;
    u32 *timg = reinterpret_cast<u32 *>(dyn_lds);
    u8 *stage = reinterpret_cast<u8 *>(timg + kChunkWords); /* [HUFD_DEC_STAGE_BYTES], 16-aligned */
    u16 *ftab = reinterpret_cast<u16 *>(stage);               /* aliases the stage until the walk starts */
    u32 *gtab = reinterpret_cast<u32 *>(stage + HUFD_DEC_STAGE_BYTES);  /* [groups][ns] */
    u32 *g_entry = gtab + kGroups * HUFD_DEC_MAX_STATES;      /* [groups] */
    u32 *g_base = g_entry + kGroups;                          /* [groups] */
    u32 *l_entry = g_base + kGroups;                          /* [lanes] */
    u32 *l_base = l_entry + HUFD_DEC_LANES;                   /* [lanes] */
    u32 *l_cnt = l_base + HUFD_DEC_LANES;                     /* [lanes] */
    u32 *blk_count = l_cnt + HUFD_DEC_LANES;                  /* [4] */
    u16 *cpt = reinterpret_cast<u16 *>(blk_count + 4);        /* [kCpRows][lanes] */
    u16 *lut = cpt + kCpRows * HUFD_DEC_LANES;

    const u32 t = threadIdx.x;
    const u32 lane = t % HUFD_DEC_LANES, q = t / HUFD_DEC_LANES;
    const u32 entry = chunk_entry[c];
    if (!(entry & 0x100u)) {
        return; /* the stream ended before this chunk */
    }
    const u32 item_index = chunk_item[c];
    const hufd_dec_item it = items[item_index];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len > chunk_off ? it.in_len - chunk_off : 0;
    const u64 cbase = chunk_base[c];

    HUFD_STAMP(1, 0);
    /* the two small tables first, so that their latency hides behind the chunk itself */
    const u16 my_cp = cp_tab[(u64)c * kCpRows * HUFD_DEC_LANES + t]; /* kCpRows * lanes == threads */
    u16 my_fn[(HUFD_DEC_MAX_STATES * HUFD_DEC_LANES + kEmitThreads - 1) / kEmitThreads];
#pragma unroll
    for (u32 j = 0; j < sizeof(my_fn) / sizeof(my_fn[0]); ++j) {
        const u32 i = t + j * kEmitThreads;
        my_fn[j] = i < ns * HUFD_DEC_LANES ? fn_tab[(u64)c * ns * HUFD_DEC_LANES + i] : (u16)0;
    }
    chunk_load<kEmitThreads>(timg, d_in + it.in_off + chunk_off, valid);
    lut_load<kEmitThreads>(lut, tb);
    cpt[t] = my_cp;
#pragma unroll
    for (u32 j = 0; j < sizeof(my_fn) / sizeof(my_fn[0]); ++j) {
        const u32 i = t + j * kEmitThreads;
        if (i < ns * HUFD_DEC_LANES) {
            ftab[i] = my_fn[j];
        }
    }
    __syncthreads();
    HUFD_STAMP(1, 1);

    /*
     * True entry state and output offset of every lane.  Nearly always every lane's true entry
     * state merges into the lane's reference walk, and then the lane's exit does not depend on
     * its entry: lane i enters in the state lane i-1's reference walk leaves in.  Assume that
     * for all lanes at once, check it for all lanes at once, and only walk the chain lane by
     * lane (the general case below) when some lane does not fit.
     */
    const u32 merged_row = (kQuarters - 1) * HUFD_DEC_LANES;
    bool lane_reached = false;
    u32 lane_state = 0, lane_count = 0, lane_stop = 0, lane_incl = 0;
    if (t < HUFD_DEC_LANES) {
        /* a stop shows as kExitStop in the lane after it */
        lane_state = t ? (u32)(cpt[merged_row + t - 1] >> 12) : (entry & 0xFFu);
        l_entry[t] = lane_state == kExitStop ? 1u : 0u; /* borrowed: flags for the search below */
    }
    if (t == 0) {
        blk_count[1] = 0; /* set by any lane that does not fit */
    }
    __syncthreads();
    if (t < HUFD_DEC_LANES) {
        /* first lane that follows a stop: lanes from there on are not reached */
        u32 first_unreached = HUFD_DEC_LANES;
        for (u32 w = HUFD_DEC_LANES / kWave; w-- > 0;) {
            const u64 b = __ballot(l_entry[w * kWave + (t & (kWave - 1))] != 0);
            first_unreached = b ? w * kWave + (u32)__builtin_ctzll(b) : first_unreached;
        }
        lane_reached = t < first_unreached;
        const u32 mine_row = cpt[merged_row + t];
        lane_stop = (mine_row >> 12) == kExitStop ? 1u : 0u;
        const bool fits = !lane_reached || (lane_state < kExitNoRef && ((mine_row >> lane_state) & 1u));
        if (!fits) {
            blk_count[1] = 1;
        }
        lane_state = (fits && lane_reached) ? lane_state : 0;
        lane_count = !lane_reached ? 0u
                     : (t != 0 && chunk_regular[c] != 0) ? (u32)lane_count_tab[(u64)c * HUFD_DEC_LANES + t]
                                                         : (u32)(ftab[lane_state * HUFD_DEC_LANES + t] & 0x7FFu);
        lane_incl = wave_inclusive_sum(lane_count, t & (kWave - 1));
        if ((t & (kWave - 1)) == kWave - 1) {
            g_base[t / kWave] = lane_incl;
        }
    }
    __syncthreads();
    const bool all_fit = blk_count[1] == 0;
    if (all_fit) {
        if (t < HUFD_DEC_LANES) {
            u32 before = 0, total = 0;
#pragma unroll
            for (u32 w = 0; w < HUFD_DEC_LANES / kWave; ++w) {
                const u32 sum = g_base[w];
                before += w < t / kWave ? sum : 0;
                total += sum;
            }
            l_base[t] = before + lane_incl - lane_count;
            l_cnt[t] = lane_count;
            l_entry[t] = entry_pack(lane_reached ? lane_state : 0u, lane_reached) | ((lane_reached && lane_stop) ? 0x200u : 0u);
            if (t == 0) {
                blk_count[0] = total;
            }
        }
    } else {
        /* the general case: fold 16 lanes per group, walk the groups, then the lanes of each group */
        if (t < kGroups * ns) {
            const u32 g = t / ns, start = t % ns;
            gtab[g * ns + start] = wide_pack(chain_fold(kGroupLanes, start, [&](u32 i, u32 stt) {
                return widen(ftab[stt * HUFD_DEC_LANES + g * kGroupLanes + i]);
            }));
        }
        __syncthreads();
        if (t == 0) {
            u32 state = entry & 0xFFu, total = 0;
            bool stopped = false;
#pragma unroll 1
            for (u32 g = 0; g < kGroups; ++g) {
                g_entry[g] = entry_pack(state, !stopped);
                g_base[g] = total;
                if (!stopped) {
                    const u32 f = gtab[g * ns + state];
                    total += wide_count(f);
                    stopped = wide_stop(f);
                    state = wide_state(f);
                }
            }
            blk_count[0] = total;
        }
        __syncthreads();
        if (t < kGroups) {
            u32 state = g_entry[t] & 0xFFu, total = g_base[t];
            bool stopped = !(g_entry[t] & 0x100u);
#pragma unroll 1
            for (u32 i = 0; i < kGroupLanes; ++i) {
                const u32 l = t * kGroupLanes + i;
                u32 ent = entry_pack(state, !stopped), cnt = 0;
                l_base[l] = total;
                if (!stopped) {
                    const u32 f = widen(ftab[state * HUFD_DEC_LANES + l]);
                    cnt = wide_count(f);
                    total += cnt;
                    stopped = wide_stop(f);
                    state = wide_state(f);
                    ent |= stopped ? 0x200u : 0u; /* the true path ends inside this sub-chunk */
                }
                l_entry[l] = ent;
                l_cnt[l] = cnt;
            }
        }
    }
    __syncthreads(); /* ftab is dead from here on: the stage may be written */

    const u32 chunk_symbols = blk_count[0];
    u8 *out_ptr = d_out + it.out_off + cbase;
    const u32 mis = (u32)((uintptr_t)out_ptr & 15u);
    const bool staged = chunk_symbols + 16 <= HUFD_DEC_STAGE_BYTES;
    /* symbols of this chunk that fit the item's capacity */
    const u64 room = it.out_cap > cbase ? it.out_cap - cbase : 0;
    const u32 writable = room < chunk_symbols ? (u32)room : chunk_symbols;

    HUFD_STAMP(1, 2);
    /*
     * The walk proper.  dec_sync already counted the symbols of the true path that start in
     * this sub-chunk, and every one of them is a valid, complete code, so the loop runs a
     * fixed count with no per-symbol stop test: window -> table -> symbol byte -> shift.
     * This thread's share: from its checkpoint (q = 0: the lane's entry state) to the next
     * usable checkpoint.  Checkpoints lie on the reference walk, so they only apply when the
     * lane's true entry state merged into it.
     */
    const u32 my_entry = l_entry[lane];
    const bool reached = (my_entry & 0x100u) != 0;
    const u32 lane_n = reached ? l_cnt[lane] : 0;
    const bool on_ref = ((cpt[(kQuarters - 1) * HUFD_DEC_LANES + lane] >> (my_entry & 0xFFu)) & 1u) != 0;
    u32 first = 0, pos = my_entry & 0xFFu; /* index of my first symbol within the lane, and its bit */
    bool mine = reached;
    if (q > 0) {
        const u32 cp = cpt[(q - 1) * HUFD_DEC_LANES + lane];
        mine = reached && on_ref && (cp & 0x8000u) != 0;
        first = lane_n - (cp & 0x7FFu);
        pos = q * kQuarterBits + ((cp >> 11) & 15u);
    }
    u32 beyond = lane_n; /* index of the first symbol that is no longer mine */
    bool last_part = true;
#pragma unroll
    for (u32 k = kQuarters - 1; k >= 1; --k) {
        const u32 cp = cpt[(k - 1) * HUFD_DEC_LANES + lane];
        if (k > q && on_ref && (cp & 0x8000u)) {
            beyond = lane_n - (cp & 0x7FFu);
            last_part = false;
        }
    }
    const u32 n = mine ? beyond - first : 0;
    const u32 base = l_base[lane] + first;
    const u32 n_store = base >= writable ? 0 : (writable - base < n ? writable - base : n);
    lane_window br;
    br.start(timg, lane, pos);
    const u32 shift = 32 - tb.lut_bits;
    if (staged) {
        u8 *dst = stage + mis + base;
        for (u32 k = 0; k < n_store; ++k) {
            const u32 e = lut[br.peek() >> shift];
            dst[k] = (u8)(e >> 8);
            pos += e & 0xFFu;
            br.skip(timg, lane, e & 0xFFu);
        }
    } else {
        u8 *dst = out_ptr + base; /* more symbols than the stage holds: straight to memory */
        for (u32 k = 0; k < n_store; ++k) {
            const u32 e = lut[br.peek() >> shift];
            dst[k] = (u8)(e >> 8);
            pos += e & 0xFFu;
            br.skip(timg, lane, e & 0xFFu);
        }
    }
    if (mine) {
        const u64 sub_bit = (chunk_off + (u64)lane * HUFD_DEC_SUB_BYTES) * 8; /* stream bit of the sub-chunk start */
        if (n_store < n) {
            if (cbase + base + n_store == it.out_cap) {
                results[item_index].cap_bit = sub_bit + pos; /* source/huffman.c:257-268 fires on this symbol */
            }
        } else if (last_part && (my_entry & 0x200u)) {
            u32 sym = 0, why = HUFD_STOP_NONE;
            (void)code_at(br.peek(), lut, tb.lut_bits, pos, clamp_remaining(valid, lane), &sym, &why);
            results[item_index].stop_kind = why;
            results[item_index].stop_bit = sub_bit + pos;
        }
    }
    HUFD_STAMP(1, 3);
    __syncthreads();
    HUFD_STAMP(1, 4);

    if (staged && writable) {
        /* stage byte b belongs at (out_ptr - mis) + b: whole 16-byte rows go out aligned */
        u8 *gbase = out_ptr - mis;
        const u32 lo = mis, hi = mis + writable;
        const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
        if (row_lo <= row_hi) {
            for (u32 b = lo + t; b < row_lo * 16; b += kEmitThreads) {
                gbase[b] = stage[b];
            }
            for (u32 r = row_lo + t; r < row_hi; r += kEmitThreads) {
                *reinterpret_cast<uint4 *>(gbase + (u64)r * 16) = *reinterpret_cast<const uint4 *>(stage + r * 16);
            }
            for (u32 b = row_hi * 16 + t; b < hi; b += kEmitThreads) {
                gbase[b] = stage[b];
            }
        } else {
            for (u32 b = lo + t; b < hi; b += kEmitThreads) {
                gbase[b] = stage[b];
            }
        }
    }
    HUFD_STAMP(1, 5);
}

/* the chunks list[0 .. *list_count), a few workgroups taking turns (list == NULL: every chunk) */
__global__ __launch_bounds__(kEmitThreads, 4) void dec_emit_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    u32 n_chunks,
    const u8 *d_in,
    u8 *d_out,
    const u16 *fn_tab,
    const u16 *cp_tab,
    const u16 *lane_count_tab,
    const u8 *chunk_regular,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    const u32 *list,
    const u32 *list_count,
    u32 *counters = nullptr /* the launch's list counters ... */,
    u32 *summary = nullptr /* ... and where this kernel, the launch's last, leaves them for the host (NULL: nowhere) */,
    u32 self_cleared = 0 /* 1: ... and clears the SYNC stage's words for the launch behind this one (nobody reads them any more;
                          * the emit stage's, which this kernel's workgroups read, are cleared by that launch's sync stage) */) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (summary) {
            for (u32 k = 0; k < HUFK_DEC_COUNTERS; ++k) {
                summary[k] = counters[k];
            }
        }
        if (self_cleared) {
            counters[HUFK_DEC_COUNT_SLOW] = 0;
            counters[HUFK_DEC_COUNT_LONG] = 0;
            counters[HUFK_DEC_COUNT_FEW] = 0;
        }
    }
    const u32 n = list ? *list_count : n_chunks;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        dec_emit_chunk(tb, items, chunk_item, d_in, d_out, fn_tab, cp_tab, lane_count_tab, chunk_regular, chunk_entry, chunk_base, results, list ? list[i] : i);
        __syncthreads(); /* image and stage are reused by the next chunk */
    }
}

/* ------------------------------------------------------------------ decode: emit, regular chunks */

/*
 * dec_emit for the chunks the sync kernels found regular, when the whole chunk fits the output and
 * the LDS stage; every other chunk is put on a list for dec_emit_kernel.  Four threads per
 * sub-chunk as there (thread (lane, q) starts at checkpoint q), but each holds its quarter of the
 * sub-chunk in registers (nine words of the lane's own 128-byte line) and walks it row by row like
 * dec_sync_one: shift, mask, table, byte store, two adds a symbol.  The table entry is
 * symbol << 16 | (0x10000 - length) & 0xFFFF; only the low half of the walk state is ever looked
 * at, so the symbol may ride along in the add.
 */
constexpr u32 kEmitChains = 2; /* quarters of sub-chunks a thread walks side by side: two independent chains per lane hide the table latency */
constexpr u32 kEmitFastThreads = kEmitThreads / kEmitChains;
constexpr u32 kEmitHalf = HUFD_DEC_LANES / kEmitChains;

template <u32 LB>
struct emit_shared {
    u32 wlut[1u << LB];
    u32 lane_base[HUFD_DEC_LANES]; /* index of the sub-chunk's first symbol within the chunk */
    u32 wave_tot[HUFD_DEC_LANES / 64];
    u32 pad[4];
    u8 dump[512]; /* where a chain that has nothing to emit writes (at most 8 rows x 32 codes) */
    u32 tail_words[2][kTailWords]; /* TAIL: the stream's last words, for the one or two careful lanes */
    /* last: a launch for chunks that cannot hold that many symbols asks for less of it (emit_lds_bytes) */
    u8 stage[HUFD_DEC_STAGE_BYTES + 32];
};

/* LDS of dec_emit_fast with room for `stage_bytes` symbols in the stage */
template <u32 LB>
__host__ __device__ constexpr u32 emit_lds_bytes(u32 stage_bytes) {
    return (u32)sizeof(emit_shared<LB>) - HUFD_DEC_STAGE_BYTES + stage_bytes;
}


template <u32 LB, bool TAIL, u32 SURE = 0> /* TAIL: a chunk that may hold the end of a stream; else one inside a stream.
                                             * SURE: the codes that are certain to start in a row, when the launch knows (0: asked of the coder at run time) */
__device__ __forceinline__ void dec_emit_fast_chunk(
    const u32 c,
    const hufd_tables &tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    u8 *d_out,
    const u16 *cp_tab,
    const u16 *lane_count,
    const u8 *chunk_regular,
    const u32 *chunk_fn,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    u32 *slow_list, /* chunks left to dec_emit_kernel */
    u32 *slow_count,
    u32 *dense_list, /* chunks with more symbols than the stage holds: left to dec_emit_big_kernel (or, the launch says, to the long way) */
    u32 *dense_count,
    u32 stage_limit /* symbols the stage of this launch holds (HUFD_DEC_STAGE_BYTES, or less: emit_lds_bytes) */) {

    emit_shared<LB> &sh = *reinterpret_cast<emit_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 t = threadIdx.x;
    /* my quarter of two sub-chunks.  In a chunk that holds the end of a stream only the first lanes have data: there a
     * thread takes two NEIGHBOURING sub-chunks and the threads are numbered sub-chunks first, so that the idle ones fill
     * whole waves, which then skip the walk (the kernel is bound by instruction issue: an idle wave's slots go to the
     * other workgroups of the CU) */
    const u32 q = TAIL ? t % kQuarters : t / kEmitHalf;
    const u32 lanes[kEmitChains] = {TAIL ? 2 * (t / kQuarters) : t % kEmitHalf,
                                    TAIL ? 2 * (t / kQuarters) + 1 : t % kEmitHalf + kEmitHalf};
    /* the table entries this thread will put into LDS: asked for first, they depend on nothing */
    /* (TAIL with a 10-bit table: the workgroup may be launched with fewer threads than 512 -- 256 at least --, see the launch) */
    constexpr bool kNarrowBlock = TAIL && LB == 10;
    constexpr u32 kLutPerThread = kNarrowBlock ? (1u << LB) / HUFD_DEC_LANES : ((1u << LB) + kEmitFastThreads - 1) / kEmitFastThreads;
    const u32 lut_stride = kNarrowBlock ? blockDim.x : kEmitFastThreads;
    u32 lut_raw[kLutPerThread];
#pragma unroll
    for (u32 j = 0; j < kLutPerThread; ++j) {
        const u32 i = t + j * lut_stride;
        lut_raw[j] = i < (1u << LB) ? tb.dec_lut[i >> (LB - tb.lut_bits)] : 0u;
    }
    const u32 centry = chunk_entry[c];
    if (!(centry & 0x100u)) {
        return; /* the stream ended before this chunk */
    }
    const u32 s0 = centry & 0xFFu;
    const hufd_chunk_rec rec = chunk_rec[c]; /* (asked for with the chunk's entry: not chunk -> item -> its record) */
    const u64 valid = rec.valid;
    /* lanes whose sub-chunk and the 8 bytes after it lie inside the stream (dec_sync_one: the others are idle or "careful") */
    if (!TAIL && valid < (u64)HUFD_DEC_CHUNK_BYTES + 8u) {
        return; /* the other instantiation's */
    }
    const u32 n_full = !TAIL ? HUFD_DEC_LANES : (valid >= 8u ? (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES) : 0u);
    const u64 cbase = chunk_base[c];
    const u16 *cpt = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES;
    const u32 merged_row = (kQuarters - 1) * HUFD_DEC_LANES;
    const u32 f0 = chunk_fn[(u64)c * ns + s0];
    const u32 chunk_symbols = wide_count(f0);
    /* all the same for the whole workgroup */
    const u32 regular = chunk_regular[c]; /* 1: all lanes whole; 2: the chunk that holds the end of the stream */
    if (TAIL && regular == 3) {
        return; /* fewer than 136 bytes: dec_emit_tail does the whole chunk */
    }
    const bool fits = regular != 0 && (regular == 2 || !wide_stop(f0)) && ((cpt[merged_row] >> s0) & 1u) != 0 &&
                      cbase + chunk_symbols <= rec.out_cap;
    const bool fast = fits && chunk_symbols + 16 <= stage_limit &&
                      (lds_offset_of(sh.wlut) & ((4u << LB) - 1u)) == 0 && SURE <= row_walk(LB, tb.max_bits).sure;
    if (!fast) {
        if (t == 0) {
            if (fits && chunk_symbols + 32 <= 2 * HUFD_DEC_STAGE_BYTES) {
                dense_list[atomicAdd(dense_count, 1u)] = c; /* short codes: a stage twice as long */
            } else {
                slow_list[atomicAdd(slow_count, 1u)] = c;
            }
        }
        return;
    }

    HUFD_STAMP(1, 0);
    /* my quarters: rows 8q .. 8q+7 and the word after them */
    constexpr u32 kRows = kSubWords / kQuarters;
    const u8 *sub[kEmitChains];
    u32 w[kEmitChains][kRows + 1];
    u32 my_cp[kEmitChains], next_cp[kEmitChains], entry_state[kEmitChains], cnt[kEmitChains];
    bool whole[kEmitChains];
#pragma unroll
    for (u32 ch = 0; ch < kEmitChains; ++ch) {
        sub[ch] = d_in + rec.src_off + (u64)lanes[ch] * HUFD_DEC_SUB_BYTES;
        whole[ch] = !TAIL || lanes[ch] < n_full;
#pragma unroll
        for (u32 j = 0; j <= kRows; ++j) {
            w[ch][j] = 0;
        }
        if (whole[ch]) {
            const unaligned_uint4 *p = reinterpret_cast<const unaligned_uint4 *>(sub[ch] + q * kRows * 4);
#pragma unroll
            for (u32 j = 0; j < kRows / 4; ++j) {
                const unaligned_uint4 v = p[j];
                w[ch][4 * j + 0] = __builtin_bswap32(v.x);
                w[ch][4 * j + 1] = __builtin_bswap32(v.y);
                w[ch][4 * j + 2] = __builtin_bswap32(v.z);
                w[ch][4 * j + 3] = __builtin_bswap32(v.w);
            }
            w[ch][kRows] = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(sub[ch] + (q + 1) * kRows * 4)->x);
        }
        my_cp[ch] = q ? cpt[(q - 1) * HUFD_DEC_LANES + lanes[ch]] : 0u;
        next_cp[ch] = q + 1 < kQuarters ? cpt[q * HUFD_DEC_LANES + lanes[ch]] : 0u;
        entry_state[ch] = lanes[ch] ? (u32)(cpt[merged_row + lanes[ch] - 1] >> 12) : s0;
        cnt[ch] = lane_count[(u64)c * HUFD_DEC_LANES + lanes[ch]];
    }
#pragma unroll
    for (u32 j = 0; j < kLutPerThread; ++j) {
        const u32 i = t + j * lut_stride;
        const u32 e = lut_raw[j];
        const u32 len = e & 0xFFu;
        if (i < (1u << LB)) {
            sh.wlut[i] = ((e >> 8) << 16) | ((0x10000u - (len ? len : kWalkDeadLen)) & 0xFFFFu);
        }
    }
    /* TAIL: the waves whose threads all stand behind the stream's whole lanes have done their share of the table and
     * leave; their slots (and, with a stage sized for what such chunks can hold, the LDS) let more workgroups onto the CU --
     * a workgroup's time is the latency of its walks, so that is what the rate follows.  Threads 0 .. 255 stay for the
     * scan below. */
    const u32 live_t = !TAIL ? kEmitFastThreads
                             : (4 * ((n_full + 1) / 2) + kWave - 1) / kWave * kWave < HUFD_DEC_LANES
                                   ? HUFD_DEC_LANES
                                   : (4 * ((n_full + 1) / 2) + kWave - 1) / kWave * kWave;
    if (TAIL && t >= live_t) {
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): the table entries are in LDS */
        return;
    }
    /* where every sub-chunk's symbols go: lane 0's count follows from the chunk's total */
    u32 incl[kEmitChains] = {0, 0};
    const u32 wl = t & (kWave - 1), half_wave = (t / kWave) & (kEmitHalf / kWave - 1);
    u32 own_cnt = 0; /* TAIL: thread t < 256 does this for sub-chunk t, whoever walks it */
    if (TAIL) {
        if (t < HUFD_DEC_LANES) {
            /* (the lanes behind the stream's last two sub-chunks hold nothing; dec_sync_pack does not even write their records) */
            own_cnt = t < n_full + 2 ? lane_count[(u64)c * HUFD_DEC_LANES + t] : 0u;
            incl[0] = wave_inclusive_sum(t ? own_cnt : 0u, wl);
            if (wl == kWave - 1) {
                sh.wave_tot[t / kWave] = incl[0];
            }
        }
    } else if (q == 0) {
#pragma unroll
        for (u32 ch = 0; ch < kEmitChains; ++ch) {
            incl[ch] = wave_inclusive_sum(lanes[ch] ? cnt[ch] : 0u, wl);
            if (wl == kWave - 1) {
                sh.wave_tot[ch * (kEmitHalf / kWave) + half_wave] = incl[ch]; /* = lanes[ch] / 64 */
            }
        }
    }
    __syncthreads();
    u32 rest = 0;
#pragma unroll
    for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
        rest += sh.wave_tot[wv];
    }
    const u32 first_count = chunk_symbols - rest; /* sub-chunk 0, entered in state s0 */
    if (TAIL) {
        if (t < HUFD_DEC_LANES) {
            u32 before = 0;
#pragma unroll
            for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
                before += wv < t / kWave ? sh.wave_tot[wv] : 0u;
            }
            sh.lane_base[t] = t ? first_count + before + incl[0] - own_cnt : 0u;
        }
    } else if (q == 0) {
#pragma unroll
        for (u32 ch = 0; ch < kEmitChains; ++ch) {
            u32 before = 0;
#pragma unroll
            for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
                before += wv < lanes[ch] / kWave ? sh.wave_tot[wv] : 0u;
            }
            sh.lane_base[lanes[ch]] = lanes[ch] ? first_count + before + incl[ch] - cnt[ch] : 0u;
        }
    }
    __syncthreads();
    HUFD_STAMP(1, 1);

    u8 *out_ptr = d_out + rec.out_off + cbase;
    const u32 mis = (u32)((uintptr_t)out_ptr & 15u);
    const row_walk rw(LB, tb.max_bits);
    u32 st[kEmitChains];
    /* where a chain's next symbol goes, as a byte offset into the workgroup's LDS record: a pointer here turns
     * the stores into flat ones with 64-bit address arithmetic */
    u8 *const lds_bytes = reinterpret_cast<u8 *>(&sh);
    const u32 stage_at = (u32)(reinterpret_cast<u8 *>(sh.stage) - lds_bytes), dump_at = (u32)(sh.dump - lds_bytes);
    const u32 stage_base = lds_offset_of(lds_bytes); /* (dst[] counts from here) */
    u32 dst[kEmitChains];
    bool idle[kEmitChains]; /* a chain with nothing to emit still walks (the two go in step): over zeros, into the dump */
    bool extend = false;
#pragma unroll
    for (u32 ch = 0; ch < kEmitChains; ++ch) {
        const u32 lane_n = lanes[ch] ? cnt[ch] : first_count;
        /* my share: from my checkpoint (q = 0: the entry state) to the next usable one; the lanes behind the whole
         * ones are not walked here */
        const bool mine = whole[ch] && (q == 0 || (my_cp[ch] & 0x8000u) != 0);
        const u32 first = q ? lane_n - (my_cp[ch] & 0x7FFu) : 0u;
        st[ch] = rw.state_at(q ? (my_cp[ch] >> 11) & 15u : entry_state[ch], 0);
        dst[ch] = mine ? stage_at + mis + sh.lane_base[lanes[ch]] + first : dump_at;
        idle[ch] = !mine;
        /* only sub-chunk 0's first checkpoint can be missing (its head is not known when the sync kernel runs) */
        if (ch == 0) {
            extend = q == 0 && lanes[0] == 0 && !(next_cp[ch] & 0x8000u);
        }
    }
    HUFD_STAMP(1, 2);
    const u8 *lut = reinterpret_cast<const u8 *>(sh.wlut);
    /* (the table sits at a multiple of its size: an entry's address is (window & mask) | table, one instruction) */
    const u32 table = lds_offset_of(sh.wlut);
    const u32 sure = SURE ? SURE : rw.sure;
    const bool wave_idle = TAIL && __all(idle[0] && idle[1]);
    if (!wave_idle) {
#pragma unroll
    for (u32 r = 0; r < kRows; ++r) {
        u64 pair[kEmitChains];
#pragma unroll
        for (u32 ch = 0; ch < kEmitChains; ++ch) {
            pair[ch] = ((u64)w[ch][r] << 32) | w[ch][r + 1];
        }
#pragma unroll
        for (u32 i = 0; i < sure; ++i) { /* the codes that are certain to start in this row, the two chains in turn */
            u32 e[kEmitChains];
#pragma unroll
            for (u32 ch = 0; ch < kEmitChains; ++ch) {
                e[ch] = lds_word_at(((u32)(pair[ch] >> (st[ch] & 63u)) & rw.mask) | table);
            }
#pragma unroll
            for (u32 ch = 0; ch < kEmitChains; ++ch) {
                lds_bytes[dst[ch]++] = (u8)(e[ch] >> 16);
                st[ch] += e[ch];
            }
        }
#pragma unroll
        for (u32 ch = 0; ch < kEmitChains; ++ch) {
            {
                u32 at = dst[ch] + stage_base; /* (as an LDS address, for the hand-written loop) */
                emit_uncertain_codes(st[ch], at, pair[ch], table, rw);
                dst[ch] = at - stage_base;
            }
            st[ch] += 32u;
            /* an idle chain starts every row afresh: whatever it decodes, its state and its writes stay in bounds */
            st[ch] = idle[ch] ? rw.state_at(0, 0) : st[ch];
            dst[ch] = idle[ch] ? dump_at : dst[ch];
        }
    }
    }
    if (extend) {
        /* rare: sub-chunk 0 on through the second quarter -- and the third and fourth when their checkpoints are missing too
         * (dec_sync_one: the walks of sub-chunk 0 met late) --, words straight from memory */
        const u32 extend_end = (cpt[1 * HUFD_DEC_LANES] & 0x8000u) ? 2 * kRows : ((cpt[2 * HUFD_DEC_LANES] & 0x8000u) ? 3 * kRows : 4 * kRows);
        u32 hi = w[0][kRows];
        for (u32 r = kRows; r < extend_end; ++r) {
            /* (sub-chunk 0: the address is rebuilt from the chunk's, so that no pointer has to stay in registers for this) */
            const u32 lo = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(d_in + rec.src_off + (r + 1) * 4)->x);
            const u64 pair = ((u64)hi << 32) | lo;
            while ((st[0] & 0xFFFFu) > rw.thr) {
                const u32 e = *reinterpret_cast<const u32 *>(lut + ((u32)(pair >> (st[0] & 63u)) & rw.mask));
                lds_bytes[dst[0]++] = (u8)(e >> 16);
                st[0] += e;
            }
            st[0] += 32u;
            hi = lo;
        }
    }
    /* (TAIL: the symbols of the one or two sub-chunks behind the whole lanes are dec_emit_tail's, straight to memory) */
    HUFD_STAMP(1, 3);
    __syncthreads();
    HUFD_STAMP(1, 4);

    {
        /* stage byte b belongs at (out_ptr - mis) + b: whole 16-byte rows go out aligned */
        u8 *gbase = out_ptr - mis;
        const u32 lo = mis, hi = mis + (TAIL && n_full < HUFD_DEC_LANES ? sh.lane_base[n_full] : chunk_symbols);
        const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
        if (row_lo <= row_hi) {
            /* (fewer than 16 bytes in front of the first whole row and behind the last: one byte a thread at most) */
            if (lo + t < row_lo * 16) {
                gbase[lo + t] = sh.stage[lo + t];
            }
#pragma unroll 2
            for (u32 r = row_lo + t; r < row_hi; r += live_t) {
                *reinterpret_cast<uint4 *>(gbase + (u64)r * 16) = *reinterpret_cast<const uint4 *>(sh.stage + r * 16);
            }
            if (row_hi * 16 + t < hi) {
                gbase[row_hi * 16 + t] = sh.stage[row_hi * 16 + t];
            }
        } else if (lo + t < hi) {
            gbase[lo + t] = sh.stage[lo + t]; /* no whole row: fewer than 31 bytes */
        }
    }
    HUFD_STAMP(1, 5);
}

template <u32 LB, bool TAIL, u32 SURE = 0> /* TAIL: the chunks listed in tail_chunks (they may hold the end of a stream); else all the others */
__global__ __launch_bounds__(kEmitFastThreads, TAIL ? 6 : 8) void dec_emit_fast_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u32 *tail_chunks,
    const u8 *d_in,
    u8 *d_out,
    const u16 *cp_tab,
    const u16 *lane_count,
    const u8 *chunk_regular,
    const u32 *chunk_fn,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    u32 *slow_list, /* chunks left to dec_emit_kernel */
    u32 *slow_count,
    u32 *dense_list, /* chunks with more symbols than the stage holds: left to dec_emit_big_kernel */
    u32 *dense_count,
    u32 stage_limit /* symbols the stage of this launch holds (HUFD_DEC_STAGE_BYTES, or less: emit_lds_bytes) */) {
    dec_emit_fast_chunk<LB, TAIL, SURE>(
        TAIL ? tail_chunks[blockIdx.x] : blockIdx.x, tb, chunk_rec, d_in, d_out, cp_tab, lane_count, chunk_regular, chunk_fn,
        chunk_entry, chunk_base, results, slow_list, slow_count, dense_list, dense_count, stage_limit);
}

/*
 * The chunks dec_emit_fast left because they hold more symbols than its stage (short codes: up to 2 x the stage's worth):
 * the same walk in ONE pass with a stage twice as long -- two workgroups per CU instead of four -- by resident workgroups
 * that take turns over the list.  (Two passes over the small stage, dec_emit_dense, cost 4.7 times dec_emit_fast's time
 * per symbol: 660 us for 256 MiB of 5.5-bit symbols.)
 */
constexpr u32 kEmitBigStage = 2 * HUFD_DEC_STAGE_BYTES;

template <u32 LB, u32 SURE>
__global__ __launch_bounds__(kEmitFastThreads, 4) void dec_emit_big_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u8 *d_in,
    u8 *d_out,
    const u16 *cp_tab,
    const u16 *lane_count,
    const u8 *chunk_regular,
    const u32 *chunk_fn,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    u32 *slow_list,
    u32 *slow_count,
    const u32 *big_list,
    const u32 *big_count) {
    const u32 n = *big_count;
    for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
        if (chunk_rec[big_list[k]].valid < HUFD_DEC_CHUNK_BYTES + 8u) {
            continue; /* (holds the end of its stream: never listed for this kernel, the long way takes those) */
        }
        /* (a chunk that does not go through here after all is left to the long way, not listed for this kernel again) */
        dec_emit_fast_chunk<LB, false, SURE>(
            big_list[k], tb, chunk_rec, d_in, d_out, cp_tab, lane_count, chunk_regular, chunk_fn, chunk_entry, chunk_base, results,
            slow_list, slow_count, slow_list, slow_count, kEmitBigStage);
        __syncthreads(); /* the table and the stage are written again */
    }
}

/* ------------------------------------------------------------------ decode: emit, several short end-of-stream chunks a workgroup */

/*
 * dec_emit_fast<TAIL> for the chunks dec_sync_pack took several to a workgroup, the same way: the workgroup's 512 threads
 * are slots of 4 x ceil(width / 2) threads (a thread a quarter of two neighbouring sub-chunks), a chunk a slot, each
 * with a stage of its own for the launch's largest chunk; the table and the barriers are shared, the symbol offsets of
 * all slots' lanes come from one scan over the lanes back to back.  The same test decides which chunks go this way as in
 * dec_emit_fast<TAIL> (dec_emit_tail, beside this kernel, applies it too); the others go on the list for the long way.
 */
template <u32 LB>
struct emit_pack_shared {
    u32 wlut[1u << LB];
    u32 lane_excl[HUFD_DEC_LANES + 4]; /* symbols of the lanes in front, all slots' lanes back to back (a slot's lane 0 counts nothing) */
    u32 wave_tot[HUFD_DEC_LANES / 64];
    u32 slot_chunk[kPackMaxSlots];     /* the slot's chunk, or HUFD_NONE32: nothing to do for the slot here */
    u32 slot_full[kPackMaxSlots];      /* ... its whole lanes */
    u8 dump[512];                      /* where a chain that has nothing to emit writes */
    __attribute__((aligned(16))) u8 stage[16]; /* slots x emit_pack_stage_bytes(stage_limit) */
};
__host__ __device__ constexpr u32 emit_pack_stage_bytes(u32 stage_limit) {
    return (stage_limit + 32u + 15u) & ~15u;
}

template <u32 LB, u32 SURE>
__global__ __launch_bounds__(kEmitFastThreads, 6) void dec_emit_pack_kernel(
    hufd_tables tb,
    const hufd_chunk_rec *chunk_rec,
    const u32 *tail_chunks,
    u32 n_tail,
    u32 width,  /* lanes a slot (dec_sync_pack's) */
    u32 slots,  /* slots a workgroup: what its threads and its LDS hold */
    const u8 *d_in,
    u8 *d_out,
    const u16 *cp_tab,
    const u16 *lane_count,
    const u8 *chunk_regular,
    const u32 *chunk_fn,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    u32 *slow_list, /* chunks left to dec_emit_kernel */
    u32 *slow_count,
    u32 stage_limit /* symbols a slot's stage holds */) {

    emit_pack_shared<LB> &sh = *reinterpret_cast<emit_pack_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 t = threadIdx.x;
    const u32 half = (width + 1) / 2, slot_threads = kQuarters * half, slot_lanes = 2 * half;
    const u32 slot = t / slot_threads, tt = t % slot_threads;
    const u32 q = tt % kQuarters;
    const u32 lanes[kEmitChains] = {2 * (tt / kQuarters), 2 * (tt / kQuarters) + 1};
    constexpr u32 kLutPerThread = ((1u << LB) + kEmitFastThreads - 1) / kEmitFastThreads;
    u32 lut_raw[kLutPerThread];
#pragma unroll
    for (u32 j = 0; j < kLutPerThread; ++j) {
        const u32 i = t + j * kEmitFastThreads;
        lut_raw[j] = i < (1u << LB) ? tb.dec_lut[i >> (LB - tb.lut_bits)] : 0u;
    }
    const u32 li = blockIdx.x * slots + slot;
    const bool have = slot < slots && li < n_tail;
    const u32 c = have ? tail_chunks[li] : 0u;
    const u32 centry = have ? chunk_entry[c] : 0u;
    const u32 s0 = centry & 0xFFu;
    const hufd_chunk_rec rec = chunk_rec[c];
    const u64 valid = rec.valid;
    const u32 n_full = valid >= 8u ? (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES) : 0u;
    const u64 cbase = have ? chunk_base[c] : 0;
    const u16 *cpt = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES;
    const u32 merged_row = (kQuarters - 1) * HUFD_DEC_LANES;
    const u32 f0 = have ? chunk_fn[(u64)c * ns + s0] : 0u;
    const u32 chunk_symbols = wide_count(f0);
    const u32 regular = have ? chunk_regular[c] : 0u;
    /* (the stream ended before this chunk; fewer than 136 bytes: dec_emit_tail does the whole chunk) */
    const bool wanted = have && (centry & 0x100u) != 0 && regular != 3;
    const bool fits = wanted && regular == 2 && ((cpt[merged_row] >> s0) & 1u) != 0 && cbase + chunk_symbols <= rec.out_cap;
    const bool fast = fits && chunk_symbols + 16 <= stage_limit && n_full + 2 <= slot_lanes &&
                      (lds_offset_of(sh.wlut) & ((4u << LB) - 1u)) == 0 && SURE <= row_walk(LB, tb.max_bits).sure;
    if (tt == 0 && slot < kPackMaxSlots) {
        sh.slot_chunk[slot] = fast ? c : HUFD_NONE32;
        sh.slot_full[slot] = n_full;
        if (wanted && !fast) {
            slow_list[atomicAdd(slow_count, 1u)] = c;
        }
    }

    /* my quarters: rows 8q .. 8q+7 and the word after them */
    constexpr u32 kRows = kSubWords / kQuarters;
    u32 w[kEmitChains][kRows + 1];
    u32 my_cp[kEmitChains], next_cp[kEmitChains], entry_state[kEmitChains], cnt[kEmitChains];
    bool whole[kEmitChains];
#pragma unroll
    for (u32 ch = 0; ch < kEmitChains; ++ch) {
        whole[ch] = fast && lanes[ch] < n_full;
#pragma unroll
        for (u32 j = 0; j <= kRows; ++j) {
            w[ch][j] = 0;
        }
        my_cp[ch] = next_cp[ch] = entry_state[ch] = cnt[ch] = 0;
        if (whole[ch]) {
            const u8 *sub = d_in + rec.src_off + (u64)lanes[ch] * HUFD_DEC_SUB_BYTES;
            const unaligned_uint4 *p = reinterpret_cast<const unaligned_uint4 *>(sub + q * kRows * 4);
#pragma unroll
            for (u32 j = 0; j < kRows / 4; ++j) {
                const unaligned_uint4 v = p[j];
                w[ch][4 * j + 0] = __builtin_bswap32(v.x);
                w[ch][4 * j + 1] = __builtin_bswap32(v.y);
                w[ch][4 * j + 2] = __builtin_bswap32(v.z);
                w[ch][4 * j + 3] = __builtin_bswap32(v.w);
            }
            w[ch][kRows] = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(sub + (q + 1) * kRows * 4)->x);
            my_cp[ch] = q ? cpt[(q - 1) * HUFD_DEC_LANES + lanes[ch]] : 0u;
            next_cp[ch] = q + 1 < kQuarters ? cpt[q * HUFD_DEC_LANES + lanes[ch]] : 0u;
            entry_state[ch] = lanes[ch] ? (u32)(cpt[merged_row + lanes[ch] - 1] >> 12) : s0;
            cnt[ch] = lane_count[(u64)c * HUFD_DEC_LANES + lanes[ch]];
        }
    }
#pragma unroll
    for (u32 j = 0; j < kLutPerThread; ++j) {
        const u32 i = t + j * kEmitFastThreads;
        const u32 e = lut_raw[j];
        const u32 len = e & 0xFFu;
        if (i < (1u << LB)) {
            sh.wlut[i] = ((e >> 8) << 16) | ((0x10000u - (len ? len : kWalkDeadLen)) & 0xFFFFu);
        }
    }
    __syncthreads();

    /* where every sub-chunk's symbols go: one scan over all slots' lanes, back to back (thread g < 256 = lane g % slot_lanes
     * of slot g / slot_lanes); the two sub-chunks a stream can end in count (dec_sync_tail wrote their symbols), a slot's
     * lane 0 does not (its count follows from the chunk's total) */
    u32 incl = 0, own = 0;
    if (t < HUFD_DEC_LANES) {
        const u32 sa = t / slot_lanes, la = t % slot_lanes;
        const u32 ca = sa < slots && sa < kPackMaxSlots ? sh.slot_chunk[sa] : HUFD_NONE32;
        own = ca != HUFD_NONE32 && la != 0 && la < sh.slot_full[sa] + 2 && la < HUFD_DEC_LANES ? lane_count[(u64)ca * HUFD_DEC_LANES + la] : 0u;
        incl = wave_inclusive_sum(own, t & (kWave - 1));
        if ((t & (kWave - 1)) == kWave - 1) {
            sh.wave_tot[t / kWave] = incl;
        }
    }
    __syncthreads();
    if (t < HUFD_DEC_LANES) {
        u32 before = 0;
#pragma unroll
        for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
            before += wv < t / kWave ? sh.wave_tot[wv] : 0u;
        }
        sh.lane_excl[t] = before + incl - own;
        if (t == HUFD_DEC_LANES - 1) {
            sh.lane_excl[HUFD_DEC_LANES] = before + incl;
        }
    }
    __syncthreads();
    if (!fast) {
        return; /* (the barrier behind the walk counts the waves that are still there) */
    }

    const u32 base_g = slot * slot_lanes; /* my slot's lane 0 among all slots' lanes */
    const u32 slot_first = sh.lane_excl[base_g], slot_end = sh.lane_excl[base_g + slot_lanes];
    const u32 first_count = chunk_symbols - (slot_end - slot_first); /* sub-chunk 0, entered in state s0 */
    const auto base_of = [&](u32 l) { return l ? first_count + sh.lane_excl[base_g + l] - slot_first : 0u; };
    u8 *out_ptr = d_out + rec.out_off + cbase;
    const u32 mis = (u32)((uintptr_t)out_ptr & 15u);
    const row_walk rw(LB, tb.max_bits);
    u8 *const lds_bytes = reinterpret_cast<u8 *>(&sh);
    const u32 stage_at = (u32)(sh.stage - lds_bytes) + slot * emit_pack_stage_bytes(stage_limit), dump_at = (u32)(sh.dump - lds_bytes);
    u32 st[kEmitChains], dst[kEmitChains];
    bool idle[kEmitChains]; /* a chain with nothing to emit still walks (the two go in step): over zeros, into the dump */
    bool extend = false;
#pragma unroll
    for (u32 ch = 0; ch < kEmitChains; ++ch) {
        const u32 lane_n = lanes[ch] ? cnt[ch] : first_count;
        const bool mine = whole[ch] && (q == 0 || (my_cp[ch] & 0x8000u) != 0);
        const u32 first = q ? lane_n - (my_cp[ch] & 0x7FFu) : 0u;
        st[ch] = rw.state_at(q ? (my_cp[ch] >> 11) & 15u : entry_state[ch], 0);
        dst[ch] = mine ? stage_at + mis + base_of(lanes[ch]) + first : dump_at;
        idle[ch] = !mine;
        if (ch == 0) {
            extend = whole[0] && q == 0 && lanes[0] == 0 && !(next_cp[ch] & 0x8000u);
        }
    }
    const u32 table = lds_offset_of(sh.wlut);
    if (!__all(idle[0] && idle[1])) {
#pragma unroll
        for (u32 r = 0; r < kRows; ++r) {
            u64 pair[kEmitChains];
#pragma unroll
            for (u32 ch = 0; ch < kEmitChains; ++ch) {
                pair[ch] = ((u64)w[ch][r] << 32) | w[ch][r + 1];
            }
#pragma unroll
            for (u32 i = 0; i < SURE; ++i) { /* the codes that are certain to start in this row, the two chains in turn */
                u32 e[kEmitChains];
#pragma unroll
                for (u32 ch = 0; ch < kEmitChains; ++ch) {
                    e[ch] = lds_word_at(((u32)(pair[ch] >> (st[ch] & 63u)) & rw.mask) | table);
                }
#pragma unroll
                for (u32 ch = 0; ch < kEmitChains; ++ch) {
                    lds_bytes[dst[ch]++] = (u8)(e[ch] >> 16);
                    st[ch] += e[ch];
                }
            }
#pragma unroll
            for (u32 ch = 0; ch < kEmitChains; ++ch) {
                while ((st[ch] & 0xFFFFu) > rw.thr) {
                    const u32 e = lds_word_at(((u32)(pair[ch] >> (st[ch] & 63u)) & rw.mask) | table);
                    lds_bytes[dst[ch]++] = (u8)(e >> 16);
                    st[ch] += e;
                }
                st[ch] += 32u;
                /* an idle chain starts every row afresh: whatever it decodes, its state and its writes stay in bounds */
                st[ch] = idle[ch] ? rw.state_at(0, 0) : st[ch];
                dst[ch] = idle[ch] ? dump_at : dst[ch];
            }
        }
    }
    if (extend) {
        /* rare: sub-chunk 0 on through the second quarter (and further while checkpoints are missing), words straight from memory */
        const u32 extend_end = (cpt[1 * HUFD_DEC_LANES] & 0x8000u) ? 2 * kRows : ((cpt[2 * HUFD_DEC_LANES] & 0x8000u) ? 3 * kRows : 4 * kRows);
        u32 hi = w[0][kRows];
        for (u32 r = kRows; r < extend_end; ++r) {
            const u32 lo = __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(d_in + rec.src_off + (r + 1) * 4)->x);
            const u64 pair = ((u64)hi << 32) | lo;
            while ((st[0] & 0xFFFFu) > rw.thr) {
                const u32 e = lds_word_at(((u32)(pair >> (st[0] & 63u)) & rw.mask) | table);
                lds_bytes[dst[0]++] = (u8)(e >> 16);
                st[0] += e;
            }
            st[0] += 32u;
            hi = lo;
        }
    }
    /* (the symbols of the one or two sub-chunks behind the whole lanes are dec_emit_tail's, straight to memory) */
    __syncthreads();

    {
        /* my slot's stage byte b belongs at (out_ptr - mis) + b: whole 16-byte rows go out aligned */
        u8 *gbase = out_ptr - mis;
        const u8 *stage = lds_bytes + stage_at;
        const u32 lo = mis, hi = mis + base_of(n_full); /* (the whole lanes' symbols: n_full + 2 <= slot_lanes) */
        const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
        if (row_lo <= row_hi) {
            /* (fewer than 16 bytes in front of the first whole row and behind the last: one byte a thread at most) */
            if (lo + tt < row_lo * 16) {
                gbase[lo + tt] = stage[lo + tt];
            }
            for (u32 r = row_lo + tt; r < row_hi; r += slot_threads) {
                *reinterpret_cast<uint4 *>(gbase + (u64)r * 16) = *reinterpret_cast<const uint4 *>(stage + r * 16);
            }
            if (row_hi * 16 + tt < hi) {
                gbase[row_hi * 16 + tt] = stage[row_hi * 16 + tt];
            }
        } else if (lo + tt < hi) {
            gbase[lo + tt] = stage[lo + tt]; /* no whole row: fewer than 31 bytes */
        }
    }
}

/*
 * The symbols of the one or two sub-chunks a stream ends in, for the chunks dec_emit_fast<TAIL> took: one THREAD
 * per chunk, straight to memory (dec_sync_tail's walk again, this time keeping the symbols), and the record of
 * where and why the true path stops (source/huffman.c:240-255).
 */
__global__ __launch_bounds__(kTailThreads) void dec_emit_tail_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u32 *tail_chunks,
    u32 n_tail,
    const u8 *d_in,
    u8 *d_out,
    const u16 *cp_tab,
    const u16 *lane_count,
    const u8 *chunk_regular,
    const u32 *chunk_fn,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results,
    u32 stage_limit /* symbols the stage of dec_emit_fast<TAIL>'s launch holds: what that kernel takes, this one finishes */) {

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
    const u32 centry = chunk_entry[c];
    const u32 kind = chunk_regular[c];
    if (!(centry & 0x100u) || (kind != 2 && kind != 3)) {
        return;
    }
    const u32 ns = tb.n_states, s0 = centry & 0xFFu;
    const u32 item_index = chunk_item[c];
    const hufd_dec_item it = items[item_index];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len - chunk_off;
    const u64 cbase = chunk_base[c];
    if (kind == 3) {
        /* the whole chunk: symbols while there is room, the start bit of the one that finds none
         * (source/huffman.c:257-268), else where and why the stream stops (:240-255) */
        u32 *tiny = sh.words[threadIdx.x];
        load_be32_run(tiny, d_in + it.in_off + chunk_off, valid, kTailWords);
        const u64 room = it.out_cap > cbase ? it.out_cap - cbase : 0;
        u8 *out = d_out + it.out_off + cbase;
        const u32 rem = (u32)(valid * 8);
        tail_reader tr;
        tr.start(tiny, s0);
        u32 pos = s0, why = HUFD_STOP_NONE;
        u64 n = 0;
        for (;;) {
            u32 sym = 0;
            const u32 len = code_at(tr.peek(), lut, tb.lut_bits, pos, rem, &sym, &why);
            if (!len) {
                results[item_index].stop_kind = why;
                results[item_index].stop_bit = chunk_off * 8 + pos;
                break;
            }
            if (n == room) {
                results[item_index].cap_bit = chunk_off * 8 + pos;
                break;
            }
            out[n++] = (u8)sym;
            tr.skip(len);
            pos += len;
        }
        return;
    }
    const u16 *cpt = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES;
    const u32 merged_row = (kQuarters - 1) * HUFD_DEC_LANES;
    const u32 f0 = chunk_fn[(u64)c * ns + s0];
    const u32 chunk_symbols = wide_count(f0);
    /* exactly the chunks dec_emit_fast<TAIL> emitted in one pass (the two-pass and the long way do their own ends) */
    if (((cpt[merged_row] >> s0) & 1u) == 0 || cbase + chunk_symbols > it.out_cap || chunk_symbols + 16 > stage_limit) {
        return; /* (the same test, with the same stage, as dec_emit_fast<TAIL>'s `fast`: the two kernels run side by side) */
    }
    const u32 n_full = (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES);
    const u32 first = n_full, second = n_full + 1;
    const u32 n_first = lane_count[(u64)c * HUFD_DEC_LANES + first];
    const u32 n_second = second < HUFD_DEC_LANES ? lane_count[(u64)c * HUFD_DEC_LANES + second] : 0u;
    const u32 entry = cpt[merged_row + first - 1] >> 12;
    const u8 *tsrc = d_in + it.in_off + chunk_off + (u64)n_full * HUFD_DEC_SUB_BYTES;
    const u64 tail_bytes = valid - (u64)n_full * HUFD_DEC_SUB_BYTES;
    u32 *words = sh.words[threadIdx.x];
    load_be32_run(words, tsrc, tail_bytes, kTailWords);
    const u32 limit = (second < HUFD_DEC_LANES ? 2u : 1u) * HUFD_DEC_SUB_BITS;
    u8 *out = d_out + it.out_off + cbase + (chunk_symbols - n_first - n_second);
    u32 stop_pos = 0, stop_why = HUFD_STOP_NONE;
    (void)tail_follow(words, lut, tb.lut_bits, entry, (u32)(tail_bytes * 8), limit, out, &stop_pos, &stop_why);
    if (stop_why != HUFD_STOP_NONE) {
        results[item_index].stop_kind = stop_why;
        results[item_index].stop_bit = (chunk_off + (u64)n_full * HUFD_DEC_SUB_BYTES) * 8 + stop_pos;
    }
}


static uint32_t dec_emit_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + HUFD_DEC_STAGE_BYTES + kGroups * HUFD_DEC_MAX_STATES * 4 + kGroups * 8 +
           HUFD_DEC_LANES * 12 + 16 + kCpRows * HUFD_DEC_LANES * 2 + (2u << tb->lut_bits) + 16;
}

} /* namespace */

using hufk_host::persistent_grid;
using hufk_host::stage_mark;
using hufk_host::current_compute_units;
using hufk_host::kBesideMinChunks;
using hufk_host::decode_launch_state;

hipError_t hufk_host::init_decode_emit(int lds_max) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(&dec_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
#define HUFK_ALLOW_BIG_LDS(LBV, SUREV)                                                                                  \
    if (e == hipSuccess) {                                                                                             \
        e = hipFuncSetAttribute(                                                                                       \
            reinterpret_cast<const void *>(&dec_emit_big_kernel<LBV, SUREV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
            lds_max);                                                                                                  \
    }
    HUFK_ALLOW_BIG_LDS(10, 3)
    HUFK_ALLOW_BIG_LDS(10, 4)
    HUFK_ALLOW_BIG_LDS(12, 2)
#undef HUFK_ALLOW_BIG_LDS
    return e;
}

void hufk_host::decode_emit_stage(const struct hufk_decode_args *a, hipStream_t st, const decode_launch_state &s) {
    const uint32_t lb_of_launch = s.lb, sure = s.sure;
    /* regular chunks that fit their output the short way; the rest through the list */
    u32 *const emit_count = a->counters + HUFK_DEC_COUNT_EMIT, *const dense_count = a->counters + HUFK_DEC_COUNT_DENSE;
#define HUFK_LAUNCH_EMIT_FAST(LBV, TAILV, SUREV, GRID, STREAMV)                                                          \
hipLaunchKernelGGL(                                                                                                \
    (dec_emit_fast_kernel<LBV, TAILV, SUREV>), dim3(GRID), dim3((TAILV) && (LBV) == 10 ? tail_block : kEmitFastThreads), \
    emit_lds_bytes<LBV>(TAILV ? tail_stage : HUFD_DEC_STAGE_BYTES), STREAMV, a->tables, a->chunk_rec,              \
    emit_single_chunks, (const u8 *)a->d_in, (u8 *)a->d_out, (const u16 *)a->cp_tab, (const u16 *)a->lane_count,   \
    (const u8 *)a->chunk_regular, (const u32 *)a->chunk_fn, (const u32 *)a->chunk_entry,                           \
    (const u64 *)a->chunk_base, a->results, a->emit_list, emit_count,                                            \
    !TAILV && big ? a->dense_list : a->emit_list, !TAILV && big ? dense_count : emit_count,                  \
    TAILV ? tail_stage : HUFD_DEC_STAGE_BYTES)
    /* dec_emit_fast<TAIL>'s workgroup: four quarters to two sub-chunks a thread over the most whole lanes a wide chunk of the
     * launch has (+ the wave that rounds it up), never fewer than the 256 threads of its scan: BASELINE configs[3]'s chunks
     * have 152 whole lanes -- 320 threads, six workgroups a CU's wave slots hold instead of four (0.83 -> 0.78 ms) */
    const uint32_t tail_live = a->tail_wide_lanes ? (kQuarters * ((a->tail_wide_lanes + 1) / 2) + kWave - 1) / kWave * kWave : kEmitFastThreads;
    const uint32_t tail_block = tail_live < HUFD_DEC_LANES ? HUFD_DEC_LANES : (tail_live < kEmitFastThreads ? tail_live : kEmitFastThreads);
    const bool some_inside = a->n_tail < a->n_chunks;
    const bool big = !a->quiet; /* (quiet: chunks of more symbols than the stage holds take the long way with the others) */
    /* (the few chunks streams end in beside the many inside streams: see the sync kernels above.  And in any case
     * dec_emit_tail beside dec_emit_fast<TAIL>: it works out for itself which chunks that kernel takes, reads
     * nothing it writes and writes other bytes) */
    const bool have_side = a->n_tail && a->side_stream && a->fork_event && a->join_event && a->n_chunks >= kBesideMinChunks;
    const bool beside = some_inside && have_side && (uint64_t)a->n_tail * 8 <= a->n_chunks;
    hipStream_t tst = beside ? (hipStream_t)a->side_stream : st;
    hipStream_t ends_st = have_side ? (hipStream_t)a->side_stream : st;
    (void)tst;
    (void)ends_st;
    if (have_side) {
        (void)hipEventRecord((hipEvent_t)a->fork_event, st);
        (void)hipStreamWaitEvent((hipStream_t)a->side_stream, (hipEvent_t)a->fork_event, 0);
    }
    /* (chunks inside a stream: with the coder's number of certain steps a row compiled in, where there is such a build) */
    const uint32_t emit_sure = sure; /* (the build for the coder: see above) */
    /* the chunks streams end in, where they are short and many: several to a workgroup (dec_emit_pack, as dec_sync_pack) */
    const uint32_t epack_width = a->tail_lanes + 2u < 16u ? 16u : a->tail_lanes + 2u;
    const uint32_t epack_threads = kQuarters * ((epack_width + 1) / 2);
    const uint32_t epack_stage = emit_pack_stage_bytes((a->tail_stage_bytes + 255u) & ~255u);
    const uint32_t epack_fixed = (uint32_t)(a->tables.lut_bits <= 10 ? sizeof(emit_pack_shared<10>) : sizeof(emit_pack_shared<12>));
    uint32_t epack_slots = kEmitFastThreads / epack_threads;
    epack_slots = epack_slots > kPackMaxSlots ? kPackMaxSlots : epack_slots;
    epack_slots = epack_slots * epack_stage + epack_fixed > 60u * 1024u ? (60u * 1024u - epack_fixed) / epack_stage : epack_slots;
    const bool epack = a->one_chunk_a_workgroup == 0 && a->n_tail_narrow >= kPackMinChunks && epack_width <= HUFD_DEC_LANES / 2 &&
                       a->tail_stage_bytes && epack_slots >= 2;
    /* the stage of the launch for the chunks streams end in: what the plan says such a chunk can hold at most */
    const uint32_t tail_stage = epack ? (a->tail_stage_bytes + 255u) & ~255u
                                : a->tail_stage_bytes >= 4096 && a->tail_stage_bytes < HUFD_DEC_STAGE_BYTES
                                    ? (a->tail_stage_bytes + 255u) & ~255u
                                    : HUFD_DEC_STAGE_BYTES;
    const uint32_t e_packed = epack ? a->n_tail_narrow : 0u, e_single = a->n_tail - e_packed;
    const u32 *emit_single_chunks = a->tail_chunks + e_packed; /* (the chunks streams end in that get a workgroup each) */
#define HUFK_LAUNCH_EMIT_PACK(LBV, SUREV)                                                                               \
hipLaunchKernelGGL(                                                                                                \
    (dec_emit_pack_kernel<LBV, SUREV>), dim3((e_packed + epack_slots - 1) / epack_slots), dim3(kEmitFastThreads),    \
    (uint32_t)sizeof(emit_pack_shared<LBV>) + epack_slots * epack_stage, tst, a->tables, a->chunk_rec, a->tail_chunks, \
    e_packed, epack_width, epack_slots, (const u8 *)a->d_in, (u8 *)a->d_out, (const u16 *)a->cp_tab,               \
    (const u16 *)a->lane_count, (const u8 *)a->chunk_regular, (const u32 *)a->chunk_fn, (const u32 *)a->chunk_entry, \
    (const u64 *)a->chunk_base, a->emit_list, emit_count, tail_stage)
    /* chunks of short codes that hold more symbols than dec_emit_fast's stage: dec_emit_big where the chunk lies
     * inside its stream; the others take the long way (dec_emit) */
    if (e_packed) {
        if (lb_of_launch == 10) {
            switch (emit_sure) {
                case 3: HUFK_LAUNCH_EMIT_PACK(10, 3); break;
                default: HUFK_LAUNCH_EMIT_PACK(10, 4); break;
            }
        } else {
            HUFK_LAUNCH_EMIT_PACK(12, 2);
        }
    }
#undef HUFK_LAUNCH_EMIT_PACK
    if (lb_of_launch == 10) {
        if (e_single) {
            switch (emit_sure) {
                case 3: HUFK_LAUNCH_EMIT_FAST(10, true, 3, e_single, tst); break;
                default: HUFK_LAUNCH_EMIT_FAST(10, true, 4, e_single, tst); break;
            }
        }
        if (some_inside) {
            switch (emit_sure) {
                case 3: HUFK_LAUNCH_EMIT_FAST(10, false, 3, a->n_chunks, st); break;
                default: HUFK_LAUNCH_EMIT_FAST(10, false, 4, a->n_chunks, st); break;
            }
        }
    } else {
        if (e_single) {
            HUFK_LAUNCH_EMIT_FAST(12, true, 2, e_single, tst);
        }
        if (some_inside) {
            HUFK_LAUNCH_EMIT_FAST(12, false, 2, a->n_chunks, st);
        }
    }
#undef HUFK_LAUNCH_EMIT_FAST
    if (a->n_tail) {
        const uint32_t lds = (uint32_t)sizeof(tail_lds) + (2u << a->tables.lut_bits);
        hipLaunchKernelGGL(
            dec_emit_tail_kernel, dim3((a->n_tail + kTailThreads - 1) / kTailThreads), dim3(kTailThreads), lds, ends_st,
            a->tables, a->items, a->chunk_item, a->tail_chunks, a->n_tail, (const u8 *)a->d_in, (u8 *)a->d_out,
            (const u16 *)a->cp_tab, (const u16 *)a->lane_count, (const u8 *)a->chunk_regular,
            (const u32 *)a->chunk_fn, (const u32 *)a->chunk_entry, (const u64 *)a->chunk_base, a->results, tail_stage);
    }
    if (have_side) {
        (void)hipEventRecord((hipEvent_t)a->join_event, (hipStream_t)a->side_stream);
        (void)hipStreamWaitEvent(st, (hipEvent_t)a->join_event, 0);
    }
    /* chunks of short codes (more symbols than one stage): resident workgroups with a stage twice as long take turns
     * over their list */
#define HUFK_LAUNCH_EMIT_BIG(LBV, SUREV)                                                                                \
do {                                                                                                               \
    const uint32_t lds = emit_lds_bytes<LBV>(kEmitBigStage);                                                       \
    hipLaunchKernelGGL(                                                                                            \
        (dec_emit_big_kernel<LBV, SUREV>),                                                                         \
        dim3(persistent_grid(dec_emit_big_kernel<LBV, SUREV>, kEmitFastThreads, lds, a->n_chunks)),                 \
        dim3(kEmitFastThreads), lds, st, a->tables, a->chunk_rec, (const u8 *)a->d_in, (u8 *)a->d_out,             \
        (const u16 *)a->cp_tab, (const u16 *)a->lane_count, (const u8 *)a->chunk_regular, (const u32 *)a->chunk_fn, \
        (const u32 *)a->chunk_entry, (const u64 *)a->chunk_base, a->results, a->emit_list, emit_count,          \
        (const u32 *)a->dense_list, (const u32 *)dense_count);                                                  \
} while (0)
    if (!big) {
        /* (nothing was listed for it) */
    } else if (lb_of_launch == 10) {
        switch (emit_sure) {
            case 3: HUFK_LAUNCH_EMIT_BIG(10, 3); break;
            default: HUFK_LAUNCH_EMIT_BIG(10, 4); break;
        }
    } else {
        HUFK_LAUNCH_EMIT_BIG(12, 2);
    }
#undef HUFK_LAUNCH_EMIT_BIG
    hipLaunchKernelGGL(
        dec_emit_kernel,
        dim3(persistent_grid(dec_emit_kernel, kEmitThreads, dec_emit_lds_bytes(&a->tables), a->n_chunks)),
        dim3(kEmitThreads), dec_emit_lds_bytes(&a->tables), st, a->tables, a->items, a->chunk_item, a->n_chunks,
        (const u8 *)a->d_in, (u8 *)a->d_out, a->fn_tab, a->cp_tab, (const u16 *)a->lane_count,
        (const u8 *)a->chunk_regular, a->chunk_entry, a->chunk_base, a->results, (const u32 *)a->emit_list,
        (const u32 *)emit_count, a->counters, a->summary, a->counters_self_cleared);
}
