/*
 * Decode of items that are not cut into chunks: a thread (dec_tiny), a wave or a workgroup (dec_deep) an item; long items of
 * coders with codes of more than 12 bits a workgroup per 32 KiB block (dec_wide_*, by transfer functions dec_wide_fn_*);
 * coders whose codes all have one length (dec_fixed_*); one host-pointer call in one launch (dec_block).  Each replays
 * reference source/huffman.c:228-268 for its items.
 */
#include "decode_common.hpp"
#include "launch_common.hpp"

namespace {

/*
 * Items of at most HUFD_DEC_TINY_BYTES encoded bytes (header-field sized strings): one THREAD per item does
 * all of source/huffman.c:228-268 for it -- from the item's first bit, a symbol per code while there is room,
 * the start bit of the first symbol that finds none, how many symbols the stream holds, where and why it
 * stops -- reading the stream from memory in aligned 16-byte blocks.  No chunks, transfer functions or scan: for
 * such items they cost far more than the symbols.
 */
constexpr u32 kTinyDecThreads = 128;
constexpr u32 kTinyDecDeepThreads = 512; /* a batch of items of a coder with long codes: the linked tables (tens of KiB) are filled per workgroup, and what a CU's LDS holds of them bounds its waves */

struct stream_reader {
    /* the stream 16 aligned bytes at a time: what limits these one-lane-one-stream walks is the number of
     * memory requests (every lane of a load touches a line of its own), not the bytes */
    const uint4 *blocks; /* 16-byte aligned, at or in front of the first byte looked at */
    u64 end;             /* bytes from there to the end of the item: what follows reads as zero */
    uint4 cur;
    u32 cur_block;
    u64 win;
    u32 nb, next, ahead;

    __device__ __forceinline__ u32 word(u32 i) {
        const u32 b = i >> 2;
        if (b != cur_block) {
            cur_block = b;
            cur = (u64)b * 16 < end ? blocks[b] : uint4{0, 0, 0, 0};
        }
        const u32 k = i & 3u;
        const u32 raw = k == 0 ? cur.x : (k == 1 ? cur.y : (k == 2 ? cur.z : cur.w));
        const u64 at = (u64)i * 4;
        if (at + 4 <= end) {
            return __builtin_bswap32(raw);
        }
        return at < end ? __builtin_bswap32(raw) & (~0u << (8 * (4 - (u32)(end - at)))) : 0u;
    }
    /* from bit `bit` (0..7) of *first, with bytes_left bytes of the item at and behind first */
    __device__ __forceinline__ void start(const u8 *first, u64 bytes_left, u32 bit) {
        const u32 lead = (u32)(reinterpret_cast<uintptr_t>(first) & 15u);
        blocks = reinterpret_cast<const uint4 *>(first - lead);
        end = lead + bytes_left;
        cur_block = ~0u;
        cur = uint4{0, 0, 0, 0};
        const u32 pos = lead * 8 + bit, r = pos >> 5;
        const u32 w0 = word(r), w1 = word(r + 1);
        win = (((u64)w0 << 32) | w1) << (pos & 31u);
        nb = 64 - (pos & 31u);
        ahead = word(r + 2);
        next = r + 3;
    }
    __device__ __forceinline__ u32 peek() const {
        return (u32)(win >> 32);
    }
    __device__ __forceinline__ void skip(u32 len) {
        win <<= len;
        nb -= len;
        if (nb <= 32) {
            win |= (u64)ahead << (32 - nb);
            nb += 32;
            ahead = word(next);
            ++next;
        }
    }
};

/* decoded symbols on their way to memory, sixteen at a time (a one-lane-one-stream walk pays per store, not per
 * byte, and a 16-byte store needs no alignment) */
struct symbol_sink {
    u8 *at; /* where the next flushed symbol goes */
    u64 lo, hi;
    u32 have;

    __device__ __forceinline__ void begin(u8 *first) {
        at = first;
        lo = hi = 0;
        have = 0;
    }
    __device__ __forceinline__ void put(u32 symbol) {
        if (have < 8) {
            lo |= (u64)symbol << (8 * have);
        } else {
            hi |= (u64)symbol << (8 * (have - 8));
        }
        if (++have == 16) {
            unaligned_uint4 v;
            v.x = (u32)lo;
            v.y = (u32)(lo >> 32);
            v.z = (u32)hi;
            v.w = (u32)(hi >> 32);
            *reinterpret_cast<unaligned_uint4 *>(at) = v;
            at += 16;
            lo = hi = 0;
            have = 0;
        }
    }
    __device__ __forceinline__ void flush() {
        for (u32 k = 0; k < have; ++k) {
            at[k] = (u8)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))));
        }
        at += have;
        have = 0;
        lo = hi = 0;
    }
};

/* the entry (symbol << 8 | length, 0 = no code) for a window, out of the linked tables of a coder with long codes */
__device__ __forceinline__ u32 deep_entry(const u32 *deep, u32 window) {
    u32 e = deep[window >> (32 - HUFD_DEEP_ROOT_BITS)], used = HUFD_DEEP_ROOT_BITS;
    while (e & HUFD_DEEP_LINK) {
        const u32 width = (e >> 16) & 0xFFu;
        e = deep[(e & 0xFFFFu) + ((window << used) >> (32 - width))];
        used += width;
    }
    return e;
}

template <bool DEEP> /* codes of more than HUFD_DEC_MAX_LUT_BITS bits: linked tables, and items of any size */
__global__ __launch_bounds__(DEEP ? kTinyDecDeepThreads : kTinyDecThreads) void dec_tiny_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *tiny_items,
    u32 n_tiny,
    const u8 *d_in,
    u8 *d_out,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {

    u16 *lut = reinterpret_cast<u16 *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds);
    if (DEEP) {
        for (u32 i = threadIdx.x; i < tb.deep_entries; i += blockDim.x) {
            deep[i] = tb.deep_lut[i];
        }
    } else {
        for (u32 i = threadIdx.x; i < (1u << tb.lut_bits); i += blockDim.x) {
            lut[i] = tb.dec_lut[i];
        }
    }
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiny) {
        return;
    }
    const u32 item = tiny_items[t];
    const hufd_dec_item it = items[item];
    /* (a thread's item is a few hundred bytes at most: positions and counts fit 32 bits, which is half the instructions
     * of the loop's arithmetic) */
    const u32 rem = (u32)(it.in_len * 8);
    const u32 cap = it.out_cap > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (u32)it.out_cap;
    u32 pos = it.first_bit;
    u32 why = HUFD_STOP_NONE;
    u32 n = 0, cap_pos = 0xFFFFFFFFu;
    {
        /* The stretch of the stream where no question but "is this a code" and "is there room" has to be asked: every
         * window lies wholly inside the stream.  The kernel is bound by the instructions a symbol costs (150 in the
         * general loop below, with its end-of-stream tests, 64-bit positions and a reader that masks what lies behind
         * the stream); here: window, table, symbol into a word of four, shift, a refill every 32 bits out of the
         * 16-byte block in registers (the block behind it already asked for).  The general loop takes over where
         * this one stops -- near the end of the stream, at a window without a code, or when the room runs out -- and
         * reports what there is to report. */
        const u32 need = DEEP ? 32u : (tb.lut_bits > tb.max_bits ? tb.lut_bits : tb.max_bits); /* (DEEP: the linked tables look at up to 32 bits) */
        const u8 *first = d_in + it.in_off;
        const u32 lead = (u32)(reinterpret_cast<uintptr_t>(first) & 15u);
        const uint4 *blocks = reinterpret_cast<const uint4 *>(first - lead);
        const u32 end_bytes = lead + (u32)it.in_len;
        if (rem >= pos + need + 64) {
            uint4 blk = blocks[0], ahead = end_bytes > 16 ? blocks[1] : uint4{0, 0, 0, 0};
            u32 wi = (lead * 8 + pos) >> 5; /* the word (of the aligned blocks) the walk starts in: in block 0 */
            const auto next_word = [&]() -> u32 {
                if ((wi & 3u) == 0 && wi != 0) {
                    blk = ahead;
                    if (((wi >> 2) + 1) * 16 < end_bytes) {
                        ahead = blocks[(wi >> 2) + 1];
                    }
                }
                const u32 k = wi & 3u;
                const u32 raw = k == 0 ? blk.x : (k == 1 ? blk.y : (k == 2 ? blk.z : blk.w));
                ++wi;
                return __builtin_bswap32(raw);
            };
            const u32 w0 = next_word(), w1 = next_word();
            const u32 off = (lead * 8 + pos) & 31u;
            u64 win = (((u64)w0 << 32) | w1) << off;
            u32 nb = 64 - off;
            u8 *outp = d_out + it.out_off;
            u32 word = 0, sh = 0, n_held = 0;
            uint4 held = uint4{0, 0, 0, 0};
            const u32 lbits = tb.lut_bits;
            while (pos + need <= rem && n < cap) {
                const u32 e = DEEP ? deep_entry(deep, (u32)(win >> 32)) : lut[(u32)(win >> (64 - lbits))];
                const u32 len = e & 0xFFu;
                if (len == 0) {
                    break;
                }
                word |= ((e >> 8) & 0xFFu) << sh;
                sh += 8;
                if (sh == 32) {
                    held.x = n_held == 0 ? word : held.x;
                    held.y = n_held == 1 ? word : held.y;
                    held.z = n_held == 2 ? word : held.z;
                    held.w = n_held == 3 ? word : held.w;
                    word = 0;
                    sh = 0;
                    if (++n_held == 4) {
                        unaligned_uint4 v = {held.x, held.y, held.z, held.w};
                        *reinterpret_cast<unaligned_uint4 *>(outp) = v;
                        outp += 16;
                        n_held = 0;
                    }
                }
                ++n;
                pos += len;
                win <<= len;
                nb -= len;
                if (nb <= 32) {
                    win |= (u64)next_word() << (32 - nb);
                    nb += 32;
                }
            }
            if (n_held > 0) {
                reinterpret_cast<unaligned_u32 *>(outp)->x = held.x;
            }
            if (n_held > 1) {
                reinterpret_cast<unaligned_u32 *>(outp + 4)->x = held.y;
            }
            if (n_held > 2) {
                reinterpret_cast<unaligned_u32 *>(outp + 8)->x = held.z;
            }
            outp += 4 * n_held;
            for (u32 k = 0; k < sh; k += 8) {
                *outp++ = (u8)(word >> k);
            }
        }
    }
    stream_reader sr = {};
    if (pos < rem) {
        sr.start(d_in + it.in_off + (pos >> 3), it.in_len - (pos >> 3), pos & 7u);
    }
    symbol_sink sink;
    sink.begin(d_out + it.out_off + n);
    for (;;) {
        /* one symbol of source/huffman.c:232-255 */
        if (pos >= rem) {
            why = HUFD_STOP_END;
            break;
        }
        const u32 window = sr.peek();
        const u32 entry = DEEP ? deep_entry(deep, window) : lut[window >> (32 - tb.lut_bits)];
        const u32 len = entry & 0xFFu;
        if (len == 0) {
            why = HUFD_STOP_INVALID;
            break;
        }
        if (pos + len > rem) {
            why = HUFD_STOP_INCOMPLETE;
            break;
        }
        if (n < cap) {
            sink.put(entry >> 8);
        } else if (n == cap) {
            cap_pos = pos; /* source/huffman.c:257-268: this symbol is not consumed */
        }
        ++n;
        sr.skip(len);
        pos += len;
    }
    sink.flush();
    hufd_dec_result rs;
    rs.total_symbols = n;
    rs.stop_bit = pos;
    rs.cap_bit = cap_pos == 0xFFFFFFFFu ? kNoBit : (u64)cap_pos;
    rs.stop_kind = why;
    rs.reserved = 0;
    results[item] = rs;
    states[item].total_symbols = n;
}

/*
 * One small WORKGROUP per item, the stream taken (lanes x lane_bytes) at a time, for two kinds of item:
 *   - HUFD_DEC_TINY_BYTES < encoded bytes <= HUFD_DEC_COOP_BYTES, any coder: one wave, the item split evenly over
 *     its 64 lanes.  A chunk's workgroup, tables and three more launches cost such an item tens of times its symbols.
 *   - longer items of a coder with long codes (DEEP): 256 lanes x 128 bytes a round.  The chunked decoder's transfer
 *     functions need a state per possible entry offset (up to 32 there).
 * This road needs no entry states: every lane walks its bytes from a guessed
 * entry (bit 0), then again from where the lane in front of it really leaves, until no lane's entry changes --
 * walks from different entries fall into step within a few codes, so that is two or three rounds, and it is
 * exact whatever the stream does, because lane 0's entry is the true one and every round settles at least one
 * more lane.  A walk that stops (end of stream, no code, code cut off: source/huffman.c:232-255) leaves the lanes
 * behind it unreached.  Then a scan of the lanes' symbol counts and one more walk that writes the symbols.
 */
constexpr u32 kDeepThreads = 256; /* at most */
constexpr u32 kDeepLaneBytes = 128;
constexpr u32 kCoopThreads = 64; /* the one-wave variant for items of up to HUFD_DEC_COOP_BYTES */
constexpr u32 kDeepStop = 0xFFu; /* a lane's exit: its walk stopped, or the lane is never reached */

struct deep_shared {
    u32 exit_of[kDeepThreads];
    u32 scan[kDeepThreads];
    u32 changed;
    u32 stop_kind;
    u64 stop_bit;
    u64 cap_bit;
};

struct deep_walked {
    u64 pos;   /* where the walk ended: the first code start at or behind `to`, or where it stopped */
    u32 count; /* symbols whose codes start in [from, to) */
    u32 why;   /* HUFD_STOP_NONE: reached `to` */
};

/* follows the codes from stream bit `from` to the first code start at or behind `to`; writes symbol number
 * index + k to out[index + k] while that is below out_cap (out == NULL: count only) */
template <bool DEEP, bool GUESS = false> /* GUESS: a window without a code is stepped over a bit at a time (a walk that only looks for where the codes fall into step) */
__device__ __forceinline__ deep_walked deep_walk(
    const u32 *deep, const u16 *lut, u32 lut_bits, const u8 *in, u64 in_len, u64 from, u64 to, u8 *out, u64 index, u64 out_cap, u64 *cap_bit) {
    const u64 rem = in_len * 8;
    stream_reader sr = {};
    if (from < rem) {
        sr.start(in + (from >> 3), in_len - (from >> 3), (u32)(from & 7));
    }
    symbol_sink sink;
    sink.begin(out ? out + index : nullptr);
    deep_walked r;
    r.pos = from;
    r.count = 0;
    r.why = HUFD_STOP_NONE;
    while (r.pos < to) {
        if (r.pos >= rem) {
            r.why = HUFD_STOP_END;
            break;
        }
        const u32 entry = DEEP ? deep_entry(deep, sr.peek()) : lut[sr.peek() >> (32 - lut_bits)];
        const u32 len = entry & 0xFFu;
        if (len == 0) {
            if (GUESS) {
                sr.skip(1);
                r.pos += 1;
                continue;
            }
            r.why = HUFD_STOP_INVALID;
            break;
        }
        if (r.pos + len > rem) {
            r.why = HUFD_STOP_INCOMPLETE;
            break;
        }
        if (out) {
            const u64 k = index + r.count;
            if (k < out_cap) {
                sink.put(entry >> 8);
            } else if (k == out_cap) {
                *cap_bit = r.pos; /* source/huffman.c:257-268: this symbol is not consumed */
            }
        }
        ++r.count;
        sr.skip(len);
        r.pos += len;
    }
    sink.flush();
    return r;
}

/* What a lane's walks from its last few entries came to.  A stream whose walks never fall into step (code lengths
 * that share a divisor) sends the news of an entry a lane a round down the block; the lanes see the same few entries
 * over and over, and with these a round costs a look instead of a walk. */
struct lane_memo {
    u32 key[4], exit[4], count[4];
    u32 n;
    __device__ __forceinline__ void clear() {
        n = 0;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            key[k] = ~0u;
        }
    }
    __device__ __forceinline__ bool find(u32 start, u32 &ex, u32 &cnt) const {
        bool hit = false;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            if (key[k] == start) {
                ex = exit[k];
                cnt = count[k];
                hit = true;
            }
        }
        return hit;
    }
    __device__ __forceinline__ void put(u32 start, u32 ex, u32 cnt) {
        const u32 slot = n & 3u;
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            if (k == slot) {
                key[k] = start;
                exit[k] = ex;
                count[k] = cnt;
            }
        }
        ++n;
    }
};

template <bool DEEP>
__global__ __launch_bounds__(kDeepThreads) void dec_deep_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *deep_items,
    u32 fixed_lane_bytes, /* 0: the item split evenly over the lanes */
    const u8 *d_in,
    u8 *d_out,
    hufd_dec_item_state *states,
    hufd_dec_result *results,
    u64 wide_from,   /* gate == NULL: items of at least this many bytes are not this launch's (dec_wide_* take them) */
    const u32 *gate) /* != NULL: one such item after all, if dec_wide_* gave it up (the word is their ctl[0]) */ {

    if (gate ? gate[0] == 0 : items[deep_items[blockIdx.x]].in_len >= wide_from) {
        return;
    }
    deep_shared &sh = *reinterpret_cast<deep_shared *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds + sizeof(deep_shared));
    u16 *lut = reinterpret_cast<u16 *>(dyn_lds + sizeof(deep_shared));
    const u32 l = threadIdx.x, lanes = blockDim.x;
    if (DEEP) {
        for (u32 i = l; i < tb.deep_entries; i += lanes) {
            deep[i] = tb.deep_lut[i];
        }
    } else {
        for (u32 i = l; i < (1u << tb.lut_bits); i += lanes) {
            lut[i] = tb.dec_lut[i];
        }
    }
    if (l == 0) {
        sh.stop_kind = HUFD_STOP_NONE;
        sh.stop_bit = kNoBit;
        sh.cap_bit = kNoBit;
    }
    __syncthreads();
    const u32 item = deep_items[blockIdx.x];
    const hufd_dec_item it = items[item];
    const u8 *in = d_in + it.in_off;
    u8 *out = d_out + it.out_off;
    u32 lane_bytes = fixed_lane_bytes;
    if (lane_bytes == 0) {
        lane_bytes = (u32)(((it.in_len + lanes - 1) / lanes + 7) & ~7ull);
        lane_bytes = lane_bytes < 16 ? 16u : lane_bytes;
    }
    const u32 lane_bits = lane_bytes * 8;
    const u64 round_bytes = (u64)lanes * lane_bytes;
    const u64 n_blocks = (it.in_len + round_bytes - 1) / round_bytes;
    u64 symbols = 0; /* on the true path in front of this block */
    u32 entry = it.first_bit;
    for (u64 b = 0; b < n_blocks && entry != kDeepStop; ++b) {
        const u64 block_bytes = it.in_len - b * round_bytes < round_bytes ? it.in_len - b * round_bytes : round_bytes;
        const u32 n_lanes = (u32)((block_bytes + lane_bytes - 1) / lane_bytes);
        const bool active = l < n_lanes;
        const u64 lane_from = (b * round_bytes + (u64)l * lane_bytes) * 8, lane_to = lane_from + lane_bits;
        u32 start = l == 0 ? entry : 0u, my_exit = kDeepStop, my_count = 0;
        bool reached = true, walk = active;
        lane_memo memo;
        memo.clear();
        for (;;) {
            if (walk && !memo.find(start, my_exit, my_count)) {
                const deep_walked r = deep_walk<DEEP>(deep, lut, tb.lut_bits, in, it.in_len, lane_from + start, lane_to, nullptr, 0, 0, nullptr);
                my_exit = r.why == HUFD_STOP_NONE ? (u32)(r.pos - lane_to) : kDeepStop;
                my_count = r.count;
                memo.put(start, my_exit, my_count);
            }
            sh.exit_of[l] = active && reached ? my_exit : kDeepStop;
            if (l == 0) {
                sh.changed = 0;
            }
            __syncthreads();
            walk = false;
            if (active && l > 0) {
                const u32 prev = sh.exit_of[l - 1];
                if (prev == kDeepStop) {
                    if (reached) {
                        reached = false;
                        sh.changed = 1;
                    }
                } else if (!reached || prev != start) {
                    reached = true;
                    start = prev;
                    walk = true;
                    sh.changed = 1;
                }
            }
            __syncthreads();
            const bool again = sh.changed != 0;
            __syncthreads(); /* everyone has seen the flag and the exits before they are written again */
            if (!again) {
                break;
            }
        }
        /* where each lane's symbols go: an exclusive scan of the counts of the lanes on the true path */
        const u32 mine = active && reached ? my_count : 0u;
        sh.scan[l] = mine;
        __syncthreads();
        for (u32 d = 1; d < lanes; d *= 2) {
            const u32 add = l >= d ? sh.scan[l - d] : 0u;
            __syncthreads();
            sh.scan[l] += add;
            __syncthreads();
        }
        const u32 before = sh.scan[l] - mine, block_symbols = sh.scan[lanes - 1];
        const u32 last_exit = sh.exit_of[n_lanes - 1];
        if (active && reached) {
            u64 cap_bit = kNoBit;
            const deep_walked r = deep_walk<DEEP>(deep, lut, tb.lut_bits, in, it.in_len, lane_from + start, lane_to, out, symbols + before, it.out_cap, &cap_bit);
            if (cap_bit != kNoBit) {
                sh.cap_bit = cap_bit;
            }
            if (r.why != HUFD_STOP_NONE) {
                sh.stop_kind = r.why;
                sh.stop_bit = r.pos;
            }
        }
        __syncthreads(); /* exit_of and scan are free for the next block */
        symbols += block_symbols;
        entry = last_exit;
    }
    __syncthreads();
    if (l == 0) {
        hufd_dec_result rs;
        rs.total_symbols = symbols;
        rs.cap_bit = sh.cap_bit;
        rs.reserved = 0;
        if (sh.stop_kind != HUFD_STOP_NONE) {
            rs.stop_kind = sh.stop_kind;
            rs.stop_bit = sh.stop_bit;
        } else {
            /* the last code ended on the last bit of the stream */
            rs.stop_kind = HUFD_STOP_END;
            rs.stop_bit = it.in_len * 8;
        }
        results[item] = rs;
        states[item].total_symbols = symbols;
    }
}

/*
 * ONE LONG item of a coder with long codes, across the chip.  dec_deep gives such an item one workgroup that takes it
 * 32 KiB at a time: 0.12 GB/s whatever its length, and HPACK's own code is such a coder.  Here every 32 KiB block is a
 * workgroup's:
 *   dec_wide_settle<0>  the block's lanes settle on their entries as dec_deep's do, from a GUESS for lane 0 (block 0:
 *                       the item's true first bit); how the block is left goes to exits[0][block];
 *   dec_wide_settle<j>  (j = 1 .. kWideFixes) lane 0 takes exits[j - 1][block - 1] for its entry and the lanes settle
 *                       again (a handful of them walk: walks from different entries fall into step within a few
 *                       codes); exits[j][block], the block's symbols, and whether the block is left differently now;
 *   dec_wide_scan       where each block's symbols go, the item's total, its result record;
 *   dec_wide_emit       the walk that writes the symbols.
 * If no block is left differently in launch j than in launch j - 1, every block had its true entry in launch j: by
 * induction from block 0, whose entry is the item's first bit.  Launch j + 1 runs only if one was (it finds the flag
 * of launch j): on the HPACK code lengths one launch does; a coder whose walks take hundreds of bits to fall into step
 * (15-, 12- and 9-bit codes with a few others in between) needs two or three.  What is still moving after kWideFixes
 * launches -- a stream whose walks never fall into step -- raises ctl[0]: dec_wide_emit does nothing then, and dec_deep,
 * queued behind it with that word as its gate, decodes the item its way.  A walk that stops
 * (source/huffman.c:232-255) says nothing to the lane or block behind it while entries are guesses; of the settled
 * lanes the first that stops ends the stream, and what lies behind it is not part of it.
 */
constexpr u32 kWideStop = 0xFFu;
constexpr u32 kWideGuessBytes = 32;
constexpr u32 kWideFixes = 8; /* (an empty launch is 3 us; the road such an item takes otherwise is a thousand times slower) */
static_assert(HUFD_WIDE_BLOCK_BYTES == kDeepThreads * kDeepLaneBytes, "a block is one round of dec_deep's lanes");

/* ctl words */
constexpr u32 kWideGaveUp = 0;   /* set by dec_wide_scan */
constexpr u32 kWideStopBlock = 1; /* the first block whose true walk stops (dec_wide_scan) */
constexpr u32 kWideMoved = 2;    /* [+ j], j = 1 .. kWideFixes: a block was left differently in launch j */
constexpr u32 kWideStops = 16;   /* [+ j]: the first block whose walk stops, as of launch j */
constexpr u32 kWideFnRoad = 26;  /* set by dec_wide_fn_scan: the item went by transfer functions, dec_wide_fn_emit writes its symbols */
constexpr u32 kWideCtlWords = 32;
static_assert(kWideMoved + kWideFixes < kWideStops && kWideStops + kWideFixes < kWideFnRoad && kWideFnRoad < kWideCtlWords,
              "the ctl words do not overlap");
constexpr u32 kWideEntries = 32; /* entry bits of a lane: a code has at most 32 bits, so the first code start in a lane is bit 0 .. 31 */

struct dec_wide_layout {
    u64 ctl;        /* u32[kWideCtlWords] */
    u64 exits;      /* u32[kWideFixes + 1][n_blocks] */
    u64 count;      /* u32[n_blocks] symbols of the block's lanes up to the first that stops */
    u64 last;       /* u32[n_blocks] that lane (kDeepThreads: none stops) */
    u64 base;       /* u64[n_blocks] symbols in front of the block */
    u64 lane_start; /* u8[n_blocks][kDeepThreads] entry bit of the lane */
    u64 lane_exit;  /* u8[n_blocks][kDeepThreads] */
    u64 lane_count; /* u16[n_blocks][kDeepThreads] */
    /* the road by transfer functions (dec_wide_fn_*), for an item whose walks never fall into step */
    u64 fn_exit;    /* u8[n_blocks][kWideEntries][kDeepThreads] how a lane entered at bit e is left (kWideStop: its walk stops) */
    u64 fn_block;   /* u64[n_blocks][kWideEntries] the same for a block: exit in the low byte, symbols above it */
    u64 fn_entry;   /* u32[n_blocks] the bit the block is truly entered at */
    u64 bytes;
};

__host__ __device__ inline dec_wide_layout dec_wide_layout_of(u64 n_blocks) {
    dec_wide_layout l;
    const u64 row = (n_blocks * 4 + 63) & ~63ull;
    u64 at = 0;
    l.ctl = at;
    at += 128;
    l.exits = at;
    at += (kWideFixes + 1) * row;
    l.count = at;
    at += row;
    l.last = at;
    at += row;
    l.base = at;
    at += 2 * row;
    l.lane_start = at;
    at += n_blocks * kDeepThreads;
    l.lane_exit = at;
    at += n_blocks * kDeepThreads;
    l.lane_count = at;
    at += n_blocks * kDeepThreads * 2;
    at = (at + 63) & ~63ull;
    l.fn_exit = at;
    at += n_blocks * kWideEntries * kDeepThreads;
    l.fn_block = at;
    at += n_blocks * kWideEntries * 8;
    l.fn_entry = at;
    at += row;
    l.bytes = at;
    return l;
}

struct wide_shared {
    u32 exit_of[kDeepThreads];
    u32 scan[kDeepThreads];
    u32 changed[2];
    u32 last_lane;
    u32 pad;
};

template <bool FIRST>
__global__ __launch_bounds__(kDeepThreads) void dec_wide_settle_kernel(
    hufd_tables tb, const hufd_dec_item *items, const u32 *the_item, const u8 *d_in, u8 *block, u32 pass, u32 fails) {

    const hufd_dec_item it = items[the_item[0]];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    u32 *ctl = reinterpret_cast<u32 *>(block + lay.ctl);
    if (!FIRST && pass > 1 && ctl[kWideMoved + pass - 1] == 0) {
        return; /* the launch before this one left every block as the one before it did: done */
    }
    const u64 row = ((n_blocks * 4 + 63) & ~63ull) / 4;
    u32 *exits_now = reinterpret_cast<u32 *>(block + lay.exits) + (FIRST ? 0 : pass) * row;
    const u32 *exits_before = reinterpret_cast<const u32 *>(block + lay.exits) + (FIRST ? 0 : pass - 1) * row;
    const u64 b = blockIdx.x;
    const u32 l = threadIdx.x;
    u8 *lane_start = block + lay.lane_start + b * kDeepThreads, *lane_exit = block + lay.lane_exit + b * kDeepThreads;
    u16 *lane_count = reinterpret_cast<u16 *>(block + lay.lane_count) + b * kDeepThreads;

    wide_shared &sh = *reinterpret_cast<wide_shared *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds + sizeof(wide_shared));
    for (u32 i = l; i < tb.deep_entries; i += kDeepThreads) {
        deep[i] = tb.deep_lut[i];
    }
    if (l == 0) {
        sh.last_lane = kDeepThreads;
    }
    __syncthreads();
    const u8 *in = d_in + it.in_off;
    const u64 block_from = b * kDeepThreads * kDeepLaneBytes;
    const u64 block_bytes = it.in_len - block_from < (u64)kDeepThreads * kDeepLaneBytes ? it.in_len - block_from : (u64)kDeepThreads * kDeepLaneBytes;
    const u32 n_lanes = (u32)((block_bytes + kDeepLaneBytes - 1) / kDeepLaneBytes);
    const bool active = l < n_lanes;
    const u64 lane_from = (block_from + (u64)l * kDeepLaneBytes) * 8, lane_to = lane_from + kDeepLaneBytes * 8;
    u32 start, my_exit = kWideStop, my_count = 0;
    bool walk;
    lane_memo memo;
    memo.clear();
    if (FIRST) {
        start = b == 0 && l == 0 ? it.first_bit : 0u;
        walk = active;
        if (active && (b | l) != 0) {
            /* the first guess: where a walk from anywhere over the 32 bytes in front of the lane crosses into it (on the
             * HPACK code lengths: right for every lane tried; over 16 bytes, for 29 in 30 -- and one wrong lane is a
             * second walk for its whole wave) */
            const deep_walked g = deep_walk<true, true>(
                deep, nullptr, 0, in, it.in_len, lane_from - kWideGuessBytes * 8, lane_from, nullptr, 0, 0, nullptr);
            start = g.why == HUFD_STOP_NONE ? (u32)(g.pos - lane_from) : 0u;
        }
    } else {
        start = lane_start[l];
        my_exit = lane_exit[l];
        my_count = lane_count[l];
        memo.put(start, my_exit, my_count);
        walk = false;
        if (l == 0 && b > 0) {
            const u32 entry = exits_before[b - 1];
            if (entry == kWideStop) {
                /* the block in front stops: if it still does when nothing moves any more, this block is not part of
                 * the stream and whatever is written for it is not looked at */
            } else if (entry != start) {
                start = entry;
                walk = true;
            }
        }
    }
    for (u32 round = 0;; ++round) {
        if (walk && !memo.find(start, my_exit, my_count)) {
            const deep_walked r = deep_walk<true>(deep, nullptr, 0, in, it.in_len, lane_from + start, lane_to, nullptr, 0, 0, nullptr);
            my_exit = r.why == HUFD_STOP_NONE ? (u32)(r.pos - lane_to) : kWideStop;
            my_count = r.count;
            memo.put(start, my_exit, my_count);
        }
        sh.exit_of[l] = active ? my_exit : kWideStop;
        if (l == 0) {
            sh.changed[round & 1u] = 0; /* (the flag of the round before last: everyone has read it) */
        }
        __syncthreads();
        walk = false;
        if (active && l > 0) {
            const u32 prev = sh.exit_of[l - 1];
            if (prev != kWideStop && prev != start) {
                start = prev;
                walk = true;
                sh.changed[round & 1u] = 1;
            }
        }
        __syncthreads();
        if (!sh.changed[round & 1u]) {
            break;
        }
    }
    if (active && my_exit == kWideStop) {
        atomicMin(&sh.last_lane, l);
    }
    __syncthreads();
    const u32 last_lane = sh.last_lane;
    const bool reached = active && l <= last_lane;
    const u32 block_exit = last_lane < n_lanes ? kWideStop : sh.exit_of[n_lanes - 1];
    lane_start[l] = (u8)start;
    lane_exit[l] = (u8)my_exit;
    lane_count[l] = (u16)my_count;
    if (FIRST) {
        if (l == 0) {
            exits_now[b] = block_exit;
        }
        return;
    }
    sh.scan[l] = reached ? my_count : 0u;
    __syncthreads();
    for (u32 d = kDeepThreads / 2; d > 0; d /= 2) {
        if (l < d) {
            sh.scan[l] += sh.scan[l + d];
        }
        __syncthreads();
    }
    if (l == 0) {
        exits_now[b] = block_exit;
        reinterpret_cast<u32 *>(block + lay.count)[b] = sh.scan[0];
        reinterpret_cast<u32 *>(block + lay.last)[b] = last_lane < n_lanes ? last_lane : kDeepThreads;
        if ((block_exit != exits_before[b] && b + 1 < n_blocks) || fails) {
            atomicOr(&ctl[kWideMoved + pass], 1u); /* the block behind this one had a wrong entry */
        }
        if (block_exit == kWideStop) {
            atomicMin(&ctl[kWideStops + pass], (u32)b);
        }
    }
}

__global__ __launch_bounds__(256) void dec_wide_scan_kernel(
    const hufd_dec_item *items, const u32 *the_item, u8 *block, hufd_dec_item_state *states, hufd_dec_result *results) {

    const u32 item = the_item[0];
    const hufd_dec_item it = items[item];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    u32 *ctl = reinterpret_cast<u32 *>(block + lay.ctl);
    const u32 l = threadIdx.x;
    /* the launch after which nothing moved (launch j ran if launches 1 .. j - 1 all saw something move) */
    u32 settled = 0;
    for (u32 j = 1; j <= kWideFixes && !settled; ++j) {
        if (ctl[kWideMoved + j] == 0) {
            settled = j;
        }
    }
    if (!settled) {
        if (l == 0) {
            ctl[kWideGaveUp] = 1;
        }
        return;
    }
    const u32 *count = reinterpret_cast<const u32 *>(block + lay.count);
    u64 *base = reinterpret_cast<u64 *>(block + lay.base);
    u64 *part = reinterpret_cast<u64 *>(dyn_lds); /* [256] */
    const u64 stop_block = ctl[kWideStops + settled]; /* blocks behind it are not part of the stream */
    const u64 counted = stop_block < n_blocks ? stop_block + 1 : n_blocks;
    const u64 per = (counted + 255) / 256, lo = l * per < counted ? l * per : counted, hi = lo + per < counted ? lo + per : counted;
    u64 sum = 0;
    for (u64 k = lo; k < hi; ++k) {
        sum += count[k];
    }
    part[l] = sum;
    __syncthreads();
    for (u32 d = 1; d < 256; d *= 2) {
        const u64 add = l >= d ? part[l - d] : 0;
        __syncthreads();
        part[l] += add;
        __syncthreads();
    }
    u64 run = part[l] - sum;
    for (u64 k = lo; k < hi; ++k) {
        base[k] = run;
        run += count[k];
    }
    if (l == 255) {
        const u64 total = part[255];
        hufd_dec_result rs;
        rs.total_symbols = total;
        rs.cap_bit = kNoBit;
        rs.reserved = 0;
        /* the lane that stops fills these in; none does: the last code ended on the last bit of the stream */
        rs.stop_kind = stop_block < n_blocks ? HUFD_STOP_NONE : HUFD_STOP_END;
        rs.stop_bit = stop_block < n_blocks ? kNoBit : it.in_len * 8;
        results[item] = rs;
        states[item].total_symbols = total;
        ctl[kWideStopBlock] = (u32)(stop_block < n_blocks ? stop_block : 0xFFFFFFFFu);
    }
}

__global__ __launch_bounds__(kDeepThreads) void dec_wide_emit_kernel(
    hufd_tables tb, const hufd_dec_item *items, const u32 *the_item, const u8 *d_in, u8 *d_out, u8 *block, hufd_dec_result *results) {

    const u32 item = the_item[0];
    const hufd_dec_item it = items[item];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    const u32 *ctl = reinterpret_cast<const u32 *>(block + lay.ctl);
    const u64 b = blockIdx.x;
    if (ctl[kWideGaveUp] || b > ctl[kWideStopBlock]) {
        return;
    }
    const u32 l = threadIdx.x;
    wide_shared &sh = *reinterpret_cast<wide_shared *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds + sizeof(wide_shared));
    for (u32 i = l; i < tb.deep_entries; i += kDeepThreads) {
        deep[i] = tb.deep_lut[i];
    }
    const bool reached = l <= reinterpret_cast<const u32 *>(block + lay.last)[b] &&
                         (b * kDeepThreads + l) * (u64)kDeepLaneBytes < it.in_len;
    const u32 start = (block + lay.lane_start + b * kDeepThreads)[l];
    const u32 mine = reached ? (reinterpret_cast<const u16 *>(block + lay.lane_count) + b * kDeepThreads)[l] : 0u;
    sh.scan[l] = mine;
    __syncthreads();
    for (u32 d = 1; d < kDeepThreads; d *= 2) {
        const u32 add = l >= d ? sh.scan[l - d] : 0u;
        __syncthreads();
        sh.scan[l] += add;
        __syncthreads();
    }
    if (!reached) {
        return;
    }
    const u64 first = reinterpret_cast<const u64 *>(block + lay.base)[b] + (sh.scan[l] - mine);
    const u64 lane_from = (b * kDeepThreads + l) * (u64)kDeepLaneBytes * 8, lane_to = lane_from + kDeepLaneBytes * 8;
    u64 cap_bit = kNoBit;
    const deep_walked r = deep_walk<true>(
        deep, nullptr, 0, d_in + it.in_off, it.in_len, lane_from + start, lane_to, d_out + it.out_off, first, it.out_cap, &cap_bit);
    if (cap_bit != kNoBit) {
        results[item].cap_bit = cap_bit;
    }
    if (r.why != HUFD_STOP_NONE) {
        results[item].stop_kind = r.why;
        results[item].stop_bit = r.pos;
    }
}

/*
 * The road for an item dec_wide_scan gave up: a stream whose walks never fall into step (code lengths that share a
 * divisor: 9, 12 and 15 bits -- three phases, each of them valid for ever; an adversary builds one from HPACK's
 * even-length codes alone).  There a block's exit is a FUNCTION of its entry, and settling sends the truth one block a
 * launch.  So the functions are computed and composed (the generator's decision tree makes every code length equally
 * cheap, source/huffman_generator/generator.c:154-214; this is what keeps every input on the whole chip here):
 *   dec_wide_fn        a lane's function: from each of the entry bits 0 .. max_bits - 1 a short walk over the lane's
 *                      first 64 bits (walks that will ever meet have mostly met by then), and from each DISTINCT bit
 *                      these land on one walk to the end of the lane -- as many long walks as the stream has phases
 *                      (three, in the example), not 32.  Exits to memory (a byte an entry and lane: a quarter of the
 *                      block's own size), counts stay in LDS for the fold over the block's lanes: the block's function.
 *   dec_wide_fn_scan   one workgroup: the blocks' functions composed in two levels (a thread a run of blocks, then the
 *                      256 runs in turn, then every run again from its true entry): each block's true entry bit, the
 *                      symbols in front of it, the first block whose walk stops, the item's result record.
 *   dec_wide_fn_emit   a block's lanes get their entries from the block's (one thread follows the stored exits), count
 *                      their own symbols with one walk and write them with a second.
 * All three look at ctl first and do nothing for an item dec_wide_settle settled.
 */
struct wide_fn_shared {
    u8 exit_of[kWideEntries][kDeepThreads];
    u16 count_of[kWideEntries][kDeepThreads];
    u32 scan[kDeepThreads];
    u32 last_lane;
    u32 pad[3];
};

__global__ __launch_bounds__(kDeepThreads) void dec_wide_fn_kernel(
    hufd_tables tb, const hufd_dec_item *items, const u32 *the_item, const u8 *d_in, u8 *block) {

    const hufd_dec_item it = items[the_item[0]];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    const u32 *ctl = reinterpret_cast<const u32 *>(block + lay.ctl);
    if (!ctl[kWideGaveUp]) {
        return;
    }
    wide_fn_shared &sh = *reinterpret_cast<wide_fn_shared *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds + sizeof(wide_fn_shared));
    const u64 b = blockIdx.x;
    const u32 l = threadIdx.x;
    for (u32 i = l; i < tb.deep_entries; i += kDeepThreads) {
        deep[i] = tb.deep_lut[i];
    }
    __syncthreads();
    const u8 *in = d_in + it.in_off;
    const u64 block_from = b * kDeepThreads * kDeepLaneBytes;
    const u64 block_bytes = it.in_len - block_from < (u64)kDeepThreads * kDeepLaneBytes ? it.in_len - block_from : (u64)kDeepThreads * kDeepLaneBytes;
    const u32 n_lanes = (u32)((block_bytes + kDeepLaneBytes - 1) / kDeepLaneBytes);
    const u32 n_entries = tb.max_bits < kWideEntries ? tb.max_bits : kWideEntries;
    const u64 lane_from = (block_from + (u64)l * kDeepLaneBytes) * 8, lane_to = lane_from + kDeepLaneBytes * 8;
    constexpr u32 kShortBits = 64;
    if (l < n_lanes) {
        /* short walks: where the walk entered at bit e stands once it is past the lane's first 64 bits */
        u32 landed = 0; /* bit o: some walk stands o bits past them */
        for (u32 e = 0; e < n_entries; ++e) {
            const deep_walked r = deep_walk<true>(deep, nullptr, 0, in, it.in_len, lane_from + e, lane_from + kShortBits, nullptr, 0, 0, nullptr);
            const bool on = r.why == HUFD_STOP_NONE;
            const u32 o = on ? (u32)(r.pos - (lane_from + kShortBits)) : 0u;
            sh.exit_of[e][l] = (u8)(on ? o : kWideStop);
            sh.count_of[e][l] = (u16)r.count;
            landed |= on ? 1u << o : 0u;
        }
        /* one long walk from every bit a walk landed on; the entries that landed there take its exit and add its count.
         * (An entry's record holds its landing bit until its long walk is done: the bits are taken in rising order and a
         * record that is through is marked in `done`, so an exit is never taken for a landing bit.) */
        u32 done = 0;
        for (u32 e = 0; e < n_entries; ++e) {
            done |= sh.exit_of[e][l] == kWideStop ? 1u << e : 0u;
        }
        while (landed) {
            const u32 o = (u32)__builtin_ctz(landed);
            landed &= landed - 1;
            const deep_walked r = deep_walk<true>(deep, nullptr, 0, in, it.in_len, lane_from + kShortBits + o, lane_to, nullptr, 0, 0, nullptr);
            const u32 ex = r.why == HUFD_STOP_NONE ? (u32)(r.pos - lane_to) : kWideStop;
            for (u32 e = 0; e < n_entries; ++e) {
                if (!((done >> e) & 1u) && sh.exit_of[e][l] == o) {
                    sh.exit_of[e][l] = (u8)ex;
                    sh.count_of[e][l] = (u16)(sh.count_of[e][l] + r.count);
                    done |= 1u << e;
                }
            }
        }
    }
    for (u32 e = (l < n_lanes ? n_entries : 0u); e < kWideEntries; ++e) {
        sh.exit_of[e][l] = (u8)kWideStop; /* (no code start there, no lane there: never looked at as an entry that goes on) */
        sh.count_of[e][l] = 0;
    }
    __syncthreads();
    u8 *fn_exit = block + lay.fn_exit + b * (u64)kWideEntries * kDeepThreads;
    for (u32 e = 0; e < kWideEntries; ++e) {
        fn_exit[e * kDeepThreads + l] = sh.exit_of[e][l];
    }
    /* the block's function: thread e follows entry e through the lanes */
    if (l < kWideEntries) {
        u32 at = l < n_entries ? l : kWideStop;
        u64 symbols = 0;
        for (u32 k = 0; k < n_lanes && at != kWideStop; ++k) {
            symbols += sh.count_of[at][k];
            at = sh.exit_of[at][k];
        }
        reinterpret_cast<u64 *>(block + lay.fn_block)[b * kWideEntries + l] = (symbols << 8) | at;
    }
}

constexpr u32 kWideFnScanThreads = 256;
struct wide_fn_scan_shared {
    u64 count_of[kWideFnScanThreads][kWideEntries + 1]; /* (+ 1: the threads' rows start in different banks) */
    u8 exit_of[kWideFnScanThreads][kWideEntries];
    u32 seg_entry[kWideFnScanThreads];
    u64 seg_base[kWideFnScanThreads];
    u64 stop_block;
    u64 total;
};

__global__ __launch_bounds__(kWideFnScanThreads) void dec_wide_fn_scan_kernel(
    const hufd_dec_item *items, const u32 *the_item, u8 *block, hufd_dec_item_state *states, hufd_dec_result *results, u32 fails) {

    const u32 item = the_item[0];
    const hufd_dec_item it = items[item];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    u32 *ctl = reinterpret_cast<u32 *>(block + lay.ctl);
    if (!ctl[kWideGaveUp] || fails >= 2) {
        return; /* (fails >= 2: this road gives the item up as well, for the test of dec_deep behind it) */
    }
    wide_fn_scan_shared &sh = *reinterpret_cast<wide_fn_scan_shared *>(dyn_lds);
    const u32 t = threadIdx.x;
    const u64 *fn_block = reinterpret_cast<const u64 *>(block + lay.fn_block);
    u32 *fn_entry = reinterpret_cast<u32 *>(block + lay.fn_entry);
    u64 *base = reinterpret_cast<u64 *>(block + lay.base);
    const u64 per = (n_blocks + kWideFnScanThreads - 1) / kWideFnScanThreads;
    const u64 lo = t * per < n_blocks ? t * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
    /* my run of blocks as a function of the bit it is entered at */
    for (u32 e = 0; e < kWideEntries; ++e) {
        u32 at = e;
        u64 symbols = 0;
        for (u64 k = lo; k < hi && at != kWideStop; ++k) {
            const u64 f = fn_block[k * kWideEntries + at];
            symbols += f >> 8;
            at = (u32)(f & 0xFFu);
        }
        sh.exit_of[t][e] = (u8)at;
        sh.count_of[t][e] = symbols;
    }
    if (t == 0) {
        sh.stop_block = n_blocks;
    }
    __syncthreads();
    if (t == 0) {
        u32 at = it.first_bit;
        u64 symbols = 0;
        for (u32 k = 0; k < kWideFnScanThreads; ++k) {
            sh.seg_entry[k] = at;
            sh.seg_base[k] = symbols;
            if (at != kWideStop) {
                symbols += sh.count_of[k][at];
                at = sh.exit_of[k][at];
            }
        }
        sh.total = symbols;
    }
    __syncthreads();
    {
        u32 at = sh.seg_entry[t];
        u64 symbols = sh.seg_base[t];
        for (u64 k = lo; k < hi; ++k) {
            fn_entry[k] = at;
            base[k] = symbols;
            if (at != kWideStop) {
                const u64 f = fn_block[k * kWideEntries + at];
                symbols += f >> 8;
                at = (u32)(f & 0xFFu);
                if (at == kWideStop) {
                    sh.stop_block = k; /* (one thread at most: behind the stop every entry is "not part of the stream") */
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        const u64 stop_block = sh.stop_block;
        hufd_dec_result rs;
        rs.total_symbols = sh.total;
        rs.cap_bit = kNoBit;
        rs.reserved = 0;
        /* the lane that stops fills these in; none does: the last code ended on the last bit of the stream */
        rs.stop_kind = stop_block < n_blocks ? HUFD_STOP_NONE : HUFD_STOP_END;
        rs.stop_bit = stop_block < n_blocks ? kNoBit : it.in_len * 8;
        results[item] = rs;
        states[item].total_symbols = sh.total;
        ctl[kWideStopBlock] = (u32)(stop_block < n_blocks ? stop_block : 0xFFFFFFFFu);
        ctl[kWideFnRoad] = 1;
        ctl[kWideGaveUp] = 0; /* dec_deep, queued behind this road with that word as its gate, stays out of it */
    }
}

__global__ __launch_bounds__(kDeepThreads) void dec_wide_fn_emit_kernel(
    hufd_tables tb, const hufd_dec_item *items, const u32 *the_item, const u8 *d_in, u8 *d_out, u8 *block, hufd_dec_result *results) {

    const u32 item = the_item[0];
    const hufd_dec_item it = items[item];
    const u64 n_blocks = (it.in_len + (u64)kDeepThreads * kDeepLaneBytes - 1) / ((u64)kDeepThreads * kDeepLaneBytes);
    const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
    const u32 *ctl = reinterpret_cast<const u32 *>(block + lay.ctl);
    const u64 b = blockIdx.x;
    if (!ctl[kWideFnRoad] || b > ctl[kWideStopBlock]) {
        return;
    }
    wide_fn_shared &sh = *reinterpret_cast<wide_fn_shared *>(dyn_lds);
    u32 *deep = reinterpret_cast<u32 *>(dyn_lds + sizeof(wide_fn_shared));
    const u32 l = threadIdx.x;
    for (u32 i = l; i < tb.deep_entries; i += kDeepThreads) {
        deep[i] = tb.deep_lut[i];
    }
    const u8 *fn_exit = block + lay.fn_exit + b * (u64)kWideEntries * kDeepThreads;
    for (u32 e = 0; e < kWideEntries; ++e) {
        sh.exit_of[e][l] = fn_exit[e * kDeepThreads + l];
    }
    const u64 block_from = b * kDeepThreads * kDeepLaneBytes;
    const u64 block_bytes = it.in_len - block_from < (u64)kDeepThreads * kDeepLaneBytes ? it.in_len - block_from : (u64)kDeepThreads * kDeepLaneBytes;
    const u32 n_lanes = (u32)((block_bytes + kDeepLaneBytes - 1) / kDeepLaneBytes);
    __syncthreads();
    /* the lanes' entries, from the block's: one thread follows the exits (sh.scan holds them for a moment) */
    if (l == 0) {
        u32 at = reinterpret_cast<const u32 *>(block + lay.fn_entry)[b];
        u32 last = kDeepThreads;
        for (u32 k = 0; k < n_lanes; ++k) {
            sh.scan[k] = at;
            at = sh.exit_of[at][k];
            if (at == kWideStop) {
                last = k; /* its walk stops: the lanes behind it are not part of the stream */
                break;
            }
        }
        sh.last_lane = last;
    }
    __syncthreads();
    const bool reached = l < n_lanes && l <= sh.last_lane;
    const u32 start = reached ? sh.scan[l] : 0u;
    __syncthreads();
    const u8 *in = d_in + it.in_off;
    const u64 lane_from = (block_from + (u64)l * kDeepLaneBytes) * 8, lane_to = lane_from + kDeepLaneBytes * 8;
    u32 mine = 0;
    if (reached) {
        mine = deep_walk<true>(deep, nullptr, 0, in, it.in_len, lane_from + start, lane_to, nullptr, 0, 0, nullptr).count;
    }
    sh.scan[l] = mine;
    __syncthreads();
    for (u32 d = 1; d < kDeepThreads; d *= 2) {
        const u32 add = l >= d ? sh.scan[l - d] : 0u;
        __syncthreads();
        sh.scan[l] += add;
        __syncthreads();
    }
    if (!reached) {
        return;
    }
    const u64 first = reinterpret_cast<const u64 *>(block + lay.base)[b] + (sh.scan[l] - mine);
    u64 cap_bit = kNoBit;
    const deep_walked r = deep_walk<true>(deep, nullptr, 0, in, it.in_len, lane_from + start, lane_to, d_out + it.out_off, first, it.out_cap, &cap_bit);
    if (cap_bit != kNoBit) {
        results[item].cap_bit = cap_bit;
    }
    if (r.why != HUFD_STOP_NONE) {
        results[item].stop_kind = r.why;
        results[item].stop_bit = r.pos;
    }
}

/*
 * A coder whose codes all have ONE length L (tables.fixed_bits): symbol k of an item starts at bit first_bit + k L.
 * Nothing has to be found -- and the walks of the chunked decoder never fall into step on such a stream (L phases,
 * every one of them valid for ever), which sends every chunk the long way at a twentieth of the speed.  Items beyond a
 * thread's work are taken 16 KiB a workgroup, 64 bytes a lane, in three launches:
 *   dec_fixed_check   the first symbol without a code, if there is one (source/huffman.c:240-247): a minimum per item;
 *   dec_fixed_finish  a thread per item: how many symbols, where and why the stream stops (:232-255), the start bit of
 *                     symbol number out_cap (:257-268);
 *   dec_fixed_emit    the symbols in front of all that.
 */
constexpr u32 kFixedThreads = 256;
constexpr u32 kFixedLaneBytes = HUFD_FIXED_BLOCK_BYTES / kFixedThreads;

/* the symbols whose codes START in the lane's bytes and lie wholly inside the stream: [k0, k1) */
struct fixed_span {
    u64 k0, k1;
};
__device__ __forceinline__ fixed_span fixed_span_of(const hufd_dec_item &it, u32 L, u64 from_byte, u64 to_byte) {
    const u64 rem = it.in_len * 8, n_full = rem > it.first_bit ? (rem - it.first_bit) / L : 0;
    const u64 from = from_byte * 8, to = to_byte * 8;
    fixed_span s;
    s.k0 = from <= it.first_bit ? 0 : (from - it.first_bit + L - 1) / L;
    s.k1 = to <= it.first_bit ? 0 : (to - it.first_bit + L - 1) / L;
    s.k0 = s.k0 < n_full ? s.k0 : n_full;
    s.k1 = s.k1 < n_full ? s.k1 : n_full;
    return s;
}

template <bool EMIT>
__global__ __launch_bounds__(kFixedThreads) void dec_fixed_kernel(
    hufd_tables tb, const hufd_dec_item *items, const u32 *blocks, const u8 *d_in, u8 *d_out, hufd_dec_item_state *states) {

    u16 *lut = reinterpret_cast<u16 *>(dyn_lds);
    for (u32 i = threadIdx.x; i < (1u << tb.lut_bits); i += kFixedThreads) {
        lut[i] = tb.dec_lut[i];
    }
    __syncthreads();
    const u32 item = blocks[2 * blockIdx.x];
    const hufd_dec_item it = items[item];
    const u32 L = tb.fixed_bits;
    const u64 from = (u64)blocks[2 * blockIdx.x + 1] * HUFD_FIXED_BLOCK_BYTES + (u64)threadIdx.x * kFixedLaneBytes;
    if (from >= it.in_len) {
        return;
    }
    const u64 to = from + kFixedLaneBytes < it.in_len ? from + kFixedLaneBytes : it.in_len;
    fixed_span sp = fixed_span_of(it, L, from, to);
    if (EMIT) {
        /* what dec_fixed_finish left: the symbols the stream holds; those with room are written */
        const u64 total = states[item].total_symbols, limit = total < it.out_cap ? total : it.out_cap;
        sp.k1 = sp.k1 < limit ? sp.k1 : limit;
    }
    if (sp.k0 >= sp.k1) {
        return;
    }
    const u64 pos = it.first_bit + sp.k0 * L;
    stream_reader sr;
    sr.start(d_in + it.in_off + (pos >> 3), it.in_len - (pos >> 3), (u32)(pos & 7));
    symbol_sink sink;
    sink.begin(EMIT ? d_out + it.out_off + sp.k0 : nullptr);
    for (u64 k = sp.k0; k < sp.k1; ++k) {
        const u32 entry = lut[sr.peek() >> (32 - tb.lut_bits)];
        if (EMIT) {
            sink.put(entry >> 8);
        } else if ((entry & 0xFFu) == 0) {
            atomicMin(reinterpret_cast<unsigned long long *>(&states[item].total_symbols), (unsigned long long)k);
            break;
        }
        sr.skip(L);
    }
    if (EMIT) {
        sink.flush();
    }
}

__global__ __launch_bounds__(256) void dec_fixed_finish_kernel(
    hufd_tables tb, const hufd_dec_item *items, u32 n_items, const u8 *d_in, hufd_dec_item_state *states, hufd_dec_result *results) {

    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_dec_item it = items[i];
    if (it.tiny != 2) {
        return;
    }
    const u32 L = tb.fixed_bits;
    const u64 rem = it.in_len * 8, n_full = rem > it.first_bit ? (rem - it.first_bit) / L : 0;
    const u64 first_bad = states[i].total_symbols; /* dec_fixed_check's minimum, all ones if every code is one */
    hufd_dec_result rs;
    rs.reserved = 0;
    if (first_bad < n_full) {
        rs.total_symbols = first_bad;
        rs.stop_kind = HUFD_STOP_INVALID;
        rs.stop_bit = it.first_bit + first_bad * L;
    } else {
        rs.total_symbols = n_full;
        const u64 pos = it.first_bit + n_full * L;
        rs.stop_bit = pos;
        if (pos >= rem) {
            rs.stop_kind = HUFD_STOP_END;
        } else {
            /* fewer than L bits left: a window without a code, or a code cut off (source/huffman.c:232-255, in that order) */
            stream_reader sr;
            sr.start(d_in + it.in_off + (pos >> 3), it.in_len - (pos >> 3), (u32)(pos & 7));
            const u32 entry = tb.dec_lut[sr.peek() >> (32 - tb.lut_bits)];
            rs.stop_kind = (entry & 0xFFu) == 0 ? HUFD_STOP_INVALID : HUFD_STOP_INCOMPLETE;
        }
    }
    rs.cap_bit = rs.total_symbols > it.out_cap ? it.first_bit + it.out_cap * L : kNoBit;
    results[i] = rs;
    states[i].total_symbols = rs.total_symbols;
}

/*
 * One host-pointer call of more than one thread's bytes and up to HUFD_DEC_BLOCK_MAX_BYTES of them (short codes),
 * HUFD_DEC_BLOCK_BYTES a turn: ONE workgroup and one launch, as enc_block is for the encoder -- a chunk's tables, lists and five more launches cost
 * such a call several times its symbols.  A lane takes 64 bits of the stream and keeps them, with the 32 behind them,
 * in registers (big-endian words, zeros behind the stream's end).  As in dec_deep the lanes settle on their
 * entries by walking again from where the lane in front really leaves until nothing changes -- exact whatever the
 * stream does, lane 0's entry being the true one; after kBlockDecRounds rounds it gives the call back instead.  What keeps that to a handful of cheap rounds: a lane remembers the
 * code starts of its last walk (a bit each), and a walk from another entry ends where it meets one of them -- walks
 * from different entries fall into step within a few codes -- so after the first round a lane's walk is two or three
 * codes, and one that leaves its lane as before stops the news from travelling on.  Then the scan of the counts and
 * the walk that writes the symbols; every walk is source/huffman.c:232-268 a code at a time, stops included.
 */
constexpr u32 kBlockDecThreads = 1024;
constexpr u32 kBlockDecLaneBits = 64;
constexpr u32 kBlockDecWaves = kBlockDecThreads / 64;
constexpr u32 kBlockDecRounds = 24; /* (the test coder's streams settle in 3 to 7) */
static_assert(HUFD_DEC_BLOCK_BYTES * 8 == kBlockDecThreads * kBlockDecLaneBits, "a lane for every 64 bits of a turn");

struct block_dec_shared {
    u8 exit_of[kBlockDecThreads];
    u32 wave_total[kBlockDecWaves];
    u32 changed[2]; /* a lane walks again: rounds take turns with the two */
    u32 last_lane;  /* the first lane whose walk from its true entry stops */
    u32 stop_kind;
    u64 stop_bit;
    u64 cap_bit;
};

/* a lane's 64 bits and the 32 behind them */
struct block_dec_bits {
    u32 w0, w1, w2;
    /* the 32 bits from bit `rel` (< 64) of the lane on */
    __device__ __forceinline__ u32 window(u32 rel) const {
        const u32 hi = rel & 32u ? w1 : w0, lo = rel & 32u ? w2 : w1;
        return (u32)((((u64)hi << 32) | lo) >> (32 - (rel & 31u)));
    }
};

/* what a lane knows of its last walk: the code starts in its bits (`seen`: they form one chain, each leads to the
 * next), how many there are, and how the chain leaves the lane (kDeepStop: it stops inside) */
struct block_dec_chain {
    u64 seen;
    u32 count, exit;
};

/* Walks from bit `rel` of the lane until it meets the chain of the walk before or leaves the lane, and makes that the
 * chain.  GUESS: a walk from anywhere, only to find a chain to meet: it steps over a window without a code a bit at a
 * time and forgets what it saw in front of it (a walk that gets there from a real entry stops there). */
template <bool GUESS>
__device__ __forceinline__ void block_dec_count(
    const u16 *lut, u32 lut_bits, const block_dec_bits &bits, u32 lane_from, u32 rem, u32 rel, block_dec_chain &c) {
    u64 fresh = 0;
    u32 count = 0, why = HUFD_STOP_NONE;
    bool met = false;
    while (rel < kBlockDecLaneBits) {
        if ((c.seen >> rel) & 1u) {
            met = true;
            count += (u32)__popcll(c.seen >> rel);
            fresh |= c.seen >> rel << rel;
            break;
        }
        if (lane_from + rel >= rem) {
            why = HUFD_STOP_END;
            break;
        }
        const u32 len = lut[bits.window(rel) >> (32 - lut_bits)] & 0xFFu;
        if (len == 0) {
            if (GUESS) {
                ++rel;
                fresh = 0;
                count = 0;
                continue;
            }
            why = HUFD_STOP_INVALID;
            break;
        }
        if (lane_from + rel + len > rem) {
            why = HUFD_STOP_INCOMPLETE;
            break;
        }
        fresh |= 1ull << rel;
        ++count;
        rel += len;
    }
    c.seen = fresh;
    c.count = count;
    if (!met) {
        c.exit = why == HUFD_STOP_NONE ? rel - kBlockDecLaneBits : kDeepStop;
    }
}

__global__ __launch_bounds__(kBlockDecThreads) void dec_block_kernel(
    hufd_tables tb,
    hufd_dec_item it, /* (by value: a record in memory is one more round trip before the stream's first byte) */
    const u8 *d_in,
    u8 *d_out,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {

    block_dec_shared &sh = *reinterpret_cast<block_dec_shared *>(dyn_lds);
    u16 *lut = reinterpret_cast<u16 *>(dyn_lds + sizeof(block_dec_shared));
    const u32 l = threadIdx.x, lane = l & 63u, wave = l >> 6;
    const u32 in_len = (u32)it.in_len; /* <= HUFD_DEC_BLOCK_MAX_BYTES: the launch's side of the bargain */
    const u32 rem = in_len * 8;
    const u8 *first = d_in + it.in_off;
    const u32 lead = (u32)(reinterpret_cast<uintptr_t>(first) & 3u);
    const u32 *words = reinterpret_cast<const u32 *>(first - lead);
    const u32 mem_words = (lead + in_len + 3) / 4; /* the aligned words that hold bytes of the stream */
    {
        /* the table, two entries a lane-load (the first turn's bits are asked for before these are waited for) */
        const u32 *lut_words = reinterpret_cast<const u32 *>(tb.dec_lut);
        u32 *lut_lds = reinterpret_cast<u32 *>(lut);
        const u32 lut_pairs = (1u << tb.lut_bits) / 2;
        static_assert((1u << HUFD_DEC_MAX_LUT_BITS) / 2 <= 2 * kBlockDecThreads, "two loads a lane hold the longest table");
        if (l < lut_pairs) {
            lut_lds[l] = lut_words[l];
        }
        if (l + kBlockDecThreads < lut_pairs) {
            lut_lds[l + kBlockDecThreads] = lut_words[l + kBlockDecThreads];
        }
    }
    if (l == 0) {
        sh.stop_kind = HUFD_STOP_NONE;
        sh.stop_bit = kNoBit;
        sh.cap_bit = kNoBit;
    }
    /* HUFD_DEC_BLOCK_BYTES a turn; a turn's lane 0 is entered the way the turn before is left */
    u32 symbols = 0, carry = it.first_bit;
    const u32 turns = (in_len + HUFD_DEC_BLOCK_BYTES - 1) / HUFD_DEC_BLOCK_BYTES;
    for (u32 turn = 0; turn < turns; ++turn) {
        const u32 turn_from = turn * HUFD_DEC_BLOCK_BYTES * 8;
        const u32 n_lanes = rem - turn_from < kBlockDecThreads * kBlockDecLaneBits
                                ? (rem - turn_from + kBlockDecLaneBits - 1) / kBlockDecLaneBits : kBlockDecThreads;
        const bool active = l < n_lanes;
        const u32 lane_from = turn_from + l * kBlockDecLaneBits;
        /* the lane's own bits and the 32 behind them, and the lane in front's for the guess, out of six aligned words
         * (the stream's first byte sits anywhere): one trip to memory */
        block_dec_bits bits, front;
        {
            const u32 w_first = lane_from / 32; /* the lane's first stream word */
            u32 m[6];
#pragma unroll
            for (u32 k = 0; k < 6; ++k) {
                m[k] = active && w_first + k >= 2 && w_first + k - 2 < mem_words ? words[w_first + k - 2] : 0u;
            }
            u32 w[5];
#pragma unroll
            for (u32 k = 0; k < 5; ++k) {
                const u32 i = w_first + k - 2; /* stream word i: bytes 4 i .. 4 i + 3, those behind the stream's end read as zero */
                const u32 raw = (u32)((((u64)m[k + 1] << 32) | m[k]) >> (8 * lead));
                const u32 have = w_first + k >= 2 && 4 * i < in_len ? (in_len - 4 * i < 4 ? in_len - 4 * i : 4u) : 0u;
                const u32 big = __builtin_bswap32(raw);
                w[k] = have == 4 ? big : (have ? big & (~0u << (8 * (4 - have))) : 0u);
            }
            front.w0 = w[0];
            front.w1 = w[1];
            front.w2 = w[2];
            bits.w0 = w[2];
            bits.w1 = w[3];
            bits.w2 = w[4];
        }
        if (l == 0) {
            sh.last_lane = kBlockDecThreads;
        }
        __syncthreads(); /* (first turn: the table is in LDS) */
        block_dec_chain chain = {0, 0, kDeepStop};
        u32 start = l == 0 ? carry : 0u;
        if (active) {
            if (l == 0) {
                block_dec_count<false>(lut, tb.lut_bits, bits, lane_from, rem, start, chain);
            } else {
                /* the first guess: how a walk from anywhere leaves the lane in front (lane 1's: the true walk) */
                block_dec_chain guess = {0, 0, kDeepStop};
                if (l == 1) {
                    block_dec_count<false>(lut, tb.lut_bits, front, lane_from - kBlockDecLaneBits, rem, carry, guess);
                } else {
                    block_dec_count<true>(lut, tb.lut_bits, front, lane_from - kBlockDecLaneBits, rem, 0u, guess);
                }
                if (guess.exit != kDeepStop) {
                    start = guess.exit;
                    block_dec_count<false>(lut, tb.lut_bits, bits, lane_from, rem, start, chain);
                } else {
                    block_dec_count<true>(lut, tb.lut_bits, bits, lane_from, rem, 0u, chain);
                    start = kDeepStop; /* (no entry yet: whatever the lane in front says first is news) */
                }
            }
        }
        /* Settling: a lane whose entry is not how the lane in front leaves walks again from there.  A walk that stops
         * says nothing to the lane behind it (from a wrong entry a window without a code is nothing special): that one
         * keeps what it has.  When nothing changes any more, lane 0 has the true entry, so has every lane up to the first
         * whose walk stops -- there the stream stops (source/huffman.c:232-255) -- and the lanes behind that one are not
         * part of it.  (A lane that never heard from the one in front is behind such a lane.) */
        for (u32 round = 0;; ++round) {
            sh.exit_of[l] = (u8)(active ? chain.exit : kDeepStop);
            if (l == 0) {
                sh.changed[round & 1u] = 0; /* (the flag of the round before last: everyone has read it) */
            }
            __syncthreads();
            if (active && l > 0) {
                const u32 prev = sh.exit_of[l - 1];
                if (prev != kDeepStop && prev != start) {
                    start = prev;
                    block_dec_count<false>(lut, tb.lut_bits, bits, lane_from, rem, start, chain);
                    sh.changed[round & 1u] = 1;
                }
            }
            __syncthreads();
            if (!sh.changed[round & 1u]) {
                break;
            }
            if (round == kBlockDecRounds) {
                /* a stream whose walks do not fall into step (codes of one length, say): the news travels a lane a
                 * round, and the chunk kernels' transfer functions are the better tool */
                if (l == 0) {
                    hufd_dec_result rs = {};
                    rs.stop_kind = HUFD_STOP_GAVE_UP;
                    results[0] = rs;
                }
                return;
            }
        }
        if (active && chain.exit == kDeepStop) {
            atomicMin(&sh.last_lane, l);
        }
        __syncthreads();
        const u32 last_lane = sh.last_lane;
        const bool reached = active && l <= last_lane;
        /* where each lane's symbols go: an exclusive scan of the counts of the lanes on the true path */
        const u32 mine = reached ? chain.count : 0u;
        const u32 upto = wave_inclusive_sum(mine, lane);
        if (lane == 63) {
            sh.wave_total[wave] = upto;
        }
        __syncthreads();
        u32 before = symbols + upto - mine;
        for (u32 w = 0; w < kBlockDecWaves; ++w) {
            const u32 t = sh.wave_total[w];
            before += w < wave ? t : 0u;
            symbols += t;
        }
        if (reached) {
            /* the symbols of the lane's chain, and what stopped it if something did */
            u8 *out = d_out + it.out_off;
            u32 rel = start, k = before, why = HUFD_STOP_NONE;
            while (rel < kBlockDecLaneBits) {
                if (lane_from + rel >= rem) {
                    why = HUFD_STOP_END;
                    break;
                }
                const u32 entry = lut[bits.window(rel) >> (32 - tb.lut_bits)];
                const u32 len = entry & 0xFFu;
                if (len == 0) {
                    why = HUFD_STOP_INVALID;
                    break;
                }
                if (lane_from + rel + len > rem) {
                    why = HUFD_STOP_INCOMPLETE;
                    break;
                }
                if (k < it.out_cap) {
                    out[k] = (u8)(entry >> 8);
                } else if (k == it.out_cap) {
                    sh.cap_bit = lane_from + rel; /* source/huffman.c:257-268: this symbol is not consumed */
                }
                ++k;
                rel += len;
            }
            if (why != HUFD_STOP_NONE) {
                sh.stop_kind = why;
                sh.stop_bit = lane_from + rel;
            }
        }
        if (last_lane < n_lanes) {
            break; /* the stream stops in this turn */
        }
        carry = sh.exit_of[n_lanes - 1];
        __syncthreads(); /* the next turn writes what this one's lanes have just read */
    }
    __syncthreads();
    if (l == 0) {
        hufd_dec_result rs;
        rs.total_symbols = symbols;
        rs.cap_bit = sh.cap_bit;
        rs.reserved = 0;
        if (sh.stop_kind != HUFD_STOP_NONE) {
            rs.stop_kind = sh.stop_kind;
            rs.stop_bit = sh.stop_bit;
        } else {
            /* the last code ended on the last bit of the stream */
            rs.stop_kind = HUFD_STOP_END;
            rs.stop_bit = rem;
        }
        results[0] = rs;
        states[0].total_symbols = symbols;
    }
}


} /* namespace */

using hufk_host::persistent_grid;
using hufk_host::stage_mark;
using hufk_host::current_compute_units;

hipError_t hufk_host::init_decode_items(int lds_max) {
    hipError_t e = hipSuccess;
    const void *kernels[] = {
        reinterpret_cast<const void *>(&dec_wide_fn_kernel), reinterpret_cast<const void *>(&dec_wide_fn_scan_kernel),
        reinterpret_cast<const void *>(&dec_wide_fn_emit_kernel)};
    for (const void *k : kernels) {
        if (e == hipSuccess) {
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        }
    }
    return e;
}

void hufk_host::decode_items_stage(const struct hufk_decode_args *a, hipStream_t st) {
    if (a->n_tiny && a->tables.deep_entries) {
        hipLaunchKernelGGL(
            dec_tiny_kernel<true>, dim3((a->n_tiny + kTinyDecDeepThreads - 1) / kTinyDecDeepThreads), dim3(kTinyDecDeepThreads),
            a->tables.deep_entries * sizeof(u32), st, a->tables, a->items, a->tiny_items, a->n_tiny, (const u8 *)a->d_in,
            (u8 *)a->d_out, a->states, a->results);
    } else if (a->n_tiny) {
        hipLaunchKernelGGL(
            dec_tiny_kernel<false>, dim3((a->n_tiny + kTinyDecThreads - 1) / kTinyDecThreads), dim3(kTinyDecThreads),
            (1u << a->tables.lut_bits) * sizeof(u16), st, a->tables, a->items, a->tiny_items, a->n_tiny, (const u8 *)a->d_in,
            (u8 *)a->d_out, a->states, a->results);
    }
    if (a->n_deep && a->tables.deep_entries) {
        const uint32_t deep_lds = (uint32_t)(sizeof(deep_shared) + a->tables.deep_entries * sizeof(u32));
        const uint32_t wide_lds = (uint32_t)(sizeof(wide_shared) + a->tables.deep_entries * sizeof(u32));
        const uint64_t wide_from = a->n_wide ? a->wide_from : ~0ull;
        hipLaunchKernelGGL(
            dec_deep_kernel<true>, dim3(a->n_deep), dim3(kDeepThreads), deep_lds, st, a->tables, a->items, a->deep_items,
            kDeepLaneBytes, (const u8 *)a->d_in, (u8 *)a->d_out, a->states, a->results, wide_from, (const u32 *)nullptr);
        /* the long ones across the chip (dec_wide_*), each with dec_deep behind it in case they give it up */
        for (uint32_t k = 0; k < a->n_wide; ++k) {
            const u32 *the_item = a->deep_items + a->wide[k].slot;
            u8 *blk = (u8 *)a->wide_block + a->wide[k].block_offset;
            const uint32_t n_blocks = a->wide[k].n_blocks;
            const dec_wide_layout lay = dec_wide_layout_of(n_blocks);
            (void)hipMemsetAsync(blk + lay.ctl, 0, 4 * kWideCtlWords, st);
            (void)hipMemsetAsync(blk + lay.ctl + 4 * (kWideStops + 1), 0xFF, 4 * kWideFixes, st);
            hipLaunchKernelGGL(
                dec_wide_settle_kernel<true>, dim3(n_blocks), dim3(kDeepThreads), wide_lds, st, a->tables, a->items, the_item,
                (const u8 *)a->d_in, blk, 0u, 0u);
            for (u32 pass = 1; pass <= kWideFixes; ++pass) {
                hipLaunchKernelGGL(
                    dec_wide_settle_kernel<false>, dim3(n_blocks), dim3(kDeepThreads), wide_lds, st, a->tables, a->items,
                    the_item, (const u8 *)a->d_in, blk, pass, a->wide_fails);
            }
            hipLaunchKernelGGL(dec_wide_scan_kernel, dim3(1), dim3(256), 256 * sizeof(u64), st, a->items, the_item, blk, a->states, a->results);
            hipLaunchKernelGGL(
                dec_wide_emit_kernel, dim3(n_blocks), dim3(kDeepThreads), wide_lds, st, a->tables, a->items, the_item,
                (const u8 *)a->d_in, (u8 *)a->d_out, blk, a->results);
            /* an item they gave up (its walks never fall into step) by transfer functions; all three return at once otherwise */
            const uint32_t fn_lds = (uint32_t)(sizeof(wide_fn_shared) + a->tables.deep_entries * sizeof(u32));
            hipLaunchKernelGGL(
                dec_wide_fn_kernel, dim3(n_blocks), dim3(kDeepThreads), fn_lds, st, a->tables, a->items, the_item,
                (const u8 *)a->d_in, blk);
            hipLaunchKernelGGL(
                dec_wide_fn_scan_kernel, dim3(1), dim3(kWideFnScanThreads), sizeof(wide_fn_scan_shared), st, a->items, the_item, blk,
                a->states, a->results, a->wide_fails);
            hipLaunchKernelGGL(
                dec_wide_fn_emit_kernel, dim3(n_blocks), dim3(kDeepThreads), fn_lds, st, a->tables, a->items, the_item,
                (const u8 *)a->d_in, (u8 *)a->d_out, blk, a->results);
            hipLaunchKernelGGL(
                dec_deep_kernel<true>, dim3(1), dim3(kDeepThreads), deep_lds, st, a->tables, a->items, the_item, kDeepLaneBytes,
                (const u8 *)a->d_in, (u8 *)a->d_out, a->states, a->results, 0ull, (const u32 *)(blk + lay.ctl));
        }
    } else if (a->n_deep) {
        hipLaunchKernelGGL(
            dec_deep_kernel<false>, dim3(a->n_deep), dim3(kCoopThreads), sizeof(deep_shared) + (1u << a->tables.lut_bits) * sizeof(u16),
            st, a->tables, a->items, a->deep_items, 0u, (const u8 *)a->d_in, (u8 *)a->d_out, a->states, a->results, ~0ull,
            (const u32 *)nullptr);
    }
    if (a->n_fixed_blocks && a->tables.fixed_bits) {
        /* (the items' state words start as "no symbol without a code": dec_fixed_check takes a minimum in them; the other
         * items' are written by their own kernels, behind this) */
        const uint32_t lds = (1u << a->tables.lut_bits) * sizeof(u16);
        if (!a->tables.fixed_complete) {
            hipLaunchKernelGGL(
                dec_fixed_kernel<false>, dim3(a->n_fixed_blocks), dim3(kFixedThreads), lds, st, a->tables, a->items,
                a->fixed_blocks, (const u8 *)a->d_in, (u8 *)a->d_out, a->states);
        }
        hipLaunchKernelGGL(
            dec_fixed_finish_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->tables, a->items, a->n_items,
            (const u8 *)a->d_in, a->states, a->results);
        hipLaunchKernelGGL(
            dec_fixed_kernel<true>, dim3(a->n_fixed_blocks), dim3(kFixedThreads), lds, st, a->tables, a->items, a->fixed_blocks,
            (const u8 *)a->d_in, (u8 *)a->d_out, a->states);
    }
}

extern "C" {

int hufk_decode_one_tiny(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream) {
    if (tables->deep_entries) {
        hipLaunchKernelGGL(
            dec_tiny_kernel<true>, dim3(1), dim3(kTinyDecThreads), tables->deep_entries * sizeof(u32), (hipStream_t)stream,
            *tables, item, zero, 1u, (const u8 *)d_in, (u8 *)d_out, state, result);
    } else {
        hipLaunchKernelGGL(
            dec_tiny_kernel<false>, dim3(1), dim3(kTinyDecThreads), (1u << tables->lut_bits) * sizeof(u16),
            (hipStream_t)stream, *tables, item, zero, 1u, (const u8 *)d_in, (u8 *)d_out, state, result);
    }
    return (int)hipGetLastError();
}

uint64_t hufk_decode_wide_bytes(uint64_t n_blocks) {
    return (dec_wide_layout_of(n_blocks).bytes + 255) & ~255ull;
}

int hufk_decode_one_block(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream) {
    if (tables->deep_entries) {
        return (int)hipErrorInvalidValue; /* (long codes: hufk_decode_one_coop) */
    }
    hipLaunchKernelGGL(
        dec_block_kernel, dim3(1), dim3(kBlockDecThreads), (uint32_t)sizeof(block_dec_shared) + (1u << tables->lut_bits) * sizeof(u16),
        (hipStream_t)stream, *tables, *item, (const u8 *)d_in, (u8 *)d_out, state, result);
    return (int)hipGetLastError();
}

int hufk_decode_one_coop(
    const struct hufd_tables *tables, const struct hufd_dec_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_dec_item_state *state, struct hufd_dec_result *result, void *stream) {
    if (tables->deep_entries) {
        hipLaunchKernelGGL(
            dec_deep_kernel<true>, dim3(1), dim3(kDeepThreads), sizeof(deep_shared) + tables->deep_entries * sizeof(u32),
            (hipStream_t)stream, *tables, item, zero, kDeepLaneBytes, (const u8 *)d_in, (u8 *)d_out, state, result, ~0ull,
            (const u32 *)nullptr);
    } else {
        /* (one wave: the lanes share the item evenly) */
        hipLaunchKernelGGL(
            dec_deep_kernel<false>, dim3(1), dim3(kCoopThreads),
            sizeof(deep_shared) + (1u << tables->lut_bits) * sizeof(u16), (hipStream_t)stream, *tables, item, zero, 0u,
            (const u8 *)d_in, (u8 *)d_out, state, result, ~0ull, (const u32 *)nullptr);
    }
    return (int)hipGetLastError();
}

} /* extern "C" */
