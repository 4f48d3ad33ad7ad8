/*
 * What the decode kernels of every translation unit share: sub-chunk geometry and record formats, the row-synchronous
 * walk (row_walk), the walk of all entry states at once (union_row_fast), LDS addresses as numbers, and the careful
 * symbol-by-symbol walk over the end of a stream (tail_follow).
 */
#ifndef HUFFMAN_AMD_DECODE_COMMON_HPP
#define HUFFMAN_AMD_DECODE_COMMON_HPP
#include "kernels_common.hpp"

namespace {

/* ------------------------------------------------------------------ decode: shared pieces */

constexpr u32 kSubWords = HUFD_DEC_SUB_BYTES / 4;    /* 32 */
constexpr u32 kSubRows = kSubWords + 2;               /* + the first two words of the next sub-chunk */
constexpr u32 kRowStride = HUFD_DEC_LANES + 1;        /* word r of lane i at r * 257 + i: coalesced loads transpose without bank conflicts */
constexpr u32 kChunkWords = (kSubRows * kRowStride + 3u) & ~3u; /* what follows it in LDS stays 16-byte aligned */
constexpr u32 kGroupLanes = 16;
constexpr u32 kGroups = HUFD_DEC_LANES / kGroupLanes;
constexpr u32 kQuarters = 4;                          /* dec_emit walks a sub-chunk with this many threads */
constexpr u32 kQuarterBits = HUFD_DEC_SUB_BITS / kQuarters;
constexpr u32 kCpRows = HUFD_DEC_CP_ROWS;                    /* kQuarters - 1 checkpoints + the merged-state mask */
constexpr u32 kEmitThreads = HUFD_DEC_LANES * kQuarters;
constexpr u32 kExitStop = 15, kExitNoRef = 14;        /* top nibble of the merged-state row (states are < 13) */
constexpr u8 kRegularFew = 4; /* chunk_regular between dec_sync_few and dec_sync_true (0: the long way, 1: regular, 2: regular up to the end of its stream, 3: a thread's work) */

/* narrow transfer-function entry (per sub-chunk): [15] stop, [14:11] exit state, [10:0] symbols */
__device__ __forceinline__ u16 fn_pack(bool stop, u32 exit_state, u32 count) {
    return (u16)((stop ? 0x8000u : 0u) | (exit_state << 11) | count);
}
/* wide entry (groups, chunks, runs): [31] stop, [30:26] exit state, [25:0] symbols */
__device__ __forceinline__ u32 wide_pack(bool stop, u32 exit_state, u32 count) {
    return (stop ? 0x80000000u : 0u) | (exit_state << 26) | count;
}
__device__ __forceinline__ u32 widen(u16 f) {
    return wide_pack((f & 0x8000u) != 0, (f >> 11) & 15u, f & 0x7FFu);
}
__device__ __forceinline__ bool wide_stop(u32 f) {
    return (f >> 31) != 0;
}
__device__ __forceinline__ u32 wide_state(u32 f) {
    return (f >> 26) & 31u;
}
__device__ __forceinline__ u32 wide_count(u32 f) {
    return f & 0x03FFFFFFu;
}

/* word r (0..33) of the lane's sub-chunk; words 32 and 33 are the next lane's words 0 and 1 */
__device__ __forceinline__ u32 chunk_word(const u32 *timg, u32 lane, u32 r) {
    return timg[r * kRowStride + lane];
}

/* the 32 stream bits starting `pos` bits into the lane's sub-chunk */
__device__ __forceinline__ u32 chunk_window(const u32 *timg, u32 lane, u32 pos) {
    const u32 r = pos >> 5;
    const u64 two = ((u64)chunk_word(timg, lane, r) << 32) | chunk_word(timg, lane, r + 1);
    return (u32)((two << (pos & 31)) >> 32);
}

/*
 * Loads one chunk (+ two words of the next) into the transposed big-endian LDS image.
 * Fast path: eight coalesced 16-byte loads per thread, all in flight before the first use;
 * uint4 number q holds words 4(q&7).. of lane q>>3, and with the 257-word row stride the
 * 32 threads of a store group land on 32 different banks.
 */
template <u32 THREADS = HUFD_DEC_LANES>
__device__ __forceinline__ void chunk_load(u32 *timg, const u8 *src, u64 valid_bytes) {
    const u32 t = threadIdx.x;
    constexpr u32 kPerThread = HUFD_DEC_CHUNK_BYTES / 16 / THREADS; /* 8 for 256 threads */
    if (((uintptr_t)src & 15u) == 0 && valid_bytes >= HUFD_DEC_CHUNK_BYTES) {
        uint4 v[kPerThread];
#pragma unroll
        for (u32 j = 0; j < kPerThread; ++j) {
            v[j] = reinterpret_cast<const uint4 *>(src)[t + THREADS * j];
        }
#pragma unroll
        for (u32 j = 0; j < kPerThread; ++j) {
            const u32 q = t + THREADS * j;
            const u32 lane = q >> 3, r0 = 4 * (q & 7);
            const u32 w0 = __builtin_bswap32(v[j].x), w1 = __builtin_bswap32(v[j].y);
            u32 *col = timg + r0 * kRowStride + lane;
            col[0] = w0;
            col[kRowStride] = w1;
            col[2 * kRowStride] = __builtin_bswap32(v[j].z);
            col[3 * kRowStride] = __builtin_bswap32(v[j].w);
            if (r0 == 0 && lane > 0) {
                timg[kSubWords * kRowStride + lane - 1] = w0;
                timg[(kSubWords + 1) * kRowStride + lane - 1] = w1;
            }
        }
    } else {
        const bool aligned = ((uintptr_t)src & 3u) == 0;
        for (u32 g = t; g < HUFD_DEC_CHUNK_BYTES / 4; g += THREADS) {
            const u32 word = load_be32(src, g, valid_bytes, aligned);
            const u32 lane = g >> 5, r = g & 31;
            timg[r * kRowStride + lane] = word;
            if (r < 2 && lane > 0) {
                timg[(kSubWords + r) * kRowStride + lane - 1] = word;
            }
        }
    }
    if (t < 2) {
        timg[(kSubWords + t) * kRowStride + HUFD_DEC_LANES - 1] =
            load_be32(src, (u64)HUFD_DEC_CHUNK_BYTES / 4 + t, valid_bytes, false);
    }
}

template <u32 THREADS = HUFD_DEC_LANES>
__device__ __forceinline__ void lut_load(u16 *lut, const hufd_tables &tb) {
    for (u32 i = threadIdx.x; i < (1u << tb.lut_bits); i += THREADS) {
        lut[i] = tb.dec_lut[i];
    }
}

/*
 * One step of the walk (source/huffman.c:232-255 for one symbol): `pos` bits into the
 * sub-chunk, `remaining` stream bits left from the sub-chunk start.  Returns the code
 * length, or 0 with *why set when the walk ends here.
 */
__device__ __forceinline__ u32 code_at(
    u32 window, const u16 *lut, u32 lut_bits, u32 pos, u32 rem, u32 *symbol, u32 *why) {
    /* `rem` = stream bits from the sub-chunk start to the end of the item, clamped to [0, 2^30] */
    if (pos >= rem) {
        *why = HUFD_STOP_END;
        return 0;
    }
    const u32 entry = lut[window >> (32 - lut_bits)];
    const u32 len = entry & 0xFFu;
    if (len == 0) {
        *why = HUFD_STOP_INVALID;
        return 0;
    }
    if (pos + len > rem) {
        *why = HUFD_STOP_INCOMPLETE;
        return 0;
    }
    *symbol = entry >> 8;
    return len;
}

__device__ __forceinline__ u32 clamp_remaining(u64 valid_bytes, u32 lane) {
    const long long rem = (long long)(valid_bytes * 8) - (long long)lane * HUFD_DEC_SUB_BITS;
    return rem <= 0 ? 0u : (rem > (1ll << 30) ? (1u << 30) : (u32)rem);
}

/*
 * Sequential bit window of one lane over its sub-chunk: the next 33..64 stream bits sit
 * at the top of `win`, refilled a word at a time from the transposed image, so a long
 * walk costs one LDS word read per 32 bits instead of two per symbol
 * (the register twin of the 64-bit window of source/huffman.c:196-211).
 */
struct bit_reader {
    u64 win;
    u32 nb;    /* valid bits in win, kept above 32 */
    u32 next;  /* index of the sub-chunk word that follows `ahead` */
    u32 ahead; /* the word that will be appended next: read one refill early, never waited for */

    __device__ __forceinline__ static u32 clamp_row(u32 r) {
        /* words past index 33 are never needed for a decision (a code starts inside the
         * sub-chunk and is at most 32 bits long); the read only has to stay in bounds */
        return r < kSubRows ? r : kSubRows - 1;
    }
    __device__ __forceinline__ void start(const u32 *timg, u32 lane, u32 pos) {
        const u32 r = pos >> 5;
        win = (((u64)chunk_word(timg, lane, r) << 32) | chunk_word(timg, lane, r + 1)) << (pos & 31);
        nb = 64 - (pos & 31);
        ahead = chunk_word(timg, lane, clamp_row(r + 2));
        next = r + 3;
    }
    __device__ __forceinline__ u32 peek() const {
        return (u32)(win >> 32);
    }
    /*
     * skip() without a branch, for loops whose lanes stop at different times (len may be 0):
     * the look-ahead word is re-read every step and selected in, so the only control flow
     * left in the caller's loop is the loop itself.
     */
    __device__ __forceinline__ void skip_predicated(const u32 *timg, u32 lane, u32 len) {
        win <<= len;
        nb -= len;
        const bool refill = nb <= 32;
        const u64 add = (u64)ahead << ((32 - nb) & 31);
        win |= refill ? add : 0;
        nb += refill ? 32u : 0u;
        next += refill ? 1u : 0u;
        ahead = chunk_word(timg, lane, clamp_row(next - 1));
    }
    __device__ __forceinline__ void skip(const u32 *timg, u32 lane, u32 len) {
        win <<= len;
        nb -= len;
        if (nb <= 32) {
            win |= (u64)ahead << (32 - nb);
            nb += 32;
            ahead = chunk_word(timg, lane, clamp_row(next));
            ++next;
        }
    }
};

/*
 * The same window for loops whose lanes stop at different times, kept as two words and a bit
 * offset so that a step is a handful of 32-bit operations and no branch: the 32 stream bits
 * at the cursor are a funnel shift of (hi:lo); crossing into the next word moves lo up and
 * takes the word that was requested one step earlier.
 */
struct lane_window {
    u32 hi, lo; /* the word the cursor is in, and the one after it */
    u32 k;      /* cursor, bits into hi: 0..31 */
    u32 next;   /* sub-chunk word index of `ahead` */
    u32 ahead;  /* word `next`, re-read every step so that it is there when the cursor crosses */

    __device__ __forceinline__ void start(const u32 *timg, u32 lane, u32 pos) {
        const u32 r = pos >> 5;
        hi = chunk_word(timg, lane, r);
        lo = chunk_word(timg, lane, r + 1);
        k = pos & 31u;
        next = r + 2;
        ahead = chunk_word(timg, lane, bit_reader::clamp_row(next));
    }
    __device__ __forceinline__ u32 peek() const {
        return (u32)(((((u64)hi << 32) | lo) << k) >> 32);
    }
    /* advance by len <= 32 bits (0 = stay) */
    __device__ __forceinline__ void skip(const u32 *timg, u32 lane, u32 len) {
        k += len;
        const bool cross = k >= 32u;
        hi = cross ? lo : hi;
        lo = cross ? ahead : lo;
        next += cross ? 1u : 0u;
        k &= 31u;
        ahead = chunk_word(timg, lane, bit_reader::clamp_row(next));
    }
};

/* result of following a run of transfer functions */
struct fold_result {
    bool stop;
    u32 state;
    u64 count;
};

/*
 * Folds `n` consecutive transfer functions from entry state `start`.
 * fn(i, state) yields the wide entry of element i.
 */
template <typename Fn>
__device__ __forceinline__ fold_result chain_fold(u32 n, u32 start, Fn fn) {
    fold_result r = {false, start, 0};
    for (u32 i = 0; i < n; ++i) {
        const u32 f = fn(i, r.state);
        r.count += wide_count(f);
        if (wide_stop(f)) {
            r.stop = true;
            r.state = 0;
            return r;
        }
        r.state = wide_state(f);
    }
    return r;
}
__device__ __forceinline__ u32 wide_pack(const fold_result &r) {
    return wide_pack(r.stop, r.state, (u32)r.count);
}

/* ------------------------------------------------------------------ decode: row-synchronous walk */

/*
 * Every decode kernel is bound by its vector-instruction count (measured: ~4 cycles per wave
 * instruction and SIMD, profiles/r01_d_*), so the walk below is written for the fewest of them
 * per symbol.  All lanes of a wave stand in the SAME 32-bit word ("row") of their sub-chunks at
 * the same time: the two words a window can touch are loaded once per row at a compile-time
 * offset, nothing is shifted between registers, and a step is
 *
 *      offset = (row pair >> s) & mask;   entry = walk_lut[offset];   state += entry;
 *
 * `state` keeps the shift amount for the next window in its low half and the symbols counted so
 * far in its high half; the table entry is 0x10000 - length, so one add moves both.  The low half
 * is 64 + (bits from the code start to the end of the pair) - (index width) - 2: the hardware
 * uses the low six bits of a shift amount, which takes the 64 off again, and the - 2 makes the
 * masked window a byte offset into the table of 32-bit entries.  A window without a code has
 * length 48 in this table: the walk leaves the row at once and lands below every position a real
 * code can produce, which is how a dead walk is told from a live one (once per row, not per step).
 *
 * Only for sub-chunks that lie wholly inside the stream with at least 8 bytes after them (every
 * code that starts in them is whole): no end-of-stream tests anywhere.
 */
constexpr u32 kWalkDeadLen = 48;

struct row_walk {
    u32 thr;    /* a code starts in the current row while (u16)state > thr */
    u32 mask;   /* index mask, times four */
    u32 floor;  /* (u16)state after a row is at least this unless the walk died */
    u32 sure;   /* codes that are certain to start in a row: taken without asking (no compare, no branch, no lane mask) */

    /* pos: the bit of the shifted pair the window's lowest bit lands on (2: the window is the byte offset of a dword entry) */
    __device__ __host__ __forceinline__ row_walk(u32 lut_bits, u32 max_bits, u32 pos = 2) {
        /* 512 = a multiple of 64 that keeps the low half positive through `sure` steps of a dead walk (48 bits each) */
        thr = 512 + (32 - lut_bits) - pos;
        mask = ((1u << lut_bits) - 1u) << pos;
        floor = thr - max_bits + 1;
        /* a row's first code starts at most max(max_bits - 1, 7) bits in (entry states go up to 7), the others max_bits apart */
        const u32 late = max_bits - 1 > 7 ? max_bits - 1 : 7;
        sure = (31 - late) / max_bits + 1;
    }
    /* state of a walk whose next code starts `k` bits into the current row, `count` symbols so far */
    __device__ __forceinline__ u32 state_at(u32 k, u32 count) const {
        return (count << 16) | (thr + 32 - k);
    }
    __device__ __forceinline__ u32 offset_of(u32 state) const { /* bits into the current row */
        return thr + 32 - (state & 0xFFFFu);
    }
    /* all codes of the walk that start in the row whose words are hi:lo */
    template <bool STEP_BY_STEP = false> /* true: ask before every step, for a walk whose count must be right even if it dies */
    __device__ __forceinline__ u32 row(u32 state, u32 hi, u32 lo, const u32 *wlut) const {
        const u64 pair = ((u64)hi << 32) | lo;
        for (u32 i = 0; !STEP_BY_STEP && i < sure; ++i) {
            const u32 off = (u32)(pair >> (state & 63u)) & mask;
            state += *reinterpret_cast<const u32 *>(reinterpret_cast<const u8 *>(wlut) + off);
        }
        while ((state & 0xFFFFu) > thr) {
            const u32 off = (u32)(pair >> (state & 63u)) & mask;
            state += *reinterpret_cast<const u32 *>(reinterpret_cast<const u8 *>(wlut) + off);
        }
        return state;
    }
    __device__ __forceinline__ bool died(u32 state) const {
        return (state & 0xFFFFu) < floor;
    }
    /* on to the next row; a walk that has died is put back on a row start so that its state stays in range */
    __device__ __forceinline__ u32 next_row(u32 state, bool dead = false) const {
        return dead ? state_at(0, 0) : state + 32u;
    }
};

/* ------------------------------------------------------------------ decode: sync, regular chunks */

/*
 * dec_sync for chunks that lie inside the stream (DESIGN.md "Decode: regular chunks"); every
 * other chunk, and every chunk that turns out not to be regular, is put on a list for
 * dec_sync_kernel (the long way, which assumes nothing).  Same tables out.
 *
 * Inside the stream a chunk is nearly always REGULAR: in every sub-chunk the walks from all
 * entry states become ONE walk after a few rows (those on a wrong phase die or fall in step),
 * and that walk reaches the end of the sub-chunk.  Then a sub-chunk's exit state does not depend
 * on its entry state, so lane i's true entry state simply IS lane i-1's exit state, and what is
 * left to find is how many symbols the walk from that entry state takes to the meeting bit:
 *   U  rows 0 .. m-1: all entry states together (head mask) until every lane of the wave is down
 *      to one head (m is the same for the wave, ~5 rows);
 *   R  rows m .. 31: the one walk, counting (row_walk: shift, mask, table, add);
 *   H  rows 0 .. m-1 again: the walk from the true entry state, counting, which must land on the
 *      lane's meeting bit.  Threads 0 .. ns-1 do the same for every entry state of sub-chunk 0,
 *      whose true entry state only dec_scan can know.
 *
 * The walks are one dependent chain per lane (window -> table -> add -> test), a couple of
 * hundred cycles a step, so what counts is how many chains a SIMD holds.  A sub-chunk therefore
 * lives in its lane's REGISTERS (33 words, loaded as the lane's own 128-byte line) and not in an
 * LDS image: eight waves per SIMD instead of four, rows at compile-time register numbers, and
 * the LDS holds only the two small tables.
 */
/*
 * The words of the one or two sub-chunks that hold the end of a stream, zero-filled past its last byte, into
 * LDS: all loads in flight together, so that the symbol-by-symbol walk over them reads LDS and not memory.
 */
constexpr u32 kTailWords = 2 * kSubWords + 4;
__device__ __forceinline__ void tail_words_load(u32 *dst, const u8 *src, u64 bytes) {
    u32 tmp[kTailWords];
#pragma unroll
    for (u32 i = 0; i < kTailWords; ++i) {
        tmp[i] = load_be32(src, i, bytes, true);
    }
#pragma unroll
    for (u32 i = 0; i < kTailWords; ++i) {
        dst[i] = tmp[i];
    }
}
__device__ __forceinline__ u32 tail_window(const u32 *words, u32 pos) {
    const u32 wi = pos >> 5;
    const u64 two = ((u64)words[wi] << 32) | words[wi + 1];
    return (u32)((two << (pos & 31u)) >> 32);
}
/*
 * The 32 stream bits at the careful walk's position, out of a 64-bit register window that is topped up a word
 * at a time; the word that will be needed next is read one refill early, so that the only LDS read a step has
 * to wait for is the table look-up (the register twin of the window of source/huffman.c:196-211).
 */
struct tail_reader {
    const u32 *words;
    u64 win;
    u32 nb, next, ahead;

    __device__ __forceinline__ u32 word(u32 i) const {
        return words[i < kTailWords ? i : kTailWords - 1]; /* words past the end are zero anyway */
    }
    __device__ __forceinline__ void start(const u32 *w, u32 pos) {
        words = w;
        const u32 r = pos >> 5;
        win = (((u64)word(r) << 32) | word(r + 1)) << (pos & 31u);
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

/* code_at() with the length taken from a walk table in LDS (low half of an entry = 0x10000 - length, 48 = no code) */
template <u32 LB>
__device__ __forceinline__ u32 code_at_walk(u32 window, const u32 *wlut, u32 pos, u32 rem, u32 *entry, u32 *why) {
    if (pos >= rem) {
        *why = HUFD_STOP_END;
        return 0;
    }
    const u32 e = wlut[window >> (32u - LB)];
    const u32 len = (0x10000u - (e & 0xFFFFu)) & 0xFFFFu;
    if (len == kWalkDeadLen) {
        *why = HUFD_STOP_INVALID;
        return 0;
    }
    if (pos + len > rem) {
        *why = HUFD_STOP_INCOMPLETE;
        return 0;
    }
    *entry = e;
    return len;
}

constexpr u32 kFastRows = kSubWords + 1;  /* a window of the last row reaches into the next sub-chunk's first word */
constexpr u32 kFastMaxMeet = 16;          /* no single head after this many rows: not regular */
constexpr u32 kFastHopelessRows = 6;      /* most lanes of a wave with several heads after this many: not regular either (dec_sync_one) */

template <u32 LB>
struct fast_shared {
    u32 wlut[1u << LB];                  /* 0x10000 - length; length 48 = no code */
    u32 exit_state[HUFD_DEC_LANES];
    u32 sub0[kFastMaxMeet + 4];          /* the first rows of sub-chunk 0, for the threads that try its entry states */
    u32 wave_sum[HUFD_DEC_LANES / 64];
    u32 bad;
    u32 pad[3];
    u16 hops[1u << LB];                  /* 1 << code length of a window (the head it sends on), 0 = no code */
};

template <u32 LB>
__device__ __forceinline__ u64 union_row_fast(u64 heads, u32 hi, u32 lo, const u16 *hops) {
    const u64 pair = ((u64)hi << 32) | lo;
    u32 here = (u32)heads, next = (u32)(heads >> 32); /* heads in this row / already in the next one */
    /* the two lowest heads a trip: two look-ups that do not wait for each other (a head that the first sends onto the
     * second, or in between the two, is simply taken again on a later trip: the heads are a set) */
    while (here) {
        const u32 j0 = (u32)__builtin_ctz(here);
        here &= here - 1;
        const bool two = here != 0;
        const u32 j1 = two ? (u32)__builtin_ctz(here) : j0;
        here &= here - 1;
        const u32 off0 = (u32)(pair >> (63u - LB - j0)) & (((1u << LB) - 1u) << 1); /* byte offset into the u16 table */
        const u32 off1 = (u32)(pair >> (63u - LB - j1)) & (((1u << LB) - 1u) << 1);
        const u64 sent0 = (u64)*reinterpret_cast<const u16 *>(reinterpret_cast<const u8 *>(hops) + off0) << j0;
        const u64 sent1 = (u64)*reinterpret_cast<const u16 *>(reinterpret_cast<const u8 *>(hops) + off1) << j1;
        const u64 sent = sent0 | (two ? sent1 : 0ull);
        here |= (u32)sent; /* no code: nothing is sent on, the walk is gone */
        next |= (u32)(sent >> 32);
    }
    return next;
}

/*
 * Phase U's first row.  The ns entry states are ns look-ups that do not wait for each other (the general loop takes a
 * head at a time, lowest first, because a head may send another into the same row): their windows lie in the row's own
 * word, where they land is an OR.  What lands on an entry state is followed already; the general loop goes on with the
 * rest of the row.
 */
template <u32 LB>
__device__ __forceinline__ u64 union_first_row(u32 ns, bool active, u32 hi, u32 lo, const u16 *hops) {
    u32 landed = 0;
#pragma unroll
    for (u32 j = 0; j < HUFD_DEC_MAX_LUT_BITS; ++j) {
        if (j < ns) {
            const u32 off = (hi >> (31u - LB - j)) & (((1u << LB) - 1u) << 1); /* byte offset into the u16 table */
            landed |= (u32)*reinterpret_cast<const u16 *>(reinterpret_cast<const u8 *>(hops) + off) << j;
        }
    }
    const u64 heads = active ? landed & ~((1u << ns) - 1u) : 0u;
    return union_row_fast<LB>(heads, hi, lo, hops);
}

/* ------------------------------------------------------------------ decode: sync, regular chunks, fewer instructions */

/*
 * An LDS address as a number, and a word read at such a number: a walk-table entry is then read at
 * (window & mask) | table, ONE instruction for the address where pointer arithmetic gives two (and + add), given a
 * table that starts at a multiple of its size.
 */
__device__ __forceinline__ u32 lds_offset_of(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
#else
    return (u32)(reinterpret_cast<const u8 *>(p) - dyn_lds);
#endif
}
__device__ __forceinline__ u32 lds_word_at(u32 byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) u32 *)(uintptr_t)byte_offset;
#else
    return *reinterpret_cast<const u32 *>(dyn_lds + byte_offset);
#endif
}

__device__ __forceinline__ u32 lds_byte_at(u32 byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) u8 *)(uintptr_t)byte_offset;
#else
    return dyn_lds[byte_offset];
#endif
}

/* ------------------------------------------------------------------ decode: the rows' loops written out for the GPU */

/* row_walk::row with the number of certain steps known to the compiler and the table given as an LDS offset.
 *
 * The loop over the codes that MAY start in the row is written out for the GPU: what the compiler makes of
 * `while ((state & 0xFFFF) > thr)` is five vector instructions a trip (shift, address, add, and, compare) and three scalar
 * ones that fold the compare into the exec mask; here the compare is v_cmpx_lt_u16 -- the low half as it stands, straight
 * into exec -- so a trip is four vector instructions and a branch.  These kernels' time follows the vector instructions a
 * step costs (profiles/r05_micro: the walk at eight waves a SIMD is held by issue as much as by the LDS).  v62 / v63 are
 * the block's own temporaries (a 64-bit shift result whose low word becomes the address, then the entry). */
template <u32 SURE, bool STEP_BY_STEP = false, bool PLAIN = false> /* PLAIN: the loop as the compiler writes it (a kernel held to 64 registers has none to set aside for the block) */
__device__ __forceinline__ u32 lean_row(u32 state, u32 hi, u32 lo, u32 table, const row_walk &rw) {
    const u64 pair = ((u64)hi << 32) | lo;
    if (!STEP_BY_STEP) {
#pragma unroll
        for (u32 i = 0; i < SURE; ++i) {
            state += lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (PLAIN) {
        while ((state & 0xFFFFu) > rw.thr) {
            state += lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
        }
        return state;
    }
    u64 saved_exec;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_u16_e32 vcc, %[thr], %[st]\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "v_lshrrev_b64 v[62:63], %[st], %[pair]\n\t"
        "v_and_or_b32 v62, v62, %[mask], %[tab]\n\t"
        "ds_read_b32 v62, v62\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_u32_e32 %[st], %[st], v62\n\t"
        "v_cmpx_lt_u16_e32 vcc, %[thr], %[st]\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [st] "+v"(state), [sv] "=&s"(saved_exec)
        : [pair] "v"(pair), [mask] "s"(rw.mask), [tab] "v"(table), [thr] "s"(rw.thr)
        : "vcc", "v62", "v63");
#else
    while ((state & 0xFFFFu) > rw.thr) {
        state += lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
    }
#endif
    return state;
}

/*
 * The codes of a row behind the certain ones for a walk whose position is the state's low BYTE and whose row is two words
 * shifted for a 32-bit funnel (dec_sync_one's one_walk): while the byte is above `thr`, the entry of the window at the
 * position's low five bits, added.  Written out for the GPU: the compiler's loop is five vector instructions a trip (funnel,
 * and, add, an AND that cuts the byte out, compare) and two scalar ones that fold the compare into the exec mask; here four
 * and a branch -- the compare looks at the byte by itself (SDWA) and writes exec.
 */
__device__ __forceinline__ u32 byte_rows_uncertain(u32 state, u32 xh, u32 xl, u32 table, u32 mask, u32 thr) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 saved_exec;
    u32 t;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_u32_sdwa vcc, %[st], %[thr] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "v_alignbit_b32 %[t], %[xh], %[xl], %[st]\n\t"
        "v_and_or_b32 %[t], %[t], %[mask], %[tab]\n\t"
        "ds_read_b32 %[t], %[t]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_u32_e32 %[st], %[st], %[t]\n\t"
        "v_cmpx_gt_u32_sdwa vcc, %[st], %[thr] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [st] "+v"(state), [t] "=&v"(t), [sv] "=&s"(saved_exec)
        : [xh] "v"(xh), [xl] "v"(xl), [mask] "s"(mask), [tab] "v"(table), [thr] "s"(thr)
        : "vcc");
#else
    while ((state & 0xFFu) > thr) {
        state += lds_word_at((funnel_by_low5(xh, xl, state) & mask) | table);
    }
#endif
    return state;
}

/*
 * The codes of a row behind the certain ones, one chain: while a code starts in the row, its table entry, its symbol to the
 * stage, the state on.  Written out for the GPU as lean_row is: what the compiler makes of the loop in C is eight vector
 * instructions a trip (shift, address, a copy of the store address, that address + 1, + the record's base, the state's add, a
 * compare through SDWA) and two scalar ones that fold the compare into the exec mask; here five and a branch -- the compare is
 * v_cmpx_lt_u16 on the state's low half, straight into exec.  1.3 of a row's 4.3 trips are such trips, for either chain.
 * `at`: where the chain's next symbol goes, as an LDS address.  v62 / v63 are the block's own temporaries.
 */
__device__ __forceinline__ void emit_uncertain_codes(u32 &state, u32 &at, u64 pair, u32 table, const row_walk &rw) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 saved_exec;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_u16_e32 vcc, %[thr], %[st]\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "v_lshrrev_b64 v[62:63], %[st], %[pair]\n\t"
        "v_and_or_b32 v62, v62, %[mask], %[tab]\n\t"
        "ds_read_b32 v62, v62\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_u32_e32 %[st], %[st], v62\n\t"
        "ds_write_b8_d16_hi %[at], v62\n\t"
        "v_add_u32_e32 %[at], 1, %[at]\n\t"
        "v_cmpx_lt_u16_e32 vcc, %[thr], %[st]\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [st] "+v"(state), [at] "+v"(at), [sv] "=&s"(saved_exec)
        : [pair] "v"(pair), [mask] "s"(rw.mask), [tab] "v"(table), [thr] "s"(rw.thr)
        : "vcc", "v62", "v63", "memory");
#else
    while ((state & 0xFFFFu) > rw.thr) {
        const u32 e = lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
        dyn_lds[at++] = (u8)(e >> 16);
        state += e;
    }
#endif
}

/* ------------------------------------------------------------------ decode: the end of a stream */

/*
 * One THREAD per chunk that holds the end of a stream (after dec_sync_one<TAIL> / dec_sync_pack, which took its whole lanes):
 * follows the true path from where the last whole lane leaves it to where the stream stops, symbol by symbol
 * with the end-of-stream tests of source/huffman.c:232-255, through the first sub-chunk behind the whole lanes
 * and the few bytes of the next one.  Then completes the chunk's tables: records of those one or two lanes,
 * and symbols + stop (or exit state) in the chunk function.  A walk of ~100 dependent steps is slow for one
 * thread and nothing for 65 536 of them side by side; inside the chunk's workgroup it held a workgroup up and cost
 * the kernel its occupancy.
 */
constexpr u32 kTailThreads = 128;

struct tail_walk {
    u32 count[2]; /* symbols that start in the first / the second sub-chunk behind the whole lanes */
    u32 exit;     /* entry state of the second one */
    u32 stop;     /* where the true path stops: 0 in the first, 1 in the second, 2 not in this chunk */
};

/* words: this thread's LDS copy of the stream's last bytes; lut: the u16 decode table in LDS */
__device__ __forceinline__ tail_walk tail_follow(
    const u32 *words, const u16 *lut, u32 lut_bits, u32 entry, u32 rem, u32 limit, u8 *out /* NULL: only count */,
    u32 *stop_pos, u32 *stop_why) {
    tail_walk r = {{0, 0}, 0, 2};
    tail_reader tr;
    tr.start(words, entry);
    u32 pos = entry, why = HUFD_STOP_NONE;
    while (pos < limit) {
        u32 sym = 0;
        const u32 len = code_at(tr.peek(), lut, lut_bits, pos, rem, &sym, &why);
        if (!len) {
            break;
        }
        tr.skip(len);
        if (out) {
            *out++ = (u8)sym;
        }
        if (pos < HUFD_DEC_SUB_BITS) { /* a symbol belongs to the sub-chunk its code starts in */
            ++r.count[0];
            if (pos + len >= HUFD_DEC_SUB_BITS) {
                r.exit = pos + len - HUFD_DEC_SUB_BITS;
            }
        } else {
            ++r.count[1];
        }
        pos += len;
    }
    r.stop = why == HUFD_STOP_NONE ? 2u : (pos < HUFD_DEC_SUB_BITS ? 0u : 1u);
    *stop_pos = pos;
    *stop_why = why;
    return r;
}

struct tail_lds {
    u32 words[kTailThreads][kTailWords + 1]; /* + 1: odd stride, the threads' copies start in different banks */
};


/* chunk entry record (dec_scan writes it, the emit kernels read it): [7:0] entry state, [8] reached */
__device__ __forceinline__ u32 entry_pack(u32 state, bool reached) {
    return state | (reached ? 0x100u : 0u);
}

/* several short end-of-stream chunks a workgroup (dec_sync_pack, dec_emit_pack): the most slots, the fewest such chunks */
constexpr u32 kPackMaxSlots = 16;
constexpr u32 kPackMinChunks = HUFD_DEC_PACK_MIN_CHUNKS;

} /* namespace */

#endif /* HUFFMAN_AMD_DECODE_COMMON_HPP */
