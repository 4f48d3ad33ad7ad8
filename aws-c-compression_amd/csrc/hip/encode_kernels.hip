/*
 * Encode: the kernels that replace the per-symbol loop of reference source/huffman.c:161-173 and the bit packer :59-105,
 * and their launches.
 *   enc_count, enc_scan_*, enc_pack*   the three-kernel road (any coder; the way back of the one-pass road)
 *   enc_onepass + enc_finish           one pass over the input for coders with codes of 4..15 bits for all 256 symbols
 *   enc_tiny                           a thread per short item
 *   enc_block                          one host-pointer call of up to 16 Ki symbols, one workgroup, one launch
 */
#include "kernels_common.hpp"
#include "launch_common.hpp"

namespace {

/* ------------------------------------------------------------------ encode: count */

constexpr u32 kGroupsPerLane = HUFD_ENC_SEG_BYTES / (HUFD_ENC_THREADS * 16); /* 16-byte groups a lane owns per segment */

/* The lane's 16-byte groups of a segment, all requested before any of them is used. */
__device__ __forceinline__ void load_segment_groups(
    const u8 *src, u32 seg_len, u32 (&gw)[kGroupsPerLane][4], u32 (&gvalid)[kGroupsPerLane]) {
    const bool aligned = ((uintptr_t)src & 15u) == 0;
#pragma unroll
    for (u32 g = 0; g < kGroupsPerLane; ++g) {
        const u32 base = (g * HUFD_ENC_THREADS + threadIdx.x) * 16;
        gvalid[g] = base < seg_len ? (seg_len - base < 16 ? seg_len - base : 16) : 0;
        gw[g][0] = gw[g][1] = gw[g][2] = gw[g][3] = 0;
        if (gvalid[g]) {
            load_group(src + base, gvalid[g], aligned, gw[g]);
        }
    }
}


/*
 * Bits per segment and its first symbol without a code.  The kernel is a stream of table
 * look-ups, and a 256-entry table read by 64 lanes at random is served at a third of the LDS rate
 * (bank conflicts), which made this kernel LDS-bound.  So every entry is kept 32 times, one copy
 * per bank: lane l reads entry b at word 32 b + (l & 31) and never shares a bank with another
 * lane.  32 KiB of table per workgroup, hence persistent workgroups (segment blockIdx.x,
 * + gridDim.x, ...) that build it once.  Entry = length | (length == 0) << 20, so one add per
 * symbol counts the bits and the symbols without a code together.
 */
constexpr u32 kCountLdsBytes = 256 * 32 * 4 + 64;

__global__ __launch_bounds__(HUFD_ENC_THREADS) void enc_count_kernel(
    hufd_tables tb,
    const hufd_enc_seg *segs,
    const u8 *d_in,
    u32 *seg_bits,
    u32 *wave_bits, /* [seg][4]: bits of each quarter of the segment (wave w of enc_pack_wave packs quarter w) */
    u32 *seg_unk,
    u32 *careful_count,
    u32 n_segs,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    u32 *tab = reinterpret_cast<u32 *>(dyn_lds); /* [256][32] */
    u32 *slots = tab + 256 * 32;                  /* [16] */

    const u32 tid = threadIdx.x;
    const u32 lane = tid & (kWave - 1), wave = tid / kWave;
    if (blockIdx.x == 0 && tid == 0) {
        *careful_count = 0; /* the scan kernels of this launch append to the list */
    }
    {
        const u32 len = (u32)(tb.enc_table[tid] >> 32);
        const u32 e = len | (len == 0 ? 1u << 20 : 0u);
#pragma unroll
        for (u32 k = 0; k < 32; ++k) {
            tab[tid * 32 + ((k + tid) & 31u)] = e; /* rotated so that the 32 stores of a group hit 32 banks */
        }
    }
    __syncthreads();
    const u8 *mine = reinterpret_cast<const u8 *>(tab) + (lane & 31u) * 4u;

    /* the next segment's symbols are asked for before this one's are counted */
    uint4 v[kGroupsPerLane], vn[kGroupsPerLane];
    bool fetched = false;
#pragma unroll
    for (u32 g = 0; g < kGroupsPerLane; ++g) {
        v[g] = vn[g] = uint4{0, 0, 0, 0};
    }
    for (u32 s = blockIdx.x; s < n_segs; s += gridDim.x) {
        const hufd_enc_seg seg = uniform_seg(&segs[s]);
        const u8 *src = d_in + seg.in_off;
        u32 sum = 0;
        const bool had = fetched;
        fetched = false;
        if (had) {
#pragma unroll
            for (u32 g = 0; g < kGroupsPerLane; ++g) {
                v[g] = vn[g];
            }
        }
        const bool whole = seg.len == HUFD_ENC_SEG_BYTES && ((uintptr_t)src & 15u) == 0;
        if (whole && !had) {
#pragma unroll
            for (u32 g = 0; g < kGroupsPerLane; ++g) {
                /* wave w counts the w-th quarter of the segment: the unit enc_pack_wave packs */
                v[g] = reinterpret_cast<const uint4 *>(src)[(wave * kGroupsPerLane + g) * kWave + lane];
            }
        }
        if (s + gridDim.x < n_segs) {
            const hufd_enc_seg nseg = uniform_seg(&segs[s + gridDim.x]);
            const u8 *nsrc = d_in + nseg.in_off;
            if (nseg.len == HUFD_ENC_SEG_BYTES && ((uintptr_t)nsrc & 15u) == 0) {
#pragma unroll
                for (u32 g = 0; g < kGroupsPerLane; ++g) {
                    vn[g] = reinterpret_cast<const uint4 *>(nsrc)[(wave * kGroupsPerLane + g) * kWave + lane];
                }
                fetched = true;
            }
        }
        if (whole) {
#pragma unroll
            for (u32 g = 0; g < kGroupsPerLane; ++g) {
                const u32 wd[4] = {v[g].x, v[g].y, v[g].z, v[g].w};
#pragma unroll
                for (u32 j = 0; j < 16; ++j) {
                    const u32 b = (wd[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                    sum += *reinterpret_cast<const u32 *>(mine + b * 128u);
                }
            }
        } else {
            /* a ragged or unaligned segment: symbol by symbol */
            u32 gw[kGroupsPerLane][4], gvalid[kGroupsPerLane];
            load_segment_groups(src, seg.len, gw, gvalid);
#pragma unroll
            for (u32 g = 0; g < kGroupsPerLane; ++g) {
                for (u32 j = 0; j < gvalid[g]; ++j) {
                    sum += *reinterpret_cast<const u32 *>(mine + group_byte(gw[g], j) * 128u);
                }
            }
        }
        u32 bits = wave_sum(sum & 0xFFFFFu);
        u32 holes = wave_sum(sum >> 20);
        if (lane == 0) {
            slots[wave] = bits;
            slots[4 + wave] = holes;
            wave_bits[4 * s + wave] = bits; /* only meaningful for whole, aligned segments: the others are packed symbol by symbol */
        }
        __syncthreads();
        bits = slots[0] + slots[1] + slots[2] + slots[3];
        holes = slots[4] + slots[5] + slots[6] + slots[7];
        u32 unk = HUFD_NONE32;
        if (holes) {
            /* rare: which symbol is the first without a code */
            for (u32 g = 0; g < kGroupsPerLane && unk == HUFD_NONE32; ++g) {
                const u32 base = (g * HUFD_ENC_THREADS + tid) * 16;
                for (u32 j = 0; j < 16 && base + j < seg.len; ++j) {
                    if ((*reinterpret_cast<const u32 *>(mine + (u32)src[base + j] * 128u) >> 20) != 0) {
                        unk = base + j;
                        break;
                    }
                }
            }
            unk = wave_min(unk);
            if (lane == 0) {
                slots[8 + wave] = unk;
            }
            __syncthreads();
            unk = slots[8];
#pragma unroll
            for (u32 wv = 1; wv < HUFD_ENC_THREADS / kWave; ++wv) {
                unk = slots[8 + wv] < unk ? slots[8 + wv] : unk;
            }
        }
        if (tid == 0) {
            seg_bits[s] = bits;
            seg_unk[s] = unk;
        }
        __syncthreads(); /* slots are reused by the next segment */
    }
}

/* ------------------------------------------------------------------ encode: scan + outcome */

/*
 * Outcome of one encode call in closed form (DESIGN.md "Encode outcome"), given the
 * item's total bit count and its first symbol without a code.  Restates the stop
 * conditions of reference source/huffman.c:149-173 without replaying the loop.
 */
__device__ void enc_finish_item(
    const hufd_enc_item &it,
    u64 total_bits,
    u32 unk_seg,
    u32 unk_idx,
    u64 unk_seg_bitoff,
    u32 unk_seg_bits,
    u32 edge_seg, /* segment with offset < capacity edge <= offset + bits, or HUFD_NONE32 */
    u32 *careful_list,
    u32 *careful_count,
    hufd_enc_item_state *state,
    hufd_enc_result *result) {

    const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
    hufd_enc_item_state st;
    hufd_enc_result rs;
    st.total_bits = total_bits;
    st.unk_seg = unk_seg;
    st.unk_idx = unk_idx;
    st.reserved = 0;
    rs.ovf_pattern = 0;
    rs.ovf_bits = 0;
    rs.reserved = 0;
    rs.consumed = 0;
    rs.produced = 0;
    rs.total_bits = total_bits;

    bool unknown_possible = unk_seg != HUFD_NONE32;
    if (unknown_possible && cap_bits <= unk_seg_bitoff) {
        /* the output fills before the bad symbol is ever read */
        unknown_possible = false;
        st.unk_seg = HUFD_NONE32;
    }

    if (unknown_possible) {
        rs.consumed = (u64)(unk_seg - it.first_seg) * HUFD_ENC_SEG_BYTES + unk_idx + 1;
        if (cap_bits > unk_seg_bitoff + unk_seg_bits) {
            st.status = HUFD_ENC_UNKNOWN; /* produced comes from the segment's workgroup */
        } else {
            st.status = HUFD_ENC_DECIDE;
        }
        rs.status = HUFD_ENC_UNKNOWN;
    } else if (unk_seg == HUFD_NONE32 && total_bits <= cap_bits) {
        st.status = HUFD_ENC_OK;
        rs.status = HUFD_ENC_OK;
        rs.consumed = it.in_len;
        rs.produced = (total_bits + 7) >> 3;
    } else {
        st.status = HUFD_ENC_SHORT;
        rs.status = HUFD_ENC_SHORT;
        rs.produced = it.out_cap;
        if (it.ovf_bits >= cap_bits) {
            /* the carried overflow alone fills the output (source/huffman.c:149-156) */
            rs.consumed = 0;
            rs.ovf_bits = (u32)(it.ovf_bits - cap_bits);
            rs.ovf_pattern = rs.ovf_bits ? (it.ovf_pattern & (u32)((1ull << rs.ovf_bits) - 1)) : 0;
        }
        /* otherwise the lane that packs the crossing symbol fills consumed / overflow */
    }
    /* the segments that need the per-symbol packer */
    const bool want_short = st.status == HUFD_ENC_SHORT || st.status == HUFD_ENC_DECIDE;
    if (want_short && edge_seg != HUFD_NONE32 && (st.unk_seg == HUFD_NONE32 || edge_seg <= st.unk_seg)) {
        careful_list[atomicAdd(careful_count, 1u)] = edge_seg;
    }
    if (st.unk_seg != HUFD_NONE32 && !(want_short && edge_seg == st.unk_seg)) {
        careful_list[atomicAdd(careful_count, 1u)] = st.unk_seg;
    }
    *state = st;
    *result = rs;
}

/* one thread per item with few segments */
__global__ __launch_bounds__(256) void enc_scan_small_kernel(
    const hufd_enc_item *items,
    u32 n_items,
    const u32 *seg_bits,
    const u32 *seg_unk,
    u64 *seg_bitoff,
    u32 *careful_list,
    u32 *careful_count,
    hufd_enc_item_state *states,
    hufd_enc_result *results,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_enc_item it = items[i];
    if (it.n_segs > HUFD_SCAN_SMALL_MAX || it.tiny) {
        return;
    }
    const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
    u64 at = it.ovf_bits;
    u32 unk_seg = HUFD_NONE32, unk_idx = 0, unk_bits = 0, edge_seg = HUFD_NONE32;
    u64 unk_off = 0;
    for (u32 k = 0; k < it.n_segs; ++k) {
        const u32 s = it.first_seg + k;
        const u32 b = seg_bits[s];
        seg_bitoff[s] = at;
        if (at < cap_bits && cap_bits <= at + b) {
            edge_seg = s;
        }
        if (unk_seg == HUFD_NONE32 && seg_unk[s] != HUFD_NONE32) {
            unk_seg = s;
            unk_idx = seg_unk[s];
            unk_off = at;
            unk_bits = b;
        }
        at += b;
    }
    enc_finish_item(
        it, at, unk_seg, unk_idx, unk_off, unk_bits, edge_seg, careful_list, careful_count, &states[i], &results[i]);
}

/*
 * Items of at most HUFD_ENC_TINY_BYTES symbols (header-field sized strings): one THREAD per item
 * replays the reference loop (source/huffman.c:149-184) as it stands -- carried overflow first, a
 * symbol only while the output has a free byte, whatever of a code does not fit becomes the
 * overflow, the last byte completed with the low bits of eos_padding -- with the code bits
 * gathered in a 64-bit accumulator and stored a byte at a time, or a word at a time once the
 * output address is word aligned.  Segments, counts, offsets and output images cost such items
 * far more than their symbols do.
 */
constexpr u32 kTinyThreads = 256;

struct tiny_sink {
    u8 *out;
    u64 cap;
    u64 produced; /* bytes that have their place in the output (the last few of them may still wait in `held`) */
    u64 acc;  /* low nacc bits: code bits not yet stored, oldest highest */
    u32 nacc;
    /* whole words on their way to ONE 16-byte store (round 4: what these one-lane-one-item kernels pay for is memory
     * requests -- a line a lane and store -- and a 57-byte item was 14 word stores): words are held from a 16-byte
     * aligned place on while 16 bytes still fit, and go out together when the fourth is in */
    uint4 held;
    u32 n_held;

    __device__ void hold(u32 word) {
        held.x = n_held == 0 ? word : held.x;
        held.y = n_held == 1 ? word : held.y;
        held.z = n_held == 2 ? word : held.z;
        held.w = n_held == 3 ? word : held.w;
        ++n_held;
        if (n_held == 4) {
            *reinterpret_cast<uint4 *>(out + produced - 16) = held;
            n_held = 0;
        }
    }
    /* the words still held, one store each (the item ends, or stops, with fewer than four) */
    __device__ void release() {
        u8 *at = out + produced - 4 * n_held;
        if (n_held > 0) {
            *reinterpret_cast<u32 *>(at) = held.x;
        }
        if (n_held > 1) {
            *reinterpret_cast<u32 *>(at + 4) = held.y;
        }
        if (n_held > 2) {
            *reinterpret_cast<u32 *>(at + 8) = held.z;
        }
        n_held = 0;
    }

    /* stores what has gathered -- whole words once the output address is word aligned and four bytes still fit
     * (fewer than 32 gathered bits then wait: a one-lane walk pays per store), single bytes otherwise; true when
     * the output filled with bits of the last code left over */
    __device__ bool drain() {
        for (;;) {
            const bool wordy = cap - produced >= 4 && ((reinterpret_cast<uintptr_t>(out) + produced) & 3) == 0;
            if (wordy) {
                if (nacc < 32) {
                    return false;
                }
                const u32 word = __builtin_bswap32((u32)(acc >> (nacc - 32)));
                const bool fresh16 = ((reinterpret_cast<uintptr_t>(out) + produced) & 15) == 0 && cap - produced >= 16;
                produced += 4;
                nacc -= 32;
                if (n_held || fresh16) {
                    hold(word);
                } else {
                    *reinterpret_cast<u32 *>(out + produced - 4) = word;
                }
            } else {
                if (nacc < 8) {
                    return false;
                }
                out[produced] = (u8)(acc >> (nacc - 8));
                nacc -= 8;
                ++produced;
            }
            if (produced == cap) {
                return nacc != 0;
            }
        }
    }
    /* the whole bytes still waiting (there is room for them: they only wait while four bytes fit) */
    __device__ void finish() {
        release();
        while (nacc >= 8) {
            out[produced++] = (u8)(acc >> (nacc - 8));
            nacc -= 8;
        }
    }
};

__global__ __launch_bounds__(kTinyThreads) void enc_tiny_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const u32 *tiny_items,
    u32 n_tiny,
    const u8 *d_in,
    u8 *d_out,
    hufd_enc_result *results,
    u32 length_only) {

    u64 *tab = reinterpret_cast<u64 *>(dyn_lds); /* [256] low half: code, high half: length */
    for (u32 i = threadIdx.x; i < 256; i += kTinyThreads) {
        tab[i] = tb.enc_table[i];
    }
    __syncthreads();
    const u32 t = blockIdx.x * kTinyThreads + threadIdx.x;
    if (t >= n_tiny) {
        return;
    }
    const u32 item = tiny_items[t];
    const hufd_enc_item it = items[item];
    const u8 *in = d_in + it.in_off;
    const u32 n = (u32)it.in_len;

    hufd_enc_result rs;
    rs.status = HUFD_ENC_OK;
    rs.ovf_pattern = 0;
    rs.ovf_bits = 0;
    rs.reserved = 0;
    rs.consumed = n;
    rs.produced = 0;
    rs.total_bits = it.ovf_bits;

    /* aligned 16-byte reads of the symbols, handed out one at a time (what counts is the number of requests) */
    const u32 lead = (u32)(reinterpret_cast<uintptr_t>(in) & 15u);
    const uint4 *blocks = reinterpret_cast<const uint4 *>(in - lead);
    uint4 block = uint4{0, 0, 0, 0};
    u32 block_at = 0xFFFFFFFFu; /* which block `block` holds (the general loop may start anywhere in the item) */
    const u32 last_block = n ? (lead + n - 1) >> 4 : 0u;
    auto symbol = [&](u32 k) -> u32 {
        const u32 a = lead + k;
        if ((a >> 4) != block_at) {
            block_at = a >> 4;
            block = blocks[block_at];
        }
        const u32 w = (a >> 2) & 3u;
        const u32 word = w == 0 ? block.x : (w == 1 ? block.y : (w == 2 ? block.z : block.w));
        return (word >> ((a & 3u) * 8)) & 0xFFu;
    };

    if (length_only) {
        u64 bits = it.ovf_bits;
        for (u32 k = 0; k < n; ++k) {
            bits += (u32)(tab[symbol(k)] >> 32);
        }
        rs.total_bits = bits;
        rs.produced = (bits + 7) >> 3;
        results[item] = rs;
        return;
    }

    /* The stretch of the item where the output has room to spare: the reference's loop (source/huffman.c:161-184) is
     * there code after code into the accumulator, a word out whenever 32 bits have gathered -- four of them as ONE
     * 16-byte store, at whatever address (the memory system takes any alignment; these one-lane-one-item kernels pay for
     * requests) -- and none of its questions about the next free byte.  The kernel is bound by the instructions a symbol
     * costs (a wave runs as long as its longest item): the symbols are taken a 16-byte block at a time, each at a place
     * in the block the compiler knows, those in front of the item, behind it or behind the stretch as codes of no bits.
     * The stretch ends at a symbol without a code, or 64 bits short of the output's end: the general loop below takes
     * over there with what has gathered, and everything the reference says about running out of room is its to say. */
    u32 fast_done = 0, fast_bits = 0, fast_produced = 0, fast_nacc = 0;
    u64 fast_acc = 0;
    if (it.ovf_bits == 0 && n != 0 && it.out_cap >= 8) {
        u8 *out = d_out + it.out_off;
        const u32 cap_bits = it.out_cap > 0x0FFFFFFFull ? 0x7FFFFFF8u : (u32)it.out_cap * 8u;
        u64 acc = 0;
        u32 nacc = 0, produced = 0, n_held = 0, bits = 0, done = 0;
        uint4 held = uint4{0, 0, 0, 0};
        bool stop = false;
        const u32 end = lead + n;
        uint4 ahead = blocks[0];
        for (u32 b = 0; b * 16 < end && !stop; ++b) {
            const uint4 blk = ahead;
            if (b < last_block) {
                ahead = blocks[b + 1]; /* (looked at sixteen symbols on: its trip to memory is not waited for) */
            }
            const u32 wds[4] = {blk.x, blk.y, blk.z, blk.w};
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                const u32 idx = b * 16 + j;
                const bool mine = idx - lead < n && !stop; /* (unsigned: also false in front of the item) */
                const u64 ent = tab[(wds[j >> 2] >> (8 * (j & 3))) & 0xFFu];
                const u32 code_len = (u32)(ent >> 32);
                const bool take = mine && code_len != 0 && bits + code_len + 64u <= cap_bits;
                stop = stop || (mine && !take);
                const u32 len = take ? code_len : 0u;
                done += take ? 1u : 0u;
                bits += len;
                acc = (acc << len) | (take ? (u32)ent : 0u);
                nacc += len;
                if (nacc >= 32) {
                    const u32 word = __builtin_bswap32((u32)(acc >> (nacc - 32)));
                    nacc -= 32;
                    held.x = n_held == 0 ? word : held.x;
                    held.y = n_held == 1 ? word : held.y;
                    held.z = n_held == 2 ? word : held.z;
                    held.w = n_held == 3 ? word : held.w;
                    if (++n_held == 4) {
                        unaligned_uint4 v = {held.x, held.y, held.z, held.w};
                        *reinterpret_cast<unaligned_uint4 *>(out + produced) = v;
                        produced += 16;
                        n_held = 0;
                    }
                }
            }
        }
        if (n_held > 0) {
            reinterpret_cast<unaligned_u32 *>(out + produced)->x = held.x;
        }
        if (n_held > 1) {
            reinterpret_cast<unaligned_u32 *>(out + produced + 4)->x = held.y;
        }
        if (n_held > 2) {
            reinterpret_cast<unaligned_u32 *>(out + produced + 8)->x = held.z;
        }
        fast_produced = produced + 4 * n_held;
        fast_done = done;
        fast_bits = bits;
        fast_nacc = nacc;
        fast_acc = nacc ? acc & ((1ull << nacc) - 1) : 0;
    }

    tiny_sink sink;
    sink.out = d_out + it.out_off;
    sink.cap = it.out_cap;
    sink.produced = fast_produced;
    sink.acc = fast_acc;
    sink.nacc = fast_nacc;
    sink.held = uint4{0, 0, 0, 0};
    sink.n_held = 0;
    bool stopped = false;
    if (it.ovf_bits) {
        if (sink.cap == 0) {
            /* no byte to put the carried bits in (source/huffman.c:150-152): they stay carried */
            rs.status = HUFD_ENC_SHORT;
            rs.consumed = 0;
            rs.ovf_bits = it.ovf_bits;
            rs.ovf_pattern = it.ovf_pattern;
            stopped = true;
        } else {
            sink.acc = it.ovf_pattern;
            sink.nacc = it.ovf_bits;
            if (sink.drain()) {
                rs.status = HUFD_ENC_SHORT;
                rs.consumed = 0;
                rs.ovf_bits = sink.nacc;
                rs.ovf_pattern = (u32)(sink.acc & ((1ull << sink.nacc) - 1));
                stopped = true;
            }
        }
    }
    u64 bits = it.ovf_bits + fast_bits;
    for (u32 k = fast_done; k < n && !stopped; ++k) {
        if (sink.produced == sink.cap) { /* source/huffman.c:162-164 */
            rs.status = HUFD_ENC_SHORT;
            rs.consumed = k;
            stopped = true;
            break;
        }
        const u64 ent = tab[symbol(k)];
        const u32 len = (u32)(ent >> 32);
        if (len == 0) { /* source/huffman.c:62-64: the symbol is consumed, the byte under construction is not written */
            rs.status = HUFD_ENC_UNKNOWN;
            rs.consumed = k + 1;
            stopped = true;
            break;
        }
        bits += len;
        sink.acc = (sink.acc << len) | (u32)ent;
        sink.nacc += len;
        if (sink.drain()) { /* source/huffman.c:88-100 */
            rs.status = HUFD_ENC_SHORT;
            rs.consumed = k + 1;
            rs.ovf_bits = sink.nacc;
            rs.ovf_pattern = (u32)(sink.acc & ((1ull << sink.nacc) - 1));
            stopped = true;
            break;
        }
    }
    if (rs.status != HUFD_ENC_SHORT) {
        sink.finish(); /* (before a symbol without a code too: the reference had written those bytes) */
    } else {
        sink.release();
    }
    if (!stopped && sink.nacc) { /* source/huffman.c:178-184 */
        const u32 room = 8 - sink.nacc;
        sink.out[sink.produced] = (u8)((sink.acc << room) | (it.eos_padding & ((1u << room) - 1)));
        ++sink.produced;
    }
    rs.produced = sink.produced;
    rs.total_bits = bits;
    results[item] = rs;
}

/*
 * One item of at most HUFD_ENC_BLOCK_MAX_BYTES symbols, one workgroup (of 256 lanes up to HUFD_ENC_BLOCK_BYTES), ONE launch: count, offsets, outcome and bits in
 * one go (the host-pointer calls' road for inputs beyond a header field: with segments the same call is a plan
 * upload and four or five launches).  A thread takes 16 symbols; the outcome is the reference's, in closed form as
 * in enc_finish_item: with `o` carried bits, T bits in all, room for A bytes, first symbol without a code `u` at bit
 * `before_u`: UNKNOWN_SYMBOL iff before_u < 8A (source/huffman.c:62-64: whole bytes in front of it stay, the byte
 * in flight is lost), else SUCCESS iff no such symbol and T <= 8A (padded, :178-184), else SHORT_BUFFER with the
 * symbol whose last bit reaches bit 8A consumed and what of its code lies behind that bit carried (:88-100).
 */
constexpr u32 kBlockEncThreads = 256;      /* up to HUFD_ENC_BLOCK_BYTES symbols */
constexpr u32 kBlockEncWideThreads = 1024; /* up to HUFD_ENC_BLOCK_MAX_BYTES (round 3): the same code, four times the lanes */
static_assert(kBlockEncThreads * 16 == HUFD_ENC_BLOCK_BYTES && kBlockEncWideThreads * 16 == HUFD_ENC_BLOCK_MAX_BYTES, "16 symbols a lane");

struct enc_block_shared {
    u64 unk_key;   /* lowest (index << 32 | bits in front) of a symbol without a code */
    u32 short_consumed, short_ovf_bits, short_ovf_pattern, pad;
    u32 slots[kBlockEncWideThreads / 64];
};

template <u32 kBlockEncThreads>
__global__ __launch_bounds__(kBlockEncThreads) void enc_block_kernel(
    hufd_tables tb,
    const hufd_enc_item *item_ptr,
    const u8 *d_in,
    u8 *d_out,
    hufd_enc_result *result,
    u32 img_words,
    u32 length_only) {

    u32 *img = reinterpret_cast<u32 *>(dyn_lds);
    u64 *tab = reinterpret_cast<u64 *>(dyn_lds + round16(img_words * 4));
    enc_block_shared *sh = reinterpret_cast<enc_block_shared *>(tab + 256);
    const u32 tid = threadIdx.x;
    if (tid < 256) {
        tab[tid] = tb.enc_table[tid];
    }
    const hufd_enc_item it = *item_ptr;
    const u32 n = (u32)it.in_len;
    const u8 *src = d_in + it.in_off;
    u8 *out = d_out + it.out_off;
    const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
    {
        const uint4 zero = {0, 0, 0, 0};
        for (u32 i = tid; i < img_words / 4; i += kBlockEncThreads) {
            reinterpret_cast<uint4 *>(img)[i] = zero;
        }
    }
    if (tid == 0) {
        sh->unk_key = kNoBit;
        sh->short_consumed = 0;
        sh->short_ovf_bits = 0;
        sh->short_ovf_pattern = 0;
    }
    const u32 base = tid * 16;
    const u32 valid = base < n ? (n - base < 16 ? n - base : 16) : 0;
    u32 gw[4] = {0, 0, 0, 0};
    if (valid) {
        load_group(src + base, valid, ((uintptr_t)src & 15u) == 0, gw);
    }
    __syncthreads();

    u64 e[16];
    u32 lane_bits = 0;
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        e[j] = j < valid ? tab[group_byte(gw, j)] : 0;
        lane_bits += (u32)(e[j] >> 32);
    }
    u32 symbols_bits;
    const u32 rel0 = it.ovf_bits + block_exclusive_sum<kBlockEncThreads>(lane_bits, sh->slots, symbols_bits);
    const u64 total = (u64)it.ovf_bits + symbols_bits;

    /* the image's bit 8 * mis is the stream's first bit: whole 16-byte rows of the output leave aligned */
    const u32 mis = (u32)((uintptr_t)out & 15u);
    u8 *gbase = out - mis;
    if (tid == 0 && it.ovf_bits && !length_only) {
        image_or_bits(img, 8 * mis, it.ovf_pattern, it.ovf_bits);
    }
    {
        u32 rel = rel0;
        u32 wi = (8 * mis + rel) >> 5, nb = (8 * mis + rel) & 31;
        u64 acc = 0;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            const u32 len = (u32)(e[j] >> 32);
            const u32 pat = (u32)e[j];
            if (j < valid) {
                if (len == 0) {
                    atomicMin(&sh->unk_key, ((u64)(base + j) << 32) | rel);
                } else {
                    const u32 after = rel + len;
                    if (rel < cap_bits && after >= cap_bits) {
                        /* the symbol whose last bit reaches the capacity edge (source/huffman.c:88-98): only one can */
                        sh->short_consumed = base + j + 1;
                        sh->short_ovf_bits = (u32)(after - cap_bits);
                        sh->short_ovf_pattern = pat & (u32)((1ull << (after - cap_bits)) - 1);
                    }
                    if (!length_only) {
                        acc = (acc << len) | pat;
                        nb += len;
                        if (nb >= 32) {
                            atomicOr(&img[wi], (u32)(acc >> (nb - 32)));
                            ++wi;
                            nb -= 32;
                            acc &= (1ull << nb) - 1;
                        }
                    }
                    rel = after;
                }
            }
        }
        if (nb && !length_only) {
            const u32 tail = (u32)(acc << (32 - nb));
            if (tail) {
                atomicOr(&img[wi], tail);
            }
        }
    }
    __syncthreads();

    const u64 unk_key = sh->unk_key;
    const bool has_unk = unk_key != kNoBit;
    const u32 unk_idx = (u32)(unk_key >> 32), unk_before = (u32)unk_key;
    hufd_enc_result rs;
    rs.reserved = 0;
    rs.ovf_pattern = 0;
    rs.ovf_bits = 0;
    rs.total_bits = total;
    if (length_only) {
        rs.status = HUFD_ENC_OK;
        rs.consumed = n;
        rs.produced = (total + 7) >> 3;
    } else if (has_unk && unk_before < cap_bits) {
        rs.status = HUFD_ENC_UNKNOWN;
        rs.consumed = unk_idx + 1;
        rs.produced = unk_before >> 3;
    } else if (!has_unk && total <= cap_bits) {
        rs.status = HUFD_ENC_OK;
        rs.consumed = n;
        rs.produced = (total + 7) >> 3;
        const u32 pad_bits = (u32)((8 - (total & 7)) & 7);
        if (tid == 0 && pad_bits) {
            image_or_bits(img, 8 * mis + (u32)total, it.eos_padding & ((1u << pad_bits) - 1), pad_bits);
        }
    } else {
        rs.status = HUFD_ENC_SHORT;
        rs.produced = it.out_cap;
        if (it.ovf_bits >= cap_bits) {
            /* the carried bits alone fill the room (source/huffman.c:149-156) */
            rs.consumed = 0;
            rs.ovf_bits = (u32)(it.ovf_bits - cap_bits);
            rs.ovf_pattern = rs.ovf_bits ? (it.ovf_pattern & (u32)((1ull << rs.ovf_bits) - 1)) : 0;
        } else {
            rs.consumed = sh->short_consumed;
            rs.ovf_bits = sh->short_ovf_bits;
            rs.ovf_pattern = sh->short_ovf_pattern;
        }
    }
    __syncthreads();
    if (!length_only && rs.produced) {
        image_store<kBlockEncThreads>(img, gbase, mis, mis + (u32)rs.produced);
    }
    if (tid == 0) {
        *result = rs;
    }
}

/*
 * One workgroup per item with many segments.  Each wave owns a contiguous range of the item's
 * segments and reads it 64 x 8 at a time, all eight loads of a lane in flight together (one load
 * per trip left this kernel waiting a memory round trip per 64 segments): first pass sums the
 * range, the 16 range sums are scanned, second pass scans inside the range with a running carry.
 */
__global__ __launch_bounds__(HUFD_SCAN_LARGE_THREADS) void enc_scan_large_kernel(
    const hufd_enc_item *items,
    const u32 *large_items,
    const u32 *seg_bits,
    const u32 *seg_unk,
    u64 *seg_bitoff,
    u32 *careful_list,
    u32 *careful_count,
    hufd_enc_item_state *states,
    hufd_enc_result *results,
    u32 all_coded /* every symbol has a code: seg_unk need not be read */,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    constexpr u32 T = HUFD_SCAN_LARGE_THREADS, W = T / kWave, U = 16; /* U independent loads a lane and trip */
    u64 *wave_tot = reinterpret_cast<u64 *>(dyn_lds);      /* [W] */
    u64 *unk_off = wave_tot + W;                            /* [1] */
    u32 *first_unk = reinterpret_cast<u32 *>(unk_off + 1);  /* [1] lowest segment with a bad symbol */
    u32 *edge_seg = first_unk + 1;                          /* [1] segment holding the capacity edge */

    const u32 i = large_items[blockIdx.x];
    const hufd_enc_item it = items[i];
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) {
        *first_unk = HUFD_NONE32;
        *unk_off = 0;
        *edge_seg = HUFD_NONE32;
    }
    const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
    /* ranges are multiples of 64 * U segments so that every read is a full coalesced row */
    const u32 per = ((it.n_segs + W - 1) / W + kWave * U - 1) / (kWave * U) * (kWave * U);
    const u32 lo = wave * per < it.n_segs ? wave * per : it.n_segs;
    const u32 hi = lo + per < it.n_segs ? lo + per : it.n_segs;
    const u32 *bits_in = seg_bits + it.first_seg, *unk_in = seg_unk + it.first_seg;

    u64 mine = 0;
    u32 my_unk = HUFD_NONE32;
    for (u32 base = lo; base < hi; base += kWave * U) {
        u32 b[U], u[U];
#pragma unroll
        for (u32 j = 0; j < U; ++j) {
            const u32 k = base + j * kWave + lane;
            b[j] = k < hi ? bits_in[k] : 0u;
            u[j] = (k < hi && !all_coded) ? unk_in[k] : HUFD_NONE32;
        }
#pragma unroll
        for (u32 j = 0; j < U; ++j) {
            mine += b[j];
            if (my_unk == HUFD_NONE32 && u[j] != HUFD_NONE32) {
                my_unk = it.first_seg + base + j * kWave + lane;
            }
        }
    }
#pragma unroll
    for (u32 d = kWave / 2; d > 0; d >>= 1) {
        mine += __shfl_xor(mine, d);
    }
    my_unk = wave_min(my_unk);
    if (lane == 0) {
        wave_tot[wave] = mine;
    }
    __syncthreads();
    if (lane == 0 && my_unk != HUFD_NONE32) {
        atomicMin(first_unk, my_unk);
    }
    u64 carry = it.ovf_bits, total = it.ovf_bits;
    for (u32 w = 0; w < W; ++w) {
        const u64 t = wave_tot[w];
        carry += w < wave ? t : 0;
        total += t;
    }
    __syncthreads();
    const u32 us = *first_unk;
    for (u32 base = lo; base < hi; base += kWave * U) {
        u32 b[U];
#pragma unroll
        for (u32 j = 0; j < U; ++j) {
            const u32 k = base + j * kWave + lane;
            b[j] = k < hi ? bits_in[k] : 0u;
        }
#pragma unroll
        for (u32 j = 0; j < U; ++j) {
            const u32 k = base + j * kWave + lane;
            const u32 incl = wave_inclusive_sum_dpp(b[j], lane);
            if (k < hi) {
                const u64 at = carry + incl - b[j];
                seg_bitoff[it.first_seg + k] = at;
                if (it.first_seg + k == us) {
                    *unk_off = at;
                }
                if (at < cap_bits && cap_bits <= at + b[j]) {
                    *edge_seg = it.first_seg + k;
                }
            }
            carry += __shfl(incl, kWave - 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const u32 ui = us != HUFD_NONE32 ? seg_unk[us] : 0;
        const u32 ub = us != HUFD_NONE32 ? seg_bits[us] : 0;
        enc_finish_item(
            it, total, us, ui, *unk_off, ub, *edge_seg, careful_list, careful_count, &states[i], &results[i]);
    }
}

/* ------------------------------------------------------------------ encode: pack */

struct enc_pack_shared {
    u64 unk_key;       /* lowest (index in segment << 32 | bit offset in segment) of a symbol without a code */
    u64 unk_before;    /* stream bit at which the item's first bad symbol sits */
    u64 short_consumed;
    u32 short_found;
    u32 short_ovf_bits;
    u32 short_ovf_pattern;
    u32 halo_unknown;
};



/* OR `len` (0..64) right-aligned bits of `value` into the image at bit q: at most three words. */
__device__ __forceinline__ void image_or_quad(u32 *img, u32 q, u64 value, u32 len) {
    const u64 left = len ? value << (64 - len) : 0;
    const u32 sh = q & 31, w = q >> 5;
    const u32 w0 = (u32)(left >> (32 + sh));
    const u32 w1 = (u32)(left >> sh);
    const u32 w2 = (u32)(left << (32 - sh));
    if (w0) {
        atomicOr(&img[w], w0);
    }
    if (w1) {
        atomicOr(&img[w + 1], w1);
    }
    if (w2) {
        atomicOr(&img[w + 2], w2);
    }
}

/* where a segment's bits go: shared by the two pack kernels */
struct pack_geometry {
    u64 p0;       /* stream bit of the segment's first code */
    u64 pend;     /* stream bit after its last code */
    u64 pa;       /* stream bit where the workgroup's image starts (0 for the item's first segment) */
    u64 cap_bits; /* the item's capacity in bits */
    u64 j0;       /* stream byte of image byte `mis` */
    u8 *gbase;    /* output address of image byte 0, 16-byte aligned */
    u32 mis;
    u32 q0;       /* image bit of stream bit p0 */
    u32 cap_rel;  /* capacity edge relative to p0; 0 disables the crossing test */
    bool last_seg, want_short, is_unk_seg, careful, skip;
};

__device__ __forceinline__ pack_geometry pack_geometry_of(
    const hufd_tables &tb,
    const hufd_enc_seg seg,
    u32 s,
    const hufd_enc_item &it,
    const hufd_enc_item_state &st,
    u64 p0,
    u32 bits,
    u8 *d_out) {
    pack_geometry g;
    g.p0 = p0;
    g.pend = p0 + bits;
    g.pa = seg.index == 0 ? 0 : p0;
    g.cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
    /* image byte 0 sits on a 16-byte boundary of the output */
    u8 *out_ptr = d_out + it.out_off;
    g.j0 = g.pa >> 3;
    g.mis = (u32)((uintptr_t)(out_ptr + g.j0) & 15u);
    g.gbase = out_ptr + g.j0 - g.mis;
    g.q0 = (u32)(p0 - 8 * g.j0) + 8 * g.mis;
    g.last_seg = (seg.flags & 2u) != 0;
    g.want_short = st.status == HUFD_ENC_SHORT || st.status == HUFD_ENC_DECIDE;
    g.cap_rel = 0;
    if (g.want_short && g.cap_bits > p0) {
        g.cap_rel = g.cap_bits - p0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(g.cap_bits - p0);
    }
    g.is_unk_seg = s == st.unk_seg;
    /*
     * Only the segment that holds the capacity edge or the first symbol without a code needs
     * to look at symbols one by one; every other segment takes the branch-free path (codes of
     * at most 16 bits: four of them always fit a 64-bit register).
     */
    const bool edge_here = g.want_short && g.cap_bits > p0 && g.cap_bits <= g.pend;
    g.careful = tb.enc_max_bits > 16 || edge_here || g.is_unk_seg;
    /* past the item's first symbol without a code the reference never gets */
    g.skip = st.unk_seg != HUFD_NONE32 && s > st.unk_seg;
    return g;
}

/*
 * Completes the last byte the workgroup owns: with the head of the next segment's codes
 * (fetched with the segment: `halo` holds its first eight symbols), or with the padding
 * when the item ends here (huffman.c:178-184).  One lane; table work only.
 */
template <typename Lookup>
__device__ __forceinline__ void pack_last_byte(
    u32 *img,
    enc_pack_shared *sh,
    const pack_geometry &g,
    const hufd_enc_seg seg,
    const hufd_enc_item &it,
    const hufd_enc_item_state &st,
    const u32 (&halo)[2],
    Lookup lookup /* symbol -> length << 32 | code */) {
    u32 need = (u32)((8 - (g.pend & 7)) & 7);
    u32 q = g.q0 + (u32)(g.pend - g.p0);
    if (need && !g.last_seg) {
        const u32 n = seg.next_len < 8 ? seg.next_len : 8;
        for (u32 j = 0; j < n && need; ++j) {
            const u64 ent = lookup((halo[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            const u32 len = (u32)(ent >> 32);
            if (len == 0) {
                sh->halo_unknown = 1;
                break;
            }
            image_or_bits(img, q, (u32)ent, len);
            q += len;
            need = len >= need ? 0 : need - len;
        }
    }
    if (need && !sh->halo_unknown && st.status == HUFD_ENC_OK) {
        /* only reachable when the item's remaining symbols ran out: pad */
        const u32 pad_bits = (u32)((8 - (st.total_bits & 7)) & 7);
        const u32 qpad = g.q0 + (u32)(st.total_bits - g.p0);
        if (pad_bits) {
            image_or_bits(img, qpad, it.eos_padding & ((1u << pad_bits) - 1), pad_bits);
        }
    }
}

/* After the barrier: copy the owned bytes out and, where this segment decides it, the result. */
__device__ __forceinline__ void pack_write_out(
    const u32 *img,
    const enc_pack_shared *sh,
    const pack_geometry &g,
    const hufd_enc_seg seg,
    const hufd_enc_item &it,
    const hufd_enc_item_state &st,
    hufd_enc_result *results) {

    u32 status = st.status;
    if (status == HUFD_ENC_DECIDE) {
        /* only the segment holding the bad symbol can tell which stop comes first;
         * for the segments before it neither limit binds */
        status = (g.is_unk_seg && sh->unk_before < g.cap_bits) ? HUFD_ENC_UNKNOWN : HUFD_ENC_SHORT;
    }
    u64 limit_bytes;
    if (status == HUFD_ENC_OK) {
        limit_bytes = (st.total_bits + 7) >> 3;
    } else if (status == HUFD_ENC_UNKNOWN && g.is_unk_seg) {
        limit_bytes = sh->unk_before >> 3; /* the partial byte in flight is lost (huffman.c:62-64) */
    } else {
        limit_bytes = it.out_cap;
    }

    u64 jhi;
    if (g.last_seg) {
        jhi = status == HUFD_ENC_OK ? (st.total_bits + 7) >> 3 : g.pend >> 3;
    } else {
        jhi = sh->halo_unknown ? g.pend >> 3 : (g.pend + 7) >> 3;
    }
    if (jhi > limit_bytes) {
        jhi = limit_bytes;
    }
    const u64 jlo = (g.pa + 7) >> 3;
    if (jhi > jlo) {
        image_store<HUFD_ENC_THREADS>(img, g.gbase, (u32)(jlo - g.j0) + g.mis, (u32)(jhi - g.j0) + g.mis);
    }

    if (threadIdx.x == 0) {
        hufd_enc_result *rs = &results[seg.item];
        if (status == HUFD_ENC_UNKNOWN && g.is_unk_seg) {
            rs->status = HUFD_ENC_UNKNOWN;
            rs->consumed = (u64)seg.index * HUFD_ENC_SEG_BYTES + (u32)(sh->unk_key >> 32) + 1;
            rs->produced = limit_bytes;
            rs->ovf_bits = 0;
            rs->ovf_pattern = 0;
        } else if (sh->short_found && status == HUFD_ENC_SHORT) {
            rs->status = HUFD_ENC_SHORT;
            rs->produced = it.out_cap;
            rs->consumed = sh->short_consumed;
            rs->ovf_bits = sh->short_ovf_bits;
            rs->ovf_pattern = sh->short_ovf_pattern;
        }
    }
}

/*
 * The per-symbol packer: any code length up to 32, finds the symbol that crosses the
 * capacity edge and the position of the item's first symbol without a code.  Used for
 * every segment of a coder with codes longer than 16 bits, and otherwise only for the
 * (at most two per item) segments listed by the scan kernel.
 *   list == NULL : workgroup b handles segment b, b + gridDim.x, ...
 *   list != NULL : the segments list[0 .. *list_count)
 */
__global__ __launch_bounds__(HUFD_ENC_THREADS) void enc_pack_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const hufd_enc_item_state *states,
    const hufd_enc_seg *segs,
    const u32 *seg_bits,
    const u64 *seg_bitoff,
    const u8 *d_in,
    u8 *d_out,
    hufd_enc_result *results,
    u32 img_words,
    u32 n_segs,
    const u32 *list,
    const u32 *list_count,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    u32 *img = reinterpret_cast<u32 *>(dyn_lds);
    u64 *tab = reinterpret_cast<u64 *>(dyn_lds + round16(img_words * 4));
    u32 *slots = reinterpret_cast<u32 *>(tab + 256); /* [8] */
    enc_pack_shared *sh = reinterpret_cast<enc_pack_shared *>(slots + 8);

    const u32 tid = threadIdx.x;
    tab[tid] = tb.enc_table[tid];
    const u32 n_work = list ? *list_count : n_segs;

    for (u32 work = blockIdx.x; work < n_work; work += gridDim.x) {
        const u32 s = list ? list[work] : work;
        const hufd_enc_seg seg = segs[s];
        const u8 *src = d_in + seg.in_off;
        u32 gw[kGroupsPerLane][4], gvalid[kGroupsPerLane];
        load_segment_groups(src, seg.len, gw, gvalid);
        u32 halo[2] = {0, 0};
        if (tid == 0 && seg.next_len) {
            const u32 n = seg.next_len < 8 ? seg.next_len : 8;
            for (u32 j = 0; j < n; ++j) {
                halo[j >> 2] |= (u32)src[HUFD_ENC_SEG_BYTES + j] << (8 * (j & 3));
            }
        }
        const hufd_enc_item it = items[seg.item];
        const hufd_enc_item_state st = states[seg.item];
        const pack_geometry g = pack_geometry_of(tb, seg, s, it, st, seg_bitoff[s], seg_bits[s], d_out);
        /* with a list the stream kernel has done every segment that is not on it */
        const bool mine = !g.skip && (list || g.careful || tb.enc_max_bits > 16);

        __syncthreads(); /* the previous segment's copy-out is done with the image */
        {
            const uint4 zero = {0, 0, 0, 0};
            for (u32 i = tid; i < img_words / 4; i += HUFD_ENC_THREADS) {
                reinterpret_cast<uint4 *>(img)[i] = zero;
            }
        }
        if (tid == 0) {
            sh->unk_key = kNoBit;
            sh->unk_before = kNoBit;
            sh->short_found = 0;
            sh->halo_unknown = 0;
        }
        __syncthreads();
        if (!mine) {
            continue;
        }
        if (tid == 0 && seg.index == 0 && it.ovf_bits) {
            image_or_bits(img, 8 * g.mis, it.ovf_pattern, it.ovf_bits);
        }

        const u64 seg_off = (u64)seg.index * HUFD_ENC_SEG_BYTES;
        u32 carry = 0; /* bits of this segment already placed */
#pragma unroll
        for (u32 iter = 0; iter < kGroupsPerLane; ++iter) {
            const u32 base = (iter * HUFD_ENC_THREADS + tid) * 16;
            const u32 valid = gvalid[iter];
            u64 e[16];
            u32 lane_bits = 0;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                e[j] = j < valid ? tab[group_byte(gw[iter], j)] : 0;
                lane_bits += (u32)(e[j] >> 32);
            }
            u32 total;
            u32 rel = carry + block_exclusive_sum<HUFD_ENC_THREADS>(lane_bits, slots, total);
            carry += total;

            /* the lane's codes go out as whole words; its first and last word are shared
             * with neighbours, so every word is OR-ed into the zeroed image */
            u32 q = g.q0 + rel;
            u32 wi = q >> 5, nb = q & 31;
            u64 acc = 0;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
                const u32 len = (u32)(e[j] >> 32);
                const u32 pat = (u32)e[j];
                if (j < valid) {
                    if (len == 0) {
                        if (g.is_unk_seg) {
                            atomicMin(&sh->unk_key, ((u64)(base + j) << 32) | rel);
                        }
                    } else {
                        const u32 after = rel + len;
                        if (rel < g.cap_rel && after >= g.cap_rel) {
                            /* first symbol whose last bit reaches the capacity edge (huffman.c:88-98) */
                            sh->short_found = 1;
                            sh->short_consumed = seg_off + base + j + 1;
                            sh->short_ovf_bits = after - g.cap_rel;
                            sh->short_ovf_pattern = pat & (u32)((1ull << (after - g.cap_rel)) - 1);
                        }
                        acc = (acc << len) | pat;
                        nb += len;
                        rel = after;
                        if (nb >= 32) {
                            atomicOr(&img[wi], (u32)(acc >> (nb - 32)));
                            ++wi;
                            nb -= 32;
                            acc &= (1ull << nb) - 1;
                        }
                    }
                }
            }
            if (nb) {
                const u32 tail = (u32)(acc << (32 - nb));
                if (tail) {
                    atomicOr(&img[wi], tail);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (sh->unk_key != kNoBit) {
                sh->unk_before = g.p0 + (u32)sh->unk_key;
            }
            pack_last_byte(img, sh, g, seg, it, st, halo, [&](u32 sym) { return tab[sym]; });
        }
        __syncthreads();
        pack_write_out(img, sh, g, seg, it, st, results);
    }
}

/* asks for a segment's symbols: 16-byte chunk c of the segment goes to inbuf + 16 c (LDS-DMA) */
__device__ __forceinline__ void stream_request(const u8 *d_in, u8 *inbuf, u64 in_off, u32 len, u32 next_len) {
    const u8 *src = d_in + in_off;
    if (((uintptr_t)src & 15u) != 0) {
        return; /* unaligned input: read with plain loads when its turn comes */
    }
    const u32 chunks = (len + 15) / 16 + (next_len ? 1 : 0);
#pragma unroll
    for (u32 j = 0; j <= kGroupsPerLane; ++j) {
        const u32 c = j * HUFD_ENC_THREADS + threadIdx.x;
        if (c < chunks && c <= HUFD_ENC_SEG_BYTES / 16) {
            lds_dma16(src + 16 * c, inbuf + 16 * c);
        }
    }
}

/*
 * The streaming packer for coders whose codes fit 16 bits (the reference's test coder
 * has at most 10).  Persistent workgroups: segment blockIdx.x, + gridDim.x, ...  While a
 * segment is packed, the symbols of the workgroup's next segment travel from HBM straight
 * into an LDS buffer (LDS-DMA, no registers), so the memory round trip hides behind the
 * packing.  Per segment and lane: 64 table lookups, codes merged pairwise to quads in
 * registers, one wave scan per 32 symbols, <= 3 LDS ORs per quad -- no per-symbol branch.
 * Segments that need the per-symbol treatment (capacity edge, symbol without a code) are
 * left to enc_pack_kernel.
 */
__global__ __launch_bounds__(HUFD_ENC_THREADS) void enc_pack_stream_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const hufd_enc_item_state *states,
    const hufd_enc_seg *segs,
    const u32 *seg_bits,
    const u64 *seg_bitoff,
    const u8 *d_in,
    u8 *d_out,
    hufd_enc_result *results,
    u32 img_words,
    u32 n_segs,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    constexpr u32 kInBytes = HUFD_ENC_SEG_BYTES + 16; /* a segment + the chunk holding the next one's head */
    u32 *img = reinterpret_cast<u32 *>(dyn_lds);
    u8 *inbuf = dyn_lds + round16(img_words * 4);
    u32 *tab32 = reinterpret_cast<u32 *>(inbuf + kInBytes); /* [256] length << 16 | code */
    u32 *slots = tab32 + 256;                                /* [8] wave totals, first half then second half */
    enc_pack_shared *sh = reinterpret_cast<enc_pack_shared *>(slots + 8);

    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    {
        const u64 ent = tb.enc_table[tid];
        tab32[tid] = ((u32)(ent >> 32) << 16) | ((u32)ent & 0xFFFFu);
    }

    u32 s = blockIdx.x;
    if (s >= n_segs) {
        return;
    }
    hufd_enc_seg seg = uniform_seg(&segs[s]);
    /* descriptors are read with a clamped index: a select between memory objects would push them to scratch */
    hufd_enc_seg seg_next = uniform_seg(&segs[s + gridDim.x < n_segs ? s + gridDim.x : n_segs - 1]);
    stream_request(d_in, inbuf, seg.in_off, seg.len, seg.next_len);
    __syncthreads(); /* tables staged, first segment landed (the barrier drains the DMA) */

    for (;;) {
        HUFD_STAMP(2, 0);
        const bool more = s + gridDim.x < n_segs;
        const hufd_enc_item it = items[seg.item];
        const hufd_enc_item_state st = states[seg.item];
        const pack_geometry g = pack_geometry_of(tb, seg, s, it, st, seg_bitoff[s], seg_bits[s], d_out);
        const u8 *src = d_in + seg.in_off;
        const bool from_lds = ((uintptr_t)src & 15u) == 0;

        /* symbols out of the buffer (or memory), image cleared */
        u32 gw[kGroupsPerLane][4], gvalid[kGroupsPerLane];
        u32 halo[2] = {0, 0};
        if (from_lds) {
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                const u32 base = (gi * HUFD_ENC_THREADS + tid) * 16;
                gvalid[gi] = base < seg.len ? (seg.len - base < 16 ? seg.len - base : 16) : 0;
                const uint4 v = *reinterpret_cast<const uint4 *>(inbuf + base);
                gw[gi][0] = v.x;
                gw[gi][1] = v.y;
                gw[gi][2] = v.z;
                gw[gi][3] = v.w;
                if (gvalid[gi] < 16) {
                    /* bytes past the segment are whatever the buffer held: mask them */
#pragma unroll
                    for (u32 c = 0; c < 4; ++c) {
                        const u32 keep = gvalid[gi] > 4 * c ? gvalid[gi] - 4 * c : 0;
                        gw[gi][c] &= keep >= 4 ? 0xFFFFFFFFu : ((1u << (8 * keep)) - 1u);
                    }
                }
            }
            if (tid == 0 && seg.next_len) {
                halo[0] = *reinterpret_cast<const u32 *>(inbuf + HUFD_ENC_SEG_BYTES);
                halo[1] = *reinterpret_cast<const u32 *>(inbuf + HUFD_ENC_SEG_BYTES + 4);
            }
        } else {
            load_segment_groups(src, seg.len, gw, gvalid);
            if (tid == 0 && seg.next_len) {
                const u32 n = seg.next_len < 8 ? seg.next_len : 8;
                for (u32 j = 0; j < n; ++j) {
                    halo[j >> 2] |= (u32)src[HUFD_ENC_SEG_BYTES + j] << (8 * (j & 3));
                }
            }
        }
        {
            const uint4 zero = {0, 0, 0, 0};
            for (u32 i = tid; i < img_words / 4; i += HUFD_ENC_THREADS) {
                reinterpret_cast<uint4 *>(img)[i] = zero;
            }
        }
        if (tid == 0) {
            sh->unk_before = kNoBit;
            sh->short_found = 0;
            sh->halo_unknown = 0;
        }
        barrier_lds(); /* every lane holds its symbols: the buffer may be refilled */
        const hufd_enc_seg seg_after = uniform_seg(&segs[s + 2 * gridDim.x < n_segs ? s + 2 * gridDim.x : n_segs - 1]);
        if (more) {
            stream_request(d_in, inbuf, seg_next.in_off, seg_next.len, seg_next.next_len);
        }
        HUFD_STAMP(2, 1);

        if (!g.skip && !g.careful) {
            if (tid == 0 && seg.index == 0 && it.ovf_bits) {
                image_or_bits(img, 8 * g.mis, it.ovf_pattern, it.ovf_bits);
            }
            u32 half_base = 0; /* bits of the groups handled by the earlier half */
#pragma unroll
            for (u32 half = 0; half < 2; ++half) {
                /* codes -> pairs (<= 32 bits) -> quads (<= 64 bits), two groups at a time */
                u64 qv[2][4];
                u32 ql[2];      /* the four quad lengths of a group, one byte each */
                u32 packed = 0; /* the lane's bit count in its two groups, 16 bits apiece */
#pragma unroll
                for (u32 gg = 0; gg < 2; ++gg) {
                    const u32 gi = 2 * half + gg;
                    u32 group_bits = 0, lens = 0;
#pragma unroll
                    for (u32 m = 0; m < 4; ++m) {
                        u32 pv[2], pl[2];
#pragma unroll
                        for (u32 h = 0; h < 2; ++h) {
                            const u32 j = 4 * m + 2 * h;
                            const u32 ea = j < gvalid[gi] ? tab32[group_byte(gw[gi], j)] : 0;
                            const u32 eb = j + 1 < gvalid[gi] ? tab32[group_byte(gw[gi], j + 1)] : 0;
                            const u32 lb = eb >> 16;
                            pv[h] = ((ea & 0xFFFFu) << lb) | (eb & 0xFFFFu);
                            pl[h] = (ea >> 16) + lb;
                        }
                        qv[gg][m] = ((u64)pv[0] << pl[1]) | pv[1];
                        lens |= (pl[0] + pl[1]) << (8 * m);
                        group_bits += pl[0] + pl[1];
                    }
                    ql[gg] = lens;
                    packed |= group_bits << (16 * gg);
                }
                if (half == 0) {
                    HUFD_STAMP(2, 2);
                }

                /* one wave scan for both groups (each 16-bit field stays below 2^16 across a wave) */
                u32 incl = packed;
#pragma unroll
                for (u32 d = 1; d < kWave; d <<= 1) {
                    const u32 up = __shfl_up(incl, d);
                    if (lane >= d) {
                        incl += up;
                    }
                }
                if (lane == kWave - 1) {
                    slots[4 * half + wave] = incl;
                }
                barrier_lds();
                u32 before[2] = {0, 0}, total[2] = {0, 0};
#pragma unroll
                for (u32 w = 0; w < HUFD_ENC_THREADS / kWave; ++w) {
                    const u32 t = slots[4 * half + w];
                    before[0] += w < wave ? (t & 0xFFFFu) : 0;
                    before[1] += w < wave ? (t >> 16) : 0;
                    total[0] += t & 0xFFFFu;
                    total[1] += t >> 16;
                }
                if (half == 0) {
                    HUFD_STAMP(2, 3);
                }
#pragma unroll
                for (u32 gg = 0; gg < 2; ++gg) {
                    const u32 mine = (packed >> (16 * gg)) & 0xFFFFu;
                    u32 q = g.q0 + half_base + (gg ? total[0] : 0) + before[gg] + ((incl >> (16 * gg)) & 0xFFFFu) - mine;
#pragma unroll
                    for (u32 m = 0; m < 4; ++m) {
                        const u32 len = (ql[gg] >> (8 * m)) & 0xFFu;
                        image_or_quad(img, q, qv[gg][m], len);
                        q += len;
                    }
                }
                half_base += total[0] + total[1];
            }
            if (tid == 0) {
                pack_last_byte(img, sh, g, seg, it, st, halo, [&](u32 sym) {
                    const u32 e = tab32[sym];
                    return ((u64)(e >> 16) << 32) | (e & 0xFFFFu);
                });
            }
        }
        HUFD_STAMP(2, 4);
        /* full barrier: the image is complete, and every wave's share of the prefetch has
         * landed (it was issued a whole packing ago) before anybody moves on */
        __syncthreads();
        HUFD_STAMP(2, 5);
        HUFD_STAMP(2, 6);
        if (!g.skip && !g.careful) {
            pack_write_out(img, sh, g, seg, it, st, results);
        }
        HUFD_STAMP(2, 7);
        if (!more) {
            break;
        }
        s += gridDim.x;
        seg = seg_next;
        seg_next = seg_after;
        barrier_lds(); /* copy-out has read the image; its stores stay in flight */
    }
}

/* ------------------------------------------------------------------ encode: pack, one wave per tile */

/*
 * The packer for whole, aligned segments of coders with codes of 4 .. 15 bits (the reference's
 * test coder: 5 .. 10).  Written around three measurements of the packer before it
 * (profiles/r01_d_*): ~16 vector instructions a symbol at ~4 cycles each were the bound, a
 * third of the LDS time went into bank conflicts of the table look-ups, and LDS atomics into a
 * zeroed image cost a zeroing pass and returned nothing.
 *
 *  - One WAVE packs one TILE: a quarter segment, 4 KiB of symbols, whose bit offset comes from
 *    enc_count's per-quarter totals.  A wave owns the output bytes whose first bit lies in its
 *    tile and completes its last byte with the first codes of the next tile (which it looks up
 *    itself), so waves share nothing: no workgroup barrier after the table is built.
 *  - The code table is kept once per LDS bank (entry b for lane l at word 32 b + l % 32): no bank
 *    conflicts.  Entry = code left-aligned in the high half | length.
 *  - A lane merges its 16 symbols pairwise in registers: codes -> pairs -> quads -> two "octs" of
 *    eight symbols (up to 120 bits, left-aligned).  One wave scan per two groups places them.
 *  - An oct becomes NW whole words at the lane's bit offset (funnel shifts), stored with PLAIN
 *    stores: an oct is at least 32 bits long, so the word a unit starts in is the only one it
 *    shares with its predecessor.  All units of a group store word k before any stores word
 *    k - 1: whatever a unit writes past its own end (zeros) is overwritten by the unit that owns
 *    that word, whose store comes later; word 0 is OR-ed in last, onto the predecessor's tail.
 *    No zeroing of the image, no atomics but that one OR.
 */
constexpr u32 kTileBytes = HUFD_ENC_SEG_BYTES / 4;
constexpr u32 kTilesPerSeg = 4;
constexpr u32 kPackWaves = 8; /* waves (= independent tiles in flight) per workgroup */
constexpr u32 kPackThreads = kPackWaves * kWave;
constexpr u32 kPackTabBytes = 256 * 32 * 4;


/* bytes of LDS one tile's bit image needs: the tile's bits, 16 bytes of alignment in front, the words a last unit spills */
__device__ __host__ inline u32 pack_region_bytes(u32 max_bits) {
    return ((kTileBytes * max_bits + 7) / 8 + 16 + 8 * 4 + 15) & ~15u;
}

/* copies region bytes [lo, hi) to gbase + b (gbase 16-byte aligned), one wave: aligned 16-byte rows, and at most 15 single bytes at either end */
__device__ __forceinline__ void region_store(const u32 *img, u8 *gbase, u32 lo, u32 hi, u32 lane) {
    if (hi <= lo) {
        return;
    }
    const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
    const u32 head_end = row_lo * 16 < hi ? row_lo * 16 : hi;          /* bytes [lo, head_end) in front of the first whole row */
    const u32 tail_at = row_hi > row_lo ? row_hi * 16 : head_end;       /* bytes [tail_at, hi) behind the last whole row */
    {
        const u32 b = lo + lane;
        if (b < head_end) {
            gbase[b] = (u8)(img[b >> 2] >> (24 - 8 * (b & 3)));
        }
    }
    uint4 *rows = reinterpret_cast<uint4 *>(__builtin_assume_aligned(gbase, 16));
    for (u32 r = row_lo + lane; r < row_hi; r += kWave) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&img[r * 4]);
        uint4 o;
        o.x = __builtin_bswap32(v.x);
        o.y = __builtin_bswap32(v.y);
        o.z = __builtin_bswap32(v.z);
        o.w = __builtin_bswap32(v.w);
        rows[r] = o;
    }
    {
        const u32 b = tail_at + lane;
        if (b < hi) {
            gbase[b] = (u8)(img[b >> 2] >> (24 - 8 * (b & 3)));
        }
    }
}

/* plain read-modify-write of the low `nbits` (1..32) bits of `pattern` into the image at bit q: one lane */
__device__ __forceinline__ void region_put_bits(u32 *img, u32 q, u32 pattern, u32 nbits) {
    const u64 left = ((u64)pattern << (64 - nbits)) >> (q & 31);
    img[q >> 5] |= (u32)(left >> 32);
    if ((u32)left) {
        img[(q >> 5) + 1] |= (u32)left;
    }
}

template <u32 NW> /* words an oct can touch: 4 for codes of at most 12 bits, 5 up to 15 */
__global__ __launch_bounds__(kPackThreads) void enc_pack_wave_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const hufd_enc_item_state *states,
    const hufd_enc_seg *segs,
    const u32 *seg_bits,
    const u32 *wave_bits,
    const u64 *seg_bitoff,
    const u8 *d_in,
    u8 *d_out,
    u32 region_bytes,
    u32 n_segs,
    u32 *careful_list,   /* segments this kernel leaves to enc_pack_kernel are added */
    u32 *careful_count,
    const u32 *gate /* NULL, or the word enc_onepass raises when a look-back wait ran out: this kernel runs only then */) {
    if (gate && gate[0] == 0) {
        return;
    }

    u32 *tab = reinterpret_cast<u32 *>(dyn_lds); /* [256][32] */
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    u32 *img = reinterpret_cast<u32 *>(dyn_lds + kPackTabBytes + wave * region_bytes);

    if (tid < 256) {
        const u64 ent = tb.enc_table[tid];
        const u32 len = (u32)(ent >> 32);
        const u32 e = len ? ((((u32)ent << (16 - len)) & 0xFFFFu) << 16) | len : 0u;
#pragma unroll
        for (u32 k = 0; k < 32; ++k) {
            tab[tid * 32 + ((k + tid) & 31u)] = e;
        }
    }
    __syncthreads();
    const u8 *mine = reinterpret_cast<const u8 *>(tab) + (lane & 31u) * 4u;
    const bool coder_ok = tb.enc_max_bits <= (NW == 4 ? 12u : 15u) && tb.enc_min_bits >= 4;

    const u32 n_tiles = n_segs * kTilesPerSeg;
    uint4 v[kGroupsPerLane], vn[kGroupsPerLane]; /* this tile's symbols and the next one's, asked for a tile ahead */
    bool fetched = false;
#pragma unroll
    for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
        v[gi] = vn[gi] = uint4{0, 0, 0, 0};
    }
    for (u32 tile = blockIdx.x * kPackWaves + wave; tile < n_tiles; tile += gridDim.x * kPackWaves) {
        const u32 s = tile / kTilesPerSeg, w4 = tile % kTilesPerSeg;
        const bool had = fetched;
        fetched = false;
        if (had) {
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                v[gi] = vn[gi];
            }
        }
        const hufd_enc_seg seg = uniform_seg(&segs[s]);
        const hufd_enc_item it = items[seg.item];
        const hufd_enc_item_state st = states[seg.item];
        const pack_geometry g = pack_geometry_of(tb, seg, s, it, st, seg_bitoff[s], seg_bits[s], d_out);
        const u8 *src = d_in + seg.in_off;
        if (g.skip || g.careful) {
            continue; /* nothing to write, or already on the list */
        }
        const bool shaped = coder_ok && seg.len == HUFD_ENC_SEG_BYTES && ((uintptr_t)src & 15u) == 0 &&
                            !(seg.index == 0 && it.ovf_bits);
        if (!shaped) {
            if (w4 == 0 && lane == 0) {
                careful_list[atomicAdd(careful_count, 1u)] = s;
            }
            continue;
        }

        /* where the tile's bits go */
        u64 bw = g.p0;
        for (u32 k = 0; k < w4; ++k) {
            bw += wave_bits[kTilesPerSeg * s + k];
        }
        const u32 tile_bits = wave_bits[kTilesPerSeg * s + w4];
        const u64 bn = bw + tile_bits;
        const bool first_tile = seg.index == 0 && w4 == 0;
        const bool last_tile = (seg.flags & 2u) != 0 && w4 == kTilesPerSeg - 1;
        u8 *out_ptr = d_out + it.out_off;
        const u64 jb = bw >> 3;                                       /* stream byte holding the tile's first bit */
        const u32 mis = (u32)((uintptr_t)(out_ptr + jb) & 15u);
        u8 *gbase = out_ptr + jb - mis;                               /* output address of image byte 0, 16-byte aligned */
        const u32 q0 = (u32)(bw - 8 * jb) + 8 * mis;                  /* image bit of the tile's first code */

        /* the tile's symbols: wave-contiguous, 16 per lane and group */
        const u8 *tsrc = src + w4 * kTileBytes;
        if (!had) {
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                v[gi] = reinterpret_cast<const uint4 *>(tsrc)[gi * kWave + lane];
            }
        }
        /* the next tile's first two symbols complete my last byte (a code is at least 4 bits, the byte lacks at most 7) */
        u32 halo_n = 0, halo0 = 0, halo1 = 0;
        if (!last_tile) {
            halo_n = w4 + 1 < kTilesPerSeg ? 2u : (seg.next_len < 2 ? seg.next_len : 2u);
            halo0 = halo_n > 0 ? tsrc[kTileBytes] : 0u;
            halo1 = halo_n > 1 ? tsrc[kTileBytes + 1] : 0u;
        }
        /* (asked for after the halo bytes: loads return in order, and the halo is needed first) */
        {
            /* the wave's next tile: on its way while this one is packed (whole, aligned segments only: the others are not packed here) */
            const u32 next = tile + gridDim.x * kPackWaves;
            if (next < n_tiles) {
                const hufd_enc_seg nseg = uniform_seg(&segs[next / kTilesPerSeg]);
                const u8 *nsrc = d_in + nseg.in_off + (next % kTilesPerSeg) * kTileBytes;
                if (nseg.len == HUFD_ENC_SEG_BYTES && ((uintptr_t)nsrc & 15u) == 0) {
#pragma unroll
                    for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                        vn[gi] = reinterpret_cast<const uint4 *>(nsrc)[gi * kWave + lane];
                    }
                    fetched = true;
                }
            }
        }
        if (lane == 0) {
            img[q0 >> 5] = 0; /* the word the first unit ORs its head into */
        }

        /* codes -> pairs -> quads -> octs */
        u64 ohi[kGroupsPerLane][2], olo[kGroupsPerLane][2];
        u32 olen[kGroupsPerLane]; /* the two oct lengths of a group, 16 bits each */
#pragma unroll
        for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
            const u32 wd[4] = {v[gi].x, v[gi].y, v[gi].z, v[gi].w};
            u32 both = 0;
#pragma unroll
            for (u32 o = 0; o < 2; ++o) {
                u64 quad[2];
                u32 qlen[2];
#pragma unroll
                for (u32 h = 0; h < 2; ++h) {
                    u32 pair[2], plen[2];
#pragma unroll
                    for (u32 m = 0; m < 2; ++m) {
                        const u32 wdv = wd[2 * o + h];
                        const u32 ea = *reinterpret_cast<const u32 *>(mine + ((wdv >> (16 * m)) & 0xFFu) * 128u);
                        const u32 eb = *reinterpret_cast<const u32 *>(mine + ((wdv >> (16 * m + 8)) & 0xFFu) * 128u);
                        /* eb's length field (< 16) falls off the low end: the shift is by at least 4 */
                        pair[m] = (ea & 0xFFFF0000u) | (eb >> (ea & 31u));
                        plen[m] = ea + eb; /* the lengths add up in the low half; what the high half holds is never looked at */
                    }
                    quad[h] = ((u64)pair[0] << 32) | (((u64)pair[1] << 32) >> (plen[0] & 63u));
                    qlen[h] = plen[0] + plen[1];
                }
                const u64 x = quad[1] >> (qlen[0] & 63u);
                ohi[gi][o] = quad[0] | x;
                olo[gi][o] = quad[1] << ((64u - qlen[0]) & 63u); /* a quad is 16 .. 60 bits */
                both |= ((qlen[0] + qlen[1]) & 0xFFFFu) << (16 * o);
            }
            olen[gi] = both;
        }

        /* bit offset of every lane's group: one wave scan for two groups (16-bit fields, < 2^16 across a wave) */
        u32 gq[kGroupsPerLane];
        {
            u32 at = q0;
#pragma unroll
            for (u32 half = 0; half < kGroupsPerLane / 2; ++half) {
                const u32 a = (olen[2 * half] & 0xFFFFu) + (olen[2 * half] >> 16);
                const u32 b = (olen[2 * half + 1] & 0xFFFFu) + (olen[2 * half + 1] >> 16);
                const u32 packed = a | (b << 16);
                const u32 incl = wave_inclusive_sum_dpp(packed, lane);
                const u32 tot = __shfl(incl, kWave - 1);
                gq[2 * half] = at + (incl & 0xFFFFu) - a;
                gq[2 * half + 1] = at + (tot & 0xFFFFu) + (incl >> 16) - b;
                at += (tot & 0xFFFFu) + (tot >> 16);
            }
        }
        wave_step(); /* img[q0 >> 5] = 0 is in place */

        /* octs -> words, highest word first */
#pragma unroll
        for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
            u32 wds[2][NW], base[2];
#pragma unroll
            for (u32 o = 0; o < 2; ++o) {
                const u32 q = gq[gi] + (o ? olen[gi] & 0xFFFFu : 0u);
                const u32 sh = q & 31u;
                base[o] = q >> 5;
                const u32 w0 = (u32)(ohi[gi][o] >> 32), w1 = (u32)ohi[gi][o], w2 = (u32)(olo[gi][o] >> 32),
                          w3 = (u32)olo[gi][o];
                wds[o][0] = w0 >> sh;
                wds[o][1] = funnel(w0, w1, sh);
                wds[o][2] = funnel(w1, w2, sh);
                if (NW == 4) {
                    wds[o][3] = funnel(w2, 0, sh);
                } else {
                    wds[o][3] = funnel(w2, w3, sh);
                    wds[o][NW - 1] = funnel(w3, 0, sh);
                }
            }
#pragma unroll
            for (u32 k = NW - 1; k >= 1; --k) {
#pragma unroll
                for (u32 o = 0; o < 2; ++o) {
                    img[base[o] + k] = wds[o][k];
                    wave_step();
                }
            }
#pragma unroll
            for (u32 o = 0; o < 2; ++o) {
                atomicOr(&img[base[o]], wds[o][0]);
                wave_step();
            }
        }

        /* the last byte: the next tile's head, or the padding when the item ends here (huffman.c:178-184) */
        bool halo_unknown = false;
        {
            const u32 need = (u32)((8 - (bn & 7)) & 7);
            const u32 e0 = halo_n > 0 ? *reinterpret_cast<const u32 *>(mine + halo0 * 128u) : 0u;
            const u32 e1 = halo_n > 1 ? *reinterpret_cast<const u32 *>(mine + halo1 * 128u) : 0u;
            const u32 l0 = e0 & 0xFFFFu, l1 = e1 & 0xFFFFu;
            /* the codes that follow, left-aligned; behind the item's last symbol the padding (ones above the low bits are masked off below) */
            u32 head = (e0 & 0xFFFF0000u) | ((e1 & 0xFFFF0000u) >> l0);
            u32 have = l0 + l1;
            halo_unknown = (halo_n > 0 && l0 == 0) || (halo_n > 1 && l0 < need && l1 == 0);
            if (have < need && halo_n < 2 && st.status == HUFD_ENC_OK) {
                const u32 pad_bits = need - have;
                head |= ((it.eos_padding & ((1u << pad_bits) - 1u)) << (32 - need));
                have = need;
            }
            if (lane == 0 && need && have >= need && !halo_unknown) {
                region_put_bits(img, q0 + tile_bits, head >> (32 - need), need);
            }
        }
        wave_step();

        /* the bytes this tile owns, within what the call may write (pack_write_out's rules) */
        const u64 limit_bytes = st.status == HUFD_ENC_OK ? (st.total_bits + 7) >> 3 : it.out_cap;
        u64 jhi = last_tile ? (st.status == HUFD_ENC_OK ? (st.total_bits + 7) >> 3 : bn >> 3)
                            : (halo_unknown ? bn >> 3 : (bn + 7) >> 3);
        jhi = jhi > limit_bytes ? limit_bytes : jhi;
        const u64 jlo = first_tile ? 0 : (bw + 7) >> 3;
        if (jhi > jlo) {
            region_store(img, gbase, (u32)(jlo - jb) + mis, (u32)(jhi - jb) + mis, lane);
        }
        wave_step(); /* the image is free for the next tile */
    }
}

/* ------------------------------------------------------------------ encode: one pass */

/*
 * Encode in ONE pass over HBM (coders whose every symbol has a code of 4 .. 15 bits): a tile's
 * symbols are read once, its bits written once -- no count kernel that reads the input a second
 * time, no scan kernel.  What a tile needs from the tiles in front of it is the number of bits they
 * hold; what this kernel is built around is that nobody waits for that number:
 *
 *  - Persistent waves take tiles (quarter segments, as enc_pack_wave) in turn: wave w of the grid
 *    packs tiles w, w + W, ...  A tile only depends on lower tiles, which are in the same turn or
 *    an earlier one; the grid is sized to be resident as a whole, every wait is bounded, and a
 *    wait that runs out sends the launch to the three-kernel path.  (Tickets from one counter
 *    would drop the residency assumption, but one word hands out ~90 tickets a microsecond --
 *    measured: 8.8 ms for the 262 144 tiles of 1 GiB.)
 *  - A wave looks a tile's symbols up, merges them to octs and scans the lane lengths exactly as
 *    enc_pack_wave does -- which gives the tile's bit total long before the tile is finished -- and
 *    publishes the total at once.  The octs then wait in registers while the wave finishes its
 *    PREVIOUS tile: only now does it ask for the offsets in front of that one, which were published
 *    a whole turn ago by waves that ran beside it (asking in the same turn made every turn a
 *    chip-wide rendezvous: the slowest of 4 096 waves set the pace and the waiting ones' polls took
 *    the memory system from the rest -- measured: 7.5 ms instead of 0.5).  Then the fresh octs go
 *    into the LDS image, at image bit 0: where the tile lies in the stream is found out a turn later.
 *  - Totals are kept on three levels so that a wave reads a few hundred bytes, not the history:
 *    tile_agg[t] (one word, flagged), group_acc[t / 64] (sum and arrival count of 64 tiles, one
 *    atomic add each, nothing returned) and round_base[r] = bits in front of round r (64 groups),
 *    stored by one wave of the grid that does nothing else.  A tile's offset = round_base + the
 *    complete groups of its round in front of it + the tiles of its group in front of it: three
 *    loads of at most 64 lanes, polled until every value is there (in the steady state: at once).
 *    item_base[item] = the same number for the item's first tile turns it into an offset inside
 *    the item.  All of it through agent-scope relaxed atomics: the data is the flag (guide:
 *    Guideline 16, R2).
 *  - The copy-out moves the image to where the offset says with one funnel shift that is the same
 *    for the whole tile (region_store_shifted).
 *
 * Every segment is packed here: a ragged tile takes the same pyramid with the entries behind its last
 * symbol set to no bits, the loads take any alignment, an item's carried overflow bits sit in the word
 * in front of image bit 0.  The tile that holds the capacity edge of an item whose output is too
 * short leaves a note for enc_finish_kernel, which finds the symbol at the edge.  Every spin is
 * bounded; a wave that gives up raises ctl[1] and the host layer redoes the launch with the
 * three-kernel path (which has no waits between workgroups).
 */
constexpr u32 kOpGroupTiles = HUFD_OP_GROUP_TILES;   /* at most 64: a lane per tile */
constexpr u32 kOpRoundGroups = HUFD_OP_ROUND_GROUPS; /* at most 64: a lane per group */
constexpr u32 kOpRoundTiles = kOpGroupTiles * kOpRoundGroups;
constexpr u64 kOpArrive = 1ull << 40; /* group_acc: arrivals above, sum of bits below */
constexpr u64 kOpSum = kOpArrive - 1;
constexpr u64 kOpReady = 1ull << 63;  /* round_base / item_base */
constexpr u32 kOpTileReady = 1u << 31; /* tile_agg */
constexpr u32 kOpSpinLimit = 1u << 13; /* polls (each a trip to memory and a sleep): milliseconds */
constexpr u32 kOpGroupStride = HUFD_OP_GROUP_STRIDE; /* u64 words from one group's counter to the next: a memory line each (the adds are done at the memory side, a line at a time) */

__device__ __forceinline__ void granule_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void word_store(u32 *p, u32 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32 word_load(const u32 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_add(u64 *p, u64 v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


/*
 * Copies stream bytes [jlo, jhi) of an item out of a tile image whose bit 0 is stream bit `bit0`
 * (any alignment), one wave: stream byte j is image bits 8 j - bit0 ...; whole 16-byte rows of
 * the output are assembled from five image words with a funnel shift that is the same for every
 * row of the tile, at most 15 single bytes at either end.
 */
__device__ __forceinline__ void region_store_shifted(const u32 *img, u8 *out_ptr, u64 bit0, u64 jlo, u64 jhi, u32 lane) {
    if (jhi <= jlo) {
        return;
    }
    /* (the lane number as a value the compiler cannot trace: what is derived from it -- the lane's place in the image --
     * is then worked out here, two instructions, instead of being kept in registers, or spilled, across the turn) */
    opaque(lane);
    /* (an item's carried bits lie in front of image bit 0, in img[-1]: image bit numbers may be negative down to -32) */
    auto byte_at = [&](u64 j) -> u8 {
        const int ib = (int)(u32)(8 * j - bit0);
        const u64 two = ((u64)img[ib >> 5] << 32) | img[(ib >> 5) + 1];
        return (u8)((two << ((u32)ib & 31u)) >> 56);
    };
    const uintptr_t a_lo = (uintptr_t)(out_ptr + jlo), a_hi = (uintptr_t)(out_ptr + jhi);
    const uintptr_t row_lo = (a_lo + 15) & ~(uintptr_t)15, row_hi = a_hi & ~(uintptr_t)15;
    if (row_lo >= row_hi) {
        /* no whole row: fewer than 31 bytes */
        if (jlo + lane < jhi) {
            out_ptr[jlo + lane] = byte_at(jlo + lane);
        }
        return;
    }
    const u64 j_row_lo = jlo + (row_lo - a_lo), j_row_hi = jlo + (row_hi - a_lo);
    if (jlo + lane < j_row_lo) {
        out_ptr[jlo + lane] = byte_at(jlo + lane);
    }
    {
        /* bits [ib0 + 128 r, + 128) of the image are row r.  With the shift written as a right shift of the
         * word pair (k - 1, k) a shift of zero needs no case of its own: it takes the pair one word down */
        const int ib0 = (int)(u32)(8 * j_row_lo - bit0);
        const u32 sh = (u32)ib0 & 31u;
        const u32 rs = (32u - sh) & 31u;
        const u32 *words = img + (ib0 >> 5) - (sh == 0 ? 1 : 0);
        const u32 rows = (u32)((row_hi - row_lo) >> 4);
        uint4 *dst = reinterpret_cast<uint4 *>(row_lo);
        for (u32 r = lane; r < rows; r += kWave) {
            const u32 *src = words + 4 * r;
            const u32 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3], x4 = src[4];
            uint4 o;
            o.x = __builtin_bswap32(funnel(x0, x1, rs));
            o.y = __builtin_bswap32(funnel(x1, x2, rs));
            o.z = __builtin_bswap32(funnel(x2, x3, rs));
            o.w = __builtin_bswap32(funnel(x3, x4, rs));
            dst[r] = o;
        }
    }
    if (j_row_hi + lane < jhi) {
        out_ptr[j_row_hi + lane] = byte_at(j_row_hi + lane);
    }
}

template <bool B> struct op_flag {
    static constexpr bool value = B;
};

/* what a wave knows about a tile (everything here is the same in all lanes) */
struct op_tile {
    hufd_enc_seg seg;
    u32 t, s, w4;
    u32 n_sym;      /* symbols of the segment that lie in this tile */
    u32 carried;    /* the item's carried overflow bits */
    u32 item_first_tile; /* the item's first tile */
    u32 bits;       /* the tile's code bits, once counted */
    u32 halo_n;     /* how many symbols behind the tile its last byte may need (their values are per-lane registers) */
    u32 carried_pattern; /* the carried bits themselves (an item's first tile puts them in front of its image) */
    bool first_tile; /* of its item */
    bool ends_item;  /* holds the item's last symbol */
    const u8 *tsrc;
};

/* SOLO: the items of one tile at most that the plan lists apart (hufd_enc_item.tiny == 2: up to HUFD_ENC_SOLO_BYTES symbols),
 * a wave an item in turn.  Such an item's bits start at its own bit 0 (or behind its carried bits): nobody in front of it to
 * ask, nobody behind it to tell -- the same turn (look-ups, octs, scan, image, last byte, copy-out, the note at a capacity
 * edge) without the look-back.  As tiles of the stream's kernel a 2 KiB item was FOUR turns (a segment is four tiles,
 * three of them empty, each waiting for its offsets like any tile): 65 536 of them took as long as 1 GiB of whole tiles. */
/* ORDERED: the way back.  Tiles by TICKET (one counter, a draw a tile) instead of by rule: a tile then only ever waits for
 * tiles that were drawn before it, by waves that run -- no wait can last for ever whatever share of the grid is resident, so
 * none is bounded and nobody gives up.  A word handing out tickets is slow (~90 a microsecond: a 1 GiB stream's 262 144
 * tiles take 3 ms, five times the kernel by rule), which is why this is not the first kernel of a launch but the one
 * queued behind it, looking at the word a wave of the first raises when a wait of its ran out (`gate`: an empty launch
 * otherwise) and doing the launch over with look-back words of its own.  (Rounds 3-5 had the three-kernel road there:
 * four empty launches of ~4 us behind every encode.) */
template <u32 NW, bool SOLO = false, bool ORDERED = false> /* NW: words an oct can touch: 4 for codes of at most 12 bits, 5 up to 15 */
__global__ __launch_bounds__(kPackThreads, 4) void enc_onepass_kernel(
    hufd_tables tb,
    const hufd_enc_item *__restrict__ items,
    const hufd_enc_seg *__restrict__ segs,
    const u8 *__restrict__ d_in,
    u8 *__restrict__ d_out,
    u32 region_bytes,
    u32 n_segs,
    u32 *ctl,          /* [1] a spin ran out, [2] careful_count */
    u32 *tile_agg,     /* [4 n_segs] zeroed */
    u64 *group_acc,    /* zeroed */
    u64 *round_base,   /* [rounds + 1] zeroed */
    u64 *item_base,    /* [n_items] zeroed */
    u64 *__restrict__ item_total,
    hufd_enc_result *__restrict__ results, /* the tile with the capacity edge leaves a note for enc_finish_kernel here */
    const u8 *__restrict__ null_tile /* kTileBytes readable bytes: what a wave "prefetches" when no tile follows */,
    u32 fail_tile /* a tile whose wave is to give up (tests of the way back); HUFD_NONE32: none */,
    const u32 *__restrict__ solo_items = nullptr /* SOLO: the items, one tile each.  ORDERED: the word that says whether the
                                                  * launch's first one-pass kernel gave up (else this one has nothing to do) --
                                                  * in this place because a parameter more costs the kernel by rule a vector
                                                  * register and eighteen scalar ones parked in vector lanes (measured: 8 us) */,
    u32 n_solo = 0) {

    static_assert(!(SOLO && ORDERED), "items a wave takes whole wait for nobody");
    if (ORDERED && solo_items[0] == 0) {
        return;
    }
    HUFD_STAMP_DECL
    HUFD_STAMP_ZERO;
    u32 *tab = reinterpret_cast<u32 *>(dyn_lds); /* [256][32] */
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = uniform32(tid / kWave);
    /* image bit 0 = the tile's first code; the four words in front of it: img[-1] = an item's carried bits (first
     * tile, right-aligned), the others only ever read along with it */
    u32 *img = reinterpret_cast<u32 *>(dyn_lds + kPackTabBytes + wave * region_bytes) + 4;

    if (tid < 256) {
        const u64 ent = tb.enc_table[tid];
        const u32 len = (u32)(ent >> 32);
        const u32 e = len ? ((((u32)ent << (16 - len)) & 0xFFFFu) << 16) | len : 0u;
#pragma unroll
        for (u32 k = 0; k < 32; ++k) {
            tab[tid * 32 + ((k + tid) & 31u)] = e;
        }
    }
    if (lane < 4) {
        img[(int)lane - 4] = 0;
    }
    __syncthreads();
    const u8 *mine = reinterpret_cast<const u8 *>(tab) + (lane & 31u) * 4u;
    /* (SOLO: tile t = tile t % 4 of listed item t / 4 -- an item of up to a segment; the tiles behind its last symbol are
     * never taken, see next_tile) */
    const u32 n_tiles = SOLO ? n_solo * kTilesPerSeg : n_segs * kTilesPerSeg;
    const bool count_only = SOLO && d_out == nullptr; /* a length query for the items a wave takes: the bits are counted as they
                                                       * are for packing -- a thread an item took 2.1 ms for 16 384 items of 16 KiB */

    /* (t is a scalar, the descriptor arrays are read-only: these are scalar loads, no vector registers, no vector-memory wait) */
    auto describe = [&](u32 t) -> op_tile {
        op_tile d;
        const u32 tc = t < n_tiles ? t : n_tiles - 1; /* (a tile past the end is never worked on) */
        d.t = t;
        if (SOLO) {
            /* the item is its own one segment */
            const u32 item = solo_items[tc / kTilesPerSeg];
            d.s = tc / kTilesPerSeg;
            d.w4 = tc % kTilesPerSeg;
            d.seg.in_off = items[item].in_off;
            d.seg.len = (u32)items[item].in_len;
            d.seg.item = item;
            d.seg.index = 0;
            d.seg.flags = 3;
            d.seg.next_len = 0;
            d.seg.reserved = 0;
        } else {
            d.s = tc / kTilesPerSeg;
            d.w4 = tc % kTilesPerSeg;
            d.seg = segs[d.s];
        }
        const u8 *src = d_in + d.seg.in_off;
        d.carried = items[d.seg.item].ovf_bits;
        d.carried_pattern = items[d.seg.item].ovf_pattern;
        d.item_first_tile = SOLO ? t - d.w4 : items[d.seg.item].first_seg * kTilesPerSeg;
        d.tsrc = src + d.w4 * kTileBytes;
        const u32 from = d.w4 * kTileBytes;
        d.n_sym = (t < n_tiles && d.seg.len > from) ? (d.seg.len - from < kTileBytes ? d.seg.len - from : kTileBytes) : 0u;
        d.first_tile = d.seg.index == 0 && d.w4 == 0;
        d.ends_item = (d.seg.flags & 2u) != 0 && d.n_sym != 0 && from + d.n_sym == d.seg.len;
        d.bits = 0;
        /* the symbols of the item behind the tile: the first two complete its last byte (a code is at least 4 bits, the byte lacks at most 7) */
        const u32 behind = d.n_sym ? (d.seg.len - from - d.n_sym) + d.seg.next_len : 0u;
        d.halo_n = behind < 2 ? behind : 2u;
        return d;
    };

    /*
     * One wave of the grid packs nothing: it watches the groups of a round arrive and publishes the next round's
     * base the moment the last one is there (a packing wave would get to it half a turn to a turn later -- measured:
     * then 9 of 10 tiles found their round's base missing at the first look and every wave polled a third of its
     * time, in step with the one wave that held the round's last tile).
     */
    if (!SOLO && blockIdx.x == 0 && wave == kPackWaves - 1) {
        const u32 full_rounds = n_tiles / kOpRoundTiles; /* (nobody asks for the base behind a round that is not full) */
        u64 base = 0;
        if (lane == 0) {
            granule_store(&round_base[0], kOpReady);
        }
        for (u32 r = 0; r < full_rounds; ++r) {
            u64 b = 0;
            u32 spins = 0;
            for (;;) {
                b = lane < kOpRoundGroups ? granule_load(&group_acc[(u64)(r * kOpRoundGroups + lane) * kOpGroupStride])
                                          : kOpGroupTiles * kOpArrive;
                if (__all((b >> 40) == kOpGroupTiles)) {
                    break;
                }
                if (!ORDERED && (++spins > kOpSpinLimit || uniform32(word_load_now(&ctl[1])) != 0)) {
                    if (lane == 0) {
                        ctl[1] = 1; /* (the waves that wait for this base give up in their turn) */
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            u64 sum = lane < kOpRoundGroups ? (b & kOpSum) : 0;
#pragma unroll
            for (u32 d = kWave / 2; d > 0; d >>= 1) {
                sum += __shfl_xor(sum, d);
            }
            base += sum;
            if (lane == 0) {
                granule_store(&round_base[r + 1], kOpReady | base);
#ifdef HUFD_STAMPS_WHY
                hufd_stamp_rows[((u64)1 * HUFD_STAMP_MAX_WG + r + 1) * 8 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
            }
        }
        return;
    }
    /* tiles in turn over the packing waves of the grid: the tiles a tile waits for belong to this turn or an earlier
     * one, so to the running waves as long as the whole grid is resident (the launch sizes it so) */
    const u32 stride = gridDim.x * kPackWaves - (SOLO ? 0u : 1u);
    /* ORDERED: the next ticket of the launch's counter (ctl[0]), the same in every lane */
    auto draw = [&]() -> u32 {
        u32 ticket = 0;
        if (lane == 0) {
            ticket = atomicAdd(&ctl[0], 1u);
        }
        return uniform32(ticket);
    };
    u32 t_new = ORDERED ? draw()
                        : (SOLO ? (blockIdx.x * kPackWaves + wave) * kTilesPerSeg : blockIdx.x * kPackWaves + wave - (blockIdx.x ? 1u : 0u));
    if (t_new >= n_tiles) {
        return;
    }
    /* the tile this wave takes behind tile `d`.  SOLO: the item's next tile while it has symbols left, then the first tile
     * of the wave's next item -- the wave packs its item's tiles one after the other and knows the bits in front of each
     * (solo_bits), which is all the look-back would tell it */
    auto next_tile = [&](const op_tile &d) -> u32 {
        if (ORDERED) {
            return draw();
        }
        if (!SOLO) {
            return d.t + stride;
        }
        const u32 from_next = (d.w4 + 1) * kTileBytes;
        return d.w4 + 1 < kTilesPerSeg && d.seg.len > from_next ? d.t + 1 : d.t - d.w4 + stride * kTilesPerSeg;
    };
    u64 solo_bits = 0; /* SOLO: the item's bits in front of the tile that is finished next */
    op_tile fresh = describe(t_new); /* the tile whose symbols are looked up in this turn ... */
    op_tile old = fresh;             /* ... and the one before it, whose image is copied out in this turn */
    bool have_old = false;
    uint4 v[kGroupsPerLane], vn[kGroupsPerLane];
#pragma unroll
    for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
        v[gi] = vn[gi] = uint4{0, 0, 0, 0};
    }
    /*
     * A tile's symbols, 16 per lane and group, from any address (the loads need no alignment).  Every lane loads (no
     * branch around a load: see ask_offsets): where a ragged tile ends inside a group, the 16 bytes that END with the
     * tile's last symbol (an item with segments is longer than 16 bytes, so they are the item's; ragged_groups shifts
     * them into place), behind that a harmless address.
     */
    auto tile_loads = [&](const op_tile &d, uint4 (&into)[kGroupsPerLane]) {
        const u8 *spare = d.n_sym >= 16 ? d.tsrc : null_tile;
#pragma unroll
        for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
            const u32 base = (gi * kWave + lane) * 16;
            const u8 *at = base + 16 <= d.n_sym ? d.tsrc + base : (base < d.n_sym ? d.tsrc + d.n_sym - 16 : spare);
            const unaligned_uint4 got = *reinterpret_cast<const unaligned_uint4 *>(at);
            into[gi] = uint4{got.x, got.y, got.z, got.w};
        }
    };
    /* how many of a lane's 16 symbols of group gi the tile holds; and its words put right where it holds only some
     * (loaded as the 16 bytes that end with the tile: down by 16 - valid bytes) */
    auto group_valid = [&](const op_tile &d, u32 gi) -> u32 {
        const u32 base = (gi * kWave + lane) * 16;
        return d.n_sym > base ? (d.n_sym - base < 16 ? d.n_sym - base : 16u) : 0u;
    };
    auto group_in_place = [&](u32 (&wd)[4], u32 valid) {
        const u32 sb = 16 - valid, ws = sb >> 2, bs8 = (sb & 3u) * 8;
        const u32 y0 = ws == 0 ? wd[0] : ws == 1 ? wd[1] : ws == 2 ? wd[2] : wd[3];
        const u32 y1 = ws == 0 ? wd[1] : ws == 1 ? wd[2] : ws == 2 ? wd[3] : 0u;
        const u32 y2 = ws == 0 ? wd[2] : ws == 1 ? wd[3] : 0u;
        const u32 y3 = ws == 0 ? wd[3] : 0u;
        const bool part = valid > 0 && valid < 16;
        wd[0] = part ? (u32)((((u64)y1 << 32) | y0) >> bs8) : wd[0];
        wd[1] = part ? (u32)((((u64)y2 << 32) | y1) >> bs8) : wd[1];
        wd[2] = part ? (u32)((((u64)y3 << 32) | y2) >> bs8) : wd[2];
        wd[3] = part ? (y3 >> bs8) : wd[3];
    };
    tile_loads(fresh, v);
    u32 base_item = HUFD_NONE32; /* the item whose base this wave has read ... */
    u64 base_value = 0;          /* ... and that base (item_base[base_item]) */
    u32 halo0 = 0, halo1 = 0, old_halo0 = 0, old_halo1 = 0; /* the first two symbols behind the fresh / the old tile */

    /* the old tile's offsets, as they stand in memory: its round's base, the complete groups of its round in front of
     * it, the tiles of its group in front of it, its item's base.  Every lane, no branch: the compiler then knows how
     * many younger loads a wait for these may leave in flight. */
    auto ask_offsets = [&](u32 &a_raw, u64 &b_raw, u64 &rb_raw, u64 &ib_raw) {
        if (SOLO) {
            a_raw = 0;
            b_raw = rb_raw = ib_raw = 0;
            return;
        }
        const u32 g = old.t / kOpGroupTiles, p = old.t % kOpGroupTiles, r = g / kOpRoundGroups, gi_r = g % kOpRoundGroups;
        a_raw = word_load(&tile_agg[g * kOpGroupTiles + (lane < p ? lane : 0u)]);
        b_raw = granule_load(&group_acc[(u64)(r * kOpRoundGroups + (lane < gi_r ? lane : 0u)) * kOpGroupStride]);
        rb_raw = granule_load(&round_base[r]);
        ib_raw = granule_load(&item_base[old.seg.item]);
    };

    /* the old tile, from the look at its offsets to the copy-out of its image; false: a wait ran out */
    auto finish_old = [&](u32 a_raw, u64 b_raw, u64 rb_raw, u64 ib_raw) -> bool {
        const u32 g = old.t / kOpGroupTiles, p = old.t % kOpGroupTiles, r = g / kOpRoundGroups, gi_r = g % kOpRoundGroups;
        /* ---- the bits in front of the old tile are there (asked for at the top of the turn); if not, ask again */
        const hufd_enc_seg seg = old.seg;
        /* (the wait for these leaves the younger loads -- the next tile's symbols -- and the arrival atomic in flight) */
        u32 a = lane < p ? a_raw : kOpTileReady;
        u64 b = lane < gi_r ? b_raw : kOpGroupTiles * kOpArrive;
        u64 rb = rb_raw;
        /* an item that starts inside my group needs no base: its tiles in front of me are among the group's (lanes
         * p - since .. p - 1); otherwise the base, read once per wave and item */
        const u32 since = old.t - old.item_first_tile; /* tiles of my item in front of me */
        const bool near = since <= p;
        u64 ib = (old.first_tile || near) ? kOpReady : (seg.item == base_item ? base_value : ib_raw);
        bool gave_up = !SOLO && !ORDERED && old.t == fail_tile;
        for (u32 spins = 0; !gave_up && !SOLO; ++spins) {
            const bool there = (a & kOpTileReady) != 0 && (b >> 40) == kOpGroupTiles && (rb & kOpReady) != 0 &&
                               (ib & kOpReady) != 0;
            if (spins == 0) {
#ifdef HUFD_STAMPS_WHY
                if ((threadIdx.x & 63u) == 0 && blockIdx.x >= 1 && blockIdx.x <= 7 && threadIdx.x == 0) {
                    hufd_stamp_rows[((u64)1 * HUFD_STAMP_MAX_WG + r) * 8 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
                }
#endif
                HUFD_STAMP_ADD(2, 7); /* the values asked for at the top of the turn are in registers */
#ifdef HUFD_STAMPS_WHY
                HUFD_STAMP_COUNT(3, __all((a & kOpTileReady) != 0) ? 0 : 1);
                HUFD_STAMP_COUNT(4, __all((b >> 40) == kOpGroupTiles) ? 0 : 1);
                HUFD_STAMP_COUNT(5, __all((rb & kOpReady) != 0) ? 0 : 1);
#endif
            }
            if (__all(there)) {
                HUFD_STAMP_COUNT(6, spins);
                break;
            }
            if (!ORDERED && (spins > kOpSpinLimit || uniform32(word_load_now(&ctl[1])) != 0)) {
                gave_up = true; /* (or somebody else has: the launch is redone anyway) */
                break;
            }
            __builtin_amdgcn_s_sleep(8); /* a poll is traffic for everybody: rarely needed, then not in a tight loop */
            if (lane < p && !(a & kOpTileReady)) {
                a = word_load_now(&tile_agg[g * kOpGroupTiles + lane]);
            }
            if (lane < gi_r && (b >> 40) != kOpGroupTiles) {
                b = granule_load_now(&group_acc[(u64)(r * kOpRoundGroups + lane) * kOpGroupStride]);
            }
            if (!(rb & kOpReady)) {
                rb = granule_load_now(&round_base[r]);
            }
            if (!(ib & kOpReady)) {
                ib = granule_load_now(&item_base[seg.item]);
            }
        }
        if (gave_up) {
            if (lane == 0) {
                ctl[1] = 1; /* the ORDERED kernel, queued behind this one, sees it and does the launch over */
            }
            return false;
        }
        HUFD_STAMP_ADD(2, 2);
        a &= ~kOpTileReady;
        /* (two sums in one scan: everything in front of me in the low half-words' place, my own item's tiles of this
         * group above bit 32 -- a group holds less than 2^22 bits) */
        const u64 part = (u64)((lane < p ? a : 0u) + (lane < gi_r ? (u32)(b & kOpSum) : 0u)) |
                         ((u64)((near && lane < p && lane + since >= p) ? a : 0u) << 32);
        u64 sums = part;
#pragma unroll
        for (u32 d = kWave / 2; d > 0; d >>= 1) {
            sums += __shfl_xor(sums, d);
        }
        const u32 in_round = (u32)sums, in_item = (u32)(sums >> 32);
        const u64 before = uniform64((rb & ~kOpReady) + in_round); /* bits of every tile of the plan in front of this one */
        u64 bw; /* stream bit (inside the item) of the tile's first code */
        if (SOLO) {
            bw = old.first_tile ? (u64)old.carried : solo_bits;
        } else if (old.first_tile) {
            bw = old.carried;
            base_value = kOpReady | before;
            base_item = seg.item;
            if (lane == 0) {
                granule_store(&item_base[seg.item], base_value);
            }
        } else if (near) {
            bw = (u64)uniform32(in_item) + old.carried;
        } else {
            base_value = uniform64(ib);
            base_item = seg.item;
            bw = before - (base_value & ~kOpReady) + old.carried;
        }
        const u64 bn = bw + old.bits;
        if (SOLO) {
            solo_bits = bn;
        }

        const u64 out_cap = uniform64(items[seg.item].out_cap);
        const u64 cap_bits = out_cap > (~0ull >> 3) ? ~0ull : out_cap * 8;
        /* ---- what enc_finish_kernel turns into the call's outcome: the item's bit total ... */
        if (lane == 0 && (SOLO ? old.ends_item : old.w4 == kTilesPerSeg - 1) && (seg.flags & 2u)) {
            item_total[seg.item] = bn; /* (tiles behind the item's last symbol hold no bits) */
        }
        if (SOLO && count_only) {
            return true; /* (a length query: the totals are all it asks for) */
        }
        {
            u8 *out_ptr = d_out + uniform64(items[seg.item].out_off);
            /* the last byte: the next tile's head, or the padding when the item ends here (huffman.c:178-184) */
            {
                const u32 need = (u32)((8 - (bn & 7)) & 7);
                const u32 e0 = old.halo_n > 0 ? *reinterpret_cast<const u32 *>(mine + old_halo0 * 128u) : 0u;
                const u32 e1 = old.halo_n > 1 ? *reinterpret_cast<const u32 *>(mine + old_halo1 * 128u) : 0u;
                const u32 l0 = e0 & 0xFFFFu, l1 = e1 & 0xFFFFu;
                u32 head = (e0 & 0xFFFF0000u) | ((e1 & 0xFFFF0000u) >> l0);
                u32 have = l0 + l1;
                /* fewer than two symbols behind the tile: the item ends inside its last byte, and whether it ends
                 * well (padding) is a matter of its total, which is then known here */
                if (have < need && old.halo_n < 2 && bn + have <= cap_bits) {
                    const u32 eos = uniform32(items[seg.item].eos_padding);
                    const u32 pad_bits = need - have;
                    head |= ((eos & ((1u << pad_bits) - 1u)) << (32 - need));
                    have = need;
                }
                if (lane == 0 && need && have >= need) {
                    /* behind the tile's last bit the image holds nothing yet (the word the bits start in was
                     * written whole by the last unit, zeros behind its end): the word after it is stored, not OR-ed */
                    const u32 q = old.bits;
                    const u64 left = ((u64)(head >> (32 - need)) << (64 - need)) >> (q & 31u);
                    img[q >> 5] |= (u32)(left >> 32);
                    img[(q >> 5) + 1] = (u32)left;
                }
            }
            wave_step();
            u64 jhi = old.ends_item ? (bn <= cap_bits ? (bn + 7) >> 3 : bn >> 3) : (bn + 7) >> 3;
            jhi = jhi > out_cap ? out_cap : jhi;
            const u64 jlo = old.first_tile ? 0 : (bw + 7) >> 3;
            region_store_shifted(img, out_ptr, bw, jlo, jhi, lane);
        }
        wave_step(); /* the image is free for the fresh tile */
        /* When the output is too short: the symbol whose last bit reaches the capacity edge (exactly one tile holds it).  The
         * wave takes its tile's symbols once more -- out of the L2, into the image's LDS, which is free now --, every lane
         * counts the bits of 64 of them, and the 64 symbols around the edge are then a symbol a lane: `consumed` (up to and
         * with that symbol) and the bits of its code that did not fit (source/huffman.c:88-98) go into the item's record,
         * where enc_finish_kernel picks them up.  (Rounds 2-4 left a note here and enc_finish read the tile again, a wave
         * an item: 143 registers, three waves a SIMD -- 61-67 us for BASELINE configs[3], a quarter of whose outputs are
         * short.) */
        if (bw < cap_bits && cap_bits <= bn) {
            const u32 target = (u32)(cap_bits - bw); /* the edge, in bits from the tile's first code: 1 .. the tile's bits */
            u8 *stage = reinterpret_cast<u8 *>(img);
            const bool staged = region_bytes >= kTileBytes + 16u; /* (codes of 8 bits or more; shorter: straight from memory) */
            if (staged) {
                for (u32 at = lane * 16; at < old.n_sym; at += kWave * 16) {
                    if (at + 16 <= old.n_sym) {
                        const unaligned_uint4 q = *reinterpret_cast<const unaligned_uint4 *>(old.tsrc + at);
                        *reinterpret_cast<uint4 *>(stage + at) = uint4{q.x, q.y, q.z, q.w};
                    } else { /* the tile's last symbols: one by one (nothing behind the item is read) */
                        for (u32 j = at; j < old.n_sym; ++j) {
                            stage[j] = old.tsrc[j];
                        }
                    }
                }
                wave_step();
            }
            const u8 *syms = staged ? stage : old.tsrc;
            /* lane l counts symbols 64 l .. 64 l + 63 */
            const u32 from = lane * 64 < old.n_sym ? lane * 64 : old.n_sym, to = from + 64 < old.n_sym ? from + 64 : old.n_sym;
            u32 sum = 0;
            if (staged) {
                /* (a word of four symbols a read: their look-ups do not wait for each other) */
                for (u32 j = from; j < to; j += 4) {
                    const u32 four = *reinterpret_cast<const u32 *>(stage + j);
#pragma unroll
                    for (u32 b = 0; b < 4; ++b) {
                        const u32 e4 = *reinterpret_cast<const u32 *>(mine + ((four >> (8 * b)) & 0xFFu) * 128u);
                        sum += j + b < to ? e4 & 0xFFFFu : 0u;
                    }
                }
            } else {
                for (u32 j = from; j < to; ++j) {
                    sum += *reinterpret_cast<const u32 *>(mine + (u32)syms[j] * 128u) & 0xFFFFu;
                }
            }
            const u32 incl = wave_inclusive_sum_dpp(sum, lane);
            /* the lane whose symbols hold the edge, and the bits in front of them */
            const u64 holds = __ballot(incl - sum < target && target <= incl);
            const u32 edge_lane = (u32)__builtin_ctzll(holds | (1ull << 63));
            const u32 edge_rel = __shfl(incl - sum, edge_lane);
            /* ... its 64 symbols, one a lane */
            const u32 j = edge_lane * 64 + lane;
            const u32 e = j < old.n_sym ? *reinterpret_cast<const u32 *>(mine + (u32)syms[j] * 128u) : 0u;
            const u32 len = e & 0xFFFFu;
            const u32 incl2 = wave_inclusive_sum_dpp(len, lane);
            const u32 rel = edge_rel + incl2 - len;
            if (len && rel < target && target <= rel + len) {
                const u32 left = rel + len - target;
                hufd_enc_result *r = &results[seg.item];
                r->consumed = (u64)seg.index * HUFD_ENC_SEG_BYTES + old.w4 * kTileBytes + j + 1;
                r->ovf_bits = left;
                r->ovf_pattern = left ? ((e >> 16) >> (16u - len)) & ((1u << left) - 1u) : 0u;
            }
            wave_step(); /* (the image is the fresh tile's from here) */
        }
        return true;
    };

    /*
     * One turn: look up and merge the FRESH tile's symbols; then finish the OLD tile (offsets in front of it --
     * published a whole turn ago --, last byte, copy-out of the image), publishing the fresh tile's bit total on the
     * way; then place the fresh tile's octs in the image.  The fresh octs wait in registers meanwhile.  The last tile
     * is finished behind the loop, so that every turn inside it asks for the same loads (see ask_offsets).
     */
    while (t_new < n_tiles) {
        HUFD_STAMP_ADD(2, 0);
        u64 ohi[kGroupsPerLane][2], olo[kGroupsPerLane][2];
        u32 olen[kGroupsPerLane];
        u32 gq[kGroupsPerLane];
        op_tile nxt = fresh;
        /* Everything the last turn asked for has had a turn to land: this tile's symbols, the copy-out stores.  Saying so
         * here (instead of leaving it to the first use) lets the wait further down be a counted one that leaves this
         * turn's own loads in flight. */
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */
        u32 a_raw;
        u64 b_raw, rb_raw, ib_raw;
        ask_offsets(a_raw, b_raw, rb_raw, ib_raw); /* (in the first turn: of the fresh tile, never looked at) */
        /* the first two symbols behind the tile (they complete its last byte): always asked for, from a harmless
         * address when there are none */
        const u8 *seg_first = d_in + fresh.seg.in_off; /* (a segment holds at least one symbol) */
        halo0 = *(fresh.halo_n > 0 ? fresh.tsrc + fresh.n_sym : seg_first);
        halo1 = *(fresh.halo_n > 1 ? fresh.tsrc + fresh.n_sym + 1 : seg_first);
        /* the tile after it: its symbols are on their way while this one is packed */
        const u32 t_next = next_tile(fresh);
        nxt = describe(t_next);
        tile_loads(nxt, vn);

        /* ---- the fresh tile's bits: codes -> pairs -> quads -> octs, one wave scan per two groups.  A ragged tile takes
         * the same way with the entries behind its last symbol set to nothing (a code of no bits). */
        auto pyramid = [&](auto ragged_tag) {
            constexpr bool RAGGED = decltype(ragged_tag)::value;
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                u32 wd[4] = {v[gi].x, v[gi].y, v[gi].z, v[gi].w};
                u32 valid = 16;
                if (RAGGED) {
                    valid = group_valid(fresh, gi);
                    group_in_place(wd, valid);
                }
                u32 both = 0;
#pragma unroll
                for (u32 o = 0; o < 2; ++o) {
                    u64 quad[2];
                    u32 qlen[2];
#pragma unroll
                    for (u32 h = 0; h < 2; ++h) {
                        u32 pair[2], plen[2];
#pragma unroll
                        for (u32 m = 0; m < 2; ++m) {
                            const u32 wdv = wd[2 * o + h];
                            u32 ea = *reinterpret_cast<const u32 *>(mine + ((wdv >> (16 * m)) & 0xFFu) * 128u);
                            u32 eb = *reinterpret_cast<const u32 *>(mine + ((wdv >> (16 * m + 8)) & 0xFFu) * 128u);
                            if (RAGGED) {
                                const u32 k = 4 * (2 * o + h) + 2 * m; /* the symbols in front of ea in its group */
                                ea = k < valid ? ea : 0u;
                                eb = k + 1 < valid ? eb : 0u;
                            }
                            /* eb's length field (< 16) falls off the low end: the shift is by at least 4 (or eb is nothing) */
                            pair[m] = (ea & 0xFFFF0000u) | (eb >> (ea & 31u));
                            plen[m] = ea + eb; /* the lengths add up in the low half; what the high half holds is never looked at */
                        }
                        quad[h] = ((u64)pair[0] << 32) | (((u64)pair[1] << 32) >> (plen[0] & 63u));
                        qlen[h] = plen[0] + plen[1];
                    }
                    const u64 x = quad[1] >> (qlen[0] & 63u);
                    ohi[gi][o] = quad[0] | x;
                    olo[gi][o] = quad[1] << ((64u - qlen[0]) & 63u); /* a quad is 16 .. 60 bits (ragged: or quad[1] is nothing) */
                    both |= ((qlen[0] + qlen[1]) & 0xFFFFu) << (16 * o);
                }
                olen[gi] = both;
            }
            u32 at = 0; /* the image starts at the tile's own first bit */
#pragma unroll
            for (u32 half = 0; half < kGroupsPerLane / 2; ++half) {
                const u32 la = (olen[2 * half] & 0xFFFFu) + (olen[2 * half] >> 16);
                const u32 lb = (olen[2 * half + 1] & 0xFFFFu) + (olen[2 * half + 1] >> 16);
                const u32 packed = la | (lb << 16);
                const u32 incl = wave_inclusive_sum_dpp(packed, lane);
                const u32 tot = __shfl(incl, kWave - 1);
                gq[2 * half] = at + (incl & 0xFFFFu) - la;
                gq[2 * half + 1] = at + (tot & 0xFFFFu) + (incl >> 16) - lb;
                at += (tot & 0xFFFFu) + (tot >> 16);
            }
            fresh.bits = uniform32(at);
        };
        /* (a segment that is not full has tiles without symbols: they only tell that they hold no bits) */
        const bool whole = fresh.n_sym == kTileBytes, empty = fresh.n_sym == 0;
        if (whole) {
            pyramid(op_flag<false>{});
        } else if (!empty) {
            pyramid(op_flag<true>{});
        } else {
            fresh.bits = 0;
        }
        HUFD_STAMP_ADD(2, 1);

        /* tell the tiles behind the fresh one (see arrival_quiet) */
        if (!SOLO && lane == 0) {
            arrival_quiet(&tile_agg[fresh.t], kOpTileReady | fresh.bits, &group_acc[(u64)(fresh.t / kOpGroupTiles) * kOpGroupStride], kOpArrive | fresh.bits);
        }
        if (have_old) {
            if (!finish_old(a_raw, b_raw, rb_raw, ib_raw)) {
                return;
            }
        } else {
            HUFD_STAMP_ADD(2, 7);
            HUFD_STAMP_ADD(2, 2);
        }
        HUFD_STAMP_ADD(2, 3);

        /* ---- the fresh octs -> words of the image */
        if (!(SOLO && count_only) && lane == 0) {
            img[0] = 0; /* the word the first unit ORs its head into */
            if (fresh.first_tile) {
                img[-1] = fresh.carried_pattern; /* stream bits 0 .. carried - 1 of the item */
            }
        }
        auto oct_words = [&](u32 gi, u32 o, u32 (&wds)[NW]) -> u32 {
            const u32 q = gq[gi] + (o ? olen[gi] & 0xFFFFu : 0u);
            const u32 sh = q & 31u;
            const u32 w0 = (u32)(ohi[gi][o] >> 32), w1 = (u32)ohi[gi][o], w2 = (u32)(olo[gi][o] >> 32),
                      w3 = (u32)olo[gi][o];
            wds[0] = w0 >> sh;
            wds[1] = funnel(w0, w1, sh);
            wds[2] = funnel(w1, w2, sh);
            if (NW == 4) {
                wds[3] = funnel(w2, 0, sh);
            } else {
                wds[3] = funnel(w2, w3, sh);
                wds[NW - 1] = funnel(w3, 0, sh);
            }
            return q >> 5;
        };
        if (SOLO && count_only) {
            /* (nothing is written by a length query) */
        } else if (whole) {
            /* highest word first: a unit's words behind its first are stored (whoever else has bits there comes later
             * in the stream and later in this order), its first word is OR-ed in at the end */
            wave_step();
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
                u32 wds[2][NW], base[2];
#pragma unroll
                for (u32 o = 0; o < 2; ++o) {
                    base[o] = oct_words(gi, o, wds[o]);
                }
#pragma unroll
                for (u32 k = NW - 1; k >= 1; --k) {
#pragma unroll
                    for (u32 o = 0; o < 2; ++o) {
                        img[base[o] + k] = wds[o][k];
                        wave_step();
                    }
                }
#pragma unroll
                for (u32 o = 0; o < 2; ++o) {
                    atomicOr(&img[base[o]], wds[o][0]);
                    wave_step();
                }
            }
        } else if (!empty) {
            /* a ragged tile has units of no bits, which own no word: the image is cleared first and every unit ORs */
            const u32 used = (fresh.bits >> 5) + NW + 2;
            for (u32 w = lane; w < used; w += kWave) {
                img[w] = 0;
            }
            wave_step();
#pragma unroll
            for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
#pragma unroll
                for (u32 o = 0; o < 2; ++o) {
                    u32 wds[NW];
                    const u32 base = oct_words(gi, o, wds);
                    const u32 len = o ? olen[gi] >> 16 : olen[gi] & 0xFFFFu;
                    if (len) {
#pragma unroll
                        for (u32 k = 0; k < NW; ++k) {
                            atomicOr(&img[base + k], wds[k]);
                        }
                    }
                }
            }
            wave_step();
        }
        HUFD_STAMP_ADD(2, 4);
        HUFD_STAMP_ADD(2, 5);

        old = fresh;
        old_halo0 = halo0;
        old_halo1 = halo1;
        have_old = true;
        fresh = nxt;
        t_new = t_next;
#pragma unroll
        for (u32 gi = 0; gi < kGroupsPerLane; ++gi) {
            v[gi] = vn[gi];
        }
    }
    /* the last tile */
    HUFD_STAMP_ADD(2, 0);
    {
        u32 a_raw;
        u64 b_raw, rb_raw, ib_raw;
        ask_offsets(a_raw, b_raw, rb_raw, ib_raw);
        HUFD_STAMP_ADD(2, 1);
        if (!finish_old(a_raw, b_raw, rb_raw, ib_raw)) {
            return;
        }
    }
    HUFD_STAMP_ADD(2, 3);
    HUFD_STAMP_ADD(2, 4);
    HUFD_STAMP_ADD(2, 5);
    HUFD_STAMP_FLUSH(2); /* (wave 0's own sums) */
}

/*
 * After the one pass: one thread per item turns the item's bit total into the outcome of the call
 * (enc_finish_item; every symbol has a code here).  For a call that ran out of room the wave that packed the
 * tile holding the capacity edge has left the answer in the item's result record: `consumed` up to and with
 * the symbol whose last bit reaches the edge, and what of its code did not fit (source/huffman.c:88-98).
 */
constexpr u32 kFinishItems = 256; /* a thread an item */
constexpr u32 kFinishLdsBytes = 0;
__global__ __launch_bounds__(256) void enc_finish_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    u32 n_items,
    const u64 *item_total,
    const u8 *d_in,
    u32 *careful_list,
    u32 *careful_count,
    hufd_enc_item_state *states,
    hufd_enc_result *results,
    const u32 *gave_up /* the word enc_onepass raises when a look-back wait ran out: totals and notes are not whole then, and
                        * the three-kernel road behind this kernel does the launch over, records included */,
    u32 which /* bit 0: the items with segments; bit 1: the items of one tile that the plan lists apart (their kernel waits
               * for nobody: whatever road the others took, theirs are whole); bit 2: the packing waves have left, for every
               * item short of room, what of it was consumed (not in a length query: nothing was packed) */,
    uint4 *clear_from = nullptr /* the look-back words of the launch (the packing kernels are through with them), both sets, ... */,
    u64 clear_vec16 = 0         /* ... so many 16-byte pieces: cleared here for the plan's next launch, whatever road this one took */,
    u32 *ctl = nullptr          /* the control words: [1] "a wait ran out" is copied to [4] for the host, then [0] and [1] are cleared
                                 * (no workgroup of this kernel reads them) */) {

    for (u64 k = (u64)blockIdx.x * kFinishItems + threadIdx.x; k < clear_vec16; k += (u64)gridDim.x * kFinishItems) {
        clear_from[k] = uint4{0, 0, 0, 0};
    }
    if (ctl && blockIdx.x == 0 && threadIdx.x == 0) {
        ctl[4] = ctl[1];
        ctl[0] = 0;
        ctl[1] = 0;
    }
    const bool with_segments = (which & 1u) && !(gave_up && gave_up[0] != 0), solo = (which & 2u) != 0;
    if (!with_segments && !solo) {
        return;
    }
    const u32 i = blockIdx.x * kFinishItems + threadIdx.x;
    const u32 road = i < n_items ? items[i].tiny : 1u; /* (1: enc_tiny's) */
    if ((road == 0 && with_segments) || (road == 2 && solo)) {
        const hufd_enc_item it = items[i];
        const u64 total = (it.n_segs || road == 2) ? item_total[i] : it.ovf_bits;
        const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
        const hufd_enc_result note = results[i];
        hufd_enc_result rs;
        /* (no segment is named as the edge's: nothing is listed for enc_pack_kernel) */
        enc_finish_item(it, total, HUFD_NONE32, 0, 0, 0, HUFD_NONE32, careful_list, careful_count, &states[i], &rs);
        if ((which & 4u) && rs.status == HUFD_ENC_SHORT && it.ovf_bits < cap_bits) {
            /* (the wave of enc_onepass that packed the tile holding the capacity edge has left these) */
            rs.consumed = note.consumed;
            rs.ovf_bits = note.ovf_bits;
            rs.ovf_pattern = note.ovf_pattern;
        }
        results[i] = rs;
    }
}

__global__ __launch_bounds__(256) void enc_plan_tiny_items_kernel(const hufd_raw_enc_item *raw, u32 n_items, hufd_enc_item *items, u32 *tiny_list) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_raw_enc_item r = raw[i];
    const u32 ob = r.ovf_bits;
    hufd_enc_item it;
    it.in_off = r.in_offset;
    it.in_len = r.in_len;
    it.out_off = r.out_offset;
    it.out_cap = r.out_capacity;
    it.ovf_bits = ob;
    it.ovf_pattern = ob == 0 ? 0u : (ob >= 32 ? r.ovf_pattern : r.ovf_pattern & ((1u << ob) - 1u));
    it.eos_padding = r.eos_padding;
    it.first_seg = 0;
    it.n_segs = 0;
    it.tiny = 1;
    items[i] = it;
    tiny_list[i] = i;
}


/* scalar registers of enc_onepass (every instantiation: the compiler uses all 102 + VCC + the rest);
 * tests/test_library_boundary.py::test_onepass_kernels_scalar_registers holds the build to it */
constexpr uint32_t kOnepassSgprs = 106;

/* layout of the block the one-pass encoder wants clear when a launch starts (all offsets multiples of 8).  It is cleared when
 * the plan gets it and then by every launch behind itself (so that a launch captured in a graph can be replayed): enc_finish,
 * which runs when the packing kernels -- the one by rule and, gated, the way back -- are through with the look-back words,
 * clears both their sets and the control words (after copying "a wait ran out" to where the host reads it).  A clearing
 * command in front of every launch was a packet of ~4 us on the queue. */
struct onepass_layout {
    uint64_t ctl, tile_agg, group_acc, round_base, item_base, null_tile, bytes;
    uint64_t again; /* from tile_agg to the same word of the way back's own set of look-back words (enc_onepass<.., ORDERED>) */
};

static onepass_layout onepass_layout_of(uint64_t n_segs, uint64_t n_items) {
    const uint64_t tiles = n_segs * kTilesPerSeg;
    const uint64_t groups = (tiles + kOpGroupTiles - 1) / kOpGroupTiles;
    const uint64_t rounds = (groups + kOpRoundGroups - 1) / kOpRoundGroups;
    onepass_layout l;
    l.ctl = 0; /* sixteen control words: [0] the way back's tickets, [1] "a wait ran out", [2] careful_count, [4] word 1 of the
                * last launch (for the host: enc_finish clears word 1 itself) */
    l.tile_agg = 64;
    l.group_acc = l.tile_agg + ((tiles * 4 + 7) & ~7ull);
    l.round_base = l.group_acc + groups * 8 * kOpGroupStride;
    l.item_base = l.round_base + (rounds + 1) * 8;
    l.null_tile = (l.item_base + n_items * 8 + 15) & ~15ull;
    l.again = l.null_tile + kTileBytes - l.tile_agg; /* (a multiple of 16) */
    l.bytes = l.tile_agg + 2 * l.again - kTileBytes;
    return l;
}

static uint32_t enc_pack_lds_bytes(uint32_t img_words) {
    return ((img_words * 4 + 15) & ~15u) + 256 * 8 + 8 * 4 + (uint32_t)sizeof(enc_pack_shared) + 16;
}

static uint32_t enc_stream_lds_bytes(uint32_t img_words) {
    return ((img_words * 4 + 15) & ~15u) + (HUFD_ENC_SEG_BYTES + 16) + 256 * 4 + 8 * 4 +
           (uint32_t)sizeof(enc_pack_shared) + 16;
}
} /* namespace */

using hufk_host::persistent_grid;
using hufk_host::stage_mark;
using hufk_host::current_compute_units;

hipError_t hufk_host::init_encode(int lds_max) {
    hipError_t e = hipSuccess;
    const void *kernels[] = {
        reinterpret_cast<const void *>(&enc_pack_kernel),      reinterpret_cast<const void *>(&enc_pack_stream_kernel),
        reinterpret_cast<const void *>(&enc_pack_wave_kernel<4>), reinterpret_cast<const void *>(&enc_pack_wave_kernel<5>),
        reinterpret_cast<const void *>(&enc_onepass_kernel<4>),   reinterpret_cast<const void *>(&enc_onepass_kernel<5>),
        reinterpret_cast<const void *>(&enc_onepass_kernel<4, true>), reinterpret_cast<const void *>(&enc_onepass_kernel<5, true>),
        reinterpret_cast<const void *>(&enc_onepass_kernel<4, false, true>), reinterpret_cast<const void *>(&enc_onepass_kernel<5, false, true>)};
    for (const void *k : kernels) {
        if (e == hipSuccess) {
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        }
    }
    return e;
}

static void encode_three_kernels(const struct hufk_encode_args *a, hipStream_t st, const u32 *gate);

/* workgroups of the way back (enc_onepass<.., ORDERED>): what it costs every launch is its EMPTY launch -- ~10 us with the
 * 512 workgroups (74 KiB of LDS each) that fill the chip, ~4 with these --, and what it can do when it runs is bounded by its
 * ticket counter (~90 tiles a microsecond) before it is by its waves: 64 x 8 waves at ~9 us a tile are 57 */
constexpr uint32_t kOrderedBlocks = 64;

/* the items of one tile that the plan lists apart (enc_onepass<.., SOLO>: a wave an item, nobody waits for anybody), and
 * their outcomes; a launch that only asks for lengths counts them a thread each */
static void encode_solo_items(const struct hufk_encode_args *a, hipStream_t st, bool finish) {
    if (!a->n_solo) {
        return;
    }
    u8 *out = a->length_only ? (u8 *)nullptr : (u8 *)a->d_out; /* (no output: the kernel counts only) */
    const onepass_layout l = onepass_layout_of(a->n_segs, a->n_items);
    const uint8_t *z = (const uint8_t *)a->zero_block;
    /* (the readable nothing at z + l.null_tile may hold anything: what a wave loads from there is never looked at) */
    const uint32_t region = pack_region_bytes(a->tables.enc_max_bits);
    const uint32_t lds = kPackTabBytes + kPackWaves * region;
    const uint32_t work = (a->n_solo + kPackWaves - 1) / kPackWaves;
#define HUFK_LAUNCH_SOLO(NWV)                                                                                          \
    hipLaunchKernelGGL(                                                                                                \
        (enc_onepass_kernel<NWV, true>), dim3(persistent_grid(enc_onepass_kernel<NWV, true>, kPackThreads, lds, work)), \
        dim3(kPackThreads), lds, st, a->tables, a->items, (const hufd_enc_seg *)nullptr, (const u8 *)a->d_in,          \
        out, region, 0u, (u32 *)nullptr, (u32 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr,     \
        a->item_total, a->results, z + l.null_tile, HUFD_NONE32, a->solo_items, a->n_solo)
    if (a->tables.enc_max_bits <= 12) {
        HUFK_LAUNCH_SOLO(4);
    } else {
        HUFK_LAUNCH_SOLO(5);
    }
#undef HUFK_LAUNCH_SOLO
    if (finish) {
        hipLaunchKernelGGL(
            enc_finish_kernel, dim3((a->n_items + kFinishItems - 1) / kFinishItems), dim3(256), kFinishLdsBytes, st, a->tables,
            a->items, a->n_items, a->item_total, (const u8 *)a->d_in, a->careful_list, a->careful_count, a->states, a->results,
            (const u32 *)nullptr, a->length_only ? 2u : 6u);
    }
}

extern "C" {

int hufk_encode_one_pass_applies(const struct hufd_tables *tb) {
    /* every symbol has a code (no stop inside a stream to look for), octs of 4 .. 15-bit codes */
    return tb->all_coded && tb->enc_max_bits <= 15 && tb->enc_min_bits >= 4;
}

int hufk_encode_plan_tiny_items(const void *raw_items, uint32_t n_items, struct hufd_enc_item *items, uint32_t *tiny_list, void *stream) {
    if (n_items == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        enc_plan_tiny_items_kernel, dim3((n_items + 255) / 256), dim3(256), 0, (hipStream_t)stream,
        (const hufd_raw_enc_item *)raw_items, n_items, items, tiny_list);
    return (int)hipGetLastError();
}

uint64_t hufk_encode_zero_bytes(uint32_t n_segs, uint32_t n_items) {
    return onepass_layout_of(n_segs, n_items).bytes;
}

uint32_t hufk_enc_image_words(uint32_t max_bits) {
    /* worst case: every symbol of the segment has the longest code, plus alignment slack,
     * carried overflow, halo codes and padding */
    const uint32_t bits = HUFD_ENC_SEG_BYTES * max_bits + 128 + 32 + 8 * 32 + 64;
    return ((bits + 31) / 32 + 3) & ~3u;
}

int hufk_encode_launch(const struct hufk_encode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_segs == 0 && a->n_items == 0) {
        return 0;
    }
    if ((a->n_segs || a->n_solo) && !a->length_only && a->single_pass && hufk_encode_one_pass_applies(&a->tables) && a->zero_block) {
        /* one pass: count + offsets + pack in one kernel (and the items of one tile by a kernel of their own, a wave each),
         * then the per-item outcome */
        onepass_layout l = onepass_layout_of(a->n_segs, a->n_items);
        uint8_t *z = (uint8_t *)a->zero_block;
        stage_mark(a->stage_events, 0, st);
        /* the block is clear: the plan's last launch left it so (enc_finish below), or the plan cleared it when it got it.
         * (Not known to be: cleared here, all of it the plan may ever have used.) */
        if (!a->zero_is_clear) {
            (void)hipMemsetAsync(a->zero_block, 0, a->zero_bytes ? a->zero_bytes : l.bytes, st);
        }
        const uint32_t region = pack_region_bytes(a->tables.enc_max_bits);
        const uint32_t lds = kPackTabBytes + kPackWaves * region;
        const uint32_t work = (a->n_segs * kTilesPerSeg + kPackWaves - 1) / kPackWaves;
#define HUFK_LAUNCH_ONEPASS(NWV)                                                                                      \
    hipLaunchKernelGGL(                                                                                                \
        enc_onepass_kernel<NWV>, dim3(persistent_grid(enc_onepass_kernel<NWV>, kPackThreads, lds, work, kOnepassSgprs)), \
        dim3(kPackThreads), lds, st, a->tables, a->items, a->segs, (const u8 *)a->d_in, (u8 *)a->d_out, region,        \
        a->n_segs, (u32 *)(z + l.ctl), (u32 *)(z + l.tile_agg), (u64 *)(z + l.group_acc),                              \
        (u64 *)(z + l.round_base), (u64 *)(z + l.item_base), a->item_total, a->results, (const u8 *)(z + l.null_tile),  \
        a->fail_tile ? a->n_segs * kTilesPerSeg / 2 : HUFD_NONE32);                                                    \
    /* the way back, on the same stream: the same kernel with its tiles by ticket (no wait of its can last for ever) and \
     * look-back words of its own, looking first at the word a wave of the kernel above raises when a wait of its ran   \
     * out -- whoever works on the output behind this launch finds it whole either way, without the host in between */ \
    hipLaunchKernelGGL(                                                                                                \
        (enc_onepass_kernel<NWV, false, true>), dim3(work < kOrderedBlocks ? work : kOrderedBlocks),                   \
        dim3(kPackThreads), lds, st, a->tables, a->items, a->segs, (const u8 *)a->d_in, (u8 *)a->d_out, region,        \
        a->n_segs, (u32 *)(z + l.ctl), (u32 *)(z + l.again + l.tile_agg), (u64 *)(z + l.again + l.group_acc),          \
        (u64 *)(z + l.again + l.round_base), (u64 *)(z + l.again + l.item_base), a->item_total, a->results,            \
        (const u8 *)(z + l.null_tile), HUFD_NONE32, (const u32 *)(z + l.ctl) + 1, 0u)
        if (!a->n_segs) {
            /* (items of one tile only) */
        } else if (a->tables.enc_max_bits <= 12) {
            HUFK_LAUNCH_ONEPASS(4);
        } else {
            HUFK_LAUNCH_ONEPASS(5);
        }
#undef HUFK_LAUNCH_ONEPASS
        encode_solo_items(a, st, false);
        stage_mark(a->stage_events, 1, st);
        {
            /* (a plan without segments -- every item a wave's or a thread's -- has no look-back words: BASELINE configs[3]'s
             * 65 536 item bases were half a megabyte cleared for nobody) */
            /* (both sets of look-back words, the readable nothing between them along with them: one stretch) */
            const uint64_t clear_vec16 = a->n_segs ? (l.bytes - l.tile_agg) / 16 : 0;
            const uint64_t item_blocks = (a->n_items + kFinishItems - 1) / kFinishItems;
            uint64_t clear_blocks = (clear_vec16 + kFinishItems * 4 - 1) / (kFinishItems * 4); /* four pieces a thread */
            clear_blocks = clear_blocks > 1024 ? 1024 : clear_blocks;
            hipLaunchKernelGGL(
                enc_finish_kernel, dim3((uint32_t)(item_blocks > clear_blocks ? item_blocks : clear_blocks)), dim3(256), kFinishLdsBytes, st,
                a->tables, a->items, a->n_items, a->item_total, (const u8 *)a->d_in, a->careful_list, a->careful_count, a->states,
                a->results, (const u32 *)nullptr /* (totals and notes are whole: the way back has seen to it) */, a->n_solo ? 7u : 5u,
                (uint4 *)(z + l.tile_agg), clear_vec16, (u32 *)(z + l.ctl));
        }
        stage_mark(a->stage_events, 2, st);
        if (a->n_tiny) {
            hipLaunchKernelGGL(
            enc_tiny_kernel, dim3((a->n_tiny + kTinyThreads - 1) / kTinyThreads), dim3(kTinyThreads), 256 * sizeof(u64), st,
            a->tables, a->items, a->tiny_items, a->n_tiny, (const u8 *)a->d_in, (u8 *)a->d_out, a->results,
            a->length_only);
        }
        /* (nothing is left for the per-symbol packer: every segment was packed by a wave, the capacity edge found by one;
         * nor for the three-kernel road, the way back of rounds 3-5: see enc_onepass<.., ORDERED>) */
        stage_mark(a->stage_events, 3, st);
        return (int)hipGetLastError();
    }
    stage_mark(a->stage_events, 0, st);
    encode_three_kernels(a, st, nullptr);
    encode_solo_items(a, st, true);
    stage_mark(a->stage_events, 3, st);
    return (int)hipGetLastError();
}

int hufk_encode_one_tiny(
    const struct hufd_tables *tables, const struct hufd_enc_item *item, const uint32_t *zero, const void *d_in, void *d_out,
    struct hufd_enc_result *result, uint32_t length_only, void *stream) {
    hipLaunchKernelGGL(
        enc_tiny_kernel, dim3(1), dim3(kTinyThreads), 256 * sizeof(u64), (hipStream_t)stream, *tables, item, zero, 1u,
        (const u8 *)d_in, (u8 *)d_out, result, length_only);
    return (int)hipGetLastError();
}

int hufk_encode_one_block_fits(const struct hufd_tables *tables, uint64_t symbols) {
    if (symbols <= HUFD_ENC_BLOCK_BYTES) {
        return 1;
    }
    const uint32_t bits = HUFD_ENC_BLOCK_MAX_BYTES * tables->enc_max_bits + 32 + 128 + 64;
    const uint32_t img_words = ((bits + 31) / 32 + 3) & ~3u;
    return symbols <= HUFD_ENC_BLOCK_MAX_BYTES &&
           ((img_words * 4 + 15) & ~15u) + 256 * 8 + (uint32_t)sizeof(enc_block_shared) <= 65536u;
}

int hufk_encode_one_block(
    const struct hufd_tables *tables, const struct hufd_enc_item *item, uint32_t symbols, const void *d_in, void *d_out,
    struct hufd_enc_result *result, uint32_t length_only, void *stream) {
    /* (the image: the symbols of the longest code, carried bits, alignment, padding) */
    const bool wide = symbols > HUFD_ENC_BLOCK_BYTES;
    const uint32_t bits = (wide ? HUFD_ENC_BLOCK_MAX_BYTES : HUFD_ENC_BLOCK_BYTES) * tables->enc_max_bits + 32 + 128 + 64;
    const uint32_t img_words = ((bits + 31) / 32 + 3) & ~3u;
    const uint32_t lds = ((img_words * 4 + 15) & ~15u) + 256 * 8 + (uint32_t)sizeof(enc_block_shared);
    if (symbols > HUFD_ENC_BLOCK_MAX_BYTES || lds > 65536u) {
        return (int)hipErrorInvalidValue; /* (hufk_encode_one_block_fits) */
    }
    if (wide) {
        hipLaunchKernelGGL(
            enc_block_kernel<kBlockEncWideThreads>, dim3(1), dim3(kBlockEncWideThreads), lds, (hipStream_t)stream, *tables, item,
            (const u8 *)d_in, (u8 *)d_out, result, img_words, length_only);
    } else {
        hipLaunchKernelGGL(
            enc_block_kernel<kBlockEncThreads>, dim3(1), dim3(kBlockEncThreads), lds, (hipStream_t)stream, *tables, item,
            (const u8 *)d_in, (u8 *)d_out, result, img_words, length_only);
    }
    return (int)hipGetLastError();
}

} /* extern "C" */

/* count + scan + pack; `gate`: NULL, or the word that says whether the kernels are to run at all (stage events: the
 * caller's, when it times them) */
static void encode_three_kernels(const struct hufk_encode_args *a, hipStream_t st, const u32 *gate) {
    void **events = gate ? nullptr : a->stage_events;
    if (a->n_segs) {
        const uint32_t grid = persistent_grid(enc_count_kernel, HUFD_ENC_THREADS, kCountLdsBytes, a->n_segs);
        hipLaunchKernelGGL(
            enc_count_kernel, dim3(grid), dim3(HUFD_ENC_THREADS), kCountLdsBytes, st, a->tables, a->segs,
            (const u8 *)a->d_in, a->seg_bits, a->wave_bits, a->seg_unk, a->careful_count, a->n_segs, gate);
    } else if (!gate) {
        (void)hipMemsetAsync(a->careful_count, 0, sizeof(uint32_t), st);
    }
    stage_mark(events, 1, st);
    if (a->n_tiny != a->n_items && a->n_large != a->n_items) { /* (a plan of thread-per-item items only has nothing to scan: a thread an item that finds that out is 10 us; an EMPTY item is not such an item -- its record is written here; a plan of long items only -- one stream -- is enc_scan_large's) */
        hipLaunchKernelGGL(
            enc_scan_small_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->items, a->n_items, a->seg_bits,
            a->seg_unk, a->seg_bitoff, a->careful_list, a->careful_count, a->states, a->results, gate);
    }
    if (a->n_large) {
        hipLaunchKernelGGL(
            enc_scan_large_kernel, dim3(a->n_large), dim3(HUFD_SCAN_LARGE_THREADS), 256, st, a->items, a->large_items,
            a->seg_bits, a->seg_unk, a->seg_bitoff, a->careful_list, a->careful_count, a->states, a->results,
            a->tables.all_coded, gate);
    }
    if (a->n_tiny && !gate) { /* (on the way back the short items are done: they wait for nobody) */
        hipLaunchKernelGGL(
            enc_tiny_kernel, dim3((a->n_tiny + kTinyThreads - 1) / kTinyThreads), dim3(kTinyThreads), 256 * sizeof(u64), st,
            a->tables, a->items, a->tiny_items, a->n_tiny, (const u8 *)a->d_in, (u8 *)a->d_out, a->results,
            a->length_only);
    }
    stage_mark(events, 2, st);
    if (a->n_segs && !a->length_only) {
        const uint32_t img_words = hufk_enc_image_words(a->tables.enc_max_bits);
        if (a->tables.enc_max_bits <= 15 && a->tables.enc_min_bits >= 4) {
            /* one wave per quarter segment for whole, aligned segments; it lists the others for the per-symbol packer */
            const uint32_t region = pack_region_bytes(a->tables.enc_max_bits);
            const uint32_t lds = kPackTabBytes + kPackWaves * region;
            if (a->tables.enc_max_bits <= 12) {
                const uint32_t grid = persistent_grid(enc_pack_wave_kernel<4>, kPackThreads, lds, (a->n_segs * kTilesPerSeg + kPackWaves - 1) / kPackWaves);
                hipLaunchKernelGGL(
                    enc_pack_wave_kernel<4>, dim3(grid), dim3(kPackThreads), lds, st, a->tables, a->items, a->states,
                    a->segs, a->seg_bits, a->wave_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out, region,
                    a->n_segs, a->careful_list, a->careful_count, gate);
            } else {
                const uint32_t grid = persistent_grid(enc_pack_wave_kernel<5>, kPackThreads, lds, (a->n_segs * kTilesPerSeg + kPackWaves - 1) / kPackWaves);
                hipLaunchKernelGGL(
                    enc_pack_wave_kernel<5>, dim3(grid), dim3(kPackThreads), lds, st, a->tables, a->items, a->states,
                    a->segs, a->seg_bits, a->wave_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out, region,
                    a->n_segs, a->careful_list, a->careful_count, gate);
            }
            const uint32_t most = a->n_segs < 1024 ? a->n_segs : 1024;
            hipLaunchKernelGGL(
                enc_pack_kernel, dim3(most), dim3(HUFD_ENC_THREADS), enc_pack_lds_bytes(img_words), st, a->tables,
                a->items, a->states, a->segs, a->seg_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out,
                a->results, img_words, a->n_segs, (const u32 *)a->careful_list, (const u32 *)a->careful_count, gate);
        } else if (a->tables.enc_max_bits <= 16) {
            /* streaming packer for everything but the listed segments, then those */
            const uint32_t lds = enc_stream_lds_bytes(img_words);
            const uint32_t grid = persistent_grid(enc_pack_stream_kernel, HUFD_ENC_THREADS, lds, a->n_segs);
            hipLaunchKernelGGL(
                enc_pack_stream_kernel, dim3(grid), dim3(HUFD_ENC_THREADS), lds, st, a->tables, a->items, a->states,
                a->segs, a->seg_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out, a->results, img_words,
                a->n_segs, gate);
            const uint32_t most = 2 * a->n_items < 1024 ? 2 * a->n_items : 1024;
            hipLaunchKernelGGL(
                enc_pack_kernel, dim3(most), dim3(HUFD_ENC_THREADS), enc_pack_lds_bytes(img_words), st, a->tables,
                a->items, a->states, a->segs, a->seg_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out,
                a->results, img_words, a->n_segs, (const u32 *)a->careful_list, (const u32 *)a->careful_count, gate);
        } else {
            hipLaunchKernelGGL(
                enc_pack_kernel, dim3(a->n_segs), dim3(HUFD_ENC_THREADS), enc_pack_lds_bytes(img_words), st, a->tables,
                a->items, a->states, a->segs, a->seg_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out,
                a->results, img_words, a->n_segs, (const u32 *)nullptr, (const u32 *)nullptr, gate);
        }
    }
}
