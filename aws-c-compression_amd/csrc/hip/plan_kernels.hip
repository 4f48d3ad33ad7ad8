/*
 * Plans made on the device: the item records, segment records and lists a launch reads, from items that are DESCRIBED
 * (a stride), that already LIE in device memory, or that are what an encode launch left (decode).  The host does O(1)
 * work: three small launches that look at the items (statistics, the thread-per-item rule of csrc/host/engine.c, counts
 * per item scanned into positions), one copy of a few totals back -- the sizes of the plan's arrays and of the launch's
 * grids are the host's to know --, then the launches that write the records.  Lists come out in item order, exactly as
 * the host's loop makes them (a plan is the same whichever way it was made: tests compare).
 *
 * Per item the meaning is unchanged: what aws_huffman_encode / aws_huffman_decode do for it (reference
 * source/huffman.c:131-187, :213-286).
 */
#include "kernels_common.hpp"
#include "launch_common.hpp"

namespace {

constexpr u32 kPlanThreads = 256;
constexpr u32 kPlanVec = 8; /* counters per item that are scanned into positions */
constexpr u32 kPlanStatsBlocks = 128; /* workgroups of the statistics pass at most (each adds its findings to one record) */
/* the longest encode item a plan made here takes: its segments fit 32 bits with room to spare (the host's loop refuses
 * a plan of 0xFFFFFFFF segments or more, csrc/host/engine.c enc_plan_fill) */
constexpr u64 kEncItemMaxBytes = (u64)(0xFFFFFFFEull - 1) * HUFD_ENC_SEG_BYTES;

/* device scratch of a planning pass: [0] statistics, [1] the decision, then a vector of sums per workgroup */
struct plan_stats {
    u64 count[2], longest_in_class[2]; /* items of at most classes[c] bytes, and the longest of them */
    u64 not_shortest, longest, largest_out_cap, worst_bits, invalid; /* not_shortest: the largest ~in_len (all zero = no item yet) */
    u64 tail_stage, tail_lanes; /* decode: of the chunks streams end in */
    u64 wide_lanes;             /* ... the most whole lanes of those that are not narrow */
    u64 pieces;                 /* all items' segments / chunks, summed in 64 bits (the scanned positions are 32 bits wide) */
    u64 pad[3];
};
struct plan_decision {
    u64 tiny_limit;
    u64 totals[kPlanVec];
    u64 pad[7];
};
static_assert(sizeof(plan_stats) == 128 && sizeof(plan_decision) == 128, "scratch layout");

struct raw_item { /* either kind of item as its source has it */
    u64 in_off, in_len, out_off, out_cap;
    u32 bits;    /* decode: first_bit; encode: overflow bits */
    u32 pattern; /* encode: overflow pattern */
    u32 eos;     /* encode */
};

template <bool ENC>
__device__ __forceinline__ raw_item load_item(const hufd_item_source &src, u32 i) {
    raw_item r;
    r.pattern = 0;
    r.eos = 0;
    if (src.kind == HUFD_ITEMS_STRIDED) {
        r.in_off = src.in_offset + (u64)i * src.in_stride;
        r.in_len = src.in_len;
        r.out_off = src.out_offset + (u64)i * src.out_stride;
        r.out_cap = src.out_capacity;
        r.bits = ENC ? 0u : src.first_bit;
        r.eos = src.eos_padding;
    } else if (!ENC && src.kind == HUFD_ITEMS_FROM_ENCODE) {
        /* item i = encode item i's output, as many bytes as its record says (never more than its room), decoded to where
         * the symbols came from */
        const hufd_enc_item e = src.enc_items[i];
        const u64 produced = src.enc_results[i].produced;
        r.in_off = e.out_off;
        r.in_len = produced < e.out_cap ? produced : e.out_cap;
        r.out_off = e.in_off;
        r.out_cap = e.in_len;
        r.bits = 0;
    } else if (ENC) {
        const hufd_raw_enc_item e = reinterpret_cast<const hufd_raw_enc_item *>(src.raw)[i];
        r.in_off = e.in_offset;
        r.in_len = e.in_len;
        r.out_off = e.out_offset;
        r.out_cap = e.out_capacity;
        r.bits = e.ovf_bits;
        r.pattern = e.ovf_pattern;
        r.eos = e.eos_padding;
    } else {
        const hufd_raw_dec_item d = reinterpret_cast<const hufd_raw_dec_item *>(src.raw)[i];
        r.in_off = d.in_offset;
        r.in_len = d.in_len;
        r.out_off = d.out_offset;
        r.out_cap = d.out_capacity;
        r.bits = d.first_bit;
    }
    return r;
}

/* ------------------------------------------------------------------ pass 1: what the thread-per-item rule asks */

template <bool ENC>
__global__ __launch_bounds__(kPlanThreads) void plan_stats_kernel(hufd_item_source src, u32 n_items, u64 class0, u64 class1, plan_stats *stats) {
    plan_stats *local = reinterpret_cast<plan_stats *>(dyn_lds);
    if (threadIdx.x == 0) {
        plan_stats z;
        memset(&z, 0, sizeof(z));
        *local = z;
    }
    __syncthreads();
    const u32 i = blockIdx.x * kPlanThreads + threadIdx.x;
    /* a thread's item as the sums and maxima it adds to; the wave folds them first and ONE lane adds to the workgroup's
     * record -- eleven LDS atomics a thread on the same seven words were 0.40 ms for a million items (the whole of a fresh
     * plan's pass was this kernel), a wave's worth of them 0.03 */
    u64 count[2] = {0, 0}, longest_in[2] = {0, 0}, not_shortest = 0, longest = 0, out_cap = 0, bits = 0, invalid = 0;
    /* (a workgroup takes items in turn: what it has to say goes into ONE record of the device with nine atomics on one
     * memory line -- from a workgroup per 256 items those were 37 000 of them for a million items, one after the other at
     * the memory side: 0.39 ms, the whole of the pass; from at most kPlanStatsBlocks workgroups 1 200) */
    for (u64 at = i; at < n_items; at += (u64)gridDim.x * kPlanThreads) {
        const raw_item r = load_item<ENC>(src, (u32)at);
        const u64 classes[2] = {class0, class1};
        for (u32 c = 0; c < 2; ++c) {
            if (r.in_len <= classes[c]) {
                count[c] += 1;
                longest_in[c] = r.in_len > longest_in[c] ? r.in_len : longest_in[c];
            }
        }
        not_shortest = ~r.in_len > not_shortest ? ~r.in_len : not_shortest;
        longest = r.in_len > longest ? r.in_len : longest;
        out_cap = r.out_cap > out_cap ? r.out_cap : out_cap;
        bits = r.bits > bits ? r.bits : bits;
        /* (an item is refused where the host's loop refuses it: a decode item holds less than 4 GiB; an encode item's
         *  segments must be a number the 32-bit counts below can hold -- more of them than that is no plan either way) */
        invalid |= (ENC ? (r.bits > 32 || r.in_len > kEncItemMaxBytes) : (r.bits > 7 || r.in_len > 0xFFFFFFFFull)) ? 1u : 0u;
    }
#pragma unroll
    for (u32 d = kWave / 2; d > 0; d >>= 1) {
#pragma unroll
        for (u32 c = 0; c < 2; ++c) {
            count[c] += __shfl_xor(count[c], d);
            const u64 o = __shfl_xor(longest_in[c], d);
            longest_in[c] = o > longest_in[c] ? o : longest_in[c];
        }
        u64 o = __shfl_xor(not_shortest, d);
        not_shortest = o > not_shortest ? o : not_shortest;
        o = __shfl_xor(longest, d);
        longest = o > longest ? o : longest;
        o = __shfl_xor(out_cap, d);
        out_cap = o > out_cap ? o : out_cap;
        o = __shfl_xor(bits, d);
        bits = o > bits ? o : bits;
        invalid |= __shfl_xor(invalid, d);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        for (u32 c = 0; c < 2; ++c) {
            if (count[c]) {
                atomicAdd(&local->count[c], count[c]);
                atomicMax(&local->longest_in_class[c], longest_in[c]);
            }
        }
        atomicMax(&local->not_shortest, not_shortest);
        atomicMax(&local->longest, longest);
        atomicMax(&local->largest_out_cap, out_cap);
        atomicMax(&local->worst_bits, bits);
        if (invalid) {
            atomicMax(&local->invalid, (u64)1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const plan_stats s = *local;
        for (u32 c = 0; c < 2; ++c) {
            if (s.count[c]) {
                atomicAdd(&stats->count[c], s.count[c]);
                atomicMax(&stats->longest_in_class[c], s.longest_in_class[c]);
            }
        }
        atomicMax(&stats->not_shortest, s.not_shortest);
        atomicMax(&stats->longest, s.longest);
        atomicMax(&stats->largest_out_cap, s.largest_out_cap);
        atomicMax(&stats->worst_bits, s.worst_bits);
        atomicMax(&stats->invalid, s.invalid);
    }
}

/* the rule of enc_tiny_limit / dec_tiny_limit (csrc/host/engine.c): the largest class of short items that holds at least
 * `per_byte` items per byte of its longest item goes to a thread per item; the class of HUFD_TINY_FEW_BYTES always */
__global__ void plan_decide_kernel(const plan_stats *stats, u64 class0, u64 class1, u64 per_byte, plan_decision *decision) {
    if (threadIdx.x || blockIdx.x) {
        return;
    }
    const u64 classes[2] = {class0, class1};
    u64 limit = HUFD_TINY_FEW_BYTES;
    for (u32 c = 0; c < 2; ++c) {
        if (stats->longest_in_class[c] > HUFD_TINY_FEW_BYTES && stats->count[c] >= per_byte * stats->longest_in_class[c]) {
            limit = classes[c];
            break;
        }
    }
    decision->tiny_limit = limit;
}

/* ------------------------------------------------------------------ pass 2: counts per item, and their positions */

/* what one decode item adds to the plan: [0] chunks, [1] thread-per-item items, [2] wave-per-item items, [3] large items,
 * [4] runs, [5] narrow end-of-stream chunks, [6] wide ones, [7] items with chunks */
struct dec_counts {
    u32 v[kPlanVec];
    u32 tail_chunk[2]; /* the item's chunks (numbered inside it) streams end in: up to two */
    u32 tail_narrow[2];
    u64 tail_stage, tail_lanes, wide_lanes;
};

__device__ __forceinline__ u64 whole_lanes_of(u64 left) {
    const u64 in_chunk = left < (u64)HUFD_DEC_CHUNK_BYTES + 8u ? left : (u64)HUFD_DEC_CHUNK_BYTES + 8u;
    return in_chunk >= 8 ? (in_chunk - 8) / HUFD_DEC_SUB_BYTES : 0;
}

__device__ __forceinline__ dec_counts count_dec_item(const raw_item &r, u64 tiny_limit, u32 shortest_code) {
    dec_counts c;
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        c.v[k] = 0;
    }
    c.tail_chunk[0] = c.tail_chunk[1] = ~0u;
    c.tail_narrow[0] = c.tail_narrow[1] = 0;
    c.tail_stage = c.tail_lanes = c.wide_lanes = 0;
    const bool tiny = r.in_len > 0 && r.in_len <= tiny_limit;
    const bool coop = !tiny && r.in_len > tiny_limit && r.in_len <= HUFD_DEC_COOP_BYTES;
    c.v[1] = tiny;
    c.v[2] = coop;
    if (tiny || coop || r.in_len == 0) {
        return c;
    }
    const u32 chunks = (u32)((r.in_len + HUFD_DEC_CHUNK_BYTES - 1) / HUFD_DEC_CHUNK_BYTES);
    c.v[0] = chunks;
    c.v[7] = 1;
    if (chunks > HUFD_SCAN_SMALL_MAX) {
        c.v[3] = 1;
        c.v[4] = (chunks + HUFD_SCAN_RUN_CHUNKS - 1) / HUFD_SCAN_RUN_CHUNKS;
    }
    u32 n = 0;
    /* (registers, not an array in memory: no index that is not a constant) */
#pragma unroll
    for (u32 t = 0; t < 2; ++t) {
        const u32 k = chunks - 2 + t; /* the last two chunks; an item of one chunk has no chunk "-1" */
        if (chunks < 2 && t == 0) {
            continue;
        }
        const u64 left = r.in_len - (u64)k * HUFD_DEC_CHUNK_BYTES;
        if (left < (u64)HUFD_DEC_CHUNK_BYTES + 8u) {
            const u64 whole = whole_lanes_of(left);
            const bool narrow = whole <= HUFD_DEC_PACK_LANES;
            if (n == 0) {
                c.tail_chunk[0] = k;
                c.tail_narrow[0] = narrow;
            } else {
                c.tail_chunk[1] = k;
                c.tail_narrow[1] = narrow;
            }
            ++n;
            c.v[5] += narrow ? 1u : 0u;
            c.v[6] += narrow ? 0u : 1u;
            u64 holds = left * 8 / (shortest_code ? shortest_code : 1) + 1;
            holds = holds < r.out_cap ? holds : r.out_cap;
            c.tail_stage = holds > c.tail_stage ? holds : c.tail_stage;
            if (narrow) {
                c.tail_lanes = whole > c.tail_lanes ? whole : c.tail_lanes;
            } else {
                c.wide_lanes = whole > c.wide_lanes ? whole : c.wide_lanes;
            }
        }
    }
    return c;
}

/* ... and one encode item: [0] segments, [1] thread-per-item items, [2] one-tile items (a wave each), [3] large items,
 * [7] items with segments */
__device__ __forceinline__ void count_enc_item(const raw_item &r, u64 tiny_limit, u64 solo_limit, u32 *v) {
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        v[k] = 0;
    }
    const bool tiny = r.in_len <= tiny_limit && (r.in_len > 0 || r.bits);
    v[1] = tiny;
    if (tiny || r.in_len == 0) {
        return; /* (an item with nothing to do and nothing carried has no segments either: (0 + 16383) / 16384) */
    }
    if (r.in_len <= solo_limit) {
        v[2] = 1;
        return;
    }
    const u32 segs = (u32)((r.in_len + HUFD_ENC_SEG_BYTES - 1) / HUFD_ENC_SEG_BYTES);
    v[0] = segs;
    v[7] = 1;
    v[3] = segs > HUFD_SCAN_SMALL_MAX;
}

/* the workgroup's sums of every counter, for the scan over workgroups */
template <bool ENC>
__global__ __launch_bounds__(kPlanThreads) void plan_count_kernel(
    hufd_item_source src, u32 n_items, const plan_decision *decision, u32 shortest_code, u64 solo_limit, plan_stats *stats, u32 *block_sums) {
    u32 *sums = reinterpret_cast<u32 *>(dyn_lds); /* [kPlanVec] */
    u64 *maxes = reinterpret_cast<u64 *>(dyn_lds + 64); /* tail_stage, tail_lanes, wide_lanes, then the pieces in 64 bits */
    if (threadIdx.x < kPlanVec) {
        sums[threadIdx.x] = 0;
    }
    if (threadIdx.x < 4) {
        maxes[threadIdx.x] = 0;
    }
    __syncthreads();
    const u32 i = blockIdx.x * kPlanThreads + threadIdx.x;
    u32 v[kPlanVec];
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        v[k] = 0;
    }
    if (i < n_items) {
        const raw_item r = load_item<ENC>(src, i);
        if (ENC) {
            count_enc_item(r, decision->tiny_limit, solo_limit, v);
        } else {
            const dec_counts c = count_dec_item(r, decision->tiny_limit, shortest_code);
#pragma unroll
            for (u32 k = 0; k < kPlanVec; ++k) {
                v[k] = c.v[k];
            }
            if (c.tail_stage) {
                atomicMax(&maxes[0], c.tail_stage);
            }
            if (c.tail_lanes) {
                atomicMax(&maxes[1], c.tail_lanes);
            }
            if (c.wide_lanes) {
                atomicMax(&maxes[2], c.wide_lanes);
            }
        }
    }
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        const u32 w = wave_sum(v[k]);
        if ((threadIdx.x & (kWave - 1)) == 0 && w) {
            atomicAdd(&sums[k], w);
        }
    }
    /* the 32-bit sums above wrap for a batch of 2^32 pieces or more; this one does not, and the host refuses by it */
    if (v[0]) {
        atomicAdd(&maxes[3], (u64)v[0]);
    }
    __syncthreads();
    if (threadIdx.x < kPlanVec) {
        block_sums[(u64)blockIdx.x * kPlanVec + threadIdx.x] = sums[threadIdx.x];
    }
    if (threadIdx.x == 0 && maxes[3]) {
        atomicAdd(&stats->pieces, maxes[3]);
    }
    if (!ENC && threadIdx.x == 0) {
        if (maxes[0]) {
            atomicMax(&stats->tail_stage, maxes[0]);
        }
        if (maxes[1]) {
            atomicMax(&stats->tail_lanes, maxes[1]);
        }
        if (maxes[2]) {
            atomicMax(&stats->wide_lanes, maxes[2]);
        }
    }
}

/* one workgroup: the sums of the workgroups in front of each (in place), and the totals.  A thread takes a stretch of
 * consecutive workgroups (its eight sums in registers), the threads' sums are scanned once a counter, and the thread writes
 * its stretch's positions: eight scans of the workgroup whatever the batch -- a scan a counter and 256 workgroups of the
 * batch was 128 of them for a million items, 83 us */
__global__ __launch_bounds__(kPlanThreads) void plan_scan_blocks_kernel(u32 *block_sums, u32 n_blocks, plan_decision *decision) {
    u32 *slots = reinterpret_cast<u32 *>(dyn_lds); /* [kPlanThreads / 64] */
    const u32 per = (n_blocks + kPlanThreads - 1) / kPlanThreads;
    const u32 lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
    u32 mine[kPlanVec];
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        mine[k] = 0;
    }
    for (u32 b = lo; b < hi; ++b) {
#pragma unroll
        for (u32 k = 0; k < kPlanVec; ++k) {
            mine[k] += block_sums[(u64)b * kPlanVec + k];
        }
    }
    u32 before[kPlanVec];
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        u32 total = 0;
        before[k] = block_exclusive_sum<kPlanThreads>(mine[k], slots, total);
        __syncthreads();
        if (threadIdx.x == 0) {
            decision->totals[k] = total;
        }
    }
    for (u32 b = lo; b < hi; ++b) {
#pragma unroll
        for (u32 k = 0; k < kPlanVec; ++k) {
            const u32 here = block_sums[(u64)b * kPlanVec + k];
            block_sums[(u64)b * kPlanVec + k] = before[k];
            before[k] += here;
        }
    }
}

/* ------------------------------------------------------------------ pass 3: the records */

/* every counter's position of this thread's item: the workgroup's base + the sums of the threads in front */
__device__ __forceinline__ void positions_of(const u32 *v, const u32 *block_base, u32 *slots, u32 *pos) {
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        u32 total = 0;
        pos[k] = block_base[k] + block_exclusive_sum<kPlanThreads>(v[k], slots, total);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kPlanThreads) void plan_dec_fill_kernel(
    hufd_item_source src, u32 n_items, const plan_decision *decision, u32 shortest_code, const u32 *block_sums, hufd_dec_item *items,
    u32 *tiny_list, u32 *tail_list, u32 *large_list, u32 *run_list) {
    u32 *slots = reinterpret_cast<u32 *>(dyn_lds);
    const u32 i = blockIdx.x * kPlanThreads + threadIdx.x;
    raw_item r;
    dec_counts c;
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        c.v[k] = 0;
    }
    if (i < n_items) {
        r = load_item<false>(src, i);
        c = count_dec_item(r, decision->tiny_limit, shortest_code);
    }
    u32 pos[kPlanVec];
    positions_of(c.v, block_sums + (u64)blockIdx.x * kPlanVec, slots, pos);
    if (i >= n_items) {
        return;
    }
    hufd_dec_item it;
    it.in_off = r.in_off;
    it.in_len = r.in_len;
    it.out_off = r.out_off;
    it.out_cap = r.out_cap;
    it.first_bit = r.bits;
    it.first_chunk = pos[0];
    it.n_chunks = c.v[0];
    it.tiny = c.v[1] || c.v[2] ? 1u : 0u;
    items[i] = it;
    if (c.v[1]) {
        tiny_list[pos[1]] = i; /* from the front, in item order */
    }
    if (c.v[2]) {
        tiny_list[n_items - 1 - pos[2]] = i; /* the items a wave takes: from the back */
    }
    if (c.v[3]) {
        large_list[2 * pos[3]] = i;
        large_list[2 * pos[3] + 1] = pos[4];
        for (u32 k = 0; k < c.v[4]; ++k) {
            run_list[2 * (pos[4] + k)] = i;
            run_list[2 * (pos[4] + k) + 1] = k;
        }
    }
    /* (the chunks with few whole lanes first: several of those share a workgroup; the wide ones behind them) */
    const u32 narrow_total = (u32)decision->totals[5];
    u32 narrow_at = pos[5], wide_at = narrow_total + pos[6];
#pragma unroll
    for (u32 t = 0; t < 2; ++t) {
        if (c.tail_chunk[t] != ~0u) {
            if (c.tail_narrow[t]) {
                tail_list[narrow_at++] = pos[0] + c.tail_chunk[t];
            } else {
                tail_list[wide_at++] = pos[0] + c.tail_chunk[t];
            }
        }
    }
}

__global__ __launch_bounds__(kPlanThreads) void plan_enc_fill_kernel(
    hufd_item_source src, u32 n_items, const plan_decision *decision, u64 solo_limit, const u32 *block_sums, hufd_enc_item *items,
    u32 *tiny_list, u32 *large_list, u32 *solo_list) {
    u32 *slots = reinterpret_cast<u32 *>(dyn_lds);
    const u32 i = blockIdx.x * kPlanThreads + threadIdx.x;
    raw_item r;
    u32 v[kPlanVec];
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        v[k] = 0;
    }
    if (i < n_items) {
        r = load_item<true>(src, i);
        count_enc_item(r, decision->tiny_limit, solo_limit, v);
    }
    u32 pos[kPlanVec];
    positions_of(v, block_sums + (u64)blockIdx.x * kPlanVec, slots, pos);
    if (i >= n_items) {
        return;
    }
    hufd_enc_item it;
    it.in_off = r.in_off;
    it.in_len = r.in_len;
    it.out_off = r.out_off;
    it.out_cap = r.out_cap;
    it.ovf_bits = r.bits;
    it.ovf_pattern = r.bits == 0 ? 0u : (r.bits >= 32 ? r.pattern : r.pattern & ((1u << r.bits) - 1u));
    it.eos_padding = r.eos;
    it.first_seg = pos[0];
    it.n_segs = v[0];
    it.tiny = v[1] ? 1u : (v[2] ? 2u : 0u);
    items[i] = it;
    if (v[1]) {
        tiny_list[pos[1]] = i;
    }
    if (v[2]) {
        solo_list[pos[2]] = i;
    }
    if (v[3]) {
        large_list[pos[3]] = i;
    }
}

/* a thread a segment: its item is the last one whose first segment is not behind it (items without segments share
 * theirs with the item behind them) */
__global__ __launch_bounds__(kPlanThreads) void plan_enc_segs_kernel(const hufd_enc_item *items, u32 n_items, u32 n_segs, hufd_enc_seg *segs) {
    const u32 s = blockIdx.x * kPlanThreads + threadIdx.x;
    if (s >= n_segs) {
        return;
    }
    u32 lo = 0, hi = n_items;
    while (hi - lo > 1) {
        const u32 mid = lo + (hi - lo) / 2;
        if (items[mid].first_seg <= s) {
            lo = mid;
        } else {
            hi = mid;
        }
    }
    const hufd_enc_item it = items[lo];
    const u32 k = s - it.first_seg;
    const u64 off = (u64)k * HUFD_ENC_SEG_BYTES;
    const u64 left = it.in_len > off ? it.in_len - off : 0;
    const u64 after = left > HUFD_ENC_SEG_BYTES ? left - HUFD_ENC_SEG_BYTES : 0;
    hufd_enc_seg sd;
    sd.in_off = it.in_off + off;
    sd.len = (u32)(left < HUFD_ENC_SEG_BYTES ? left : HUFD_ENC_SEG_BYTES);
    sd.item = lo;
    sd.index = k;
    sd.flags = (k == 0 ? 1u : 0u) | (k + 1 == it.n_segs ? 2u : 0u);
    sd.next_len = (u32)(after < HUFD_ENC_SEG_BYTES ? after : HUFD_ENC_SEG_BYTES);
    sd.reserved = 0;
    segs[s] = sd;
}

u32 plan_blocks(u32 n_items) {
    return (n_items + kPlanThreads - 1) / kPlanThreads;
}

template <bool ENC>
int plan_count(
    const hufd_item_source *src, u32 n_items, u64 class0, u64 class1, u64 per_byte, u32 shortest_code, u64 solo_limit, void *scratch,
    hufk_plan_totals *out, hipStream_t st) {
    plan_stats *stats = reinterpret_cast<plan_stats *>(scratch);
    plan_decision *decision = reinterpret_cast<plan_decision *>(reinterpret_cast<u8 *>(scratch) + sizeof(plan_stats));
    u32 *block_sums = reinterpret_cast<u32 *>(reinterpret_cast<u8 *>(scratch) + sizeof(plan_stats) + sizeof(plan_decision));
    const u32 blocks = plan_blocks(n_items);
    hipError_t e = hipMemsetAsync(stats, 0, sizeof(plan_stats), st);
    if (e != hipSuccess) {
        return (int)e;
    }
    hipLaunchKernelGGL(
        (plan_stats_kernel<ENC>), dim3(blocks < kPlanStatsBlocks ? blocks : kPlanStatsBlocks), dim3(kPlanThreads), sizeof(plan_stats), st, *src,
        n_items, class0, class1, stats);
    hipLaunchKernelGGL(plan_decide_kernel, dim3(1), dim3(64), 0, st, stats, class0, class1, per_byte, decision);
    hipLaunchKernelGGL((plan_count_kernel<ENC>), dim3(blocks), dim3(kPlanThreads), 128, st, *src, n_items, decision, shortest_code, solo_limit, stats, block_sums);
    hipLaunchKernelGGL(plan_scan_blocks_kernel, dim3(1), dim3(kPlanThreads), 128, st, block_sums, blocks, decision);
    struct {
        plan_stats s;
        plan_decision d;
    } host;
    e = hipMemcpyAsync(&host, scratch, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) {
        e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        return (int)e;
    }
#pragma unroll
    for (u32 k = 0; k < kPlanVec; ++k) {
        out->totals[k] = host.d.totals[k];
    }
    out->tiny_limit = host.d.tiny_limit;
    out->shortest = ~host.s.not_shortest;
    out->longest = host.s.longest;
    out->largest_out_cap = host.s.largest_out_cap;
    out->worst_bits = (uint32_t)host.s.worst_bits;
    out->invalid = (uint32_t)host.s.invalid;
    out->totals[0] = host.s.pieces; /* (the 64-bit sum: what the scan's 32 bits say only while it is below 2^32) */
    out->tail_stage = host.s.tail_stage;
    out->tail_lanes = host.s.tail_lanes;
    out->wide_lanes = host.s.wide_lanes;
    return (int)hipGetLastError();
}

} /* namespace */

extern "C" {

uint64_t hufk_plan_scratch_bytes(uint64_t n_items) {
    return sizeof(plan_stats) + sizeof(plan_decision) + (uint64_t)plan_blocks((u32)n_items) * kPlanVec * sizeof(u32) + 256;
}

int hufk_decode_plan_count(
    const struct hufd_item_source *src, uint32_t n_items, uint64_t per_byte, uint32_t shortest_code_bits, void *scratch,
    struct hufk_plan_totals *totals, void *stream) {
    return plan_count<false>(src, n_items, HUFD_DEC_COOP_BYTES, HUFD_DEC_TINY_BYTES, per_byte, shortest_code_bits, 0, scratch, totals, (hipStream_t)stream);
}

int hufk_decode_plan_fill(
    const struct hufd_item_source *src, uint32_t n_items, uint32_t shortest_code_bits, const void *scratch, struct hufd_dec_item *items,
    uint32_t *tiny_list, uint32_t *tail_list, uint32_t *large_list, uint32_t *run_list, void *stream) {
    const plan_decision *decision = reinterpret_cast<const plan_decision *>(reinterpret_cast<const u8 *>(scratch) + sizeof(plan_stats));
    const u32 *block_sums = reinterpret_cast<const u32 *>(reinterpret_cast<const u8 *>(scratch) + sizeof(plan_stats) + sizeof(plan_decision));
    hipLaunchKernelGGL(
        plan_dec_fill_kernel, dim3(plan_blocks(n_items)), dim3(kPlanThreads), 64, (hipStream_t)stream, *src, n_items, decision,
        shortest_code_bits, block_sums, items, tiny_list, tail_list, large_list, run_list);
    return (int)hipGetLastError();
}

int hufk_encode_plan_count(
    const struct hufd_item_source *src, uint32_t n_items, uint64_t class0, uint64_t class1, uint64_t per_byte, uint64_t solo_limit,
    void *scratch, struct hufk_plan_totals *totals, void *stream) {
    return plan_count<true>(src, n_items, class0, class1, per_byte, 0, solo_limit, scratch, totals, (hipStream_t)stream);
}

int hufk_encode_plan_fill(
    const struct hufd_item_source *src, uint32_t n_items, uint32_t n_segs, uint64_t solo_limit, const void *scratch,
    struct hufd_enc_item *items, struct hufd_enc_seg *segs, uint32_t *tiny_list, uint32_t *large_list, uint32_t *solo_list, void *stream) {
    const plan_decision *decision = reinterpret_cast<const plan_decision *>(reinterpret_cast<const u8 *>(scratch) + sizeof(plan_stats));
    const u32 *block_sums = reinterpret_cast<const u32 *>(reinterpret_cast<const u8 *>(scratch) + sizeof(plan_stats) + sizeof(plan_decision));
    hipLaunchKernelGGL(
        plan_enc_fill_kernel, dim3(plan_blocks(n_items)), dim3(kPlanThreads), 64, (hipStream_t)stream, *src, n_items, decision,
        solo_limit, block_sums, items, tiny_list, large_list, solo_list);
    if (n_segs) {
        hipLaunchKernelGGL(
            plan_enc_segs_kernel, dim3((n_segs + kPlanThreads - 1) / kPlanThreads), dim3(kPlanThreads), 0, (hipStream_t)stream,
            (const hufd_enc_item *)items, n_items, n_segs, segs);
    }
    return (int)hipGetLastError();
}

} /* extern "C" */
