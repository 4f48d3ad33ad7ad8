/*
 * HIP kernels of the Huffman hot path for gfx950 (MI355X, CDNA4, wave64).
 *
 * Encode  (replaces the per-symbol loop of reference source/huffman.c:161-173 and
 *          the bit packer :59-105):
 *   enc_count   per segment: sum of code lengths, first symbol without a code
 *   enc_scan_*  per item: exclusive bit offset of every segment, outcome of the call
 *               (closed form of the reference's stop conditions, DESIGN.md "Encode")
 *   enc_pack    per segment: codes -> bitstream image in LDS -> aligned 16-byte stores
 *
 * Decode  (replaces the window/walk loop of reference source/huffman.c:230-281 and
 *          the refill :196-211):
 *   dec_sync    per sub-chunk: transfer function entry state -> (exit state, symbols),
 *               folded per chunk
 *   dec_scan_*  per item: true entry state and output offset of every chunk
 *   dec_emit    per chunk: true entry state of every lane, table walk, symbols staged
 *               in LDS, aligned 16-byte stores
 *
 * No MFMA anywhere: this is byte/bit work bound by HBM and LDS, not a contraction.
 * All LDS lives in the dynamic region with 16-byte carves (guide: Guideline 17).
 */
#include <hip/hip_runtime.h>
#include <numeric>
#include <type_traits>

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "device_types.h"
#include "huffman_kernels.h"

namespace {

typedef unsigned int u32;
typedef uint64_t u64;
typedef unsigned short u16;
typedef unsigned char u8;

constexpr u32 kWave = 64;
constexpr u64 kNoBit = ~0ull;

HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, dyn_lds)

/*
 * Diagnostic build only (-DHUFD_STAMPS, profiles/tools/stamps.py): wave 0 of every
 * workgroup adds the shader clock at phase boundaries into a table of its own; differences
 * of the sums / workgroups = average phase length.  Never compiled into the product.
 */
#ifdef HUFD_STAMPS
/* one private row of 8 clocks per workgroup and kernel: plain stores, no contention */
__device__ unsigned long long *hufd_stamp_rows; /* [3][HUFD_STAMP_MAX_WG][8], set by hufk_stamps_attach */
#define HUFD_STAMP_MAX_WG 131072u
#define HUFD_STAMP(kernel, phase)                                                                                      \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < HUFD_STAMP_MAX_WG) {                                                      \
            hufd_stamp_rows[((u64)(kernel)*HUFD_STAMP_MAX_WG + blockIdx.x) * 8 + (phase)] =                            \
                (unsigned long long)clock64();                                                                         \
        }                                                                                                              \
    } while (0)
/* the same, summed over the turns of a persistent workgroup: kept in LDS (a stamp must not add a memory round trip to
 * the phase it measures), written out once by HUFD_STAMP_FLUSH */
#define HUFD_STAMP_DECL __shared__ unsigned long long hufd_stamp_acc[8];
#define HUFD_STAMP_ZERO                                                                                                \
    do {                                                                                                               \
        if (threadIdx.x < 8) {                                                                                         \
            hufd_stamp_acc[threadIdx.x] = 0;                                                                           \
        }                                                                                                              \
    } while (0)
#ifdef HUFD_STAMPS_WHY /* slots 3 .. 5 count events instead of clocks */
#define HUFD_STAMP_TIMED(phase) ((phase) < 3 || (phase) > 5)
#else
#define HUFD_STAMP_TIMED(phase) true
#endif
#define HUFD_STAMP_ADD(kernel, phase)                                                                                  \
    do {                                                                                                               \
        if (threadIdx.x == 0 && HUFD_STAMP_TIMED(phase)) {                                                             \
            hufd_stamp_acc[phase] += (unsigned long long)clock64();                                                    \
        }                                                                                                              \
    } while (0)
#define HUFD_STAMP_COUNT(phase, n)                                                                                     \
    do {                                                                                                               \
        if (threadIdx.x == 0) {                                                                                        \
            hufd_stamp_acc[phase] += (unsigned long long)(n);                                                          \
        }                                                                                                              \
    } while (0)
#define HUFD_STAMP_FLUSH(kernel)                                                                                       \
    do {                                                                                                               \
        if (threadIdx.x < 8 && blockIdx.x < HUFD_STAMP_MAX_WG) {                                                       \
            hufd_stamp_rows[((u64)(kernel)*HUFD_STAMP_MAX_WG + blockIdx.x) * 8 + threadIdx.x] =                        \
                hufd_stamp_acc[threadIdx.x];                                                                           \
        }                                                                                                              \
    } while (0)
#else
#define HUFD_STAMP(kernel, phase)
#define HUFD_STAMP_ADD(kernel, phase)
#define HUFD_STAMP_DECL
#define HUFD_STAMP_ZERO
#define HUFD_STAMP_COUNT(phase, n)
#define HUFD_STAMP_FLUSH(kernel)
#endif

/* 16 / 4 bytes at any address (one load: the memory system takes any alignment) */
struct __attribute__((packed, aligned(1))) unaligned_uint4 {
    u32 x, y, z, w;
};
struct __attribute__((packed, aligned(1))) unaligned_u32 {
    u32 x;
};

__device__ __forceinline__ u32 round16(u32 x) {
    return (x + 15u) & ~15u;
}

/* ------------------------------------------------------------------ wave / block primitives */

__device__ __forceinline__ u32 wave_inclusive_sum(u32 v, u32 lane) {
#pragma unroll
    for (u32 d = 1; d < kWave; d <<= 1) {
        const u32 up = __shfl_up(v, d);
        if (lane >= d) {
            v += up;
        }
    }
    return v;
}

/* the same sum with data-parallel-primitive moves instead of LDS permutes: six adds, no LDS traffic */
__device__ __forceinline__ u32 wave_inclusive_sum_dpp(u32 v, u32 lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); /* row_bcast:15 into rows 1 and 3 */
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); /* row_bcast:31 into rows 2 and 3 */
    return v;
#else
    return wave_inclusive_sum(v, lane);
#endif
}

__device__ __forceinline__ u32 wave_min(u32 v) {
#pragma unroll
    for (u32 d = kWave / 2; d > 0; d >>= 1) {
        const u32 o = __shfl_xor(v, d);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (u32 d = kWave / 2; d > 0; d >>= 1) {
        v += __shfl_xor(v, d);
    }
    return v;
}

/* Exclusive sum over the workgroup; `slots` is LDS scratch of THREADS/64 words. */
template <u32 THREADS>
__device__ __forceinline__ u32 block_exclusive_sum(u32 v, u32 *slots, u32 &total) {
    constexpr u32 kWaves = THREADS / kWave;
    const u32 lane = threadIdx.x & (kWave - 1);
    const u32 wave = threadIdx.x / kWave;
    const u32 incl = wave_inclusive_sum(v, lane);
    if (lane == kWave - 1) {
        slots[wave] = incl;
    }
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (u32 w = 0; w < kWaves; ++w) {
        const u32 t = slots[w];
        before += w < wave ? t : 0;
        all += t;
    }
    __syncthreads();
    total = all;
    return before + incl - v;
}

/* ------------------------------------------------------------------ input helpers */

/* 16 input symbols of one lane: an aligned 16-byte load when possible. */
__device__ __forceinline__ void load_group(const u8 *src, u32 valid, bool aligned, u32 (&w)[4]) {
    if (valid == 16 && aligned) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src);
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
        return;
    }
    w[0] = w[1] = w[2] = w[3] = 0;
    for (u32 j = 0; j < valid; ++j) {
        w[j >> 2] |= (u32)src[j] << (8 * (j & 3));
    }
}

__device__ __forceinline__ u32 group_byte(const u32 (&w)[4], u32 j) {
    return (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
}

/* big-endian 32-bit word `index` of a byte range, zero past `valid_bytes` */
__device__ __forceinline__ u32 load_be32(const u8 *base, u64 index, u64 valid_bytes, bool aligned) {
    const u64 at = index * 4;
    if (aligned && at + 4 <= valid_bytes) {
        return __builtin_bswap32(reinterpret_cast<const unaligned_u32 *>(base + at)->x); /* (whole words: one load, at any address) */
    }
    u32 v = 0;
#pragma unroll
    for (u32 b = 0; b < 4; ++b) {
        if (at + b < valid_bytes) {
            v |= (u32)base[at + b] << (24 - 8 * b);
        }
    }
    return v;
}

/* the first `n_words` big-endian words of `bytes` bytes at `src` (zeros behind them) into `dst`: 16 bytes a load where
 * 16 lie inside (a thread that reads a stream's end on its own pays per request: 9 instead of 34 for 135 bytes) */
__device__ __forceinline__ void load_be32_run(u32 *dst, const u8 *src, u64 bytes, u32 n_words) {
    u32 k = 0;
    for (; k + 4 <= n_words && (u64)k * 4 + 16 <= bytes; k += 4) {
        const unaligned_uint4 v = *reinterpret_cast<const unaligned_uint4 *>(src + k * 4);
        dst[k + 0] = __builtin_bswap32(v.x);
        dst[k + 1] = __builtin_bswap32(v.y);
        dst[k + 2] = __builtin_bswap32(v.z);
        dst[k + 3] = __builtin_bswap32(v.w);
    }
    for (; k < n_words; ++k) {
        dst[k] = (u64)k * 4 < bytes ? load_be32(src, k, bytes, true) : 0u;
    }
}

/* ------------------------------------------------------------------ LDS bit image */

/* OR the low `nbits` (1..32) bits of `pattern` into the MSB-first bit image at bit `q`. */
__device__ __forceinline__ void image_or_bits(u32 *img, u32 q, u32 pattern, u32 nbits) {
    const u64 left = ((u64)pattern << (64 - nbits)) >> (q & 31);
    const u32 hi = (u32)(left >> 32), lo = (u32)left;
    atomicOr(&img[q >> 5], hi);
    if (lo) {
        atomicOr(&img[(q >> 5) + 1], lo);
    }
}

/*
 * Copies image bytes [lo, hi) to global memory.  Image byte b lives in bits
 * 31-8*(b&3).. of word b>>2 and belongs at gbase + b, where gbase is 16-byte
 * aligned, so whole 16-byte rows go out as aligned dwordx4 stores.
 */
template <u32 THREADS>
__device__ __forceinline__ void image_store(const u32 *img, u8 *gbase, u32 lo, u32 hi) {
    if (hi <= lo) {
        return;
    }
    const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
    if (row_lo <= row_hi) {
        for (u32 b = lo + threadIdx.x; b < row_lo * 16; b += THREADS) {
            gbase[b] = (u8)(img[b >> 2] >> (24 - 8 * (b & 3)));
        }
        /* four rows per thread in flight: the LDS reads are issued together, then the stores */
        for (u32 r = row_lo + threadIdx.x; r < row_hi; r += 4 * THREADS) {
            uint4 v[4];
#pragma unroll
            for (u32 u = 0; u < 4; ++u) {
                const u32 ru = r + u * THREADS;
                v[u] = *reinterpret_cast<const uint4 *>(&img[(ru < row_hi ? ru : r) * 4]);
            }
#pragma unroll
            for (u32 u = 0; u < 4; ++u) {
                const u32 ru = r + u * THREADS;
                if (ru < row_hi) {
                    uint4 o;
                    o.x = __builtin_bswap32(v[u].x);
                    o.y = __builtin_bswap32(v[u].y);
                    o.z = __builtin_bswap32(v[u].z);
                    o.w = __builtin_bswap32(v[u].w);
                    *reinterpret_cast<uint4 *>(gbase + (u64)ru * 16) = o;
                }
            }
        }
        for (u32 b = row_hi * 16 + threadIdx.x; b < hi; b += THREADS) {
            gbase[b] = (u8)(img[b >> 2] >> (24 - 8 * (b & 3)));
        }
    } else {
        for (u32 b = lo + threadIdx.x; b < hi; b += THREADS) {
            gbase[b] = (u8)(img[b >> 2] >> (24 - 8 * (b & 3)));
        }
    }
}

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

/* A segment descriptor is the same in every lane: say so, and it lives in scalar registers. */
__device__ __forceinline__ hufd_enc_seg uniform_seg(const hufd_enc_seg *p) {
    hufd_enc_seg d = *p;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)d.in_off);
    const u32 hi = __builtin_amdgcn_readfirstlane((u32)(d.in_off >> 32));
    d.in_off = ((u64)hi << 32) | lo;
    d.len = __builtin_amdgcn_readfirstlane(d.len);
    d.item = __builtin_amdgcn_readfirstlane(d.item);
    d.index = __builtin_amdgcn_readfirstlane(d.index);
    d.flags = __builtin_amdgcn_readfirstlane(d.flags);
    d.next_len = __builtin_amdgcn_readfirstlane(d.next_len);
#endif
    return d;
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

/*
 * Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also
 * drains the vector-memory counter, which would stall on an LDS-DMA prefetch or on the
 * copy-out stores that are meant to stay in flight (guide: "Pipelining across barriers").
 */
__device__ __forceinline__ void barrier_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    __syncthreads();
#endif
}

/* 16 bytes global -> LDS without a register in between (global_load_lds_dwordx4) */
__device__ __forceinline__ void lds_dma16(const void *global_src, void *lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)global_src, (__attribute__((address_space(3))) void *)lds_dst,
        16, 0, 0);
#else
    memcpy(lds_dst, global_src, 16);
#endif
}

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

/* lanes of a wave take turns in program order: nothing on the GPU, a rendezvous of the wave's fibers under tests/emu */
__device__ __forceinline__ void wave_step() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_wave_barrier();
#else
    (void)__ballot(1);
#endif
}

__device__ __forceinline__ u32 funnel(u32 hi, u32 lo, u32 shift /* 0..31 */) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, shift);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> shift);
#endif
}

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
/*
 * The same loads for a poll loop: load and wait in one piece of assembly, so that the compiler sees a value, not a
 * load in flight.  (A load it knows of inside the loop makes every wait behind the loop a wait for everything --
 * among it the next tile's symbols, which are meant to stay in flight.)  Polling with atomics (which are carried out
 * at the memory side and cannot be served from a cache) was tried: 2 000 waves asking for one word that way take
 * turns at ~12 ns each -- 18 ms instead of 0.6.
 */
__device__ __forceinline__ u32 word_load_now(const u32 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
#else
    return *p;
#endif
}
__device__ __forceinline__ u64 granule_load_now(const u64 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
#else
    return *p;
#endif
}
/*
 * A tile's arrival (flagged word + add to its group), issued without the compiler knowing of a store in flight: with
 * loads and a store outstanding together it makes every wait a wait for everything (it has to assume that the two
 * kinds complete in any order), and the wait behind this is for the old tile's offsets only -- the next tile's symbols
 * are meant to stay in flight.  A counted wait that does not count these two still covers the loads it is for: at
 * most two of the operations it lets stand are these, the others are loads, which complete in order.
 */
__device__ __forceinline__ void arrival_quiet(u32 *flag_word, u32 flagged, u64 *group, u64 add) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dword %0, %1, off sc1\n\tglobal_atomic_add_x2 %2, %3, off"
                 :
                 : "v"(flag_word), "v"(flagged), "v"(group), "v"(add)
                 : "memory");
#else
    *flag_word = flagged;
    *group += add;
#endif
}
__device__ __forceinline__ u64 granule_add(u64 *p, u64 v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* a value that is the same in every lane, as a scalar */
__device__ __forceinline__ u32 uniform32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return __shfl(x, 0);
#endif
}
__device__ __forceinline__ u64 uniform64(u64 x) {
    return ((u64)uniform32((u32)(x >> 32)) << 32) | uniform32((u32)x);
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
#if defined(__HIP_DEVICE_COMPILE__)
    /* (the lane number as a value the compiler cannot trace: what is derived from it -- the lane's place in the image --
     * is then worked out here, two instructions, instead of being kept in registers, or spilled, across the turn) */
    asm volatile("" : "+v"(lane));
#endif
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

template <u32 NW> /* words an oct can touch: 4 for codes of at most 12 bits, 5 up to 15 */
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
    u32 fail_tile /* a tile whose wave is to give up (tests of the way back); HUFD_NONE32: none */) {

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
    const u32 n_tiles = n_segs * kTilesPerSeg;

    /* (t is a scalar, the descriptor arrays are read-only: these are scalar loads, no vector registers, no vector-memory wait) */
    auto describe = [&](u32 t) -> op_tile {
        op_tile d;
        const u32 tc = t < n_tiles ? t : n_tiles - 1; /* (a tile past the end is never worked on) */
        d.t = t;
        d.s = tc / kTilesPerSeg;
        d.w4 = tc % kTilesPerSeg;
        d.seg = segs[d.s];
        const u8 *src = d_in + d.seg.in_off;
        d.carried = items[d.seg.item].ovf_bits;
        d.carried_pattern = items[d.seg.item].ovf_pattern;
        d.item_first_tile = items[d.seg.item].first_seg * kTilesPerSeg;
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
    if (blockIdx.x == 0 && wave == kPackWaves - 1) {
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
                if (++spins > kOpSpinLimit || uniform32(word_load_now(&ctl[1])) != 0) {
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
    const u32 stride = gridDim.x * kPackWaves - 1;
    u32 t_new = blockIdx.x * kPackWaves + wave - (blockIdx.x ? 1u : 0u);
    if (t_new >= n_tiles) {
        return;
    }
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
        bool gave_up = old.t == fail_tile;
        for (u32 spins = 0; !gave_up; ++spins) {
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
            if (spins > kOpSpinLimit || uniform32(word_load_now(&ctl[1])) != 0) {
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
                ctl[1] = 1; /* the kernels of the three-kernel road, queued behind this one, see it and do the launch over */
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
        if (old.first_tile) {
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

        const u64 out_cap = uniform64(items[seg.item].out_cap);
        const u64 cap_bits = out_cap > (~0ull >> 3) ? ~0ull : out_cap * 8;
        /* ---- what enc_finish_kernel turns into the call's outcome: the item's bit total ... */
        if (lane == 0 && old.w4 == kTilesPerSeg - 1 && (seg.flags & 2u)) {
            item_total[seg.item] = bn; /* (tiles behind the item's last symbol hold no bits) */
        }
        /* ... and, when the output is too short, which tile holds the symbol whose last bit reaches the capacity edge (exactly
         * one does) and where that tile's bits start: enc_finish_kernel looks the symbol up */
        if (lane == 0 && bw < cap_bits && cap_bits <= bn) {
            hufd_enc_result *r = &results[seg.item];
            r->consumed = (u64)seg.index * HUFD_ENC_SEG_BYTES + old.w4 * kTileBytes; /* the item's symbols in front of the tile */
            r->total_bits = bw;
            r->ovf_bits = old.n_sym;
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
        nxt = describe(t_new + stride);
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
        if (lane == 0) {
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
        if (lane == 0) {
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
        if (whole) {
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
        t_new += stride;
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
 * (enc_finish_item; every symbol has a code here).  For a call that ran out of room the tile holding
 * the capacity edge has left a note in the item's result record -- the item's symbols in front of
 * the tile (consumed), the stream bit its codes start at (total_bits), its symbols (ovf_bits) -- and
 * a wave of this workgroup reads that tile again to find the symbol whose last bit reaches the edge:
 * `consumed` counts up to and with it, the overflow is what of its code did not fit
 * (source/huffman.c:88-98).
 */
constexpr u32 kFinishItems = 64; /* per workgroup of 256: a wave of it per 16 items that may each need a tile read again */
constexpr u32 kFinishLdsBytes = 256 * (8 + 8 + 4 + 4 + 4) + 16;
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
                        * the three-kernel road behind this kernel does the launch over, records included */) {

    if (gave_up[0] != 0) {
        return;
    }
    u64 *note_first = reinterpret_cast<u64 *>(dyn_lds), *note_bit = note_first + 256; /* kFinishLdsBytes */
    u32 *code_len = reinterpret_cast<u32 *>(note_bit + 256), *noted = code_len + 256, *note_syms = noted + 256;
    u32 &n_noted = note_syms[256];
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const u32 i = blockIdx.x * kFinishItems + tid;
    if (tid == 0) {
        n_noted = 0;
    }
    code_len[tid] = (u32)(tb.enc_table[tid] >> 32);
    __syncthreads();
    if (tid < kFinishItems && i < n_items && !items[i].tiny /* enc_tiny's */) {
        const hufd_enc_item it = items[i];
        const u64 total = it.n_segs ? item_total[i] : it.ovf_bits;
        const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;
        const hufd_enc_result note = results[i];
        hufd_enc_result rs;
        /* (no segment is named as the edge's: nothing is listed for enc_pack_kernel) */
        enc_finish_item(it, total, HUFD_NONE32, 0, 0, 0, HUFD_NONE32, careful_list, careful_count, &states[i], &rs);
        results[i] = rs;
        if (rs.status == HUFD_ENC_SHORT && it.ovf_bits < cap_bits) {
            const u32 k = atomicAdd(&n_noted, 1u);
            noted[k] = i;
            note_first[k] = note.consumed;
            note_bit[k] = note.total_bits;
            note_syms[k] = note.ovf_bits;
        }
    }
    __syncthreads();
    for (u32 k = wave; k < n_noted; k += blockDim.x / kWave) {
        const hufd_enc_item it = items[noted[k]];
        const u8 *src = d_in + it.in_off + note_first[k];
        const u32 n_sym = note_syms[k];
        const u32 target = (u32)(it.out_cap * 8 - note_bit[k]); /* the edge, in bits from the tile's first code: 1 .. the tile's bits */
        /* lane l counts symbols 64 l .. 64 l + 63 (four loads, all on their way before the first look-up), the lane that
         * holds the edge walks them once more */
        const u32 from = lane * 64 < n_sym ? lane * 64 : n_sym, to = from + 64 < n_sym ? from + 64 : n_sym;
        u32 wd[16];
        if (to - from == 64) {
#pragma unroll
            for (u32 g = 0; g < 4; ++g) {
                const unaligned_uint4 q = *reinterpret_cast<const unaligned_uint4 *>(src + from + 16 * g);
                wd[4 * g] = q.x, wd[4 * g + 1] = q.y, wd[4 * g + 2] = q.z, wd[4 * g + 3] = q.w;
            }
        } else {
            /* the tile's last symbols: one by one (nothing behind the item is read) */
#pragma unroll
            for (u32 g = 0; g < 16; ++g) {
                wd[g] = 0;
            }
            for (u32 j = from; j < to; ++j) {
                const u32 at = j - from;
                const u32 v = (u32)src[j] << (8 * (at & 3u));
#pragma unroll
                for (u32 g = 0; g < 16; ++g) {
                    wd[g] |= g == (at >> 2) ? v : 0u;
                }
            }
        }
        const u32 mine_n = to - from;
        u32 sum = 0;
#pragma unroll
        for (u32 b = 0; b < 64; ++b) {
            sum += b < mine_n ? code_len[(wd[b >> 2] >> (8 * (b & 3u))) & 0xFFu] : 0u;
        }
        const u32 incl = wave_inclusive_sum_dpp(sum, lane);
        u32 rel = incl - sum;
        if (rel < target && target <= incl) {
#pragma unroll
            for (u32 b = 0; b < 64; ++b) {
                const u32 sym = (wd[b >> 2] >> (8 * (b & 3u))) & 0xFFu;
                const u32 len = b < mine_n ? code_len[sym] : 0u;
                if (rel < target && target <= rel + len) {
                    const u32 left = rel + len - target;
                    hufd_enc_result *r = &results[noted[k]];
                    r->consumed = note_first[k] + from + b + 1;
                    r->ovf_bits = left;
                    r->ovf_pattern = left ? ((u32)tb.enc_table[sym] & ((1u << left) - 1u)) : 0u;
                }
                rel += len;
            }
        }
    }
}

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
    const u32 *list_count) {
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
constexpr u32 kFastHopelessRows = 6;      /* most lanes of a wave with several heads after this many: not regular either (dec_sync_lean) */

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

/* row_walk::row with the number of certain steps known to the compiler and the table given as an LDS offset */
template <u32 SURE, bool STEP_BY_STEP = false>
__device__ __forceinline__ u32 lean_row(u32 state, u32 hi, u32 lo, u32 table, const row_walk &rw) {
    const u64 pair = ((u64)hi << 32) | lo;
    if (!STEP_BY_STEP) {
#pragma unroll
        for (u32 i = 0; i < SURE; ++i) {
            state += lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
        }
    }
    while ((state & 0xFFFFu) > rw.thr) {
        state += lds_word_at(((u32)(pair >> (state & 63u)) & rw.mask) | table);
    }
    return state;
}

/*
 * The sync kernel for regular chunks (round 1's dec_sync_fast, retired in round 5, with fewer instructions: it was bound by them, 297 M vector
 * instructions per GiB, 17.7 per symbol, at one per 4 cycles and SIMD).  Same phases, same tables out; what is
 * different is what a step of a walk costs: the number of certain steps a row is known to the compiler (no loop
 * around them), a table entry's address is one instruction (lean_row), the words are byte-swapped once.
 */
template <u32 LB, u32 SURE, bool TAIL = false> /* TAIL: the chunks listed in tail_chunks (a stream ends in them) */
__global__ __launch_bounds__(HUFD_DEC_LANES, 8) void dec_sync_lean_kernel(
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
    u32 *slow_list, /* chunks inside a stream that are not regular by this kernel's rules but whose first sub-chunk's walks
                     * do meet: dec_sync_guess tries them its way */
    u32 *slow_count,
    u32 *long_list, /* the others that are not regular: dec_sync's (may be the same list as slow_list) */
    u32 *long_count) {

    HUFD_STAMP(0, 0);
    lean_shared<LB> &sh = *reinterpret_cast<lean_shared<LB> *>(dyn_lds);
    const u32 ns = tb.n_states;
    const u32 lane = threadIdx.x;
    const u32 c = TAIL ? tail_chunks[blockIdx.x] : blockIdx.x;
    const hufd_chunk_rec rec = chunk_rec[c]; /* (one load: not chunk -> item -> its record) */
    const u64 valid = rec.valid;
    const u8 *src = d_in + rec.src_off;
    if (!TAIL && valid < (u64)HUFD_DEC_CHUNK_BYTES + 8u) {
        return; /* holds the end of its stream: the other instantiation's */
    }
    /* the lanes whose sub-chunk and the 8 bytes behind it lie inside the stream */
    const u32 n_full = !TAIL ? HUFD_DEC_LANES : (valid >= 8u ? (u32)((valid - 8u) / HUFD_DEC_SUB_BYTES) : 0u);
    const bool active = !TAIL || lane < n_full;
    if (TAIL && n_full == 0 && tb.lut_bits <= HUFD_DEC_MAX_LUT_BITS) {
        /* fewer than 136 bytes: no lane is whole, and the whole chunk is one thread's work in dec_sync_tail / dec_emit_tail */
        if (lane == 0) {
            chunk_regular[c] = 3;
        }
        return;
    }
    const row_walk rw(LB, tb.max_bits);
    const u32 table = lds_offset_of(sh.wlut);
    /* (a chunk may lie at any address: the loads need no alignment) */
    const bool eligible = n_full >= 1 && tb.lut_bits <= LB && tb.max_bits <= HUFD_DEC_MAX_LUT_BITS &&
                          rw.sure >= SURE && (table & ((4u << LB) - 1u)) == 0;
    if (!eligible) {
        if (lane == 0) {
            chunk_regular[c] = 0;
            long_list[atomicAdd(long_count, 1u)] = c;
        }
        return;
    }
    u16 *cp = cp_tab + (u64)c * kCpRows * HUFD_DEC_LANES + lane;
    /* The sub-chunk's words and the table entries this lane will put into LDS are asked for together, at most four
     * entries at a time -- a workgroup's time is mostly the latency of what it loads in front of its first walk, and a
     * loop of load, wait, store over the table was a fifth of that. */
    constexpr u32 kLutPerLane = (1u << LB) / HUFD_DEC_LANES, kLutBatch = 4;
    const auto table_share = [&](u32 j0) {
        u32 lut_raw[kLutBatch];
#pragma unroll
        for (u32 j = 0; j < kLutBatch; ++j) {
            lut_raw[j] = tb.dec_lut[(lane + (j0 + j) * HUFD_DEC_LANES) >> (LB - tb.lut_bits)];
        }
#pragma unroll
        for (u32 j = 0; j < kLutBatch; ++j) {
            const u32 len = lut_raw[j] & 0xFFu;
            sh.wlut[lane + (j0 + j) * HUFD_DEC_LANES] = 0x10000u - (len ? len : kWalkDeadLen);
            sh.hops[lane + (j0 + j) * HUFD_DEC_LANES] = (u16)(len ? 1u << len : 0u);
        }
    };
    if (TAIL && lane >= kWave && (lane & ~(kWave - 1)) >= n_full) {
        /* a wave wholly behind the stream's whole lanes: never reached, as far as this kernel knows (dec_sync_tail follows
         * the true path through the one or two sub-chunks the stream ends in and rewrites their records).  Its share of
         * the table done, it leaves -- the barriers below count the waves that are still there -- and its slots go to
         * another workgroup. */
#pragma unroll
        for (u32 j0 = 0; j0 < kLutPerLane; j0 += kLutBatch) {
            table_share(j0);
        }
#pragma unroll
        for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
            cp[qq * HUFD_DEC_LANES] = 0;
        }
        lane_count[(u64)c * HUFD_DEC_LANES + lane] = 0;
        cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)(kExitStop << 12);
        return;
    }
    u32 w[kFastRows];
    {
        /* (TAIL: a lane behind the stream's whole lanes reads sub-chunk 0 again -- no branch, no second set of
         * registers for "nothing", and words that are codes; what it makes of them is never looked at) */
        const u32 mine = active ? lane : 0u;
        const unaligned_uint4 *line = reinterpret_cast<const unaligned_uint4 *>(src + (u64)mine * HUFD_DEC_SUB_BYTES);
#pragma unroll
        for (u32 q = 0; q < kSubWords / 4; ++q) {
            const unaligned_uint4 v = line[q];
            w[4 * q + 0] = v.x;
            w[4 * q + 1] = v.y;
            w[4 * q + 2] = v.z;
            w[4 * q + 3] = v.w;
        }
        w[kSubWords] = reinterpret_cast<const unaligned_u32 *>(src + (u64)(mine + 1) * HUFD_DEC_SUB_BYTES)->x;
    }
#pragma unroll
    for (u32 j0 = 0; j0 < kLutPerLane; j0 += kLutBatch) {
        table_share(j0);
    }
#pragma unroll
    for (u32 r = 0; r < kFastRows; ++r) {
        w[r] = __builtin_bswap32(w[r]);
    }
    if (lane == 0) {
        sh.bad = 0;
#pragma unroll
        for (u32 r = 0; r <= kFastMaxMeet; ++r) {
            sh.sub0[r] = w[r];
        }
#pragma unroll
        for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
            sh.wave_sum[wv] = 0; /* (TAIL: of the waves that have left) */
        }
    }
    __syncthreads();
    HUFD_STAMP(0, 1);

    /* U: all entry states as one mask of heads per row, until every lane of the wave is down to one */
    u64 heads = active ? (1ull << ns) - 1ull : 0ull;
    u32 meet_row = 0; /* the same for the whole wave */
    bool one = false, settled = false;
    /* a wave most of whose lanes still follow several walks after kFastHopelessRows rows is looking at a stream whose
     * walks do not fall into step (one symbol over and over: as many walks as its code has bits, for ever) -- ten more
     * rows of all of them, and then the one walk, were 0.8 ms of a 1.8 ms decode of 256 MiB of such symbols */
    bool hopeless = false;
#pragma unroll
    for (u32 r = 0; r < kFastMaxMeet; ++r) {
        if (!settled && !hopeless) {
            heads = r == 0 ? union_first_row<LB>(ns, active, w[0], w[1], sh.hops) : union_row_fast<LB>(heads, w[r], w[r + 1], sh.hops);
            one = heads != 0 && (heads & (heads - 1)) == 0;
            meet_row = r + 1;
            settled = __all(one || heads == 0);
            if (r + 1 == kFastHopelessRows) {
                hopeless = !settled && __popcll(__ballot(!one && heads != 0)) > kWave - kWave / 8;
            }
        }
    }

    HUFD_STAMP(0, 2);
    const u32 meet_bit = one ? (u32)__builtin_ctzll(heads) : 0u; /* bits into row meet_row */
    bool ok = !active || (one && settled);
    if (lane == 0) {
        sh.pad[2] = one; /* sub-chunk 0's own walks have met (whatever the wave's other lanes' have): what dec_sync_guess needs of a chunk */
    }

    /* R: the one walk from the meeting bit to the end of the sub-chunk */
    u32 state = rw.state_at(meet_bit, 0);
    u32 cp_state[kQuarters - 1] = {0, 0, 0};
    bool dead = false;
#pragma unroll
    for (u32 r = 1; r < kSubWords; ++r) {
        if (r >= meet_row && !hopeless) { /* (hopeless: no row of this walk, none of the head walks below -- the chunk is not regular) */
            if (r % (kSubWords / kQuarters) == 0) {
                cp_state[r / (kSubWords / kQuarters) - 1] = state;
            }
            state = lean_row<SURE>(state, w[r], w[r + 1], table, rw);
            dead = dead || rw.died(state);
            /* (a walk that has died drifts: the chunk is not regular then and nothing of this is kept.  TAIL: the lanes
             * behind the stream walk zeros, which need not be a code -- drifting, their state would wrap and the row
             * loop run for thousands of steps: they are put back on a row start) */
            state = TAIL ? rw.next_row(state, dead) : state + 32u;
        }
    }
    const u32 ref_count = state >> 16; /* symbols from the meeting bit to the end of the sub-chunk */
    const u32 ref_exit = rw.offset_of(state);
    ok = ok && (!active || (!dead && ref_exit < ns));
    sh.exit_state[lane] = ref_exit;
    HUFD_STAMP(0, 3);
    __syncthreads();
    HUFD_STAMP(0, 4);

    /* H: my own sub-chunk from my true entry state, to the meeting bit */
    const u32 entry = lane ? sh.exit_state[lane - 1] : 0u;
    u32 count;
    u32 head_cp = 0; /* the head walk where it enters the second quarter, when the meeting row lies behind that */
    const bool late = meet_row > kSubWords / kQuarters;
    {
        u32 st = rw.state_at(entry < ns ? entry : 0u, 0);
        bool dd = false;
#pragma unroll
        for (u32 r = 0; r < kFastMaxMeet; ++r) {
            if (r < meet_row && !hopeless) {
                if (r == kSubWords / kQuarters) {
                    head_cp = st;
                }
                st = lean_row<SURE>(st, w[r], w[r + 1], table, rw);
                dd = dd || rw.died(st);
                st = TAIL ? rw.next_row(st, dd) : st + 32u;
            }
        }
        const bool reached = !dd && rw.offset_of(st) == meet_bit;
        ok = ok && (lane == 0 || !active || reached);
        count = active ? (st >> 16) + ref_count : 0u; /* symbols of the true path that start in my sub-chunk (lanes >= 1) */
    }

    /* H: sub-chunk 0 from every entry state the chunk may be entered in (threads 0 .. ns-1), step by step: the count
     * of a walk that dies has to be right */
    u32 cand_count = 0, cand_dead = 0;
    bool cand_reached = false;
    u64 cand_alive = 0;
    if (lane < kWave && !hopeless) {
        const u32 target = __shfl(meet_bit, 0), tail0 = __shfl(ref_count, 0); /* sub-chunk 0 is lane 0's */
        u32 st = rw.state_at(lane < ns ? lane : 0u, 0);
        bool dd = false;
        u32 hi = sh.sub0[0];
        for (u32 r = 0; r < meet_row; ++r) {
            const u32 lo = sh.sub0[r + 1];
            st = lean_row<SURE, true>(st, hi, lo, table, rw);
            const bool now = rw.died(st) && !dd;
            cand_dead = now ? (st >> 16) - 1u : cand_dead; /* the step that found no code is not a symbol */
            dd = dd || now;
            st = rw.next_row(st, dd);
            hi = lo;
        }
        cand_reached = !dd && lane < ns && rw.offset_of(st) == target;
        cand_alive = __ballot(cand_reached);
        cand_count = (st >> 16) + tail0;
    }

    HUFD_STAMP(0, 5);
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
            chunk_regular[c] = 0;
            if (TAIL || !sh.pad[2]) {
                long_list[atomicAdd(long_count, 1u)] = c;
            } else {
                slow_list[atomicAdd(slow_count, 1u)] = c;
            }
        }
        return;
    }

    /* the tables dec_scan and dec_emit read (the regular chunks' format) */
    u16 *fn_out = fn_tab + (u64)c * ns * HUFD_DEC_LANES;
    if (active) {
#pragma unroll
    for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
        /* a checkpoint in front of the meeting row is not on the one walk: the first one is then taken from the head
         * walk (not for lane 0, whose head is only known to dec_scan) */
        const bool usable = (qq + 1) * (kSubWords / kQuarters) >= meet_row;
        u32 tail = ref_count - (cp_state[qq] >> 16), bits = rw.offset_of(cp_state[qq]);
        bool have = usable;
        if (qq == 0 && late && lane != 0) {
            tail = count - (head_cp >> 16);
            bits = rw.offset_of(head_cp);
            have = true;
        }
        cp[qq * HUFD_DEC_LANES] = (u16)(have ? 0x8000u | (bits << 11) | tail : 0u);
    }
    lane_count[(u64)c * HUFD_DEC_LANES + lane] = (u16)(lane ? count : ref_count);
    cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)((lane ? 1u << entry : (u32)cand_alive) | (ref_exit << 12));
    } else {
        /* (TAIL) behind the whole lanes, in a wave that has some: as for the waves that left */
#pragma unroll
        for (u32 qq = 0; qq + 1 < kQuarters; ++qq) {
            cp[qq * HUFD_DEC_LANES] = 0;
        }
        lane_count[(u64)c * HUFD_DEC_LANES + lane] = 0;
        cp[(kQuarters - 1) * HUFD_DEC_LANES] = (u16)(kExitStop << 12);
    }
    if (TAIL && lane + 1 == n_full) {
        tail_entry[c] = ref_exit;
    }
    if (lane == 0) {
        chunk_regular[c] = TAIL ? 2 : 1;
    }
    if (lane < ns) {
        u32 rest = 0;
#pragma unroll
        for (u32 wv = 0; wv < HUFD_DEC_LANES / kWave; ++wv) {
            rest += sh.wave_sum[wv];
        }
        const u32 first_exit = sh.exit_state[0];
        const u32 last_exit = sh.exit_state[HUFD_DEC_LANES - 1];
        fn_out[(u64)lane * HUFD_DEC_LANES] =
            cand_reached ? fn_pack(false, first_exit, cand_count & 0x7FFu) : fn_pack(true, 0, cand_dead);
        /* (TAIL: symbols of the whole lanes only, and no exit yet: dec_sync_tail adds the stream's last symbols and how it ends) */
        chunk_fn[(u64)c * ns + lane] =
            cand_reached ? wide_pack(false, TAIL ? 0u : last_exit, cand_count + rest) : wide_pack(true, 0, cand_dead);
    }
    HUFD_STAMP(0, 6);
}

/* ------------------------------------------------------------------ decode: sync, several short end-of-stream chunks a workgroup */

/*
 * A batch of items of a few KiB each is all chunks that streams END in, one per item, with a handful of whole lanes: 19
 * of 256 for a 2 KiB item.  dec_sync_lean<TAIL> gives such a chunk a workgroup of its own -- one wave of 19 lanes, a table
 * of 4 KiB filled, three barriers -- and the batch decodes at a seventh of a stream's rate (bench.py, the mid_items leg).
 * Here a workgroup takes SEVERAL such chunks: its 256 threads are `slots` of `width` lanes (the most whole lanes any
 * end-of-stream chunk of the launch has, at least 16), a chunk a slot, so that the waves are full and the table and the
 * barriers are shared.  Same phases per lane, same records out as dec_sync_lean<TAIL>; what is per chunk there (the
 * candidates' walks of sub-chunk 0, the sum of the lanes' symbols, the verdict) is per slot here, through LDS words
 * instead of wave votes, because a slot need not start on a wave.
 */
constexpr u32 kPackMaxSlots = 16;
constexpr u32 kPackMinChunks = HUFD_DEC_PACK_MIN_CHUNKS;

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
 * dec_sync_lean wants ALL entry states of EVERY sub-chunk to fall into one walk within 16 rows.  The test coder does
 * that; a coder that synchronises slowly on its own kind of data does not (codes of 4 .. 12 bits on symbols drawn to
 * match them: a quarter of the sub-chunks still have several heads after 16 rows), so none of its chunks is regular
 * and all of them take the long way at a tenth of the speed.  This kernel takes the chunks dec_sync_lean gave up on
 * (its list) and asks less: only sub-chunk 0, whose entry state nobody in the chunk can know, goes through phase U
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
#if defined(__HIP_DEVICE_COMPILE__)
            /* (the words as values the compiler cannot trace through the rounds: it otherwise builds every row's 64-bit
             * window register pair once, in front of the loop -- twice the registers, and a value spilled inside this
             * divergent loop has come back wrong on this toolchain) */
#pragma unroll
            for (u32 r = 0; r < kFastRows; ++r) {
                asm volatile("" : "+v"(w[r]));
            }
#endif
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
            slow_list[atomicAdd(slow_count, 1u)] = c; /* (chunk_regular[c] is 0 already: dec_sync_lean's) */
        }
        return;
    }

    /* the tables dec_scan and dec_emit read (the regular chunks' format).  (The lane number as a value the compiler cannot trace:
     * where the records go is worked out here, not in front of the walks where the registers are needed.) */
    u32 lane_o = lane;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane_o));
#endif
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
    const u32 *given_up,       /* dec_sync_lean's list ... */
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
 * The chunks inside a stream that dec_sync_lean and dec_sync_guess gave up: in some sub-chunk the walks from the
 * possible entry bits do not become one -- one symbol over and over (as many walks as its code has bits, each valid
 * for ever), two symbols of one length taking turns, any stream whose code lengths share a divisor.  An adversary picks
 * those; the long way (dec_sync) follows every entry's walk a bit of the stream at a time out of an LDS image, 1 ms for
 * 160 MB, and the emit kernel behind it has no checkpoints to start threads at.  Here, as for the long-code coders
 * (dec_wide_fn):
 *   dec_sync_few    a lane's sub-chunk in registers as in dec_sync_lean; from every entry bit a walk over the first two
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

/* one row of a walk whose count has to be right when it dies (dec_sync_lean's walks of sub-chunk 0's entries) */
template <u32 LB>
__device__ __forceinline__ u32 few_row(u32 st, u32 hi, u32 lo, u32 table, const row_walk &rw, bool &dd, u32 &dead_count) {
    st = lean_row<0, true>(st, hi, lo, table, rw);
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

/* ------------------------------------------------------------------ decode: the end of a stream */

/*
 * One THREAD per chunk that holds the end of a stream (after dec_sync_lean<TAIL> / dec_sync_pack, which took its whole lanes):
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

/* ------------------------------------------------------------------ decode: scan */

/* chunk entry record: [7:0] entry state, [8] reached */
__device__ __forceinline__ u32 entry_pack(u32 state, bool reached) {
    return state | (reached ? 0x100u : 0u);
}

__device__ void dec_finish_item(
    const hufd_dec_item &it,
    u64 total,
    bool stopped,
    hufd_dec_item_state *state,
    hufd_dec_result *result) {
    state->total_symbols = total;
    result->total_symbols = total;
    result->cap_bit = kNoBit;
    result->reserved = 0;
    if (!stopped) {
        /* the last code ended exactly on the last bit of the last chunk (or the item is empty) */
        result->stop_kind = HUFD_STOP_END;
        result->stop_bit = it.in_len * 8;
    } else {
        result->stop_kind = HUFD_STOP_NONE; /* the lane that stops fills these in */
        result->stop_bit = kNoBit;
    }
}

__global__ __launch_bounds__(256) void dec_scan_small_kernel(
    const hufd_dec_item *items,
    u32 n_items,
    u32 ns,
    const u32 *chunk_fn,
    u32 *chunk_entry,
    u64 *chunk_base,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {

    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_dec_item it = items[i];
    if (it.tiny) {
        return;
    }
    if (it.n_chunks > HUFD_SCAN_SMALL_MAX) {
        return;
    }
    u32 state = it.first_bit;
    u64 total = 0;
    bool stopped = false;
    for (u32 k = 0; k < it.n_chunks; ++k) {
        const u32 c = it.first_chunk + k;
        chunk_entry[c] = entry_pack(state, !stopped);
        chunk_base[c] = total;
        if (!stopped) {
            const u32 f = chunk_fn[(u64)c * ns + state];
            total += wide_count(f);
            stopped = wide_stop(f);
            state = wide_state(f);
        }
    }
    dec_finish_item(it, total, stopped, &states[i], &results[i]);
}

/*
 * Items with many chunks are scanned in RUNS of HUFD_SCAN_RUN_CHUNKS chunks, one workgroup per
 * run, in three short launches (the walk along an item is a chain of dependent table look-ups:
 * what matters is that every look-up is an LDS read and every chain is short):
 *   dec_scan_runs   the run's chunk functions into LDS, folded 16 at a time and then once more:
 *                   the run's own transfer function
 *   dec_scan_top    per item: the true path through its run functions (in LDS) -> entry state and
 *                   symbol offset of every run, outcome of the item
 *   dec_scan_apply  per run: the same fold again, then the true path through the 16 sub-runs and
 *                   through the chunks of each -> entry state and symbol offset of every chunk
 */
constexpr u32 kRunChunks = HUFD_SCAN_RUN_CHUNKS, kSubRun = HUFD_SCAN_SUB_CHUNKS, kSubRuns = kRunChunks / kSubRun;

/* chunk functions of run `k` of item `it` -> fn[chunk][state]; sub-run functions -> sub[sub-run][state].  Returns the run's chunk count. */
__device__ __forceinline__ u32 scan_run_load(const hufd_dec_item &it, u32 k, u32 ns, const u32 *chunk_fn, u32 *fn, u32 *sub) {
    const u32 lo = k * kRunChunks;
    const u32 n = it.n_chunks - lo < kRunChunks ? it.n_chunks - lo : kRunChunks;
    const u32 *src = chunk_fn + (u64)(it.first_chunk + lo) * ns;
    for (u32 i = threadIdx.x; i < n * ns; i += blockDim.x) {
        fn[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x < kSubRuns * ns) {
        const u32 j = threadIdx.x / ns, start = threadIdx.x % ns;
        const u32 first = j * kSubRun;
        const u32 cnt = first < n ? (n - first < kSubRun ? n - first : kSubRun) : 0;
        sub[j * ns + start] =
            wide_pack(chain_fold(cnt, start, [&](u32 i, u32 stt) { return fn[(first + i) * ns + stt]; }));
    }
    __syncthreads();
    return n;
}

static uint32_t scan_run_lds_bytes(uint32_t ns) {
    return kRunChunks * ns * 4 + kSubRuns * ns * 4 + kSubRuns * 4 + kSubRuns * 8 + 16;
}

__global__ __launch_bounds__(256) void dec_scan_runs_kernel(
    const hufd_dec_item *items, const u32 *runs, u32 ns, const u32 *chunk_fn, u32 *run_fn) {
    u32 *fn = reinterpret_cast<u32 *>(dyn_lds);
    u32 *sub = fn + kRunChunks * ns;
    const u32 run = blockIdx.x;
    const hufd_dec_item it = items[runs[2 * run]];
    (void)scan_run_load(it, runs[2 * run + 1], ns, chunk_fn, fn, sub);
    if (threadIdx.x < ns) {
        run_fn[(u64)run * ns + threadIdx.x] = wide_pack(
            chain_fold(kSubRuns, threadIdx.x, [&](u32 j, u32 stt) { return sub[j * ns + stt]; }));
    }
}

constexpr u32 kTopTile = 1024; /* run functions of an item held in LDS at a time */

__global__ __launch_bounds__(256) void dec_scan_top_kernel(
    const hufd_dec_item *items,
    const u32 *large_items,
    u32 ns,
    const u32 *run_fn,
    u32 *run_entry,
    u64 *run_base,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {
    u32 *fn = reinterpret_cast<u32 *>(dyn_lds); /* [kTopTile][ns] */
    const u32 i = large_items[2 * blockIdx.x], run0 = large_items[2 * blockIdx.x + 1];
    const hufd_dec_item it = items[i];
    const u32 n_runs = (it.n_chunks + kRunChunks - 1) / kRunChunks;
    u32 state = it.first_bit;
    u64 total = 0;
    bool stopped = false;
    for (u32 base = 0; base < n_runs; base += kTopTile) {
        const u32 n = n_runs - base < kTopTile ? n_runs - base : kTopTile;
        __syncthreads();
        for (u32 k = threadIdx.x; k < n * ns; k += blockDim.x) {
            fn[k] = run_fn[(u64)(run0 + base) * ns + k];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (u32 k = 0; k < n; ++k) {
                run_entry[run0 + base + k] = entry_pack(state, !stopped);
                run_base[run0 + base + k] = total;
                if (!stopped) {
                    const u32 f = fn[k * ns + state];
                    total += wide_count(f);
                    stopped = wide_stop(f);
                    state = wide_state(f);
                }
            }
        }
    }
    if (threadIdx.x == 0) {
        dec_finish_item(it, total, stopped, &states[i], &results[i]);
    }
}

__global__ __launch_bounds__(256) void dec_scan_apply_kernel(
    const hufd_dec_item *items,
    const u32 *runs,
    u32 ns,
    const u32 *chunk_fn,
    const u32 *run_entry,
    const u64 *run_base,
    u32 *chunk_entry,
    u64 *chunk_base) {
    u32 *fn = reinterpret_cast<u32 *>(dyn_lds);
    u32 *sub = fn + kRunChunks * ns;
    u32 *sub_entry = sub + kSubRuns * ns;                           /* [kSubRuns] */
    u64 *sub_base = reinterpret_cast<u64 *>(sub_entry + kSubRuns);  /* [kSubRuns] */
    const u32 run = blockIdx.x;
    const hufd_dec_item it = items[runs[2 * run]];
    const u32 k = runs[2 * run + 1];
    const u32 n = scan_run_load(it, k, ns, chunk_fn, fn, sub);
    if (threadIdx.x == 0) {
        u32 state = run_entry[run] & 0xFFu;
        bool stopped = !(run_entry[run] & 0x100u);
        u64 total = run_base[run];
        for (u32 j = 0; j < kSubRuns; ++j) {
            sub_entry[j] = entry_pack(state, !stopped);
            sub_base[j] = total;
            if (!stopped) {
                const u32 f = sub[j * ns + state];
                total += wide_count(f);
                stopped = wide_stop(f);
                state = wide_state(f);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kSubRuns) {
        const u32 j = threadIdx.x;
        u32 state = sub_entry[j] & 0xFFu;
        bool stopped = !(sub_entry[j] & 0x100u);
        u64 total = sub_base[j];
        const u32 first = j * kSubRun;
        const u32 cnt = first < n ? (n - first < kSubRun ? n - first : kSubRun) : 0;
        for (u32 q = 0; q < cnt; ++q) {
            const u32 c = it.first_chunk + k * kRunChunks + first + q;
            chunk_entry[c] = entry_pack(state, !stopped);
            chunk_base[c] = total;
            if (!stopped) {
                const u32 f = fn[(first + q) * ns + state];
                total += wide_count(f);
                stopped = wide_stop(f);
                state = wide_state(f);
            }
        }
    }
}

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
    const u32 *list_count) {
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
 * dec_sync_lean: shift, mask, table, byte store, two adds a symbol.  The table entry is
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
    constexpr u32 kLutPerThread = ((1u << LB) + kEmitFastThreads - 1) / kEmitFastThreads;
    u32 lut_raw[kLutPerThread];
#pragma unroll
    for (u32 j = 0; j < kLutPerThread; ++j) {
        const u32 i = t + j * kEmitFastThreads;
        lut_raw[j] = i < (1u << LB) ? tb.dec_lut[i >> (LB - tb.lut_bits)] : 0u;
    }
    const u32 centry = chunk_entry[c];
    if (!(centry & 0x100u)) {
        return; /* the stream ended before this chunk */
    }
    const u32 s0 = centry & 0xFFu;
    const hufd_chunk_rec rec = chunk_rec[c]; /* (asked for with the chunk's entry: not chunk -> item -> its record) */
    const u64 valid = rec.valid;
    /* lanes whose sub-chunk and the 8 bytes after it lie inside the stream (dec_sync_lean: the others are idle or "careful") */
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
        const u32 i = t + j * kEmitFastThreads;
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
        /* rare: sub-chunk 0 on through the second quarter, words straight from memory */
        u32 hi = w[0][kRows];
        for (u32 r = kRows; r < 2 * kRows; ++r) {
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
        /* rare: sub-chunk 0 on through the second quarter, words straight from memory */
        u32 hi = w[0][kRows];
        for (u32 r = kRows; r < 2 * kRows; ++r) {
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

/* ------------------------------------------------------------------ decode plans: the per-chunk records */

/*
 * What a decode plan holds per CHUNK (which item it belongs to, hufd_chunk_rec) follows from the item records: built here,
 * a thread a chunk, instead of by a loop on the host -- for BASELINE configs[3] that loop and the copies of its arrays
 * were most of the time it took to make a plan, more than the launch the plan is for.  The item of chunk c is the last
 * one whose first chunk is not behind c (items without chunks share their first chunk with the item behind them).
 */
__global__ __launch_bounds__(256) void dec_plan_chunks_kernel(
    const hufd_dec_item *items, u32 n_items, u32 n_chunks, u32 *chunk_item, hufd_chunk_rec *chunk_rec) {
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) {
        return;
    }
    u32 lo = 0, hi = n_items; /* items[lo].first_chunk <= c < items[hi].first_chunk (hi == n_items: no such item) */
    while (hi - lo > 1) {
        const u32 mid = lo + (hi - lo) / 2;
        if (items[mid].first_chunk <= c) {
            lo = mid;
        } else {
            hi = mid;
        }
    }
    const hufd_dec_item it = items[lo];
    const u64 off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES, left = it.in_len - off;
    hufd_chunk_rec rec;
    rec.src_off = it.in_off + off;
    rec.out_off = it.out_off;
    rec.out_cap = it.out_cap;
    rec.valid = left < 0xFFFFFFFFull ? (u32)left : 0xFFFFFFFFu;
    rec.item = lo;
    chunk_item[c] = lo;
    chunk_rec[c] = rec;
}

/*
 * A plan whose items are ALL one thread's work (header-sized strings, the reference's production use): nothing of it
 * needs the host's attention per item -- the caller's records are copied up as they are and turned into the kernels'
 * records here; the list of thread-per-item items is every item.  (A million such items cost the host loop 27 ms.)
 */
__global__ __launch_bounds__(256) void dec_plan_tiny_items_kernel(const hufd_raw_dec_item *raw, u32 n_items, hufd_dec_item *items, u32 *tiny_list) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_raw_dec_item r = raw[i];
    hufd_dec_item it;
    it.in_off = r.in_offset;
    it.in_len = r.in_len;
    it.out_off = r.out_offset;
    it.out_cap = r.out_capacity;
    it.first_bit = r.first_bit;
    it.first_chunk = 0;
    it.n_chunks = 0;
    it.tiny = 1;
    items[i] = it;
    tiny_list[i] = i;
}

/* The decode plan of what an encode launch left (aws_huffman_amd_decode_plan_from_encode): item i is encode item i's
 * output -- where it was written, as many bytes as its record says were -- decoded to where the symbols came from.  The
 * lengths never leave the device. */
__global__ __launch_bounds__(256) void dec_plan_from_encode_kernel(
    const hufd_enc_item *enc_items, const hufd_enc_result *enc_results, u32 n_items, hufd_dec_item *items, u32 *tiny_list) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_enc_item e = enc_items[i];
    hufd_dec_item it;
    it.in_off = e.out_off;
    /* (never more than the item's room: the record of a plan that was not launched yet is whatever the memory held) */
    const u64 produced = enc_results[i].produced;
    it.in_len = produced < e.out_cap ? produced : e.out_cap;
    it.out_off = e.in_off;
    it.out_cap = e.in_len;
    it.first_bit = 0;
    it.first_chunk = 0;
    it.n_chunks = 0;
    it.tiny = 1;
    items[i] = it;
    tiny_list[i] = i;
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

/* ------------------------------------------------------------------ synthetic input */

__global__ __launch_bounds__(256) void splitmix64_fill_kernel(u8 *dst, u64 len, u64 seed) {
    const u64 draws = (len + 7) / 8;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < draws; i += (u64)gridDim.x * blockDim.x) {
        u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (i * 8 + 8 <= len && ((uintptr_t)dst & 7u) == 0) {
            reinterpret_cast<u64 *>(dst)[i] = z;
        } else {
            for (u32 b = 0; b < 8 && i * 8 + b < len; ++b) {
                dst[i * 8 + b] = (u8)(z >> (8 * b));
            }
        }
    }
}

} /* namespace */

/* ------------------------------------------------------------------ launch wrappers */

/* per device: compute units (sizes the grids of the persistent kernels) and whether hufk_init has run there */
constexpr int kMaxDevices = 64;
static int s_compute_units[kMaxDevices];
static bool s_device_ready[kMaxDevices];
static pthread_mutex_t s_init_lock = PTHREAD_MUTEX_INITIALIZER;

static int current_compute_units() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices || s_compute_units[device] <= 0) {
        return 256;
    }
    return s_compute_units[device];
}

/* workgroups of a persistent kernel that one launch keeps resident: CUs x blocks per CU */
template <typename Kernel>
static uint32_t persistent_grid(Kernel kernel, uint32_t threads, uint32_t lds_bytes, uint32_t work_items, uint32_t sgprs = 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds_bytes) != hipSuccess ||
        per_cu < 1) {
        per_cu = 1;
    }
    /* The occupancy query knows nothing of the scalar registers: a SIMD has 800 of them and a wave is given its count
     * rounded up to 16, plus 16, so the hardware admits floor(800 / that) waves a SIMD -- one block a CU fewer than the
     * query says in two bands of the count (MI355X_MICROARCH.md, "Residency and cooperative launch").  A grid whose
     * workgroups WAIT for each other must not be larger than what is resident: those kernels say how many they use
     * (kOnepassSgprs: read off the build, profiles/tools/spill_census.py prints it), and the smaller number counts. */
    if (sgprs) {
        const uint32_t waves_per_simd = 800u / ((sgprs + 15u) / 16u * 16u + 16u);
        const uint32_t waves_per_block_and_simd = (threads + 255u) / 256u;
        const uint32_t by_sgprs = waves_per_simd / waves_per_block_and_simd;
        per_cu = by_sgprs >= 1 && (int)by_sgprs < per_cu ? (int)by_sgprs : per_cu;
    }
    const uint64_t resident = (uint64_t)current_compute_units() * (uint32_t)per_cu;
    return (uint32_t)(work_items < resident ? work_items : resident);
}

/* scalar registers of enc_onepass (every instantiation: the compiler uses all 102 + VCC + the rest);
 * tests/test_library_boundary.py::test_onepass_kernels_scalar_registers holds the build to it */
constexpr uint32_t kOnepassSgprs = 106;

/* layout of the block the one-pass encoder wants zeroed before every launch (all offsets multiples of 8) */
struct onepass_layout {
    uint64_t ctl, tile_agg, group_acc, round_base, item_base, null_tile, bytes;
};

static onepass_layout onepass_layout_of(uint64_t n_segs, uint64_t n_items) {
    const uint64_t tiles = n_segs * kTilesPerSeg;
    const uint64_t groups = (tiles + kOpGroupTiles - 1) / kOpGroupTiles;
    const uint64_t rounds = (groups + kOpRoundGroups - 1) / kOpRoundGroups;
    onepass_layout l;
    l.ctl = 0;
    l.tile_agg = 32;
    l.group_acc = l.tile_agg + ((tiles * 4 + 7) & ~7ull);
    l.round_base = l.group_acc + groups * 8 * kOpGroupStride;
    l.item_base = l.round_base + (rounds + 1) * 8;
    l.null_tile = (l.item_base + n_items * 8 + 15) & ~15ull;
    l.bytes = l.null_tile + kTileBytes;
    return l;
}

extern "C" {

/* for the calling thread's current device; every device an engine is made on gets its own call (the opt-ins below are
 * per device, and so is the number of compute units), threads may race here */
int hufk_init(void) {
    /* a workgroup may use up to 160 KiB of LDS on gfx950, but dynamic LDS above 64 KiB is opt-in */
#ifdef HUFD_STAMPS
    const int lds_max = 160 * 1024 - 1024; /* the diagnostic build keeps its stamp sums in static LDS */
#else
    const int lds_max = 160 * 1024;
#endif
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) {
        return (int)hipErrorInvalidDevice;
    }
    pthread_mutex_lock(&s_init_lock);
    if (s_device_ready[device]) {
        pthread_mutex_unlock(&s_init_lock);
        return 0;
    }
    hipDeviceProp_t prop;
    s_compute_units[device] = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        s_compute_units[device] = prop.multiProcessorCount;
    }
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(&dec_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_emit_fast_kernel<10, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_emit_fast_kernel<12, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
#define HUFK_ALLOW_BIG_LDS(LBV, SUREV)                                                                                  \
    if (e == hipSuccess) {                                                                                             \
        e = hipFuncSetAttribute(                                                                                       \
            reinterpret_cast<const void *>(&dec_emit_big_kernel<LBV, SUREV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
            lds_max);                                                                                                  \
    }
    HUFK_ALLOW_BIG_LDS(10, 3)
    HUFK_ALLOW_BIG_LDS(10, 4)
    HUFK_ALLOW_BIG_LDS(10, 5)
    HUFK_ALLOW_BIG_LDS(12, 2)
#undef HUFK_ALLOW_BIG_LDS
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_sync_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_sync_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_sync_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_pack_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
            lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_pack_wave_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_pack_wave_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_onepass_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_onepass_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_wide_fn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_wide_fn_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_wide_fn_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    s_device_ready[device] = e == hipSuccess;
    pthread_mutex_unlock(&s_init_lock);
    return (int)e;
}

int hufk_encode_one_pass_applies(const struct hufd_tables *tb) {
    /* every symbol has a code (no stop inside a stream to look for), octs of 4 .. 15-bit codes */
    return tb->all_coded && tb->enc_max_bits <= 15 && tb->enc_min_bits >= 4;
}

int hufk_decode_plan_chunks(
    const struct hufd_dec_item *items, uint32_t n_items, uint32_t n_chunks, uint32_t *chunk_item, struct hufd_chunk_rec *chunk_rec,
    void *stream) {
    if (n_chunks == 0 || n_items == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        dec_plan_chunks_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, (hipStream_t)stream, items, n_items, n_chunks, chunk_item,
        chunk_rec);
    return (int)hipGetLastError();
}

int hufk_decode_plan_tiny_items(const void *raw_items, uint32_t n_items, struct hufd_dec_item *items, uint32_t *tiny_list, void *stream) {
    if (n_items == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        dec_plan_tiny_items_kernel, dim3((n_items + 255) / 256), dim3(256), 0, (hipStream_t)stream,
        (const hufd_raw_dec_item *)raw_items, n_items, items, tiny_list);
    return (int)hipGetLastError();
}

int hufk_decode_plan_from_encode(
    const struct hufd_enc_item *enc_items, const struct hufd_enc_result *enc_results, uint32_t n_items, struct hufd_dec_item *items,
    uint32_t *tiny_list, void *stream) {
    if (n_items == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        dec_plan_from_encode_kernel, dim3((n_items + 255) / 256), dim3(256), 0, (hipStream_t)stream, enc_items, enc_results, n_items,
        items, tiny_list);
    return (int)hipGetLastError();
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

static uint32_t enc_pack_lds_bytes(uint32_t img_words) {
    return ((img_words * 4 + 15) & ~15u) + 256 * 8 + 8 * 4 + (uint32_t)sizeof(enc_pack_shared) + 16;
}

static uint32_t enc_stream_lds_bytes(uint32_t img_words) {
    return ((img_words * 4 + 15) & ~15u) + (HUFD_ENC_SEG_BYTES + 16) + 256 * 4 + 8 * 4 +
           (uint32_t)sizeof(enc_pack_shared) + 16;
}

static uint32_t dec_sync_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + kGroups * tb->n_states * 4 + (2u << tb->lut_bits);
}

static uint32_t dec_emit_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + HUFD_DEC_STAGE_BYTES + kGroups * HUFD_DEC_MAX_STATES * 4 + kGroups * 8 +
           HUFD_DEC_LANES * 12 + 16 + kCpRows * HUFD_DEC_LANES * 2 + (2u << tb->lut_bits) + 16;
}

static void stage_mark(void **events, int index, hipStream_t st) {
    if (events) {
        (void)hipEventRecord((hipEvent_t)events[index], st);
    }
}

} /* extern "C" */
static void encode_three_kernels(const struct hufk_encode_args *a, hipStream_t st, const u32 *gate);
extern "C" {

int hufk_encode_launch(const struct hufk_encode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_segs == 0 && a->n_items == 0) {
        return 0;
    }
    if (a->n_segs && !a->length_only && a->single_pass && hufk_encode_one_pass_applies(&a->tables) && a->zero_block) {
        /* one pass: count + offsets + pack in one kernel, then the per-item outcome */
        const onepass_layout l = onepass_layout_of(a->n_segs, a->n_items);
        uint8_t *z = (uint8_t *)a->zero_block;
        stage_mark(a->stage_events, 0, st); /* (the clearing of the look-back words is part of what is timed) */
        (void)hipMemsetAsync(a->zero_block, 0, l.bytes, st);
        const uint32_t region = pack_region_bytes(a->tables.enc_max_bits);
        const uint32_t lds = kPackTabBytes + kPackWaves * region;
        const uint32_t work = (a->n_segs * kTilesPerSeg + kPackWaves - 1) / kPackWaves;
#define HUFK_LAUNCH_ONEPASS(NWV)                                                                                      \
    hipLaunchKernelGGL(                                                                                                \
        enc_onepass_kernel<NWV>, dim3(persistent_grid(enc_onepass_kernel<NWV>, kPackThreads, lds, work, kOnepassSgprs)), \
        dim3(kPackThreads), lds, st, a->tables, a->items, a->segs, (const u8 *)a->d_in, (u8 *)a->d_out, region,        \
        a->n_segs, (u32 *)(z + l.ctl), (u32 *)(z + l.tile_agg), (u64 *)(z + l.group_acc),                              \
        (u64 *)(z + l.round_base), (u64 *)(z + l.item_base), a->item_total, a->results, (const u8 *)(z + l.null_tile),  \
        a->fail_tile ? a->n_segs * kTilesPerSeg / 2 : HUFD_NONE32)
        if (a->tables.enc_max_bits <= 12) {
            HUFK_LAUNCH_ONEPASS(4);
        } else {
            HUFK_LAUNCH_ONEPASS(5);
        }
#undef HUFK_LAUNCH_ONEPASS
        stage_mark(a->stage_events, 1, st);
        hipLaunchKernelGGL(
            enc_finish_kernel, dim3((a->n_items + kFinishItems - 1) / kFinishItems), dim3(256), kFinishLdsBytes, st, a->tables, a->items, a->n_items,
            a->item_total, (const u8 *)a->d_in, a->careful_list, a->careful_count, a->states, a->results,
            (const u32 *)(z + l.ctl) + 1);
        stage_mark(a->stage_events, 2, st);
        if (a->n_tiny) {
            hipLaunchKernelGGL(
            enc_tiny_kernel, dim3((a->n_tiny + kTinyThreads - 1) / kTinyThreads), dim3(kTinyThreads), 256 * sizeof(u64), st,
            a->tables, a->items, a->tiny_items, a->n_tiny, (const u8 *)a->d_in, (u8 *)a->d_out, a->results,
            a->length_only);
        }
        /* (nothing is left for the per-symbol packer: every segment was packed by a wave, the capacity edge found by one) */
        /* The way back, on the same stream: the three-kernel road (no waits between workgroups) queued behind the one
         * pass, every kernel of it looking first at the word a wave raises when a look-back wait runs out -- whoever
         * works on the output behind this launch finds it whole either way, without the host in between. */
        encode_three_kernels(a, st, (const u32 *)(z + l.ctl) + 1);
        stage_mark(a->stage_events, 3, st);
        return (int)hipGetLastError();
    }
    stage_mark(a->stage_events, 0, st);
    encode_three_kernels(a, st, nullptr);
    stage_mark(a->stage_events, 3, st);
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
    if (a->n_tiny != a->n_items) { /* (a plan of thread-per-item items only has nothing to scan: a thread an item that finds that out is 10 us; an EMPTY item is not such an item -- its record is written here) */
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

extern "C" {

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

constexpr uint32_t kBesideMinChunks = 1024; /* launches of fewer chunks keep to one stream */

int hufk_decode_launch(const struct hufk_decode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_items == 0) {
        return 0;
    }
    const uint32_t ns = a->tables.n_states;
    bool few = false; /* dec_sync_few ran: dec_sync_true follows behind the scan */
    stage_mark(a->stage_events, 0, st);
    if (a->n_fixed_blocks && a->tables.fixed_bits) {
        (void)hipMemsetAsync(a->states, 0xFF, (size_t)a->n_items * sizeof(hufd_dec_item_state), st);
    }
    /* the builds of the row-synchronous kernels: a decode table of up to 10 bits with 3, 4 or 5 certain steps a row (codes
     * of up to 10, 8, 6 bits; a coder of shorter codes still has more certain steps than that: the rest are asked for),
     * of 11 or 12 bits with 2 */
    const uint32_t lb_of_launch = a->tables.lut_bits <= 10 ? 10u : 12u;
    const uint32_t sure_of_coder = row_walk(lb_of_launch, a->tables.max_bits).sure;
    const uint32_t sure = lb_of_launch == 10 ? (sure_of_coder > 5 ? 5u : sure_of_coder) : (sure_of_coder > 2 ? 2u : sure_of_coder);
    if (a->n_chunks && (a->tables.max_bits > HUFD_DEC_MAX_LUT_BITS || (lb_of_launch == 10 ? sure < 3 : sure != 2))) {
        return (int)hipErrorInvalidValue; /* (a plan has chunks only for a decode table of up to 12 bits; see row_walk for the steps) */
    }
    if (a->n_chunks) {
        /* chunks inside the stream the short way; the rest, and those that turn out irregular, through the list */
        const auto sync = ns <= 8 ? dec_sync_kernel<8> : (ns <= 10 ? dec_sync_kernel<10> : dec_sync_kernel<12>);
        (void)hipMemsetAsync(a->slow_count, 0, sizeof(uint32_t), st);
        const bool some_inside = a->n_tail < a->n_chunks; /* chunks with a whole chunk + 8 bytes of stream left */
        /* two lists of chunks that are not regular by dec_sync_lean's rules: the ones dec_sync_guess may still take
         * (inside a stream, first sub-chunk's walks meet) and the ones for the long way.  The second is the emit stage's
         * list, free until then; one list where there is no dec_sync_guess for the launch. */
        const bool guessing = some_inside;
        u32 *lean_long_list = guessing ? a->emit_list : a->slow_list;
        u32 *lean_long_count = guessing ? a->emit_count : a->slow_count;
        if (guessing) {
            (void)hipMemsetAsync(a->emit_count, 0, sizeof(uint32_t), st);
        }
        /* A few chunks that streams end in beside many inside streams (one long stream: ONE): their kernels are tiny
         * and, one after the other behind the big ones, cost a tenth of the decode time in launch and drain.  They run on
         * a second stream of the engine's, beside the big kernels, forked off and joined with events. */
        /* (not for a launch of a few chunks: the fork and the join are four commands, ~30 us of a small call) */
        const bool beside = some_inside && a->n_tail && a->side_stream && a->fork_event && a->join_event &&
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
    if (n_packed) {                                                                                                    \
        hipLaunchKernelGGL(                                                                                            \
            (dec_sync_pack_kernel<LBV, SUREV>), dim3((n_packed + pack_slots - 1) / pack_slots), dim3(HUFD_DEC_LANES),   \
            (uint32_t)sizeof(pack_shared<LBV>), tst, a->tables, a->chunk_rec, a->tail_chunks, n_packed, pack_width,     \
            (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
            lean_long_list, lean_long_count);                                                                          \
    }                                                                                                                  \
    if (n_single) {                                                                                                    \
        hipLaunchKernelGGL(                                                                                            \
            (dec_sync_lean_kernel<LBV, SUREV, true>), dim3(n_single), dim3(HUFD_DEC_LANES),                             \
            (uint32_t)sizeof(lean_shared<LBV>), tst, a->tables, a->chunk_rec, single_chunks,                            \
            (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
            a->slow_list, a->slow_count, lean_long_list, lean_long_count);                                             \
    }                                                                                                                  \
    if (some_inside) {                                                                                                 \
        hipLaunchKernelGGL(                                                                                            \
            (dec_sync_lean_kernel<LBV, SUREV, false>), dim3(a->n_chunks), dim3(HUFD_DEC_LANES),                         \
            (uint32_t)sizeof(lean_shared<LBV>), st, a->tables, a->chunk_rec, a->tail_chunks,                            \
            (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, a->tail_entry,     \
            a->slow_list, a->slow_count, lean_long_list, lean_long_count);                                             \
    }
        if (lb_of_launch == 10) {
            switch (sure) {
                case 3: HUFK_LAUNCH_SYNC_LEAN(10, 3); break;
                case 4: HUFK_LAUNCH_SYNC_LEAN(10, 4); break;
                default: HUFK_LAUNCH_SYNC_LEAN(10, 5); break;
            }
        } else {
            HUFK_LAUNCH_SYNC_LEAN(12, 2);
        }
#undef HUFK_LAUNCH_SYNC_LEAN
        if (a->n_tail) {
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
        /* the chunks inside streams that dec_sync_lean gave up on: a second chance that asks less of the coder
         * (dec_sync_guess); what that gives up on goes on a second list (the emit stage's, free until then) */
        const u32 *long_list = a->slow_list, *long_count = a->slow_count;
        if (guessing) {
#define HUFK_LAUNCH_SYNC_GUESS(LBV, SUREV)                                                                              \
    hipLaunchKernelGGL(                                                                                                \
        (dec_sync_guess_kernel<LBV, SUREV>),                                                                           \
        dim3(persistent_grid(dec_sync_guess_kernel<LBV, SUREV>, HUFD_DEC_LANES, (uint32_t)sizeof(lean_shared<LBV>),     \
                             a->n_chunks)),                                                                            \
        dim3(HUFD_DEC_LANES), (uint32_t)sizeof(lean_shared<LBV>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,     \
        a->fn_tab, a->cp_tab, a->chunk_fn, a->lane_count, a->chunk_regular, (const u32 *)a->slow_list,                 \
        (const u32 *)a->slow_count, a->emit_list, a->emit_count)
            if (lb_of_launch == 10) {
                switch (sure) {
                    case 3: HUFK_LAUNCH_SYNC_GUESS(10, 3); break;
                    case 4: HUFK_LAUNCH_SYNC_GUESS(10, 4); break;
                    default: HUFK_LAUNCH_SYNC_GUESS(10, 5); break;
                }
            } else {
                HUFK_LAUNCH_SYNC_GUESS(12, 2);
            }
#undef HUFK_LAUNCH_SYNC_GUESS
            long_list = a->emit_list;
            long_count = a->emit_count;
            /* of those, the chunks inside streams whose walks do not fall into step: a few walks a lane, not the long
             * way's every bit (dec_sync_few; its list -- dec_sync_lean's, used up by now -- is for dec_sync_true below) */
            if (a->few_walks) {
                few = true;
                (void)hipMemsetAsync(a->slow_count, 0, sizeof(uint32_t), st);
                if (a->tables.lut_bits <= 10) {
                    hipLaunchKernelGGL(
                        (dec_sync_few_kernel<10>),
                        dim3(persistent_grid(dec_sync_few_kernel<10>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<10>), a->n_chunks)),
                        dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<10>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                        a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count, a->slow_list, a->slow_count);
                } else {
                    hipLaunchKernelGGL(
                        (dec_sync_few_kernel<12>),
                        dim3(persistent_grid(dec_sync_few_kernel<12>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<12>), a->n_chunks)),
                        dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<12>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                        a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count, a->slow_list, a->slow_count);
                }
            }
        }
        hipLaunchKernelGGL(
            sync, dim3(persistent_grid(sync, HUFD_DEC_LANES, dec_sync_lds_bytes(&a->tables), a->n_chunks)),
            dim3(HUFD_DEC_LANES), dec_sync_lds_bytes(&a->tables), st, a->tables, a->items, a->chunk_item, a->n_chunks,
            (const u8 *)a->d_in, a->fn_tab, a->cp_tab, a->chunk_fn, a->chunk_regular, long_list, long_count);
    }
    stage_mark(a->stage_events, 1, st);
    if (a->n_tiny != a->n_items) { /* (as in the encoder: nothing to scan, and no empty item's record to write, in a plan of thread-per-item items only) */
        hipLaunchKernelGGL(
            dec_scan_small_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->items, a->n_items, ns, a->chunk_fn,
            a->chunk_entry, a->chunk_base, a->states, a->results);
    }
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
    if (a->n_large) {
        const uint32_t lds = scan_run_lds_bytes(ns);
        hipLaunchKernelGGL(
            dec_scan_runs_kernel, dim3(a->n_runs), dim3(256), lds, st, a->items, a->runs, ns, a->chunk_fn, a->run_fn);
        hipLaunchKernelGGL(
            dec_scan_top_kernel, dim3(a->n_large), dim3(256), kTopTile * ns * 4, st, a->items, a->large_items, ns,
            (const u32 *)a->run_fn, a->run_entry, a->run_base, a->states, a->results);
        hipLaunchKernelGGL(
            dec_scan_apply_kernel, dim3(a->n_runs), dim3(256), lds, st, a->items, a->runs, ns, a->chunk_fn,
            (const u32 *)a->run_entry, (const u64 *)a->run_base, a->chunk_entry, a->chunk_base);
    }
    if (few) {
        /* dec_sync_few's chunks, now that dec_scan has said where each is entered: the true walk's records (dec_sync_true) */
        if (a->tables.lut_bits <= 10) {
            hipLaunchKernelGGL(
                (dec_sync_true_kernel<10>),
                dim3(persistent_grid(dec_sync_true_kernel<10>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<10>), a->n_chunks)),
                dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<10>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                (const u16 *)a->fn_tab, a->cp_tab, a->lane_count, a->chunk_regular, (const u32 *)a->chunk_entry,
                (const u32 *)a->slow_list, (const u32 *)a->slow_count);
        } else {
            hipLaunchKernelGGL(
                (dec_sync_true_kernel<12>),
                dim3(persistent_grid(dec_sync_true_kernel<12>, HUFD_DEC_LANES, (uint32_t)sizeof(few_shared<12>), a->n_chunks)),
                dim3(HUFD_DEC_LANES), (uint32_t)sizeof(few_shared<12>), st, a->tables, a->chunk_rec, (const u8 *)a->d_in,
                (const u16 *)a->fn_tab, a->cp_tab, a->lane_count, a->chunk_regular, (const u32 *)a->chunk_entry,
                (const u32 *)a->slow_list, (const u32 *)a->slow_count);
        }
    }
    stage_mark(a->stage_events, 2, st);
    if (a->n_chunks) {
        /* regular chunks that fit their output the short way; the rest through the list */
        (void)hipMemsetAsync(a->emit_count, 0, sizeof(uint32_t), st);
        (void)hipMemsetAsync(a->dense_count, 0, sizeof(uint32_t), st);
#define HUFK_LAUNCH_EMIT_FAST(LBV, TAILV, SUREV, GRID, STREAMV)                                                          \
    hipLaunchKernelGGL(                                                                                                \
        (dec_emit_fast_kernel<LBV, TAILV, SUREV>), dim3(GRID), dim3(kEmitFastThreads),                                  \
        emit_lds_bytes<LBV>(TAILV ? tail_stage : HUFD_DEC_STAGE_BYTES), STREAMV, a->tables, a->chunk_rec,              \
        emit_single_chunks, (const u8 *)a->d_in, (u8 *)a->d_out, (const u16 *)a->cp_tab, (const u16 *)a->lane_count,   \
        (const u8 *)a->chunk_regular, (const u32 *)a->chunk_fn, (const u32 *)a->chunk_entry,                           \
        (const u64 *)a->chunk_base, a->results, a->emit_list, a->emit_count,                                            \
        !TAILV ? a->dense_list : a->emit_list, !TAILV ? a->dense_count : a->emit_count,                                \
        TAILV ? tail_stage : HUFD_DEC_STAGE_BYTES)
        const bool some_inside = a->n_tail < a->n_chunks;
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
        (const u64 *)a->chunk_base, a->emit_list, a->emit_count, tail_stage)
        /* chunks of short codes that hold more symbols than dec_emit_fast's stage: dec_emit_big where the chunk lies
         * inside its stream; the others take the long way (dec_emit) */
        if (e_packed) {
            if (lb_of_launch == 10) {
                switch (emit_sure) {
                    case 3: HUFK_LAUNCH_EMIT_PACK(10, 3); break;
                    case 4: HUFK_LAUNCH_EMIT_PACK(10, 4); break;
                    default: HUFK_LAUNCH_EMIT_PACK(10, 5); break;
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
                    case 4: HUFK_LAUNCH_EMIT_FAST(10, true, 4, e_single, tst); break;
                    default: HUFK_LAUNCH_EMIT_FAST(10, true, 5, e_single, tst); break;
                }
            }
            if (some_inside) {
                switch (emit_sure) {
                    case 3: HUFK_LAUNCH_EMIT_FAST(10, false, 3, a->n_chunks, st); break;
                    case 4: HUFK_LAUNCH_EMIT_FAST(10, false, 4, a->n_chunks, st); break;
                    default: HUFK_LAUNCH_EMIT_FAST(10, false, 5, a->n_chunks, st); break;
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
            (const u32 *)a->chunk_entry, (const u64 *)a->chunk_base, a->results, a->emit_list, a->emit_count,          \
            (const u32 *)a->dense_list, (const u32 *)a->dense_count);                                                  \
    } while (0)
        if (lb_of_launch == 10) {
            switch (emit_sure) {
                case 3: HUFK_LAUNCH_EMIT_BIG(10, 3); break;
                case 4: HUFK_LAUNCH_EMIT_BIG(10, 4); break;
                default: HUFK_LAUNCH_EMIT_BIG(10, 5); break;
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
            (const u32 *)a->emit_count);
    }
    stage_mark(a->stage_events, 3, st);
    return (int)hipGetLastError();
}

#ifdef HUFD_STAMPS
/* diagnostic build: hand the kernels a buffer for their clock stamps (3 * 131072 * 8 u64) */
int hufk_stamps_attach(void *device_buffer) {
    unsigned long long *p = (unsigned long long *)device_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(hufd_stamp_rows), &p, sizeof(p));
}
#endif

int hufk_fill_splitmix64(void *dst, uint64_t len, uint64_t seed, void *stream) {
    if (len == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        splitmix64_fill_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (u8 *)dst, (u64)len, (u64)seed);
    return (int)hipGetLastError();
}

} /* extern "C" */
