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

#include <stdint.h>

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
        return __builtin_bswap32(*reinterpret_cast<const u32 *>(base + at));
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
        for (u32 r = row_lo + threadIdx.x; r < row_hi; r += THREADS) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&img[r * 4]);
            uint4 o;
            o.x = __builtin_bswap32(v.x);
            o.y = __builtin_bswap32(v.y);
            o.z = __builtin_bswap32(v.z);
            o.w = __builtin_bswap32(v.w);
            *reinterpret_cast<uint4 *>(gbase + (u64)r * 16) = o;
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

__global__ __launch_bounds__(HUFD_ENC_THREADS) void enc_count_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const u32 *seg_item,
    const u8 *d_in,
    u32 *seg_bits,
    u32 *seg_unk) {

    u32 *len_tab = reinterpret_cast<u32 *>(dyn_lds); /* [256] */
    u32 *slots = len_tab + 256;                       /* [8] */

    const u32 tid = threadIdx.x;
    len_tab[tid] = (u32)(tb.enc_table[tid] >> 32);
    __syncthreads();

    const u32 s = blockIdx.x;
    const hufd_enc_item it = items[seg_item[s]];
    const u64 seg_off = (u64)(s - it.first_seg) * HUFD_ENC_SEG_BYTES;
    const u32 seg_len = it.in_len > seg_off
                            ? (u32)(it.in_len - seg_off < HUFD_ENC_SEG_BYTES ? it.in_len - seg_off : HUFD_ENC_SEG_BYTES)
                            : 0u;
    const u8 *src = d_in + it.in_off + seg_off;
    const bool aligned = ((uintptr_t)src & 15u) == 0;

    u32 bits = 0, unk = HUFD_NONE32;
    for (u32 base = tid * 16; base < seg_len; base += HUFD_ENC_THREADS * 16) {
        const u32 valid = seg_len - base < 16 ? seg_len - base : 16;
        u32 w[4];
        load_group(src + base, valid, aligned, w);
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            if (j < valid) {
                const u32 len = len_tab[group_byte(w, j)];
                bits += len;
                if (len == 0 && unk == HUFD_NONE32) {
                    unk = base + j;
                }
            }
        }
    }

    bits = wave_sum(bits);
    unk = wave_min(unk);
    const u32 lane = tid & (kWave - 1), wave = tid / kWave;
    if (lane == 0) {
        slots[wave] = bits;
        slots[4 + wave] = unk;
    }
    __syncthreads();
    if (tid == 0) {
        u32 b = 0, u = HUFD_NONE32;
        for (u32 w = 0; w < HUFD_ENC_THREADS / kWave; ++w) {
            b += slots[w];
            u = slots[4 + w] < u ? slots[4 + w] : u;
        }
        seg_bits[s] = b;
        seg_unk[s] = u;
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
    hufd_enc_item_state *states,
    hufd_enc_result *results) {

    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    const hufd_enc_item it = items[i];
    if (it.n_segs > HUFD_SCAN_SMALL_MAX) {
        return;
    }
    u64 at = it.ovf_bits;
    u32 unk_seg = HUFD_NONE32, unk_idx = 0, unk_bits = 0;
    u64 unk_off = 0;
    for (u32 k = 0; k < it.n_segs; ++k) {
        const u32 s = it.first_seg + k;
        const u32 b = seg_bits[s];
        seg_bitoff[s] = at;
        if (unk_seg == HUFD_NONE32 && seg_unk[s] != HUFD_NONE32) {
            unk_seg = s;
            unk_idx = seg_unk[s];
            unk_off = at;
            unk_bits = b;
        }
        at += b;
    }
    enc_finish_item(it, at, unk_seg, unk_idx, unk_off, unk_bits, &states[i], &results[i]);
}

/* one workgroup per item with many segments */
__global__ __launch_bounds__(HUFD_SCAN_LARGE_THREADS) void enc_scan_large_kernel(
    const hufd_enc_item *items,
    const u32 *large_items,
    const u32 *seg_bits,
    const u32 *seg_unk,
    u64 *seg_bitoff,
    hufd_enc_item_state *states,
    hufd_enc_result *results) {

    u32 *slots = reinterpret_cast<u32 *>(dyn_lds);                 /* [16] wave totals */
    u32 *first_unk = slots + 16;                                    /* [1] lowest segment with a bad symbol */
    u64 *unk_off = reinterpret_cast<u64 *>(dyn_lds + 128);         /* [1] */

    const u32 i = large_items[blockIdx.x];
    const hufd_enc_item it = items[i];
    const u32 tid = threadIdx.x;
    if (tid == 0) {
        *first_unk = HUFD_NONE32;
        *unk_off = 0;
    }
    __syncthreads();

    u64 carry = it.ovf_bits;
    for (u32 base = 0; base < it.n_segs; base += HUFD_SCAN_LARGE_THREADS) {
        const u32 k = base + tid;
        const bool live = k < it.n_segs;
        const u32 b = live ? seg_bits[it.first_seg + k] : 0;
        u32 total;
        const u32 excl = block_exclusive_sum<HUFD_SCAN_LARGE_THREADS>(b, slots, total);
        if (live) {
            seg_bitoff[it.first_seg + k] = carry + excl;
            if (seg_unk[it.first_seg + k] != HUFD_NONE32) {
                atomicMin(first_unk, it.first_seg + k);
            }
        }
        carry += total;
    }
    __syncthreads();
    const u32 us = *first_unk;
    if (us != HUFD_NONE32) {
        /* the owner of that segment republishes its offset */
        for (u32 base = 0; base < it.n_segs; base += HUFD_SCAN_LARGE_THREADS) {
            if (it.first_seg + base + tid == us) {
                *unk_off = seg_bitoff[us];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const u32 ui = us != HUFD_NONE32 ? seg_unk[us] : 0;
        const u32 ub = us != HUFD_NONE32 ? seg_bits[us] : 0;
        enc_finish_item(it, carry, us, ui, *unk_off, ub, &states[i], &results[i]);
    }
}

/* ------------------------------------------------------------------ encode: pack */

struct enc_pack_shared {
    u64 unk_before;    /* stream bit at which the item's first bad symbol sits */
    u64 short_consumed;
    u32 short_found;
    u32 short_ovf_bits;
    u32 short_ovf_pattern;
    u32 halo_unknown;
};

__global__ __launch_bounds__(HUFD_ENC_THREADS) void enc_pack_kernel(
    hufd_tables tb,
    const hufd_enc_item *items,
    const hufd_enc_item_state *states,
    const u32 *seg_item,
    const u32 *seg_bits,
    const u64 *seg_bitoff,
    const u8 *d_in,
    u8 *d_out,
    hufd_enc_result *results,
    u32 img_words) {

    u32 *img = reinterpret_cast<u32 *>(dyn_lds);
    u64 *tab = reinterpret_cast<u64 *>(dyn_lds + round16(img_words * 4));
    u32 *slots = reinterpret_cast<u32 *>(tab + 256);
    enc_pack_shared *sh = reinterpret_cast<enc_pack_shared *>(slots + 8);

    const u32 tid = threadIdx.x;
    const u32 s = blockIdx.x;
    const u32 item_index = seg_item[s];
    const hufd_enc_item it = items[item_index];
    const hufd_enc_item_state st = states[item_index];
    if (st.unk_seg != HUFD_NONE32 && s > st.unk_seg) {
        return; /* past the bad symbol: the reference never gets here */
    }

    const u32 k = s - it.first_seg;
    const u64 seg_off = (u64)k * HUFD_ENC_SEG_BYTES;
    const u32 seg_len = it.in_len > seg_off
                            ? (u32)(it.in_len - seg_off < HUFD_ENC_SEG_BYTES ? it.in_len - seg_off : HUFD_ENC_SEG_BYTES)
                            : 0u;
    const bool last_seg = k + 1 == it.n_segs;
    const u8 *src = d_in + it.in_off + seg_off;
    const bool aligned = ((uintptr_t)src & 15u) == 0;

    const u64 p0 = seg_bitoff[s];          /* stream bit of this segment's first code */
    const u64 pa = k == 0 ? 0 : p0;        /* stream bit where this workgroup's image starts */
    const u64 pend = p0 + seg_bits[s];
    const u64 cap_bits = it.out_cap > (~0ull >> 3) ? ~0ull : it.out_cap * 8;

    /* image byte 0 sits on a 16-byte boundary of the output */
    u8 *out_ptr = d_out + it.out_off;
    const u64 j0 = pa >> 3;
    const u32 mis = (u32)((uintptr_t)(out_ptr + j0) & 15u);
    u8 *gbase = out_ptr + j0 - mis;
    const u32 q0 = (u32)(p0 - 8 * j0) + 8 * mis; /* image bit of stream bit p0 */

    for (u32 i = tid; i < img_words; i += HUFD_ENC_THREADS) {
        img[i] = 0;
    }
    tab[tid] = tb.enc_table[tid];
    if (tid == 0) {
        sh->unk_before = kNoBit;
        sh->short_found = 0;
        sh->halo_unknown = 0;
    }
    __syncthreads();

    if (tid == 0 && k == 0 && it.ovf_bits) {
        image_or_bits(img, 8 * mis, it.ovf_pattern, it.ovf_bits);
    }

    const bool want_short = st.status == HUFD_ENC_SHORT || st.status == HUFD_ENC_DECIDE;
    /* capacity edge relative to p0; 0 disables the crossing test for this segment */
    u32 cap_rel = 0;
    if (want_short && cap_bits > p0) {
        cap_rel = cap_bits - p0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(cap_bits - p0);
    }
    const bool is_unk_seg = s == st.unk_seg;

    u32 carry = 0; /* bits of this segment already placed */
    for (u32 iter = 0; iter < HUFD_ENC_SEG_BYTES / (HUFD_ENC_THREADS * 16); ++iter) {
        const u32 base = (iter * HUFD_ENC_THREADS + tid) * 16;
        const u32 valid = base < seg_len ? (seg_len - base < 16 ? seg_len - base : 16) : 0;
        u32 w[4] = {0, 0, 0, 0};
        if (valid) {
            load_group(src + base, valid, aligned, w);
        }
        u64 e[16];
        u32 lane_bits = 0;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            e[j] = j < valid ? tab[group_byte(w, j)] : 0;
            lane_bits += (u32)(e[j] >> 32);
        }
        u32 total;
        u32 rel = carry + block_exclusive_sum<HUFD_ENC_THREADS>(lane_bits, slots, total);
        carry += total;

        /* the lane's codes go out as whole words; its first and last word are shared
         * with neighbours, so every word is OR-ed into the zeroed image */
        u32 q = q0 + rel;
        u32 wi = q >> 5, nb = q & 31;
        u64 acc = 0;
#pragma unroll
        for (u32 j = 0; j < 16; ++j) {
            const u32 len = (u32)(e[j] >> 32);
            const u32 pat = (u32)e[j];
            if (j < valid) {
                if (len == 0) {
                    if (is_unk_seg && base + j == st.unk_idx) {
                        sh->unk_before = p0 + rel;
                    }
                } else {
                    const u32 after = rel + len;
                    if (rel < cap_rel && after >= cap_rel) {
                        /* first symbol whose last bit reaches the capacity edge (huffman.c:88-98) */
                        sh->short_found = 1;
                        sh->short_consumed = seg_off + base + j + 1;
                        sh->short_ovf_bits = after - cap_rel;
                        sh->short_ovf_pattern = pat & (u32)((1ull << (after - cap_rel)) - 1);
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

    /* Complete the last byte this workgroup owns: with the head of the next segment's
     * codes, or with the padding when the item ends here (huffman.c:178-184). */
    if (tid == 0) {
        u32 need = (u32)((8 - (pend & 7)) & 7);
        u32 q = q0 + (u32)(pend - p0);
        if (need && !last_seg) {
            const u64 next_off = seg_off + HUFD_ENC_SEG_BYTES;
            const u64 next_len = it.in_len > next_off ? it.in_len - next_off : 0;
            const u8 *nxt = d_in + it.in_off + next_off;
            for (u32 j = 0; j < 8 && j < next_len && need; ++j) {
                const u64 ent = tab[nxt[j]];
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
            const u32 qpad = q0 + (u32)(st.total_bits - p0);
            if (pad_bits) {
                image_or_bits(img, qpad, it.eos_padding & ((1u << pad_bits) - 1), pad_bits);
            }
        }
    }
    __syncthreads();

    /* which bytes this workgroup may write */
    u32 status = st.status;
    if (status == HUFD_ENC_DECIDE) {
        /* only the segment holding the bad symbol can tell which stop comes first;
         * for the segments before it neither limit binds */
        status = (is_unk_seg && sh->unk_before < cap_bits) ? HUFD_ENC_UNKNOWN : HUFD_ENC_SHORT;
    }
    u64 limit_bytes;
    if (status == HUFD_ENC_OK) {
        limit_bytes = (st.total_bits + 7) >> 3;
    } else if (status == HUFD_ENC_UNKNOWN && is_unk_seg) {
        limit_bytes = sh->unk_before >> 3; /* the partial byte in flight is lost (huffman.c:62-64) */
    } else {
        limit_bytes = it.out_cap;
    }

    u64 jhi;
    if (last_seg) {
        jhi = status == HUFD_ENC_OK ? (st.total_bits + 7) >> 3 : pend >> 3;
    } else {
        jhi = sh->halo_unknown ? pend >> 3 : (pend + 7) >> 3;
    }
    if (jhi > limit_bytes) {
        jhi = limit_bytes;
    }
    const u64 jlo = (pa + 7) >> 3;
    if (jhi > jlo) {
        image_store<HUFD_ENC_THREADS>(img, gbase, (u32)(jlo - j0) + mis, (u32)(jhi - j0) + mis);
    }

    if (tid == 0) {
        hufd_enc_result *rs = &results[item_index];
        if (status == HUFD_ENC_UNKNOWN && is_unk_seg) {
            rs->status = HUFD_ENC_UNKNOWN;
            rs->produced = limit_bytes;
            rs->ovf_bits = 0;
            rs->ovf_pattern = 0;
        } else if (sh->short_found && (status == HUFD_ENC_SHORT)) {
            rs->status = HUFD_ENC_SHORT;
            rs->produced = it.out_cap;
            rs->consumed = sh->short_consumed;
            rs->ovf_bits = sh->short_ovf_bits;
            rs->ovf_pattern = sh->short_ovf_pattern;
        }
    }
}

/* ------------------------------------------------------------------ decode: shared pieces */

constexpr u32 kSubWords = HUFD_DEC_SUB_BYTES / 4;    /* 32 */
constexpr u32 kRowStride = HUFD_DEC_LANES + 1;        /* transposed chunk image, one pad column */
constexpr u32 kChunkWords = kSubWords * kRowStride;
constexpr u32 kMergeWords = 8;                        /* reference-path bitmap covers the first 256 bits */
constexpr u32 kGroupLanes = 16;
constexpr u32 kGroups = HUFD_DEC_LANES / kGroupLanes;

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

/* word r (0..32) of lane's sub-chunk in the transposed image; word 32 is the next lane's word 0 */
__device__ __forceinline__ u32 chunk_word(const u32 *timg, u32 lane, u32 r) {
    return timg[(r & (kSubWords - 1)) * kRowStride + lane + (r >> 5)];
}

/* the 32 stream bits starting `pos` bits into the lane's sub-chunk */
__device__ __forceinline__ u32 chunk_window(const u32 *timg, u32 lane, u32 pos) {
    const u32 r = pos >> 5;
    const u64 two = ((u64)chunk_word(timg, lane, r) << 32) | chunk_word(timg, lane, r + 1);
    return (u32)((two << (pos & 31)) >> 32);
}

/* Loads one chunk (+ one word of the next) into the transposed big-endian image. */
__device__ __forceinline__ void chunk_load(u32 *timg, const u8 *src, u64 valid_bytes) {
    const bool aligned = ((uintptr_t)src & 3u) == 0;
    for (u32 g = threadIdx.x; g < HUFD_DEC_CHUNK_BYTES / 4 + 1; g += HUFD_DEC_LANES) {
        const u32 word = load_be32(src, g, valid_bytes, aligned);
        const u32 lane = g >> 5, r = g & 31;
        timg[r * kRowStride + lane] = word; /* g == 8192 lands on lane 256, row 0: the halo column */
    }
}

__device__ __forceinline__ void lut_load(u16 *lut, const hufd_tables &tb) {
    for (u32 i = threadIdx.x; i < (1u << tb.lut_bits); i += HUFD_DEC_LANES) {
        lut[i] = tb.dec_lut[i];
    }
}

/*
 * One step of the walk (source/huffman.c:232-255 for one symbol): `pos` bits into the
 * sub-chunk, `remaining` stream bits left from the sub-chunk start.  Returns the code
 * length, or 0 with *why set when the walk ends here.
 */
__device__ __forceinline__ u32 walk_step(
    const u32 *timg, const u16 *lut, u32 lut_bits, u32 lane, u32 pos, long long remaining, u32 *symbol, u32 *why) {
    const long long rem = remaining - (long long)pos;
    if (rem <= 0) {
        *why = HUFD_STOP_END;
        return 0;
    }
    const u32 entry = lut[chunk_window(timg, lane, pos) >> (32 - lut_bits)];
    const u32 len = entry & 0xFFu;
    if (len == 0) {
        *why = HUFD_STOP_INVALID;
        return 0;
    }
    if ((long long)len > rem) {
        *why = HUFD_STOP_INCOMPLETE;
        return 0;
    }
    *symbol = entry >> 8;
    return len;
}

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

/* ------------------------------------------------------------------ decode: sync */

__global__ __launch_bounds__(HUFD_DEC_LANES) void dec_sync_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u8 *d_in,
    u16 *fn_tab,   /* [chunk][state][lane] */
    u32 *chunk_fn) /* [chunk][state] */ {

    const u32 ns = tb.n_states;
    u32 *timg = reinterpret_cast<u32 *>(dyn_lds);
    u32 *bitmap = timg + kChunkWords;                                   /* [kMergeWords][lanes] */
    u16 *cnt_at = reinterpret_cast<u16 *>(bitmap + kMergeWords * HUFD_DEC_LANES); /* [kMergeWords][lanes] */
    u16 *ftab = cnt_at + kMergeWords * HUFD_DEC_LANES;                 /* [ns][lanes] */
    u32 *gtab = reinterpret_cast<u32 *>(ftab + ns * HUFD_DEC_LANES); /* [groups][ns] */
    u16 *lut = reinterpret_cast<u16 *>(gtab + kGroups * HUFD_DEC_MAX_STATES);

    const u32 lane = threadIdx.x;
    const u32 c = blockIdx.x;
    const hufd_dec_item it = items[chunk_item[c]];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len > chunk_off ? it.in_len - chunk_off : 0;

    chunk_load(timg, d_in + it.in_off + chunk_off, valid);
    lut_load(lut, tb);
    for (u32 w = 0; w < kMergeWords; ++w) {
        bitmap[w * HUFD_DEC_LANES + lane] = 0;
        cnt_at[w * HUFD_DEC_LANES + lane] = 0;
    }
    __syncthreads();

    /* stream bits left, counted from this lane's sub-chunk start (may be <= 0) */
    const long long remaining = (long long)(valid * 8) - (long long)lane * HUFD_DEC_SUB_BITS;

    /* Phase A: the reference path from entry state 0; remember where it stepped. */
    u32 ref_count = 0, ref_exit = 0;
    bool ref_stop = false;
    {
        u32 pos = 0, cur_word = 0, cur_mask = 0;
        while (pos < HUFD_DEC_SUB_BITS) {
            if (pos < kMergeWords * 32) {
                const u32 w = pos >> 5;
                if (w != cur_word) {
                    bitmap[cur_word * HUFD_DEC_LANES + lane] = cur_mask;
                    cur_word = w;
                    cur_mask = 0;
                    cnt_at[w * HUFD_DEC_LANES + lane] = (u16)ref_count;
                }
                cur_mask |= 1u << (pos & 31);
            }
            u32 sym, why;
            const u32 len = walk_step(timg, lut, tb.lut_bits, lane, pos, remaining, &sym, &why);
            if (!len) {
                ref_stop = true;
                break;
            }
            pos += len;
            ++ref_count;
        }
        bitmap[cur_word * HUFD_DEC_LANES + lane] = cur_mask;
        ref_exit = ref_stop ? 0 : pos - HUFD_DEC_SUB_BITS;
        ftab[lane] = fn_pack(ref_stop, ref_exit, ref_count);
    }

    /* Phase B: the other entry states, one after another per lane, each until it falls
     * onto the reference path, dies, or leaves the sub-chunk on its own. */
    {
        u32 state = 1, pos = 1, steps = 0;
        while (state < ns) {
            bool done = false;
            u16 res = 0;
            if (pos >= HUFD_DEC_SUB_BITS) {
                res = fn_pack(false, pos - HUFD_DEC_SUB_BITS, steps);
                done = true;
            } else {
                if (pos < kMergeWords * 32) {
                    const u32 m = bitmap[(pos >> 5) * HUFD_DEC_LANES + lane];
                    if ((m >> (pos & 31)) & 1u) {
                        const u32 before = cnt_at[(pos >> 5) * HUFD_DEC_LANES + lane] +
                                           __popc(m & ((1u << (pos & 31)) - 1u));
                        res = fn_pack(ref_stop, ref_exit, steps + ref_count - before);
                        done = true;
                    }
                }
                if (!done) {
                    u32 sym, why;
                    const u32 len = walk_step(timg, lut, tb.lut_bits, lane, pos, remaining, &sym, &why);
                    if (!len) {
                        res = fn_pack(true, 0, steps);
                        done = true;
                    } else {
                        pos += len;
                        ++steps;
                    }
                }
            }
            if (done) {
                ftab[state * HUFD_DEC_LANES + lane] = res;
                ++state;
                pos = state;
                steps = 0;
            }
        }
    }
    __syncthreads();

    /* publish the per-lane functions for dec_emit */
    for (u32 sidx = 0; sidx < ns; ++sidx) {
        fn_tab[((u64)c * ns + sidx) * HUFD_DEC_LANES + lane] = ftab[sidx * HUFD_DEC_LANES + lane];
    }

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
 * One workgroup per item with many chunks.  Thread t owns a run of consecutive
 * chunks; run functions are folded 32 at a time, the true path is followed through
 * the 32 groups, then through each group's threads, then through each run.
 */
__global__ __launch_bounds__(HUFD_SCAN_LARGE_THREADS) void dec_scan_large_kernel(
    const hufd_dec_item *items,
    const u32 *large_items,
    u32 ns,
    const u32 *chunk_fn,
    u32 *chunk_entry,
    u64 *chunk_base,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {

    constexpr u32 T = HUFD_SCAN_LARGE_THREADS, G = 32, PER = T / G;
    u64 *grp_cnt = reinterpret_cast<u64 *>(dyn_lds);          /* [G][ns] symbols of a group of runs */
    u64 *t_base = grp_cnt + G * HUFD_DEC_MAX_STATES;           /* [T] */
    u64 *g_base = t_base + T;                                  /* [G] */
    u64 *fin_total = g_base + G;                               /* [2] */
    u32 *run_fn = reinterpret_cast<u32 *>(fin_total + 2);      /* [T][ns] wide entries, counts < 2^26 */
    u32 *grp_st = run_fn + T * HUFD_DEC_MAX_STATES;            /* [G][ns] entry_pack(exit state, !stop) */
    u32 *t_entry = grp_st + G * HUFD_DEC_MAX_STATES;           /* [T] */
    u32 *g_entry = t_entry + T;                                /* [G] */
    u32 *fin_stop = g_entry + G;                               /* [1] */

    const u32 i = large_items[blockIdx.x];
    const hufd_dec_item it = items[i];
    const u32 tid = threadIdx.x;
    const u32 per_thread = (it.n_chunks + T - 1) / T;
    const u32 run_lo = tid * per_thread < it.n_chunks ? tid * per_thread : it.n_chunks;
    const u32 run_hi = run_lo + per_thread < it.n_chunks ? run_lo + per_thread : it.n_chunks;

    for (u32 start = 0; start < ns; ++start) {
        run_fn[tid * ns + start] = wide_pack(chain_fold(run_hi - run_lo, start, [&](u32 k, u32 stt) {
            return chunk_fn[(u64)(it.first_chunk + run_lo + k) * ns + stt];
        }));
    }
    __syncthreads();
    if (tid < G * ns) {
        const u32 g = tid / ns, start = tid % ns;
        const fold_result r =
            chain_fold(PER, start, [&](u32 k, u32 stt) { return run_fn[(g * PER + k) * ns + stt]; });
        grp_cnt[g * ns + start] = r.count;
        grp_st[g * ns + start] = entry_pack(r.state, !r.stop);
    }
    __syncthreads();
    if (tid == 0) {
        u32 state = it.first_bit;
        u64 total = 0;
        bool stopped = false;
        for (u32 g = 0; g < G; ++g) {
            g_entry[g] = entry_pack(state, !stopped);
            g_base[g] = total;
            if (!stopped) {
                const u32 f = grp_st[g * ns + state];
                total += grp_cnt[g * ns + state];
                stopped = !(f & 0x100u);
                state = f & 0xFFu;
            }
        }
        *fin_total = total;
        *fin_stop = stopped;
    }
    __syncthreads();
    if (tid < G) {
        u32 state = g_entry[tid] & 0xFFu;
        bool stopped = !(g_entry[tid] & 0x100u);
        u64 total = g_base[tid];
        for (u32 k = 0; k < PER; ++k) {
            const u32 t = tid * PER + k;
            t_entry[t] = entry_pack(state, !stopped);
            t_base[t] = total;
            if (!stopped) {
                const u32 f = run_fn[t * ns + state];
                total += wide_count(f);
                stopped = wide_stop(f);
                state = wide_state(f);
            }
        }
    }
    __syncthreads();
    {
        u32 state = t_entry[tid] & 0xFFu;
        bool stopped = !(t_entry[tid] & 0x100u);
        u64 total = t_base[tid];
        for (u32 k = run_lo; k < run_hi; ++k) {
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
    }
    if (tid == 0) {
        dec_finish_item(it, *fin_total, *fin_stop != 0, &states[i], &results[i]);
    }
}

/* ------------------------------------------------------------------ decode: emit */

__global__ __launch_bounds__(HUFD_DEC_LANES) void dec_emit_kernel(
    hufd_tables tb,
    const hufd_dec_item *items,
    const u32 *chunk_item,
    const u8 *d_in,
    u8 *d_out,
    const u16 *fn_tab,
    const u32 *chunk_entry,
    const u64 *chunk_base,
    hufd_dec_result *results) {

    const u32 ns = tb.n_states;
    u32 *timg = reinterpret_cast<u32 *>(dyn_lds);
    u8 *stage = reinterpret_cast<u8 *>(timg + kChunkWords); /* [HUFD_DEC_STAGE_BYTES], 16-aligned */
    u16 *ftab = reinterpret_cast<u16 *>(stage);               /* aliases the stage until the walk starts */
    u32 *gtab = reinterpret_cast<u32 *>(stage + HUFD_DEC_STAGE_BYTES);  /* [groups][ns] */
    u32 *g_entry = gtab + kGroups * HUFD_DEC_MAX_STATES;      /* [groups] */
    u32 *g_base = g_entry + kGroups;                          /* [groups] */
    u32 *l_entry = g_base + kGroups;                          /* [lanes] */
    u32 *l_base = l_entry + HUFD_DEC_LANES;                   /* [lanes] */
    u32 *blk_count = l_base + HUFD_DEC_LANES;                 /* [4] */
    u16 *lut = reinterpret_cast<u16 *>(blk_count + 4);

    const u32 lane = threadIdx.x;
    const u32 c = blockIdx.x;
    const u32 entry = chunk_entry[c];
    if (!(entry & 0x100u)) {
        return; /* the stream ended before this chunk */
    }
    const u32 item_index = chunk_item[c];
    const hufd_dec_item it = items[item_index];
    const u64 chunk_off = (u64)(c - it.first_chunk) * HUFD_DEC_CHUNK_BYTES;
    const u64 valid = it.in_len > chunk_off ? it.in_len - chunk_off : 0;
    const u64 cbase = chunk_base[c];

    chunk_load(timg, d_in + it.in_off + chunk_off, valid);
    lut_load(lut, tb);
    for (u32 sidx = 0; sidx < ns; ++sidx) {
        ftab[sidx * HUFD_DEC_LANES + lane] = fn_tab[((u64)c * ns + sidx) * HUFD_DEC_LANES + lane];
    }
    __syncthreads();

    /* true entry state and output offset of every lane: groups, then lanes */
    if (lane < kGroups * ns) {
        const u32 g = lane / ns, start = lane % ns;
        gtab[g * ns + start] = wide_pack(chain_fold(kGroupLanes, start, [&](u32 i, u32 stt) {
            return widen(ftab[stt * HUFD_DEC_LANES + g * kGroupLanes + i]);
        }));
    }
    __syncthreads();
    if (lane == 0) {
        u32 state = entry & 0xFFu, total = 0;
        bool stopped = false;
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
    if (lane < kGroups) {
        u32 state = g_entry[lane] & 0xFFu, total = g_base[lane];
        bool stopped = !(g_entry[lane] & 0x100u);
        for (u32 i = 0; i < kGroupLanes; ++i) {
            const u32 l = lane * kGroupLanes + i;
            l_entry[l] = entry_pack(state, !stopped);
            l_base[l] = total;
            if (!stopped) {
                const u32 f = widen(ftab[state * HUFD_DEC_LANES + l]);
                total += wide_count(f);
                stopped = wide_stop(f);
                state = wide_state(f);
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

    const u32 my_entry = l_entry[lane];
    if (my_entry & 0x100u) {
        const long long remaining = (long long)(valid * 8) - (long long)lane * HUFD_DEC_SUB_BITS;
        const u64 sub_bit = (chunk_off + (u64)lane * HUFD_DEC_SUB_BYTES) * 8; /* stream bit of the sub-chunk start */
        u32 pos = my_entry & 0xFFu;
        u32 idx = l_base[lane]; /* symbol number inside the chunk */
        while (pos < HUFD_DEC_SUB_BITS) {
            u32 sym = 0, why = 0;
            const u32 len = walk_step(timg, lut, tb.lut_bits, lane, pos, remaining, &sym, &why);
            if (!len) {
                hufd_dec_result *rs = &results[item_index];
                rs->stop_kind = why;
                rs->stop_bit = sub_bit + pos;
                break;
            }
            if (idx < writable) {
                if (staged) {
                    stage[mis + idx] = (u8)sym;
                } else {
                    out_ptr[idx] = (u8)sym;
                }
            } else if (cbase + idx == it.out_cap) {
                results[item_index].cap_bit = sub_bit + pos; /* source/huffman.c:257-268 fires on this symbol */
                break;
            } else {
                break;
            }
            pos += len;
            ++idx;
        }
    }
    __syncthreads();

    if (staged && writable) {
        /* stage byte b belongs at (out_ptr - mis) + b: whole 16-byte rows go out aligned */
        u8 *gbase = out_ptr - mis;
        const u32 lo = mis, hi = mis + writable;
        const u32 row_lo = (lo + 15) >> 4, row_hi = hi >> 4;
        if (row_lo <= row_hi) {
            for (u32 b = lo + lane; b < row_lo * 16; b += HUFD_DEC_LANES) {
                gbase[b] = stage[b];
            }
            for (u32 r = row_lo + lane; r < row_hi; r += HUFD_DEC_LANES) {
                *reinterpret_cast<uint4 *>(gbase + (u64)r * 16) = *reinterpret_cast<const uint4 *>(stage + r * 16);
            }
            for (u32 b = row_hi * 16 + lane; b < hi; b += HUFD_DEC_LANES) {
                gbase[b] = stage[b];
            }
        } else {
            for (u32 b = lo + lane; b < hi; b += HUFD_DEC_LANES) {
                gbase[b] = stage[b];
            }
        }
    }
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

extern "C" {

int hufk_init(void) {
    /* a workgroup may use up to 160 KiB of LDS on gfx950, but dynamic LDS above 64 KiB is opt-in */
    const int lds_max = 160 * 1024;
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(&dec_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_sync_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&dec_scan_large_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&enc_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    }
    return (int)e;
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

static uint32_t dec_sync_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + kMergeWords * HUFD_DEC_LANES * 4 + kMergeWords * HUFD_DEC_LANES * 2 +
           tb->n_states * HUFD_DEC_LANES * 2 + kGroups * HUFD_DEC_MAX_STATES * 4 + (2u << tb->lut_bits) + 16;
}

static uint32_t dec_emit_lds_bytes(const hufd_tables *tb) {
    return kChunkWords * 4 + HUFD_DEC_STAGE_BYTES + kGroups * HUFD_DEC_MAX_STATES * 4 + kGroups * 8 +
           HUFD_DEC_LANES * 8 + 16 + (2u << tb->lut_bits) + 16;
}

static void stage_mark(void **events, int index, hipStream_t st) {
    if (events) {
        (void)hipEventRecord((hipEvent_t)events[index], st);
    }
}

int hufk_encode_launch(const struct hufk_encode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_segs == 0 && a->n_items == 0) {
        return 0;
    }
    stage_mark(a->stage_events, 0, st);
    if (a->n_segs) {
        hipLaunchKernelGGL(
            enc_count_kernel, dim3(a->n_segs), dim3(HUFD_ENC_THREADS), 256 * 4 + 8 * 4, st, a->tables, a->items,
            a->seg_item, (const u8 *)a->d_in, a->seg_bits, a->seg_unk);
    }
    stage_mark(a->stage_events, 1, st);
    hipLaunchKernelGGL(
        enc_scan_small_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->items, a->n_items, a->seg_bits,
        a->seg_unk, a->seg_bitoff, a->states, a->results);
    if (a->n_large) {
        hipLaunchKernelGGL(
            enc_scan_large_kernel, dim3(a->n_large), dim3(HUFD_SCAN_LARGE_THREADS), 256, st, a->items, a->large_items,
            a->seg_bits, a->seg_unk, a->seg_bitoff, a->states, a->results);
    }
    stage_mark(a->stage_events, 2, st);
    if (a->n_segs && !a->length_only) {
        const uint32_t img_words = hufk_enc_image_words(a->tables.max_bits);
        hipLaunchKernelGGL(
            enc_pack_kernel, dim3(a->n_segs), dim3(HUFD_ENC_THREADS), enc_pack_lds_bytes(img_words), st, a->tables,
            a->items, a->states, a->seg_item, a->seg_bits, a->seg_bitoff, (const u8 *)a->d_in, (u8 *)a->d_out,
            a->results, img_words);
    }
    stage_mark(a->stage_events, 3, st);
    return (int)hipGetLastError();
}

int hufk_decode_launch(const struct hufk_decode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_items == 0) {
        return 0;
    }
    const uint32_t ns = a->tables.n_states;
    stage_mark(a->stage_events, 0, st);
    if (a->n_chunks) {
        hipLaunchKernelGGL(
            dec_sync_kernel, dim3(a->n_chunks), dim3(HUFD_DEC_LANES), dec_sync_lds_bytes(&a->tables), st, a->tables,
            a->items, a->chunk_item, (const u8 *)a->d_in, a->fn_tab, a->chunk_fn);
    }
    stage_mark(a->stage_events, 1, st);
    hipLaunchKernelGGL(
        dec_scan_small_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->items, a->n_items, ns, a->chunk_fn,
        a->chunk_entry, a->chunk_base, a->states, a->results);
    if (a->n_large) {
        const uint32_t T = HUFD_SCAN_LARGE_THREADS, G = 32;
        const uint32_t lds = G * HUFD_DEC_MAX_STATES * 8 + T * 8 + G * 8 + 16 + T * HUFD_DEC_MAX_STATES * 4 +
                             G * HUFD_DEC_MAX_STATES * 4 + T * 4 + G * 4 + 16;
        hipLaunchKernelGGL(
            dec_scan_large_kernel, dim3(a->n_large), dim3(HUFD_SCAN_LARGE_THREADS), lds, st, a->items, a->large_items,
            ns, a->chunk_fn, a->chunk_entry, a->chunk_base, a->states, a->results);
    }
    stage_mark(a->stage_events, 2, st);
    if (a->n_chunks) {
        hipLaunchKernelGGL(
            dec_emit_kernel, dim3(a->n_chunks), dim3(HUFD_DEC_LANES), dec_emit_lds_bytes(&a->tables), st, a->tables,
            a->items, a->chunk_item, (const u8 *)a->d_in, (u8 *)a->d_out, a->fn_tab, a->chunk_entry, a->chunk_base,
            a->results);
    }
    stage_mark(a->stage_events, 3, st);
    return (int)hipGetLastError();
}

int hufk_fill_splitmix64(void *dst, uint64_t len, uint64_t seed, void *stream) {
    if (len == 0) {
        return 0;
    }
    hipLaunchKernelGGL(
        splitmix64_fill_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (u8 *)dst, (u64)len, (u64)seed);
    return (int)hipGetLastError();
}

} /* extern "C" */
