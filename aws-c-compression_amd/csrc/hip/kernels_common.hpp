/*
 * HIP kernels of the Huffman hot path for gfx950 (MI355X, CDNA4, wave64): what every translation unit of csrc/hip
 * starts from -- types, the diagnostic stamps, wave and block primitives, input helpers, the LDS bit image.
 *
 * The kernels by path: encode_kernels.hip (count / scan / pack, one pass, short items, one-block calls),
 * decode_sync_kernels.hip (transfer functions of sub-chunks: the long way, regular chunks, packed end-of-stream chunks,
 * second chances, the end of a stream), decode_items_kernels.hip (items without chunks: a thread, a wave, a workgroup,
 * blocks across the chip, long codes, codes of one length, one-block calls), decode_emit_kernels.hip (symbols out),
 * decode_launch.hip (entry scan, plans on the device, the launch sequence).
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
#ifndef HUFFMAN_AMD_KERNELS_COMMON_HPP
#define HUFFMAN_AMD_KERNELS_COMMON_HPP
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

/* a value that is the same in every lane of the wave, as a scalar (what is derived from it -- a record's address, its
 * fields -- then lives in scalar registers and is loaded by the scalar unit) */
__device__ __forceinline__ u32 wave_uniform(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}
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

/* ------------------------------------------------------------------ what the GPU build and the CPU test build (tests/emu) spell differently:
 * every `#if defined(__HIP_DEVICE_COMPILE__)` of the kernels is in this header or in decode_common.hpp, none in a kernel */

/* a value the compiler cannot trace any further back (what it would otherwise work out early and keep in registers, or build
 * twice, is then worked out where it is used); nothing under tests/emu */
__device__ __forceinline__ void opaque(u32 &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
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

__device__ __forceinline__ u32 funnel(u32 hi, u32 lo, u32 shift /* 0..31 */) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, shift);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> shift);
#endif
}

/* the same with the shift taken from the low five bits of a word that holds other things above them (v_alignbit_b32 looks at
 * those five bits only: no AND in front of it) */
__device__ __forceinline__ u32 funnel_by_low5(u32 hi, u32 lo, u32 word) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, word);
#else
    return (u32)(((((u64)hi) << 32) | lo) >> (word & 31u));
#endif
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

/* lanes of a wave take turns in program order (what one lane wrote to LDS, another reads behind this): nothing on the GPU
 * but a line the compiler does not move LDS accesses across, a rendezvous of the wave's fibers under tests/emu */
__device__ __forceinline__ void wave_step() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_wave_barrier();
#else
    (void)__ballot(1);
#endif
}

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


} /* namespace */

#endif /* HUFFMAN_AMD_KERNELS_COMMON_HPP */
