// Round-5 hardware probe (gfx950): the decode walk's step with a SMALL length table.
//
// Round 4 found the count-only walk of dec_sync_lean bound by LDS bank conflicts of its one 4 KiB table (1024 windows x
// dword, read at 32 random places a half wave: 23.5 cycles / 13.7 ns a step and SIMD at eight waves, 9.75 / 7.3 with a
// table per bank -- which for 1024 dwords is 128 KiB).  A coder's window -> length map is a step function with a dozen
// steps: indexed by the window's top 6 bits (after adding an offset that moves the block edges between neighbouring
// steps) it is 64 entries of (lower length, upper length, threshold inside the block), 256 bytes -- 8 KiB with a copy
// per bank, or one VGPR read by ds_bpermute_b32.  This probe times the R phase of the sync kernel (the one walk through
// a lane's 128-byte sub-chunk, rows in registers, SURE certain steps a row and a loop for the rest) with
//   M0  today's table: 1024 dwords, (window & mask) | table                                   3 VALU + ds_read_b32
//   M1  64 dwords, a copy per bank (8 KiB), threshold by v_cmp_sdwa + v_cndmask_sdwa          6 VALU + ds_read_b32
//   M2  64 dwords in ONE VGPR across the wave, read by ds_bpermute_b32                        6 VALU + ds_bpermute
//   M3  64 dwords once in LDS (256 B: two entries a bank)                                     7 VALU + ds_read_b32
//   M5  as M1 without the offset add (5 VALU; wrong lengths in the one block with two steps: timing only)
// on (i) a real stream of the test coder (uniform symbols, every lane entered at its true first code) and (ii) uniform
// random words (what walks on a wrong phase and an adversary's bytes look like: 24 % of the windows have no code).
// Every variant's symbol counts and exits are checked against a host walk of the same words.
// Build: make -C profiles/tools/micro   (needs tests/golden/test_coder_table.json for the coder's rows)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint16_t u16;
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct coder_row { u16 pattern; u8 bits; };
static const coder_row kCoder[256] = {
#include "build/test_coder_rows.inc"
};

extern __shared__ __attribute__((aligned(16))) u8 dyn_lds[];
__device__ __forceinline__ u32 lds_u32(u32 off) { return *(const __attribute__((address_space(3))) u32 *)(uintptr_t)off; }
__device__ __forceinline__ u32 lds_base() { return (u32)(uintptr_t)(const __attribute__((address_space(3))) void *)dyn_lds; }

constexpr u32 kLB = 10, kMaxBits = 10, kDeadLen = 48, kSubWords = 32, kLanes = 256, kSure = 3;
constexpr u32 kK = 6;                 /* index bits of the small table */
constexpr u32 kLow = kLB - kK;        /* window bits below the index: 4 */

struct walk_consts {
    u32 thr, mask, floor;
    __host__ __device__ walk_consts(u32 pos) {
        thr = 512 + (32 - kLB) - pos;
        mask = ((1u << kLB) - 1u) << pos;
        floor = thr - kMaxBits + 1;
    }
    __host__ __device__ u32 state_at(u32 k) const { return thr + 32 - k; }
    __host__ __device__ u32 offset_of(u32 s) const { return thr + 32 - (s & 0xFFFFu); }
};

/* the window's lowest bit in the shifted pair, per mode */
template <int MODE> struct pos_of { static constexpr u32 v = MODE == 0 ? 2 : (MODE == 2 || MODE == 3 ? 2 : 3); };

template <int MODE>
__device__ __forceinline__ u32 step(u32 state, u64 pair, u32 table, u32 bank4, u32 cadd, u32 treg, u32 k_ff00) {
    const u32 t = (u32)(pair >> (state & 63u));
    if (MODE == 0) {
        return state + lds_u32((t & (((1u << kLB) - 1u) << 2)) | table);
    }
    u32 u = MODE == 5 ? t : t + cadd, e, d;
    if (MODE == 1 || MODE == 5) {
        e = lds_u32((u & (((1u << kK) - 1u) << 7)) | bank4); /* window at [12:3]: index at [12:7] = the 128-byte row of the entry's copies */
    } else if (MODE == 2) {
        e = (u32)__builtin_amdgcn_ds_bpermute((int)(u >> 4), (int)treg); /* window at [11:2]: index at [11:6] -> lane * 4 */
    } else {
        e = lds_u32(((u >> 4) & (((1u << kK) - 1u) << 2)) | table);
    }
    /* byte 0 of u = (index's low bits) | low window bits | bits below the window; byte 2 of e = the same index bits |
     * threshold - 1 | ones: u.b0 > e.b2 <=> low window bits >= threshold.  Then byte 1 (upper) or byte 0 (lower). */
    asm volatile("v_cmp_gt_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_2\n\t"
                 "v_cndmask_b32_sdwa %0, %2, %2, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1"
                 : "=v"(d)
                 : "v"(u), "v"(e)
                 : "vcc");
    return state + k_ff00 + d;
}

template <int MODE, bool RANDOM, bool COUNT>
__global__ __launch_bounds__(kLanes, 8) void walk_kernel(const u8 *stream, const u8 *entries, const u32 *table_in, u32 table_words, u32 cadd,
                                                         u32 *out, int iters, u64 *clocks, u32 *trips_out) {
    const u32 lane = threadIdx.x;
    const u32 base = lds_base();
    constexpr u32 pos = pos_of<MODE>::v;
    const walk_consts wc(pos);
    if (MODE == 1 || MODE == 5) {
#pragma unroll 1
        for (u32 i = lane; i < (1u << kK) * 32u; i += kLanes) reinterpret_cast<u32 *>(dyn_lds)[i] = table_in[i >> 5];
    } else if (MODE != 2) {
#pragma unroll 1
        for (u32 i = lane; i < table_words; i += kLanes) reinterpret_cast<u32 *>(dyn_lds)[i] = table_in[i];
    }
    __syncthreads();
    const u32 treg = MODE == 2 ? table_in[lane & 63u] : 0u;
    const u64 sub = (u64)blockIdx.x * kLanes + lane;
    const u32 entry = RANDOM ? 0u : entries[sub];
    const u32 bank4 = base | ((lane & 31u) << 2);
    u32 k_ff00 = 0xFF00u, trips = 0;
    asm volatile("" : "+s"(k_ff00));
    u32 total = 0, exit_state = 0;
    u64 walked = 0;
    for (int it = 0; it < iters; ++it) {
        /* (the words are loaded again every round, as the sync kernel loads them once: a loop around the walk alone
         * makes the compiler build all 32 register pairs in front of it, 66 registers where 8 waves a SIMD have 64) */
        const u32 *src = reinterpret_cast<const u32 *>(stream + sub * 128u);
        asm volatile("" : "+v"(src));
        u32 w[kSubWords + 1];
#pragma unroll
        for (u32 q = 0; q < kSubWords / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
            w[4 * q + 0] = __builtin_bswap32(v.x);
            w[4 * q + 1] = __builtin_bswap32(v.y);
            w[4 * q + 2] = __builtin_bswap32(v.z);
            w[4 * q + 3] = __builtin_bswap32(v.w);
        }
        w[kSubWords] = __builtin_bswap32(src[kSubWords]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u64 t0 = __builtin_amdgcn_s_memtime();
        u32 state = wc.state_at(entry);
        bool dead = false;
#pragma unroll
        for (u32 r = 0; r < kSubWords; ++r) {
            const u64 pair = ((u64)w[r] << 32) | w[r + 1];
#pragma unroll
            for (u32 i = 0; i < kSure; ++i) {
                state = step<MODE>(state, pair, base, bank4, cadd, treg, k_ff00);
            }
            trips += kSure;
            if (COUNT) { /* (the wave's trips, counted in a loop the whole wave leaves together: the untimed launch) */
                for (;;) {
                    const bool more = (state & 0xFFFFu) > wc.thr;
                    if (!__any(more)) break;
                    if (more) state = step<MODE>(state, pair, base, bank4, cadd, treg, k_ff00);
                    ++trips;
                }
            } else {
                while ((state & 0xFFFFu) > wc.thr) {
                    state = step<MODE>(state, pair, base, bank4, cadd, treg, k_ff00);
                }
            }
            const bool now = (state & 0xFFFFu) < wc.floor;
            dead = dead || now;
            state = RANDOM && now ? (state & 0xFFFF0000u) | wc.state_at(0) : state + 32u; /* a walk that died: back on a row start, the count kept */
        }
        total += state >> 16;
        exit_state = dead ? 99u : wc.offset_of(state);
        asm volatile("" : "+v"(total));
        walked += __builtin_amdgcn_s_memtime() - t0;
    }
    out[sub] = (total << 8) | exit_state;
    if ((lane & 63u) == 0) {
        clocks[sub >> 6] = walked;
        trips_out[sub >> 6] = trips;
    }
}

// ------------------------------------------------------------------ host
static std::vector<u32> g_len(1024); /* 0 = no code */

static void build_len() {
    for (const coder_row &r : kCoder) {
        for (u32 w = (u32)r.pattern << (kLB - r.bits); w < ((u32)r.pattern + 1u) << (kLB - r.bits); ++w) g_len[w] = r.bits;
    }
}

/* the small table: for offset c, block b = windows w with (w + c) >> kLow == b (mod 2^kK), low = (w + c) & (2^kLow - 1).
 * Returns false when a block has more than one step. */
static bool build_small(u32 c, u32 cmp_pos, u32 cmp_idx_bits, std::vector<u32> &tab) {
    const u32 nb = 1u << kK, bl = 1u << kLow;
    tab.assign(nb, 0);
    for (u32 b = 0; b < nb; ++b) {
        u32 lens[1u << kLow];
        for (u32 low = 0; low < bl; ++low) {
            const u32 w = ((b << kLow) + low - c) & ((1u << kLB) - 1u);
            lens[low] = g_len[w] ? g_len[w] : kDeadLen;
        }
        u32 thr = bl, steps = 0;
        for (u32 low = 1; low < bl; ++low) {
            if (lens[low] != lens[low - 1]) {
                thr = low;
                ++steps;
            }
        }
        if (steps > 1) return false;
        const u32 a = 0x100u - lens[0], bb = 0x100u - lens[bl - 1];
        /* compare byte: index bits that share byte 0 with the low window bits | (thr - 1) << cmp_pos | ones below */
        const u32 idx_part = (b & ((1u << cmp_idx_bits) - 1u)) << (cmp_pos + kLow);
        const u32 cmp = idx_part | ((thr - 1u) << cmp_pos) | ((1u << cmp_pos) - 1u);
        tab[b] = (a & 0xFFu) | ((bb & 0xFFu) << 8) | ((cmp & 0xFFu) << 16);
    }
    return true;
}

struct result { double ms, med_cycles, trips; };

template <typename F, typename G>
static result timed(F launch, G count_launch, int n_waves, u64 *d_clocks, u32 *d_trips) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    count_launch();
    CK(hipDeviceSynchronize());
    std::vector<u32> tr(n_waves);
    CK(hipMemcpy(tr.data(), d_trips, n_waves * sizeof(u32), hipMemcpyDeviceToHost));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<u64> ck(n_waves);
    CK(hipMemcpy(ck.data(), d_clocks, n_waves * sizeof(u64), hipMemcpyDeviceToHost));
    std::sort(ck.begin(), ck.end());
    double tsum = 0;
    for (u32 t : tr) tsum += t;
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return {best, (double)ck[n_waves / 2], tsum / n_waves};
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"probe\":\"device\",\"cus\":%d,\"clock_khz\":%d}\n", cus, prop.clockRate);
    build_len();
    const u32 n_chunks = argc > 1 ? (u32)atoi(argv[1]) : (u32)cus * 8u; /* one resident round by default */
    const int iters = argc > 2 ? atoi(argv[2]) : 8;
    const u64 n_sub = (u64)n_chunks * kLanes, bytes = n_sub * 128u + 256u;

    /* (i) a real stream: uniform symbols through the test coder, MSB first; the first code that starts in every sub-chunk */
    std::vector<u8> real(bytes, 0), entries(n_sub + 1, 0);
    std::vector<u32> true_count(n_sub + 1, 0);
    {
        u64 x = 88172645463325252ull, bit = 0;
        u64 next_sub = 0;
        while (bit < (n_sub * 128u + 200u) * 8u) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const coder_row &r = kCoder[(x >> 24) & 0xFF];
            while (next_sub <= n_sub && bit >= next_sub * 1024u) {
                entries[next_sub] = (u8)(bit - next_sub * 1024u);
                ++next_sub;
            }
            if (bit / 1024u < n_sub) true_count[bit / 1024u]++;
            for (int b = r.bits - 1; b >= 0; --b, ++bit) {
                if ((r.pattern >> b) & 1u) real[bit >> 3] |= (u8)(0x80u >> (bit & 7));
            }
        }
    }
    /* (ii) uniform random words */
    std::vector<u8> rnd(bytes);
    {
        u64 x = 0x9E3779B97F4A7C15ull;
        for (u64 i = 0; i < bytes; i += 8) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            memcpy(&rnd[i], &x, std::min<u64>(8, bytes - i));
        }
    }
    /* host walk of the random words with the full table: what every variant has to count */
    const walk_consts hw(2);
    auto host_walk = [&](const std::vector<u8> &s, u64 sub, u32 entry, bool random, u32 &count, u32 &exit_state) {
        u32 state = hw.state_at(entry);
        bool dead = false;
        auto word = [&](u32 r) {
            const u8 *p = &s[sub * 128u + 4u * r];
            return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | p[3];
        };
        for (u32 r = 0; r < kSubWords; ++r) {
            const u64 pair = ((u64)word(r) << 32) | word(r + 1);
            auto one = [&] {
                const u32 w = (u32)(pair >> (state & 63u)) >> 2 & 0x3FFu;
                state += 0x10000u - (g_len[w] ? g_len[w] : kDeadLen);
            };
            for (u32 i = 0; i < kSure; ++i) one();
            while ((state & 0xFFFFu) > hw.thr) one();
            const bool now = (state & 0xFFFFu) < hw.floor;
            dead = dead || now;
            state = random && now ? (state & 0xFFFF0000u) | hw.state_at(0) : state + 32u;
        }
        count = state >> 16;
        exit_state = dead ? 99u : hw.offset_of(state);
    };

    u8 *d_real, *d_rnd, *d_entries;
    u32 *d_out, *d_tab, *d_trips;
    u64 *d_clocks;
    CK(hipMalloc(&d_real, bytes));
    CK(hipMalloc(&d_rnd, bytes));
    CK(hipMalloc(&d_entries, n_sub + 1));
    CK(hipMalloc(&d_out, n_sub * sizeof(u32)));
    CK(hipMalloc(&d_tab, 4096));
    CK(hipMalloc(&d_trips, n_sub / 64 * sizeof(u32)));
    CK(hipMalloc(&d_clocks, n_sub / 64 * sizeof(u64)));
    CK(hipMemcpy(d_real, real.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rnd, rnd.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_entries, entries.data(), n_sub + 1, hipMemcpyHostToDevice));

    /* tables */
    std::vector<u32> full(1024);
    for (u32 w = 0; w < 1024; ++w) full[w] = 0x10000u - (g_len[w] ? g_len[w] : kDeadLen);
    u32 c_found = ~0u;
    std::vector<u32> small3, small2;
    for (u32 c = 0; c < (1u << kLow); ++c) {
        if (build_small(c, 3, 1, small3)) { c_found = c; break; }
    }
    if (c_found == ~0u) { printf("no offset gives one step a block\n"); return 1; }
    build_small(c_found, 2, 2, small2);
    printf("{\"probe\":\"small_table\",\"index_bits\":%u,\"offset\":%u}\n", kK, c_found);

    const int n_waves = (int)(n_sub / 64);
    auto run = [&](const char *name, int mode, bool random, auto kernel, auto count_kernel, const std::vector<u32> &tab, u32 cadd, u32 lds, bool check) {
        CK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(u32), hipMemcpyHostToDevice));
        const u8 *s = random ? d_rnd : d_real;
        auto r = timed([&] { hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(kLanes), lds, 0, s, d_entries, d_tab, (u32)tab.size(), cadd, d_out, iters, d_clocks, d_trips); },
                       [&] { hipLaunchKernelGGL(count_kernel, dim3(n_chunks), dim3(kLanes), lds, 0, s, d_entries, d_tab, (u32)tab.size(), cadd, d_out, iters, d_clocks, d_trips); },
                       n_waves, d_clocks, d_trips);
        std::vector<u32> out(n_sub);
        CK(hipMemcpy(out.data(), d_out, n_sub * sizeof(u32), hipMemcpyDeviceToHost));
        u64 wrong = 0;
        for (u64 sub = 0; sub < n_sub; sub += 37) {
            u32 cnt, ex;
            host_walk(random ? rnd : real, sub, random ? 0u : entries[sub], random, cnt, ex);
            if ((out[sub] >> 8) != cnt * (u32)iters || (out[sub] & 0xFFu) != ex) ++wrong;
            if (!random && (cnt != true_count[sub] || ex != entries[sub + 1])) ++wrong;
        }
        /* 8 workgroups of 4 waves a CU: 8 waves a SIMD when the grid fills the chip */
        printf("{\"probe\":\"walk_r05\",\"table\":\"%s\",\"mode\":%d,\"input\":\"%s\",\"chunks\":%u,\"iters\":%d,\"ms\":%.4f,\"wave_trips\":%.0f,"
               "\"wave_cycles_per_trip\":%.1f,\"cycles_per_trip_per_simd_at_8_waves\":%.2f,\"ns_per_trip_per_simd\":%.3f,\"wrong\":%llu%s}\n",
               name, mode, random ? "uniform words" : "test coder stream", n_chunks, iters, r.ms, r.trips, r.med_cycles / r.trips, r.med_cycles / r.trips / 8.0,
               r.ms * 1e6 / (r.trips * (n_waves / (4.0 * cus))), (unsigned long long)wrong, check ? "" : ",\"timing_only\":true");
    };
    for (int random = 0; random < 2; ++random) {
        if (random) {
            run("1024 dwords, one copy (today)", 0, true, walk_kernel<0, true, false>, walk_kernel<0, true, true>, full, 0, 4096, true);
            run("64 dwords, a copy per bank, offset add", 1, true, walk_kernel<1, true, false>, walk_kernel<1, true, true>, small3, c_found << 3, 8192, true);
            run("64 dwords in a VGPR, ds_bpermute", 2, true, walk_kernel<2, true, false>, walk_kernel<2, true, true>, small2, c_found << 2, 0, true);
            run("64 dwords, one copy", 3, true, walk_kernel<3, true, false>, walk_kernel<3, true, true>, small2, c_found << 2, 256, true);
            run("64 dwords, a copy per bank, no offset add", 5, true, walk_kernel<5, true, false>, walk_kernel<5, true, true>, small3, 0, 8192, false);
        } else {
            run("1024 dwords, one copy (today)", 0, false, walk_kernel<0, false, false>, walk_kernel<0, false, true>, full, 0, 4096, true);
            run("64 dwords, a copy per bank, offset add", 1, false, walk_kernel<1, false, false>, walk_kernel<1, false, true>, small3, c_found << 3, 8192, true);
            run("64 dwords in a VGPR, ds_bpermute", 2, false, walk_kernel<2, false, false>, walk_kernel<2, false, true>, small2, c_found << 2, 0, true);
            run("64 dwords, one copy", 3, false, walk_kernel<3, false, false>, walk_kernel<3, false, true>, small2, c_found << 2, 256, true);
            run("64 dwords, a copy per bank, no offset add", 5, false, walk_kernel<5, false, false>, walk_kernel<5, false, true>, small3, 0, 8192, false);
        }
    }
    return 0;
}
