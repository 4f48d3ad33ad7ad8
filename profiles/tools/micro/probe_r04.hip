// Round-4 hardware probes (gfx950): what bounds a table-driven code walk.
//   (i)   vector-instruction issue: independent / dependent v_add_u32, v_lshrrev_b64, v_alignbit_b32 at 1, 2, 4, 8
//         waves per SIMD;
//   (ii)  the walk's dependent step (shift -> address -> ds_read -> add) with the table read at random addresses
//         (one 4 KiB table, as the product kernels have it) against conflict-free ones (a copy per LDS bank: dword
//         entries on an 8-bit window, byte entries and nibble entries on the 10-bit window), 64-bit shift against
//         v_alignbit_b32, at 4 and 8 waves per SIMD;
//   (iii) the same with two independent chains per lane;
//   (iv)  byte stores to an LDS stage as dec_emit_fast does them, against dword stores;
//   (v)   scattered global stores: every lane a run of ~27 bytes (what a lane-quarter of the emit kernel produces).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 probe_r04.hip -o build/probe_r04   (make -C profiles/tools/micro)
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
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

extern __shared__ __attribute__((aligned(16))) u8 dyn_lds[];

__device__ __forceinline__ u32 lds_u32(u32 off) {
    return *(const __attribute__((address_space(3))) u32 *)(uintptr_t)off;
}
__device__ __forceinline__ u32 lds_u8(u32 off) {
    return *(const __attribute__((address_space(3))) u8 *)(uintptr_t)off;
}
__device__ __forceinline__ u32 lds_base() {
    return (u32)(uintptr_t)(const __attribute__((address_space(3))) void *)dyn_lds;
}

// ------------------------------------------------------------------ (i) vector-instruction issue
// OP 0 v_add_u32, 1 v_lshrrev_b64, 2 v_alignbit_b32, 3 v_and_or_b32, 4 v_bfe_u32, 5 v_perm_b32, 6 v_lshl_or_b32
template <int OP, int CHAINS> // CHAINS independent accumulators (1 = a dependent chain)
__global__ __launch_bounds__(1024, 8) void valu_kernel(u32 *out, int iters, u64 *clocks) {
    u32 a[8];
    u64 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 3 + i;
        q[i] = ((u64)threadIdx.x << 33) | (u32)(i * 77 + 1);
    }
    const u32 k = (threadIdx.x & 7) + 1, m = 0x00fff0f0u | threadIdx.x;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = j % CHAINS;
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(k));
            if (OP == 1) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[c]) : "v"(k));
            if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
            if (OP == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
            if (OP == 4) asm volatile("v_bfe_u32 %0, %0, %1, 9" : "+v"(a[c]) : "v"(k));
            if (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(k));
            if (OP == 6) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[c]) : "v"(k));
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (u32)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ------------------------------------------------------------------ (ii), (iii) the walk
// Every lane has ROWS words of "stream" in registers and walks them again and again: SURE steps a row, then on to
// the next row.  A step is what a count-only walk does: window -> table entry -> state += entry.
//   MODE 0  one 1024-entry dword table (4 KiB, random banks); window by v_lshrrev_b64       (dec_sync_lean today)
//   MODE 1  the same table; window by v_alignbit_b32 (stream kept LSB-first, words shifted by two at load)
//   MODE 3  1024-entry byte table, a copy per bank (32 KiB): ds_read_u8 at (w >> 2) << 7 | bank << 2 | (w & 3)
//   MODE 4  1024-entry nibble table, a copy per bank (16 KiB): ds_read_b32 at (w >> 3) << 7 | bank << 2, v_bfe
//   MODE 5  256-entry dword table, a copy per bank (32 KiB): (window8 << 7) | bank << 2 -- what a conflict-free dword
//           read costs at best (one address instruction; an 8-bit window is not enough for the product's coders)
constexpr int kRows = 8;
constexpr int kSure = 3;

__device__ const u32 *g_table_ptr;
template <int MODE>
__device__ __forceinline__ u32 walk_step(u32 state, u32 hi, u32 lo, u32 table, u32 bank4, const u32 *gtab = nullptr) {
    if (MODE == 8) { /* a 1 KiB table of bytes in global memory: 8 cache lines instead of 32 */
        const u64 pair = ((u64)hi << 32) | lo;
        return state + 0xFF00u + *(reinterpret_cast<const u8 *>(gtab) + ((u32)(pair >> (state & 63u)) & 0x3FFu));
    }
    if (MODE == 9) { /* a 2 KiB table of 16-bit entries in global memory */
        const u64 pair = ((u64)hi << 32) | lo;
        return state + 0xFF00u + *reinterpret_cast<const unsigned short *>(reinterpret_cast<const u8 *>(gtab) + ((u32)(pair >> (state & 63u)) & 0x7FEu));
    }
    if (MODE == 6) { /* the 4 KiB table in global memory (L1-resident): the vector memory pipe instead of the LDS */
        const u64 pair = ((u64)hi << 32) | lo;
        return state + *reinterpret_cast<const u32 *>(reinterpret_cast<const u8 *>(gtab) + ((u32)(pair >> (state & 63u)) & 0xFFCu));
    }
    if (MODE == 0) {
        const u64 pair = ((u64)hi << 32) | lo;
        return state + lds_u32(((u32)(pair >> (state & 63u)) & 0xFFCu) | table);
    } else if (MODE == 1) {
        const u32 t = __builtin_amdgcn_alignbit(hi, lo, state);
        return state + lds_u32((t & 0xFFCu) | table);
    } else if (MODE == 2 || MODE == 5) {
        const u32 t = __builtin_amdgcn_alignbit(hi, lo, state);
        return state + lds_u32((t & (0xFFu << 7)) | bank4); /* bank4 = table | (lane & 31) << 2 */
    } else if (MODE == 3) {
        const u32 t = __builtin_amdgcn_alignbit(hi, lo, state); /* window at t[14:5] */
        const u32 low = (t >> 5) & 3u;                          /* v_bfe_u32 */
        const u32 a = (t & (0xFFu << 7)) | bank4;               /* v_and_or_b32 */
        return state - lds_u8(a | low);                         /* v_or, ds_read_u8, v_sub */
    } else {
        const u32 t = __builtin_amdgcn_alignbit(hi, lo, state); /* window at t[11:2] */
        const u32 nib = t & 0x1Cu;
        const u32 a = ((t & 0xFE0u) << 2) | bank4;
        const u32 wd = lds_u32(a);
        return state - __builtin_amdgcn_ubfe(wd, nib, 4);
    }
}

template <int MODE, int CHAINS>
__global__ __launch_bounds__(1024, 8) void walk_kernel(const u32 *words, u32 *out, int iters, u32 lds_bytes, u64 *clocks, const u32 *gtab = nullptr, int global_every = 0) {
    const u32 lane = threadIdx.x;
    const u32 base = lds_base();
    /* fill the whole dynamic LDS with entries: a "length" of 5..10 per byte / nibble / dword, whatever the mode reads */
    for (u32 i = lane; i < lds_bytes / 4; i += blockDim.x) {
        u32 h = i * 2654435761u;
        h ^= h >> 15;
        u32 v;
        if (MODE == 3) {
            v = 0;
            for (int b = 0; b < 4; ++b) v |= (5u + ((h >> (5 * b)) % 6u)) << (8 * b);
        } else if (MODE == 4) {
            v = 0;
            for (int b = 0; b < 8; ++b) v |= (5u + ((h >> (3 * b)) % 6u)) << (4 * b);
        } else {
            v = 0x10000u - (5u + h % 6u);
        }
        reinterpret_cast<u32 *>(dyn_lds)[i] = v;
    }
    u32 w[CHAINS][kRows + 1];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
#pragma unroll
        for (int r = 0; r <= kRows; ++r) {
            w[c][r] = words[((blockIdx.x * blockDim.x + lane) * CHAINS + c) * (kRows + 1) + r];
        }
    }
    __syncthreads();
    const u32 bank4 = base | ((lane & 31u) << 2);
    u32 st[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) st[c] = 37u + lane + c;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    /* MODE 7: every `global_every`-th wave of the workgroup looks up through global memory, the others through the LDS */
    const bool by_memory = MODE == 6 || MODE >= 8 || (MODE == 7 && global_every && (threadIdx.x / 64) % global_every == 0);
    if (MODE >= 6 && by_memory) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
#pragma unroll
                for (int s = 0; s < kSure; ++s) {
#pragma unroll
                    for (int c = 0; c < CHAINS; ++c) {
                        st[c] = walk_step<(MODE >= 8 ? MODE : 6)>(st[c], w[c][r], w[c][r + 1], base, bank4, gtab);
                    }
                }
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) st[c] += 32u;
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
#pragma unroll
                for (int s = 0; s < kSure; ++s) {
#pragma unroll
                    for (int c = 0; c < CHAINS; ++c) {
                        st[c] = walk_step<(MODE >= 6 ? 0 : MODE)>(st[c], w[c][r], w[c][r + 1], base, bank4);
                    }
                }
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) st[c] += 32u;
            }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += st[c];
    out[blockIdx.x * blockDim.x + lane] = s;
    if ((lane & 63) == 0) clocks[(blockIdx.x * blockDim.x + lane) >> 6] = t1 - t0;
}

struct __attribute__((packed, aligned(1))) u32u { u32 v; };
struct __attribute__((packed, aligned(1))) u16u { unsigned short v; };
// ------------------------------------------------------------------ (iv) LDS stage stores
// MODE 0: a byte per step at slot + count (lanes' slots 108 bytes apart: random banks), as dec_emit_fast
// MODE 1: a dword every fourth step at an aligned slot (stride 33 dwords: a bank of its own per lane)
// MODE 2: a dword EVERY step at the aligned slot (rewriting the dword until it is full)
template <int MODE>
__global__ __launch_bounds__(1024, 8) void stage_kernel(u32 *out, int iters, u64 *clocks) {
    const u32 lane = threadIdx.x;
    const u32 base = lds_base();
    const u32 slot = base + (MODE == 0 || MODE >= 3 ? lane * 27u : lane * 33u * 4u % (27u * 1024u));
    u32 acc = lane;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (u32 k = 0; k < 24; ++k) {
            acc = acc * 5u + 1u;
            if (MODE == 0) {
                *(__attribute__((address_space(3))) u8 *)(uintptr_t)(slot + k) = (u8)acc;
            } else if (MODE == 1) {
                if ((k & 3) == 3) *(__attribute__((address_space(3))) u32 *)(uintptr_t)(slot + (k & ~3u)) = acc;
            } else if (MODE == 3) {
                /* a row's three certain symbols as ONE unaligned dword store (the fourth byte is overwritten by what follows) */
                if (k % 3 == 2) ((__attribute__((address_space(3))) u32u *)(uintptr_t)(slot + k - 2))->v = acc;
            } else if (MODE == 4) {
                /* two symbols as one unaligned 16-bit store, the third as a byte */
                if (k % 3 == 1) ((__attribute__((address_space(3))) u16u *)(uintptr_t)(slot + k - 1))->v = (unsigned short)acc;
                if (k % 3 == 2) *(__attribute__((address_space(3))) u8 *)(uintptr_t)(slot + k) = (u8)acc;
            } else {
                *(__attribute__((address_space(3))) u32 *)(uintptr_t)(slot + (k & ~3u)) = acc;
            }
        }
        asm volatile("" ::: "memory");
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + lane] = acc + lds_u32(slot & ~3u);
    if ((lane & 63) == 0) clocks[(blockIdx.x * blockDim.x + lane) >> 6] = t1 - t0;
}

// ------------------------------------------------------------------ (v) scattered global stores
// every lane owns `run` bytes of the output (lanes back to back); MODE 0: unaligned dword stores, MODE 1: byte stores,
// MODE 2: unaligned 8-byte stores 4 bytes apart (overlapping: a row's symbols with slack), MODE 3: 16-byte unaligned
struct __attribute__((packed, aligned(1))) u64u { u64 v; };
struct __attribute__((packed, aligned(1))) u128u { u32 x, y, z, w; };
template <int MODE>
__global__ void scatter_kernel(u8 *out, u32 run, u64 total) {
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 g = gid; g * run + run + 16 <= total; g += stride) {
        u8 *p = out + g * run;
        const u32 v = (u32)g * 2654435761u;
        if (MODE == 0) {
            for (u32 k = 0; k + 4 <= run; k += 4) reinterpret_cast<u32u *>(p + k)->v = v + k;
        } else if (MODE == 1) {
            for (u32 k = 0; k < run; ++k) p[k] = (u8)(v + k);
        } else if (MODE == 2) {
            for (u32 k = 0; k + 8 <= run; k += 4) reinterpret_cast<u64u *>(p + k)->v = ((u64)v << 32) | (v + k);
        } else {
            for (u32 k = 0; k + 16 <= run; k += 16) {
                u128u x = {v, v + k, v ^ k, k};
                *reinterpret_cast<u128u *>(p + k) = x;
            }
        }
    }
}

// ------------------------------------------------------------------ host
static int g_cus = 256;
static u32 *d_out;
static u64 *d_clocks;
static u32 *d_words;

struct result {
    double ms, med_cycles;
};

template <typename F>
static result timed(F launch, int n_waves) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
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
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return {best, (double)ck[n_waves / 2]};
}

// block of `threads` threads, `blocks_per_cu` resident per CU (forced by the dynamic LDS size), one resident round
static u32 lds_for(int blocks_per_cu, u32 need) {
    u32 cap = (160u * 1024u) / blocks_per_cu;
    cap &= ~1023u;
    if (cap < need) {
        printf("  (LDS: %u needed, %u per block at %d blocks per CU)\n", need, cap, blocks_per_cu);
        exit(1);
    }
    /* more than half of what one block fewer could have: exactly blocks_per_cu fit */
    return cap;
}

template <int OP, int CHAINS>
static void run_valu(const char *name) {
    const int iters = 1024;
    for (int wps : {1, 2, 4, 8}) {
        const int threads = wps >= 4 ? 1024 : 256 * wps, bpc = wps >= 4 ? wps / 4 : 1;
        const int grid = g_cus * bpc, n_waves = grid * threads / 64;
        auto r = timed([&] { hipLaunchKernelGGL((valu_kernel<OP, CHAINS>), dim3(grid), dim3(threads), 0, 0, d_out, iters, d_clocks); }, n_waves);
        const double instr = 64.0 * iters; /* per wave */
        printf("{\"probe\":\"valu\",\"op\":\"%s\",\"chains\":%d,\"waves_per_simd\":%d,\"ms\":%.4f,\"cycles_per_instr_per_simd\":%.2f,\"wave_cycles_per_instr\":%.2f}\n",
               name, CHAINS, wps, r.ms, r.med_cycles / (instr * wps), r.med_cycles / instr);
    }
}

static u32 *d_gtab;
template <int MODE, int CHAINS>
static void run_walk(const char *name, u32 table_bytes, int global_every = 0) {
    const int iters = 400;
    struct shape { int threads, bpc; };
    for (shape s : {shape{256, 4}, shape{256, 8}, shape{1024, 1}, shape{1024, 2}}) {
        const int wps = s.threads / 256 * s.bpc;
        if (table_bytes * s.bpc > 160u * 1024u) continue; /* (32 KiB x 8 does not fit) */
        const u32 lds = lds_for(s.bpc, table_bytes);
        const int grid = g_cus * s.bpc, n_waves = grid * s.threads / 64;
        CK(hipFuncSetAttribute((const void *)walk_kernel<MODE, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto r = timed([&] { hipLaunchKernelGGL((walk_kernel<MODE, CHAINS>), dim3(grid), dim3(s.threads), lds, 0, d_words, d_out, iters, table_bytes, d_clocks, d_gtab, global_every); }, n_waves);
        const double steps = (double)iters * kRows * kSure * CHAINS; /* per wave */
        printf("{\"probe\":\"walk\",\"table\":\"%s\",\"chains\":%d,\"block\":%d,\"blocks_per_cu\":%d,\"waves_per_simd\":%d,\"ms\":%.4f,"
               "\"cycles_per_step_per_simd\":%.2f,\"wave_cycles_per_step\":%.1f,\"ns_per_step_per_simd\":%.3f}\n",
               name, CHAINS, s.threads, s.bpc, wps, r.ms, r.med_cycles / (steps * wps), r.med_cycles / steps, r.ms * 1e6 / (steps * wps));
    }
}

template <int MODE>
static void run_stage(const char *name) {
    const int iters = 2000;
    struct shape { int threads, bpc; };
    for (shape s : {shape{1024, 1}, shape{1024, 2}}) {
        const int wps = s.threads / 256 * s.bpc;
        const u32 lds = lds_for(s.bpc, 36 * 1024);
        const int grid = g_cus * s.bpc, n_waves = grid * s.threads / 64;
        CK(hipFuncSetAttribute((const void *)stage_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        auto r = timed([&] { hipLaunchKernelGGL((stage_kernel<MODE>), dim3(grid), dim3(s.threads), lds, 0, d_out, iters, d_clocks); }, n_waves);
        const double steps = (double)iters * 24;
        printf("{\"probe\":\"stage\",\"store\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"cycles_per_symbol_step_per_simd\":%.2f}\n", name, wps, r.ms,
               r.med_cycles / (steps * wps));
    }
}

template <int MODE>
static void run_scatter(const char *name, u8 *d_big, u64 total) {
    for (u32 run : {27u, 108u}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((scatter_kernel<MODE>), dim3(g_cus * 8), dim3(256), 0, 0, d_big, run, total);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((scatter_kernel<MODE>), dim3(g_cus * 8), dim3(256), 0, 0, d_big, run, total);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\":\"scatter\",\"store\":\"%s\",\"run_bytes\":%u,\"ms_per_GiB\":%.4f,\"TB_per_s\":%.2f}\n", name, run, ms * (double)(1ull << 30) / total,
               total / (ms * 1e-3) / 1e12);
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_khz\":%d}\n", prop.name, g_cus, prop.clockRate);
    const size_t n_threads = (size_t)g_cus * 2048;
    CK(hipMalloc(&d_out, n_threads * sizeof(u32)));
    CK(hipMalloc(&d_clocks, n_threads / 64 * sizeof(u64)));
    std::vector<u32> words(n_threads * 2 * (kRows + 1));
    u64 x = 88172645463325252ull;
    for (auto &v : words) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        v = (u32)(x >> 16);
    }
    CK(hipMalloc(&d_words, words.size() * sizeof(u32)));
    CK(hipMemcpy(d_words, words.data(), words.size() * sizeof(u32), hipMemcpyHostToDevice));
    const std::string what = argc > 1 ? argv[1] : "all";

    if (what == "all" || what == "valu") {
        run_valu<0, 8>("v_add_u32");
        run_valu<0, 1>("v_add_u32");
        run_valu<1, 8>("v_lshrrev_b64");
        run_valu<1, 1>("v_lshrrev_b64");
        run_valu<2, 8>("v_alignbit_b32");
        run_valu<2, 1>("v_alignbit_b32");
        run_valu<3, 8>("v_and_or_b32");
        run_valu<4, 8>("v_bfe_u32");
        run_valu<5, 8>("v_perm_b32");
        run_valu<6, 8>("v_lshl_or_b32");
    }
    {
        std::vector<u32> gt(1024);
        for (u32 i = 0; i < 1024; ++i) {
            u32 h = i * 2654435761u;
            h ^= h >> 15;
            gt[i] = 0x10000u - (5u + h % 6u);
        }
        CK(hipMalloc(&d_gtab, 4096));
        CK(hipMemcpy(d_gtab, gt.data(), 4096, hipMemcpyHostToDevice));
    }
    if (what == "all" || what == "mixed") {
        run_walk<0, 1>("shared dword 4K in LDS", 4096);
        run_walk<6, 1>("dword 4K in global memory (L1)", 4096);
        run_walk<8, 1>("byte 1K in global memory (L1)", 4096);
        run_walk<9, 1>("u16 2K in global memory (L1)", 4096);
        run_walk<7, 1>("LDS, every 4th wave through global memory", 4096, 4);
        run_walk<7, 1>("LDS, every 2nd wave through global memory", 4096, 2);
        run_walk<7, 1>("LDS, every 8th wave through global memory", 4096, 8);
    }
    if (what == "all" || what == "walk") {
        run_walk<0, 1>("shared dword 4K, lshr_b64", 4096);
        run_walk<1, 1>("shared dword 4K, alignbit", 4096);
        run_walk<5, 1>("dword per bank (address conflict-free)", 32768);
        run_walk<3, 1>("byte per bank 32K", 32768);
        run_walk<4, 1>("nibble per bank 16K", 16384);
        run_walk<0, 2>("shared dword 4K, lshr_b64", 4096);
        run_walk<1, 2>("shared dword 4K, alignbit", 4096);
        run_walk<5, 2>("dword per bank (address conflict-free)", 32768);
        run_walk<3, 2>("byte per bank 32K", 32768);
        run_walk<4, 2>("nibble per bank 16K", 16384);
    }
    if (what == "all" || what == "stage") {
        run_stage<0>("byte per step");
        run_stage<1>("dword per 4 steps");
        run_stage<2>("dword per step");
        run_stage<3>("unaligned dword per 3 steps");
        run_stage<4>("unaligned 16 bits + a byte per 3 steps");
    }
    if (what == "all" || what == "scatter") {
        const u64 total = 1ull << 30;
        u8 *d_big;
        CK(hipMalloc(&d_big, total + 64));
        run_scatter<0>("dword unaligned", d_big, total);
        run_scatter<1>("byte", d_big, total);
        run_scatter<2>("8 bytes every 4", d_big, total);
        run_scatter<3>("16 bytes unaligned", d_big, total);
        CK(hipFree(d_big));
    }
    return 0;
}
