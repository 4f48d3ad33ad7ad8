// Round-6 hardware probe (gfx950): what the emit stage's byte stores cost the LDS beside the table look-ups, and whether a
// row's three certain symbols could leave as an aligned 16-bit store + a byte store (which of the two comes first depends on
// the parity of the lane's stage address: two pairs of stores under exec masks) instead of three byte stores.
//
// dec_emit_fast's step is one look-up at a random place of a 4 KiB table (ds_read_b32) and one byte to the lane's place in the
// stage (ds_write_b8_d16_hi); a row's first three steps are certain for every lane.  Modes, each 8 waves a SIMD, 3 "steps" a
// trip of the loop:
//   M0  3 look-ups + 3 byte stores               (the kernel's row as it is)
//   M1  3 look-ups + {b16 + b8 | b8 + b16} by parity of the address, under exec masks
//   M2  3 look-ups, no store
//   M3  3 byte stores, no look-up
//   M4  the parity-split stores, no look-up
//   M5  3 look-ups + ONE byte store
// Prints ns a trip and CU (the kernel's time x CUs' share), and LDS-bound cycles at the clock given (default 2.0 GHz).
// Build: make -C profiles/tools/micro build/probe_r06 ; run on the GPU box: build/probe_r06 [trips=4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32;
#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
constexpr u32 kThreads = 512, kTable = 4096, kStage = 16384;

__device__ __forceinline__ u32 lds_read(u32 off) {
    u32 v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    return v;
}
__device__ __forceinline__ void st_b8(u32 at, u32 v) { asm volatile("ds_write_b8 %0, %1" ::"v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void st_b8o(u32 at, u32 v, int) { asm volatile("ds_write_b8 %0, %1 offset:1" ::"v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void st_b8o2(u32 at, u32 v) { asm volatile("ds_write_b8 %0, %1 offset:2" ::"v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void st_b16(u32 at, u32 v) { asm volatile("ds_write_b16 %0, %1" ::"v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void st_b16o1(u32 at, u32 v) { asm volatile("ds_write_b16 %0, %1 offset:1" ::"v"(at), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(kThreads, 8) void probe(u32 trips, u32 *out) {
    const u32 t = threadIdx.x;
    for (u32 i = t; i < kTable / 4; i += kThreads) {
        reinterpret_cast<u32 *>(dyn_lds)[i] = (i * 2654435761u) >> 7;
    }
    __syncthreads();
    /* a chain's place in the stage: ~13.5 bytes a chain, as the kernel's (two chains a thread would double the loop) */
    const u32 base = kTable + (t * 27u) / 2u;
    u32 x = t * 747796405u + blockIdx.x * 2891336453u + 1u, acc = 0;
    for (u32 it = 0; it < trips; ++it) {
        const u32 at = base + 3u * (it & 3u);
        u32 e0 = 0, e1 = 0, e2 = 0;
        if (MODE != 3 && MODE != 4) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            e0 = lds_read(x & 0xFFCu);
            e1 = lds_read((x >> 10) & 0xFFCu);
            e2 = lds_read(((x >> 20) ^ e1) & 0xFFCu); /* (a dependent one, as the walk's are) */
        } else {
            e0 = it; e1 = it + 1; e2 = it + 2;
        }
        if (MODE == 0 || MODE == 3) {
            st_b8(at, e0);
            st_b8o(at, e1, 0);
            st_b8o2(at, e2);
        } else if (MODE == 1 || MODE == 4) {
            const u32 lo = (e0 & 0xFFu) | (e1 << 8), hi = (e1 & 0xFFu) | (e2 << 8);
            if (at & 1u) {
                st_b8(at, e0);
                st_b16o1(at, hi);
            } else {
                st_b16(at, lo);
                st_b8o2(at, e2);
            }
        } else if (MODE == 5) {
            st_b8(at, e0 ^ e1 ^ e2);
        }
        acc += e0 + e1 + e2;
    }
    __syncthreads();
    if (acc == 0x12345u) {
        out[blockIdx.x * kThreads + t] = acc + dyn_lds[kTable + t];
    }
}

template <int MODE>
static void run(const char *what, u32 trips, u32 *d_out, int cus, double ghz) {
    const u32 grid = (u32)cus * 4u * 4u; /* four workgroups a CU resident, four rounds of them */
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(kThreads), kTable + kStage, 0, trips, d_out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    /* a CU sees grid / cus workgroups x 8 waves x trips wave-trips */
    const double wave_trips = (double)grid / cus * (kThreads / 64) * trips;
    const double ns = best * 1e6 / wave_trips;
    printf("M%d %-58s %8.3f ms  %6.2f ns a wave-trip and CU  = %5.1f cycles at %.1f GHz\n", MODE, what, best, ns, ns * ghz, ghz);
}

int main(int argc, char **argv) {
    const u32 trips = argc > 1 ? (u32)atoi(argv[1]) : 4000u;
    const double ghz = argc > 2 ? atof(argv[2]) : 2.0;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    u32 *d_out;
    CK(hipMalloc(&d_out, (size_t)p.multiProcessorCount * 16 * kThreads * 4));
    printf("%s, %d CUs, %u trips a wave; a trip = three steps\n", p.name, p.multiProcessorCount, trips);
    run<2>("3 look-ups", trips, d_out, p.multiProcessorCount, ghz);
    run<3>("3 byte stores", trips, d_out, p.multiProcessorCount, ghz);
    run<4>("b16 + b8 by parity (two pairs under exec masks)", trips, d_out, p.multiProcessorCount, ghz);
    run<0>("3 look-ups + 3 byte stores (the kernel's row)", trips, d_out, p.multiProcessorCount, ghz);
    run<1>("3 look-ups + b16 + b8 by parity", trips, d_out, p.multiProcessorCount, ghz);
    run<5>("3 look-ups + 1 byte store", trips, d_out, p.multiProcessorCount, ghz);
    return 0;
}
