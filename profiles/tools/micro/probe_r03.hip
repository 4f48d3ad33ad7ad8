// Round-3 hardware probes (gfx950): unaligned LDS access, 64-bit shift rate, gated-empty launch cost,
// unaligned 16-byte global stores.  Build: hipcc -O3 --offload-arch=gfx950 probe_r03.hip -o probe_r03
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct __attribute__((packed, aligned(1))) u128u { u32 x, y, z, w; };
struct __attribute__((packed, aligned(1))) u64u { u64 v; };
struct __attribute__((packed, aligned(1))) u32u { u32 v; };

__global__ void lds_unaligned(u8 *out, u32 off, int width) {
    __shared__ __attribute__((aligned(16))) u8 buf[64 * 48 + 64];
    const u32 l = threadIdx.x;
    for (u32 i = l; i < sizeof(buf); i += 64) buf[i] = 0xEE;
    __syncthreads();
    u8 *p = buf + l * 48 + off;
    if (width == 16) { u128u v = {0x03020100u + l, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu}; *reinterpret_cast<u128u *>(p) = v; }
    if (width == 8) { u64u v = {0x0706050403020100ull + l}; *reinterpret_cast<u64u *>(p) = v; }
    if (width == 4) { u32u v = {0x03020100u + l}; *reinterpret_cast<u32u *>(p) = v; }
    __syncthreads();
    for (u32 i = l; i < sizeof(buf); i += 64) out[i] = buf[i];
    __syncthreads();
    // unaligned read back through the wide type
    if (width == 16) { u128u v = *reinterpret_cast<u128u *>(p); reinterpret_cast<u32 *>(out + 4096)[l * 4 + 0] = v.x; reinterpret_cast<u32 *>(out + 4096)[l * 4 + 3] = v.w; }
}

// timing: many unaligned vs aligned LDS 16-byte writes+reads
__global__ void lds_rate(u32 *out, u32 off, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) u8 dyn[];
    const u32 l = threadIdx.x;
    u8 *p = dyn + l * 144 + off;
    u128u v = {l, l + 1, l + 2, l + 3};
    u32 acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<u128u *>(p + 16 * j) = v;
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { u128u r = *reinterpret_cast<u128u *>(p + 16 * j); acc += r.x + r.w; }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) *reinterpret_cast<u32u *>(p + 4 * j) = u32u{v.x + (u32)j};
        }
        v.x += acc;
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + l] = acc + v.x;
}

template <int MODE>
__global__ void valu_rate(u32 *out, int iters) {
    u32 a = threadIdx.x, b = blockIdx.x * 7 + 1, c = a ^ 0x55, d = a + 3;
    u64 pa = ((u64)a << 32) | b, pb = ((u64)c << 32) | d;
    u32 s0 = a & 31, s1 = c & 31, s2 = b & 31, s3 = d & 31;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { // 64-bit shift, 4 independent chains
                s0 = (u32)(pa >> (s0 & 63)) & 63; s1 = (u32)(pb >> (s1 & 63)) & 63; s2 = (u32)(pa >> (s2 & 63)) & 63; s3 = (u32)(pb >> (s3 & 63)) & 63;
            } else if (MODE == 1) { // alignbit
                s0 = __builtin_amdgcn_alignbit(a, b, s0) & 31; s1 = __builtin_amdgcn_alignbit(c, d, s1) & 31; s2 = __builtin_amdgcn_alignbit(a, b, s2) & 31; s3 = __builtin_amdgcn_alignbit(c, d, s3) & 31;
            } else if (MODE == 2) { // plain adds+and (2 ops as the other modes)
                s0 = (s0 + a) & 31; s1 = (s1 + c) & 31; s2 = (s2 + b) & 31; s3 = (s3 + d) & 31;
            } else { // 32-bit shift + and
                s0 = (a >> s0) & 31; s1 = (c >> s1) & 31; s2 = (b >> s2) & 31; s3 = (d >> s3) & 31;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
}

__global__ void gated_empty(const u32 *gate, u32 *out) {
    if (*gate == 0) return;
    out[blockIdx.x] = 1;
}

__global__ void store16(u8 *dst, u32 shift, u64 bytes_per_block) {
    u8 *base = dst + (u64)blockIdx.x * bytes_per_block + shift;
    u128u v = {threadIdx.x, 1, 2, 3};
    for (u64 o = (u64)threadIdx.x * 16; o + 16 <= bytes_per_block - 16; o += 256 * 16) *reinterpret_cast<u128u *>(base + o) = v;
}

int main() {
    u8 *d; CK(hipMalloc(&d, 1 << 20)); std::vector<u8> h(8192);
    for (int width : {16, 8, 4}) for (u32 off : {0u, 1u, 2u, 3u, 5u, 13u}) {
        CK(hipMemset(d, 0, 8192));
        hipLaunchKernelGGL(lds_unaligned, dim3(1), dim3(64), 0, 0, d, off, width);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("lds_unaligned width %d off %u: ERROR %s\n", width, off, hipGetErrorString(e)); return 1; }
        CK(hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost));
        int bad = 0;
        for (u32 l = 0; l < 64; ++l) for (u32 i = 0; i < 48; ++i) {
            u8 want = 0xEE; if (i >= off && i < off + (u32)width) { u32 k = i - off; want = (u8)(k == 0 ? (0 + l) & 0xFF : k); if (k == 0) want = (u8)l; }
            // byte 0 carries +l (may carry into byte 1 for l up to 63: no carry since base byte0 = 0)
            if (h[l * 48 + i] != want) ++bad;
        }
        u32 rb = 0; if (width == 16) memcpy(&rb, h.data() + 4096 + 5 * 16, 4);
        printf("lds_unaligned width %2d off %2u: %s (bad bytes %d) readback lane5.x=%08x\n", width, off, bad ? "MISMATCH" : "ok", bad, rb);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32 *o32; CK(hipMalloc(&o32, 64 << 20));
    for (int mode = 0; mode < 3; ++mode) for (u32 off : {0u, 4u, 1u}) {
        const int iters = 200; float ms;
        hipLaunchKernelGGL(lds_rate, dim3(1024), dim3(256), 256 * 144 + 64, 0, o32, off, 10, mode);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(lds_rate, dim3(1024), dim3(256), 256 * 144 + 64, 0, o32, off, iters, mode); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        double ops = 1024.0 * 4 * iters * (mode == 2 ? 32 : 8); // wave-instructions
        printf("lds_rate mode %d (0=w128,1=r128,2=w32) off %u: %.3f ms, %.2f ns per wave-instr per CU-slot (%.1f cycles@2.4GHz per CU)\n", mode, off, ms, ms * 1e6 / (ops / 256), ms * 1e6 / (ops / 256) * 2.4);
    }
    {
        const int iters = 2000; float ms[4];
        for (int m = 0; m < 4; ++m) {
            auto k = m == 0 ? valu_rate<0> : m == 1 ? valu_rate<1> : m == 2 ? valu_rate<2> : valu_rate<3>;
            hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, o32, 10);
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, o32, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[m], e0, e1));
            // per SIMD: 8 waves; pairs per wave = iters*16*4
            double pairs_per_simd = 8.0 * iters * 16 * 4;
            printf("valu_rate mode %d (0=lshr64+and,1=alignbit+and,2=add+and,3=lshr32+and): %.3f ms, %.2f cycles@2.4GHz per pair per SIMD\n", m, ms[m], ms[m] * 1e-3 * 2.4e9 / pairs_per_simd);
        }
    }
    {
        u32 *gate; CK(hipMalloc(&gate, 4)); CK(hipMemset(gate, 0, 4)); float ms;
        for (u32 grid : {2048u, 39322u, 65536u}) {
            hipLaunchKernelGGL(gated_empty, dim3(grid), dim3(256), 0, 0, gate, o32);
            CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gated_empty, dim3(grid), dim3(256), 0, 0, gate, o32); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("gated_empty grid %u x256: %.2f us per launch (back to back)\n", grid, ms * 1000 / 20);
            CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gated_empty, dim3(grid), dim3(512), 0, 0, gate, o32); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("gated_empty grid %u x512: %.2f us per launch (back to back)\n", grid, ms * 1000 / 20);
        }
    }
    {
        u8 *big; const u64 total = 1ull << 30; CK(hipMalloc(&big, total + 4096)); float ms;
        const u32 blocks = 8192; const u64 per = total / blocks;
        for (u32 shift : {0u, 1u, 4u, 8u, 15u}) {
            hipLaunchKernelGGL(store16, dim3(blocks), dim3(256), 0, 0, big, shift, per);
            CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store16, dim3(blocks), dim3(256), 0, 0, big, shift, per); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("store16 shift %2u: %.3f ms per GiB = %.2f TB/s\n", shift, ms / 5, 1.0737 / (ms / 5));
        }
    }
    return 0;
}
