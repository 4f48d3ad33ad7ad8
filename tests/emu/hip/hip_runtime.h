/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * A single-header stand-in for <hip/hip_runtime.h> that runs HIP kernels on the
 * CPU, one workgroup at a time, every work-item as a fiber (ucontext).  It
 * exists so that the kernel sources under aws-c-compression_amd/csrc/hip can be
 * compiled unchanged with g++ and exercised -- with UBSan, and with barrier
 * divergence detection -- in the container that has no GPU (GPU sanitizers are
 * not available on the pool).  tests/emu/Makefile builds
 * tests/emu/libaws-c-compression-emu.so from the same sources with
 * -Itests/emu in front of the include path.
 *
 * The product library (built by hipcc) never sees this file, and nothing here is
 * a fallback for it: the emulated library is only ever loaded by the CPU-side
 * logic tests in tests/test_emulated_kernels.py.  Parity claims rest on the
 * `-m gpu` tests, which run the hipcc build on an MI355X.
 *
 * Semantics: work-items of a workgroup run to their next barrier in a shuffled
 * order (seeded), so code that forgets a __syncthreads() tends to fail here
 * too.  Wave-level primitives (__shfl*, __ballot) synchronise the 64 lanes of a
 * wave.  A barrier that not all live work-items reach aborts with a message.
 */
#pragma once

#include <ucontext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 {
    unsigned x, y, z, w;
};

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct hip_emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorInvalidDevice = 101 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hip_emu {

constexpr unsigned kWave = 64;
constexpr size_t kStackBytes = 256 * 1024;
constexpr size_t kLdsBytes = 160 * 1024;

enum class Wait { kNone, kBlock, kWave, kDone };

struct Fiber {
    ucontext_t ctx;
    void *stack = nullptr;
    Wait wait = Wait::kNone;
    unsigned tid = 0;
    unsigned long long xchg = 0; /* value published for wave exchanges */
};

struct State {
    alignas(16) unsigned char lds[kLdsBytes];
    std::vector<Fiber> fibers;
    ucontext_t scheduler;
    Fiber *cur = nullptr;
    dim3 block_idx, block_dim, grid_dim;
    const std::function<void()> *body = nullptr;
    std::mt19937 rng{12345};
    int last_error = 0;
};

inline State &state() {
    static State s;
    return s;
}

inline unsigned char *lds() {
    return state().lds;
}

inline void fiber_entry() {
    State &s = state();
    (*s.body)();
    s.cur->wait = Wait::kDone;
    swapcontext(&s.cur->ctx, &s.scheduler);
}

inline void yield(Wait why) {
    State &s = state();
    s.cur->wait = why;
    swapcontext(&s.cur->ctx, &s.scheduler);
}

/* Runs one workgroup to completion. */
inline void run_block(unsigned threads) {
    State &s = state();
    if (s.fibers.size() < threads) {
        s.fibers.resize(threads);
    }
    for (unsigned t = 0; t < threads; ++t) {
        Fiber &f = s.fibers[t];
        if (!f.stack) {
            f.stack = std::malloc(kStackBytes);
        }
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        f.wait = Wait::kNone;
        f.tid = t;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    std::vector<unsigned> order(threads);
    for (unsigned t = 0; t < threads; ++t) {
        order[t] = t;
    }
    for (;;) {
        std::shuffle(order.begin(), order.end(), s.rng);
        bool ran = false;
        for (unsigned t : order) {
            Fiber &f = s.fibers[t];
            if (f.wait != Wait::kNone) {
                continue;
            }
            ran = true;
            s.cur = &f;
            swapcontext(&s.scheduler, &f.ctx);
        }
        /* release barriers whose every live participant has arrived */
        unsigned done = 0, at_block = 0;
        for (unsigned t = 0; t < threads; ++t) {
            done += s.fibers[t].wait == Wait::kDone;
            at_block += s.fibers[t].wait == Wait::kBlock;
        }
        if (done == threads) {
            return;
        }
        bool released = false;
        if (at_block && at_block + done == threads) {
            for (unsigned t = 0; t < threads; ++t) {
                if (s.fibers[t].wait == Wait::kBlock) {
                    s.fibers[t].wait = Wait::kNone;
                }
            }
            released = true;
        }
        for (unsigned w = 0; w * kWave < threads; ++w) {
            const unsigned lo = w * kWave, hi = std::min(threads, lo + kWave);
            unsigned at_wave = 0, gone = 0;
            for (unsigned t = lo; t < hi; ++t) {
                at_wave += s.fibers[t].wait == Wait::kWave;
                gone += s.fibers[t].wait == Wait::kDone;
            }
            if (at_wave && at_wave + gone == hi - lo) {
                for (unsigned t = lo; t < hi; ++t) {
                    if (s.fibers[t].wait == Wait::kWave) {
                        s.fibers[t].wait = Wait::kNone;
                    }
                }
                released = true;
            }
        }
        if (!ran && !released) {
            std::fprintf(
                stderr,
                "hip_emu: deadlock in block %u: %u of %u work-items wait at a workgroup barrier the others never reach\n",
                s.block_idx.x, at_block, threads);
            std::abort();
        }
    }
}

inline void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
    /* one launch at a time: the fibers, the LDS and the block index are one global state, and host threads with an
     * engine each (the sharded driver) may launch side by side */
    static std::mutex one_launch;
    std::lock_guard<std::mutex> hold(one_launch);
    State &s = state();
    if (lds_bytes > kLdsBytes || block.x * block.y * block.z > 1024) {
        s.last_error = hipErrorInvalidValue;
        return;
    }
    s.body = &body;
    s.block_dim = block;
    s.grid_dim = grid;
    for (unsigned b = 0; b < grid.x; ++b) {
        s.block_idx = dim3(b, 0, 0);
        /* poison the LDS so reads of never-written shared memory stand out */
        std::memset(s.lds, 0xA5, lds_bytes ? lds_bytes : 1);
        run_block(block.x);
    }
}

struct IdxProxy {
    unsigned x, y, z;
};

inline unsigned long long wave_exchange(unsigned long long mine, unsigned from_lane, bool *valid) {
    State &s = state();
    const unsigned tid = s.cur->tid;
    const unsigned base = tid & ~(kWave - 1);
    s.cur->xchg = mine;
    yield(Wait::kWave);
    unsigned long long got = mine;
    *valid = false;
    if (from_lane < kWave && base + from_lane < s.block_dim.x) {
        got = s.fibers[base + from_lane].xchg;
        *valid = true;
    }
    yield(Wait::kWave); /* nobody republishes before everyone has read */
    return got;
}

/* one exchange for the whole wave: publish the predicate, then read every lane's */
inline unsigned long long wave_ballot(bool predicate) {
    State &s = state();
    const unsigned base = s.cur->tid & ~(kWave - 1);
    s.cur->xchg = predicate ? 1 : 0;
    yield(Wait::kWave);
    unsigned long long mask = 0;
    for (unsigned l = 0; l < kWave && base + l < s.block_dim.x; ++l) {
        /* lanes that have left the kernel count as inactive */
        if (s.fibers[base + l].wait != Wait::kDone && s.fibers[base + l].xchg) {
            mask |= 1ull << l;
        }
    }
    yield(Wait::kWave);
    return mask;
}

} /* namespace hip_emu */

#define threadIdx (hip_emu::IdxProxy{hip_emu::state().cur->tid, 0, 0})
#define blockIdx (hip_emu::IdxProxy{hip_emu::state().block_idx.x, 0, 0})
#define blockDim (hip_emu::IdxProxy{hip_emu::state().block_dim.x, 1, 1})
#define gridDim (hip_emu::IdxProxy{hip_emu::state().grid_dim.x, 1, 1})

#define HIP_DYNAMIC_SHARED(type, var) static type *const var = reinterpret_cast<type *>(hip_emu::lds());

/* HIP_EMU_TRACE=1: the name of every kernel launched, on stderr (tests that must know which road a call took) */
namespace hip_emu {
inline void trace_launch(const char *name) {
    static const bool on = std::getenv("HIP_EMU_TRACE") != nullptr;
    if (on) {
        std::fprintf(stderr, "hip_emu launch %s\n", name);
    }
}
} // namespace hip_emu
#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...)                                                \
    (hip_emu::trace_launch(#kernel), hip_emu::launch((grid), (block), (lds_bytes), [&]() { (kernel)(__VA_ARGS__); }))

inline void __syncthreads() {
    hip_emu::yield(hip_emu::Wait::kBlock);
}

template <typename T>
inline T __shfl_up(T v, unsigned delta) {
    const unsigned lane = hip_emu::state().cur->tid & (hip_emu::kWave - 1);
    bool ok;
    const unsigned long long got = hip_emu::wave_exchange((unsigned long long)v, lane >= delta ? lane - delta : ~0u, &ok);
    return ok ? (T)got : v;
}
template <typename T>
inline T __shfl_down(T v, unsigned delta) {
    const unsigned lane = hip_emu::state().cur->tid & (hip_emu::kWave - 1);
    bool ok;
    const unsigned long long got = hip_emu::wave_exchange((unsigned long long)v, lane + delta, &ok);
    return ok ? (T)got : v;
}
template <typename T>
inline T __shfl_xor(T v, unsigned mask) {
    const unsigned lane = hip_emu::state().cur->tid & (hip_emu::kWave - 1);
    bool ok;
    const unsigned long long got = hip_emu::wave_exchange((unsigned long long)v, lane ^ mask, &ok);
    return ok ? (T)got : v;
}
template <typename T>
inline T __shfl(T v, unsigned src_lane) {
    bool ok;
    const unsigned long long got = hip_emu::wave_exchange((unsigned long long)v, src_lane & (hip_emu::kWave - 1), &ok);
    return ok ? (T)got : v;
}
inline unsigned long long __ballot(int predicate) {
    return hip_emu::wave_ballot(predicate != 0);
}
inline int __any(int predicate) {
    return __ballot(predicate) != 0;
}
inline int __all(int predicate) {
    return __ballot(!predicate) == 0;
}

/* ((hi:lo) << (shift & 31)) >> 32, as the HIP device function of the same name */
inline unsigned __funnelshift_l(unsigned lo, unsigned hi, unsigned shift) {
    const unsigned k = shift & 31u;
    return k ? (hi << k) | (lo >> (32u - k)) : hi;
}
inline int __popc(unsigned v) {
    return __builtin_popcount(v);
}
inline int __popcll(unsigned long long v) {
    return __builtin_popcountll(v);
}

template <typename T>
inline T atomicOr(T *p, T v) {
    const T old = *p;
    *p = old | v;
    return old;
}
template <typename T>
inline T atomicAdd(T *p, T v) {
    const T old = *p;
    *p = old + v;
    return old;
}
template <typename T>
inline T atomicMin(T *p, T v) {
    const T old = *p;
    *p = v < old ? v : old;
    return old;
}
template <typename T>
inline T atomicMax(T *p, T v) {
    const T old = *p;
    *p = v > old ? v : old;
    return old;
}

/* clang's scoped atomic builtins, as the kernels spell their agent-scope hand-offs: one workgroup runs at a
 * time here and its work-items are fibers of one thread, so a plain access is what they come to */
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
template <typename T>
inline T __hip_atomic_load(const T *p, int, int) {
    return *p;
}
template <typename T, typename V>
inline void __hip_atomic_store(T *p, V v, int, int) {
    *p = (T)v;
}
template <typename T, typename V>
inline T __hip_atomic_fetch_add(T *p, V v, int, int) {
    const T old = *p;
    *p = old + (T)v;
    return old;
}

/* ------------------------------------------------------------------ runtime API: host memory stands in for device memory */

inline hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int) {
    return hipSuccess;
}
inline const char *hipGetErrorString(hipError_t e) {
    return e == hipSuccess ? "hipSuccess (emulated)" : "hip error (emulated)";
}
inline hipError_t hipGetLastError() {
    const int e = hip_emu::state().last_error;
    hip_emu::state().last_error = 0;
    return e;
}
inline hipError_t hipMalloc(void **p, size_t n) {
    /* fill with a pattern: device memory is not zeroed either */
    *p = std::malloc(n + 64);
    if (!*p) {
        return hipErrorOutOfMemory;
    }
    std::memset(*p, 0xCD, n + 64);
    return hipSuccess;
}
inline hipError_t hipFree(void *p) {
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) {
    *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipHostFree(void *p) {
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
    std::memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = reinterpret_cast<hipStream_t>(0x1);
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) {
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize() {
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) {
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = reinterpret_cast<hipEvent_t>(std::malloc(8));
    return hipSuccess;
}
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) {
    return hipEventCreate(e);
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    std::free(e);
    return hipSuccess;
}
/* (launches run to completion where they are made: every stream is already in order with every other) */
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) {
    return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) {
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) {
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
    *ms = 0.0f;
    return hipSuccess;
}
inline hipError_t hipFuncSetAttribute(const void *, int, int) {
    return hipSuccess;
}
struct hipDeviceProp_t {
    int multiProcessorCount;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int) {
    /* one "CU" with one resident workgroup: workgroups run one after another here, so a
     * persistent kernel whose workgroups wait for each other (look-back) must fit in one */
    prop->multiProcessorCount = 1;
    return hipSuccess;
}
template <typename Kernel>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *blocks, Kernel, int, size_t) {
    *blocks = 1;
    return hipSuccess;
}
