/*
 * Host side of the launches, shared by the translation units of csrc/hip: the size of a resident grid, the stage events,
 * and the pieces of a decode launch that live with their kernels (hufk_decode_launch, decode_launch.hip, runs them in
 * order on the launch's stream).
 */
#ifndef HUFFMAN_AMD_LAUNCH_COMMON_HPP
#define HUFFMAN_AMD_LAUNCH_COMMON_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "huffman_kernels.h"

namespace hufk_host {

/* compute units of the calling thread's current device (hufk_init has counted them; 256 before that) */
int current_compute_units();

/* what the occupancy query said of a kernel at a block size and an LDS size, remembered: a launch sizes the grids of some
 * ten resident kernels, and the query is a few microseconds of the host's time each (the first one of a kernel a good deal
 * more) -- a third of what a small call costs.  (One table a process: every device of a process is the same chip.) */
int blocks_per_cu_remembered(const void *kernel, uint32_t threads, uint32_t lds_bytes); /* -1: not asked yet */
void blocks_per_cu_remember(const void *kernel, uint32_t threads, uint32_t lds_bytes, int per_cu);

/* workgroups of a persistent kernel that one launch keeps resident: CUs x blocks per CU */
template <typename Kernel>
inline uint32_t persistent_grid(Kernel kernel, uint32_t threads, uint32_t lds_bytes, uint32_t work_items, uint32_t sgprs = 0) {
    int per_cu = blocks_per_cu_remembered(reinterpret_cast<const void *>(kernel), threads, lds_bytes);
    if (per_cu < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)threads, lds_bytes) != hipSuccess ||
            per_cu < 1) {
            per_cu = 1;
        }
        blocks_per_cu_remember(reinterpret_cast<const void *>(kernel), threads, lds_bytes, per_cu);
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

inline void stage_mark(void **events, int index, hipStream_t st) {
    if (events) {
        (void)hipEventRecord((hipEvent_t)events[index], st);
    }
}

constexpr uint32_t kBesideMinChunks = 1024; /* launches of fewer chunks keep to one stream */

/* a FEW chunks that streams end in among many inside streams (one long stream: one): they are workgroups of the big
 * kernels' own grids (dec_sync_one_mixed_kernel, dec_emit_fast_mixed_kernel), not launches of their own */
inline bool tails_are_folded(const struct hufk_decode_args *a) {
    return a->n_tail && a->n_tail < a->n_chunks && (uint64_t)a->n_tail * 8 <= a->n_chunks && !a->tails_apart;
}

/* which builds of the row-synchronous kernels a launch takes, and what its sync stage tells the stages behind it */
struct decode_launch_state {
    uint32_t lb;   /* 10 or 12: the decode table's bits as compiled */
    uint32_t sure; /* certain steps a row as compiled: 3, 4 (10-bit tables) or 2 (12-bit) */
    bool few;      /* dec_sync_few ran: dec_sync_true follows behind the scan */
};

/* dynamic LDS above 64 KiB is opt-in per kernel: every translation unit opts its own in */
hipError_t init_encode(int lds_max);
hipError_t init_decode_sync(int lds_max);
hipError_t init_decode_items(int lds_max);
hipError_t init_decode_emit(int lds_max);

void decode_sync_stage(const struct hufk_decode_args *a, hipStream_t st, decode_launch_state &s);
void decode_sync_true_stage(const struct hufk_decode_args *a, hipStream_t st, const decode_launch_state &s);
void decode_items_stage(const struct hufk_decode_args *a, hipStream_t st);
void decode_emit_stage(const struct hufk_decode_args *a, hipStream_t st, const decode_launch_state &s);

} /* namespace hufk_host */

#endif /* HUFFMAN_AMD_LAUNCH_COMMON_HPP */
