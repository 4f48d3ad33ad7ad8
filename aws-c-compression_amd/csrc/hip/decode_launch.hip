/*
 * The decode launch: what runs in which order on the launch's stream, the entry scan between the two passes (dec_scan_*:
 * the true path through the chunk functions of every item), the plans made on the device, the per-device set-up.
 */
#include "decode_common.hpp"
#include "launch_common.hpp"

namespace {

/* ------------------------------------------------------------------ decode: scan */

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
 * run, in two short launches (the walk along an item is a chain of dependent table look-ups:
 * what matters is that every look-up is an LDS read and every chain is short):
 *   dec_scan_runs   the run's chunk functions into LDS, folded 16 at a time and then once more:
 *                   the run's own transfer function
 *   dec_scan_apply  per run: the true path through the run functions of the item's runs in front of it (in LDS, in
 *                   groups) -> the run's entry; the same fold of its chunks again, then the true path through the 16
 *                   sub-runs and through the chunks of each -> entry state and symbol offset of every chunk; the
 *                   item's last run writes the item's outcome
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

__device__ __host__ constexpr u32 scan_run_lds_bytes_device(u32 ns) {
    return kRunChunks * ns * 4 + kSubRuns * ns * 4 + kSubRuns * 4 + kSubRuns * 8 + 16;
}
static uint32_t scan_run_lds_bytes(uint32_t ns) {
    return scan_run_lds_bytes_device(ns);
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

#ifndef HUFD_SCAN_TOP_TILE /* (tests/emu: a few runs a tile, two a group) */
#define HUFD_SCAN_TOP_TILE 512u
#define HUFD_SCAN_TOP_GROUP 16u
#endif
constexpr u32 kTopTile = HUFD_SCAN_TOP_TILE; /* run functions of an item held in LDS at a time (an item has at most 512 runs: 4 GiB) */
constexpr u32 kTopGroup = HUFD_SCAN_TOP_GROUP;  /* ... and walked in groups of this many: the true path is a chain of dependent look-ups, ~0.1 us
                                * each -- 152 runs of the 1 GiB stream one after the other were 18 us between the sync and the
                                * emit kernels; 16 (every group from every state, side by side) + 10 (the groups) are 3 */
constexpr u32 kTopGroups = kTopTile / kTopGroup;

struct scan_point { /* the true path at a run's first chunk */
    u32 state;
    bool stopped;
    u64 total;
};

/*
 * The true path of item `it` up to its run `k` (0: the item's first bit), through the run functions of the runs in front:
 * the whole workgroup, every thread gets the answer.  `fn`: kTopTile * ns words of LDS, `group`: 2 * kTopGroups * ns.
 * (Round 4 had a kernel of its own for this, a workgroup an item, between dec_scan_runs and dec_scan_apply: a launch more
 * on every decode's critical path.  Every run's workgroup now walks the runs in front of its own -- at most 511 -- itself.)
 */
__device__ __forceinline__ scan_point scan_path_to_run(const hufd_dec_item &it, const u32 *run_fn_of_item, u32 k, u32 ns, u32 *fn, u32 *group) {
    u32 *group_to = group;                       /* [kTopGroups][ns] where a group leaves: stop << 31 | state */
    u32 *group_count = group + kTopGroups * ns;  /* [kTopGroups][ns] ... and its symbols on the way (16 runs: < 2^30) */
    scan_point p = {it.first_bit, false, 0};
    for (u32 base = 0; base < k; base += kTopTile) {
        const u32 n = k - base < kTopTile ? k - base : kTopTile;
        __syncthreads();
        for (u32 q = threadIdx.x; q < n * ns; q += blockDim.x) {
            fn[q] = run_fn_of_item[(u64)base * ns + q];
        }
        __syncthreads();
        const u32 groups = (n + kTopGroup - 1) / kTopGroup;
        /* every group from every state */
        for (u32 idx = threadIdx.x; idx < groups * ns; idx += blockDim.x) {
            const u32 g = idx / ns, first = g * kTopGroup;
            const u32 cnt = n - first < kTopGroup ? n - first : kTopGroup;
            const fold_result r = chain_fold(cnt, idx % ns, [&](u32 q, u32 stt) { return fn[(first + q) * ns + stt]; });
            group_to[idx] = (r.stop ? 0x80000000u : 0u) | r.state;
            group_count[idx] = (u32)r.count;
        }
        __syncthreads();
        /* the true path over the groups */
        for (u32 g = 0; g < groups && !p.stopped; ++g) {
            const u32 to = group_to[g * ns + p.state];
            p.total += group_count[g * ns + p.state];
            p.stopped = (to >> 31) != 0;
            p.state = to & 0x7FFFFFFFu;
        }
    }
    __syncthreads();
    return p;
}

static uint32_t scan_apply_lds_bytes(uint32_t ns) {
    return scan_run_lds_bytes(ns) + kTopTile * ns * 4 + 2 * kTopGroups * ns * 4;
}

__global__ __launch_bounds__(256) void dec_scan_apply_kernel(
    const hufd_dec_item *items,
    const u32 *runs,
    u32 ns,
    const u32 *chunk_fn,
    const u32 *run_fn,
    u32 *chunk_entry,
    u64 *chunk_base,
    hufd_dec_item_state *states,
    hufd_dec_result *results) {
    u32 *fn = reinterpret_cast<u32 *>(dyn_lds);
    u32 *sub = fn + kRunChunks * ns;
    u32 *sub_entry = sub + kSubRuns * ns;                           /* [kSubRuns] */
    u64 *sub_base = reinterpret_cast<u64 *>(sub_entry + kSubRuns);  /* [kSubRuns] */
    u32 *path_fn = reinterpret_cast<u32 *>(dyn_lds + scan_run_lds_bytes_device(ns));
    u32 *path_group = path_fn + kTopTile * ns;
    const u32 run = blockIdx.x;
    const u32 item = runs[2 * run], k = runs[2 * run + 1];
    const hufd_dec_item it = items[item];
    /* (an item's runs are listed one after the other: its first is k in front of this one) */
    const scan_point at = scan_path_to_run(it, run_fn + (u64)(run - k) * ns, k, ns, path_fn, path_group);
    const u32 n = scan_run_load(it, k, ns, chunk_fn, fn, sub);
    if (threadIdx.x == 0) {
        u32 state = at.state;
        bool stopped = at.stopped;
        u64 total = at.total;
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
        if ((k + 1) * kRunChunks >= it.n_chunks) {
            dec_finish_item(it, total, stopped, &states[item], &results[item]); /* the item's last run: its outcome */
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
    rec.entry_bit = c == it.first_chunk ? it.first_bit : HUFD_NONE32;
    rec.reserved = 0;
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


/* per device: compute units (sizes the grids of the persistent kernels) and whether hufk_init has run there */
constexpr int kMaxDevices = 64;
static int s_compute_units[kMaxDevices];
static bool s_device_ready[kMaxDevices];
static pthread_mutex_t s_init_lock = PTHREAD_MUTEX_INITIALIZER;

} /* namespace */

namespace {
struct occupancy_note {
    const void *kernel;
    uint32_t threads, lds_bytes;
    int per_cu;
};
constexpr int kOccupancyNotes = 256;
occupancy_note s_occupancy[kOccupancyNotes];
int s_occupancy_count = 0;
pthread_mutex_t s_occupancy_lock = PTHREAD_MUTEX_INITIALIZER;
} /* namespace */

int hufk_host::blocks_per_cu_remembered(const void *kernel, uint32_t threads, uint32_t lds_bytes) {
    int found = -1;
    pthread_mutex_lock(&s_occupancy_lock);
    for (int i = 0; i < s_occupancy_count; ++i) {
        if (s_occupancy[i].kernel == kernel && s_occupancy[i].threads == threads && s_occupancy[i].lds_bytes == lds_bytes) {
            found = s_occupancy[i].per_cu;
            break;
        }
    }
    pthread_mutex_unlock(&s_occupancy_lock);
    return found;
}

void hufk_host::blocks_per_cu_remember(const void *kernel, uint32_t threads, uint32_t lds_bytes, int per_cu) {
    pthread_mutex_lock(&s_occupancy_lock);
    if (s_occupancy_count < kOccupancyNotes) { /* (a full table: the query is asked again, as before) */
        s_occupancy[s_occupancy_count++] = occupancy_note{kernel, threads, lds_bytes, per_cu};
    }
    pthread_mutex_unlock(&s_occupancy_lock);
}

int hufk_host::current_compute_units() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices || s_compute_units[device] <= 0) {
        return 256;
    }
    return s_compute_units[device];
}

using hufk_host::persistent_grid;
using hufk_host::stage_mark;
using hufk_host::current_compute_units;
using hufk_host::decode_launch_state;

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
    hipError_t e = hufk_host::init_encode(lds_max);
    if (e == hipSuccess) {
        e = hufk_host::init_decode_sync(lds_max);
    }
    if (e == hipSuccess) {
        e = hufk_host::init_decode_items(lds_max);
    }
    if (e == hipSuccess) {
        e = hufk_host::init_decode_emit(lds_max);
    }
    s_device_ready[device] = e == hipSuccess;
    pthread_mutex_unlock(&s_init_lock);
    return (int)e;
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

int hufk_decode_launch(const struct hufk_decode_args *a, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (a->n_items == 0) {
        return 0;
    }
    const uint32_t ns = a->tables.n_states;
    stage_mark(a->stage_events, 0, st);
    if (a->n_fixed_blocks && a->tables.fixed_bits) {
        (void)hipMemsetAsync(a->states, 0xFF, (size_t)a->n_items * sizeof(hufd_dec_item_state), st);
    }
    /* the builds of the row-synchronous kernels: a decode table of up to 10 bits with 3 or 4 certain steps a row (codes
     * of up to 10, 8 bits; a coder of shorter codes has more certain steps than that: the rest are asked for -- round 6
     * took the builds with 5 out, for codes of at most 6 bits, when the kernels for a few stream ends among many chunks came in),
     * of 11 or 12 bits with 2 */
    const uint32_t lb_of_launch = a->tables.lut_bits <= 10 ? 10u : 12u;
    const uint32_t sure_of_coder = row_walk(lb_of_launch, a->tables.max_bits).sure;
    const uint32_t sure = lb_of_launch == 10 ? (sure_of_coder > 4 ? 4u : sure_of_coder) : (sure_of_coder > 2 ? 2u : sure_of_coder);
    if (a->n_chunks && (a->tables.max_bits > HUFD_DEC_MAX_LUT_BITS || (lb_of_launch == 10 ? sure < 3 : sure != 2))) {
        return (int)hipErrorInvalidValue; /* (a plan has chunks only for a decode table of up to 12 bits; see row_walk for the steps) */
    }
    decode_launch_state state = {lb_of_launch, sure, false};
    if (a->n_chunks) {
        if (!a->counters_self_cleared) {
            (void)hipMemsetAsync(a->counters, 0, HUFK_DEC_COUNTERS * sizeof(uint32_t), st);
        }
        hufk_host::decode_sync_stage(a, st, state);
    }
    stage_mark(a->stage_events, 1, st);
    if (a->n_tiny != a->n_items && a->n_large != a->n_items) { /* (as in the encoder: nothing to scan, and no empty item's record to write, in a plan of thread-per-item items only; nor in one of long items only -- one stream --, which are dec_scan_runs / _top / _apply's) */
        hipLaunchKernelGGL(
            dec_scan_small_kernel, dim3((a->n_items + 255) / 256), dim3(256), 0, st, a->items, a->n_items, ns, a->chunk_fn,
            a->chunk_entry, a->chunk_base, a->states, a->results);
    }
    hufk_host::decode_items_stage(a, st);
    if (a->n_large) {
        const uint32_t lds = scan_run_lds_bytes(ns);
        hipLaunchKernelGGL(
            dec_scan_runs_kernel, dim3(a->n_runs), dim3(256), lds, st, a->items, a->runs, ns, a->chunk_fn, a->run_fn);
        hipLaunchKernelGGL(
            dec_scan_apply_kernel, dim3(a->n_runs), dim3(256), scan_apply_lds_bytes(ns), st, a->items, a->runs, ns, a->chunk_fn,
            (const u32 *)a->run_fn, a->chunk_entry, a->chunk_base, a->states, a->results);
    }
    hufk_host::decode_sync_true_stage(a, st, state);
    stage_mark(a->stage_events, 2, st);
    if (a->n_chunks) {
        hufk_host::decode_emit_stage(a, st, state);
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
