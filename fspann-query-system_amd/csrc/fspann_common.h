// fspann_common.h — context, error plumbing and small helpers shared by the
// gfx950 kernels and the C ABI (include/fspann.h).  Product code: never includes
// or links anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fspann.h"

namespace fspann {

// ---- thread-local error message ------------------------------------------------
inline std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define FSP_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::fspann::fail(_e == hipErrorOutOfMemory ? FSPANN_E_NOMEM : FSPANN_E_DEVICE,    \
                                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                  __LINE__);                                                       \
    } while (0)

// ---- order-key bit budget (DESIGN.md "Java order key") ----------------------------
// key = score(10) | bucket(20) | seq(22); seq = (td*P + step)*S + pos is unique per tuple.
constexpr int kSeqBits = 22;
constexpr int kBucketBits = 20;
constexpr int kScoreBits = 10;
constexpr uint32_t kSeqMask = (1u << kSeqBits) - 1;
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

// One (t,d) table inside the concatenated device arrays.
struct RouteTable {
    int64_t part_base;  // first partition of this table in keys2 / rep
    int64_t off_base;   // first entry of this table in id_off (nparts+1 entries per table)
    int64_t ids_base;   // first id of this table in ids
    int32_t nparts;
    int32_t dir_base;   // first entry pair of this table in the radix directory (RouteParams::dir), in pairs
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    unsigned gen = 0;   // bumped by every (re)allocation: "same address" does not mean "same contents"
};

}  // namespace fspann

// The opaque context of include/fspann.h.
constexpr int kDirBitsAuto = 99;   // knob_dir_extra_bits: size the probe's radix directory by a memory budget (fspann_api.hip)
struct fspann_ctx {
    // Calls on one context are serialised INSIDE the library: every entry point that takes a context holds this
    // lock for its duration (recursive: entry points call each other).  Different contexts — e.g. the clones of one index —
    // run concurrently.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    fspann_cfg cfg{};
    int TD = 0, W = 0, bits = 0, P_total = 0;  // P_total = TD*m projections
    int hard_cap = 0;                          // max(maxGlobalCandidates, refinementLimit), PIS:612-615
    int cap0 = 0;                              // tableSizeFor(min(HARD_CAP, 1<<16)), PIS:619
    int num_cus = 256;
    int lds_limit = 160 * 1024;
    bool frozen = false;
    bool have_g = false;
    long long* dbg_route = nullptr;  // per-block phase stamps; only reachable in FSPANN_DEBUG_STAMPS builds (tools/)
    // tuning / test knobs, read from the environment ONCE at fspann_ctx_create (never per call)
    int knob_ht_x4 = 0;              // FSPANN_ROUTE_HT_X4=1: hash load factor <= 0.5 instead of <= 0.8
    int knob_threads = 0;            // FSPANN_ROUTE_THREADS: 512 / 1024 force the full select's workgroup size (0: by list length)
    int knob_lazy_cap = 0;           // FSPANN_ROUTE_LAZY_CAP: entries one query may hold in the bounded select (tests)
    int knob_fused_probe = 1;        // FSPANN_ROUTE_FUSED_PROBE=0: separate probe kernel in front of the bounded select
    int knob_refine_dc = 0;          // FSPANN_REFINE_DC: dims per LDS tile of the refinement scan (tools/refine_bench.py)
    int knob_dir_extra_bits = kDirBitsAuto;   // FSPANN_ROUTE_DIR_EXTRA_BITS: finer (+) or coarser (-) radix directory than four partitions per entry; unset: auto
    bool knob_lazy_small = true;     // FSPANN_ROUTE_LAZY_SMALL: 512-entry size class of the bounded select for limit <= 256 (0: always 1024)
    bool knob_probe_dir = true;      // FSPANN_ROUTE_DIR: radix directory + one-round window in the probe (0: plain G-ary search)
    int knob_refine_stream = -1;     // FSPANN_REFINE_STREAM: workgroups per CU of the streaming refinement scan (-1: 4 dense / 3 gather, 0: one workgroup per query)
    int knob_tick_refine = 1;        // FSPANN_TICK_REFINE: refine workgroups per CU inside a tick (each streams several queries)
    int knob_gpu_cut = 1;            // FSPANN_GPU_CUT=0: fspann_build_index cuts the partitions on host threads (std::sort) instead of the GPU radix sort
    int knob_wave_sort = 1;          // FSPANN_ROUTE_WAVE_SORT=0: long lists' groups are sorted by the whole workgroup one by one (dev A/B)
    int knob_tick_fuse = 1;          // FSPANN_TICK_FUSE=0: fspann_tick_dev always uses the stand-alone kernels
    int knob_tick_front = 100;       // FSPANN_TICK_FRONT: percent of a tick's Route workgroups that head the grid
    int knob_bincheck = -1;          // FSPANN_ROUTE_BINCHECK: the bounded select's exact treeify check (-1: on for opaque ids, off for decimal ordinals; 0 / 1 force)
    bool knob_shape_spec = true;     // FSPANN_ROUTE_SHAPE_SPEC=0: the bounded select's build with run-time tables x probes also for 16 x 5 (dev A/B)
    int knob_mfma_tile = 0;          // FSPANN_ENCODE_MFMA_TILE=1: the 32 x 128 block tile of the MFMA encode for every batch size (dev A/B)
    int knob_encode_qb = 0;          // FSPANN_ENCODE_QB: query rows per workgroup of the exact encode (0 = by batch size; dev A/B)
    bool knob_zero_copy = true;      // FSPANN_ZERO_COPY=0: tiny host-pointer calls go through copy commands like larger ones (dev A/B)
    int knob_route_lds_kb = 0;       // FSPANN_ROUTE_LDS_KB: LDS the full select may plan with (0 = all of it); less leaves room for scan workgroups beside it
    int knob_route_wgs = 0;          // FSPANN_ROUTE_WGS: workgroups per CU of the full select's grid (0 = what fits, at most 4)
    int knob_devflags = 0;           // FSPANN_ROUTE_DEVFLAGS: dev A/B switches of the full select (route.hip.h)
    bool knob_refine_run = true;     // FSPANN_REFINE_RUN=0: long lists keep one partial top-k list per 256-row chunk (dev A/B)
    bool knob_slice = true;          // FSPANN_ROUTE_SLICE=0: the full select's global-arena mode builds its hash in the arena (dev A/B)
    int last_tick_fused = 0;

    // GFunctions: alphaT[dim][P_total] fp64 (transposed for coalescing), r/omega[P_total]
    double* d_alphaT = nullptr;
    double* d_alpha_rows = nullptr;   // alpha as the JVM hands it over, [P][d]: the re-check of the MFMA path reads a projection's row 64 dimensions at a time
    double* d_r = nullptr;
    double* d_omega = nullptr;
    std::vector<double> h_alpha, h_r, h_omega;  // host copies (export)
    float* d_alphaT32 = nullptr;               // fp32 copy of alphaT for the MFMA fast path
    double alpha_norm_max = 1.0;               // max_j ||alpha_j||_2 (error bound of the fast path)
    int encode_mode = 0;                       // 0 auto, 1 exact fp64, 2 MFMA fp32 + exact re-check
    unsigned long long fix_cap_last = 0;       // capacity of the re-check list of the last MFMA-path call
    bool mfma_last = false;                    // the last encode took the MFMA path
    fspann::DevBuf ws_fix;                     // fix list + counter + int32 hashes of the MFMA path

    // frozen index (concatenated over td)
    std::vector<fspann::RouteTable> h_tables;
    std::vector<std::vector<int64_t>> h_min, h_max, h_off;  // host mirror per td (export / rebuild)
    std::vector<std::vector<uint64_t>> h_rep;
    std::vector<std::vector<int32_t>> h_ids;
    std::vector<char> h_table_set;
    bool dev_index_dirty = true;
    fspann::RouteTable* d_tables = nullptr;
    int64_t* d_recs = nullptr;    // [total_parts][rec_words] partition records {minKey, maxKey, rep[W], id offset | size << 32}
    int rec_words = 0;
    int2* d_dir = nullptr;        // radix directory of the probe: [TD][2^dir_bits + 1] {first maxKey >= bound, first minKey >= bound}
    int dir_bits = 0;
    int32_t* d_ids = nullptr;
    int32_t* d_inv = nullptr;        // [TD][n_ids] inverse id map for the bounded select (null: a table holds an id twice)
    uint64_t* d_ids_bk = nullptr;    // per partition: (id << 32 | bucket field) sorted by bucket, for the bounded select
    uint16_t* d_bin16 = nullptr;     // [total_parts][1 << bin16_shift] HashMap bin (at cap0) of every id, partition order, padded per partition:
    int bin16_shift = 0;             //   what the bounded select's exact treeify check reads (one 128-byte line per 64-id partition)
    int meta_epoch = 0, bk_epoch = -1;  // d_ids_bk / d_inv are valid for the id metadata of bk_epoch
    int route_mode = 0;              // 0 auto, 1 always route_select_kernel, 2 bounded select whenever its preconditions hold
    fspann::DevBuf ws_ovf;
    // kernel-attached HIP events of the refinement scan (fspann_refine_timing_begin / _end)
    std::vector<hipEvent_t> rt_events;   // pairs: start, stop
    size_t rt_used = 0;
    int rt_every = 1, rt_seen = 0;       // events go on every rt_every-th dispatch
    bool rt_on = false;
    fspann::DevBuf ws_search;        // codes / F_q ids / counts of fspann_search_store_dev
    unsigned attr_mask = 0;          // kernels whose dynamic-LDS ceiling has been raised on this context's device
    int ovf_flip = 0;                // which of the two overflow counters the last bounded select used
    unsigned ovf_gen_seen = 0;       // ws_ovf.gen whose counters have been zeroed (0 = never)
    int last_route_lazy = 0;         // 1 if the last fspann_route[_dev] ran the bounded select
    int32_t* d_unmodelled = nullptr; // device counter: queries flagged "HashMap bin treeified" (out_count = -1) since the last reset
    int64_t total_parts = 0, total_ids = 0;

    // id metadata
    int64_t n_ids = 0;
    int32_t* d_java_hash = nullptr;
    // mirror of metadata.isDeleted (PIS:739), one bit per handle; nullptr => nothing deleted so far.  Owned by the index owner;
    // a clone reads the OWNER's pointer at every call (fspann_set_deleted may allocate it while clones are alive), so it is atomic.
    std::atomic<uint32_t*> d_deleted_bits{nullptr};
    std::vector<uint32_t> h_deleted_bits;   // host mirror (owner only; guarded by deleted_mu): the host replay of a query reads it
    std::mutex deleted_mu;
    std::vector<int32_t> h_java_hash;
    bool decimal_ids = false;  // ids are Long.toString(handle): String.hashCode computed in-kernel

    // plaintext store (test / bench harness)
    void* d_store = nullptr;
    bool store_owned = false;        // false: rows attached from caller-owned device memory (fspann_store_attach_dev)
    int store_dtype = FSPANN_F32;
    int64_t store_n = 0;

    // scratch arenas (grown on demand, reused across calls)
    fspann::DevBuf ws_route;   // global hash/sort fallback for the route kernel
    fspann::DevBuf ws_probe;   // probe lists handed from route_probe_kernel to route_select_kernel
    fspann::DevBuf ws_refine;  // per-chunk partial top-k
    fspann::DevBuf ws_gt;      // [query chunk][n] fp64 distance matrix of fspann_groundtruth_dev
    fspann::DevBuf ws_tickfix; // arenas of the full select run by the refine role of a tick for PENDING queries
    static constexpr int kFixSlots = 8;
    void* d_fixparams = nullptr;         // kFixSlots RouteParams blocks in device memory (parameters of that redo) ...
    std::vector<unsigned char> h_fixparams;   // ... and what each slot holds
    unsigned fix_valid = 0;
    int fix_next = 0;
    fspann::DevBuf ws_io[8];   // staging for the host-pointer entry points
    void* h_pin = nullptr;     // pinned host block (kPinBytes) for the small transfers of the host-pointer entry points: one H2D and one D2H
    void* d_pin = nullptr;     // the same block as the device sees it (mapped host memory): kernels of the tiniest calls read and write it directly
    void* h_rows = nullptr;    // fspann_host_buffer: pinned host memory the adapter packs decrypted rows into (grown on demand)
    size_t h_rows_bytes = 0;
                               //   per call instead of one synchronous pageable copy per argument (a per-query caller pays each of them)
    // transient, set by fspann_tick_dev around a refine-only tick: device-resident RouteParams of the batch's hand-over buffer
    // (the streaming scan then finishes PENDING queries itself), the LDS its full select needs, and whether a launch took it
    const void* refine_fix_dev = nullptr;
    size_t refine_fix_lds = 0;
    bool refine_fix_used = false;

    // fspann_ctx_clone: a clone reads its parent's GFunctions, frozen index, id metadata and store in place (no second copy in
    // HBM, and ONE working set in the caches however many contexts serve it); it owns its stream and work areas.
    fspann_ctx* share_parent = nullptr;  // non-null: the index arrays above belong to that context
    std::atomic<int> share_children{0};  // clones alive; the shared state may not change while > 0 (changed under the owner's `mu`)
    bool zombie = false;                 // destroyed by its owner while clones were alive: freed with the last clone
    std::atomic<int> comm_refs{0};       // fspann_comm objects that hold this context (it must outlive them)
    std::atomic<bool> destroy_deferred{false};   // fspann_ctx_destroy came while a communicator still held the context: the last fspann_comm_destroy finishes it
    // incremental Setup (fspann_build_begin / _append / _finish): codes of the rows appended so far stay in HBM
    int64_t bld_n = 0, bld_done = -1;    // bld_done < 0: no build in progress
    fspann::DevBuf bld_codes;
};

namespace fspann {

inline int ensure(fspann_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return FSPANN_OK;
    if (b.p) {
        FSP_HIP(hipStreamSynchronize(c->stream));
        FSP_HIP(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    FSP_HIP(hipMalloc(&b.p, want));
    b.bytes = want;
    b.gen++;
    return FSPANN_OK;
}

// java.util.HashMap.tableSizeFor
inline int table_size_for(int cap) {
    uint32_t c = static_cast<uint32_t>(cap - 1);
    int nlz = (c == 0) ? 32 : __builtin_clz(c);
    int32_t n = static_cast<int32_t>(0xFFFFFFFFu >> (nlz & 31));
    if (n < 0) return 1;
    if (n >= (1 << 30)) return 1 << 30;
    return n + 1;
}

// String.hashCode of Long.toString(v) (ids are decimal ordinals, ForwardSecureANNSystem.java:515)
inline int32_t decimal_string_hash(int64_t v) {
    char buf[24];
    int n = snprintf(buf, sizeof(buf), "%lld", static_cast<long long>(v));
    uint32_t h = 0;
    for (int i = 0; i < n; i++) h = 31u * h + static_cast<unsigned char>(buf[i]);
    return static_cast<int32_t>(h);
}

}  // namespace fspann
