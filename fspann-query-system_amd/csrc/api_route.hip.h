// api_route.hip.h — Route entry points: the plan of a Route call, fspann_route[_dev], the select switch and its counters (PIS:592-715, QSI:169-214)
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

namespace {

struct RoutePlan {
    int P, S, S_shift, max_tuples, maxcand, ht_size, ht_shift, sort_cap, nbins, need_cap, lds_mode;
    size_t lds_bytes, arena_bytes;
    int grid;
    int threads;
    int64_t g_sort_stride;
    // bounded select (route_lazy.hip.h)
    int lds_sort_words;
    bool long_lists;
    int lazy, lazy_cap, lz_ht_size, lz_grid, lz_entries;
    bool bincheck;                 // the bounded select runs its exact treeify check (bin16)
    int slice_bits, slice_ht;      // sliced hash build of the full select in global-arena mode (0 / 0: off)
    size_t lz_lds_bytes, small_bytes;
};

int plan_route(fspann_ctx* c, int probe_override, int64_t nq, int32_t limit, RoutePlan& pl, bool want_counters = true, bool for_tick = false) {
    pl.P = effective_probes(c, probe_override);
    pl.S = c->cfg.block_size;
    pl.S_shift = ((pl.S & (pl.S - 1)) == 0) ? __builtin_ctz(pl.S) : -1;
    const int64_t mt = static_cast<int64_t>(c->TD) * pl.P * pl.S;
    if (mt > (1LL << kSeqBits)) return fail(FSPANN_E_RANGE, "T*D*probes*blockSize = %lld exceeds 2^%d tuple slots", (long long)mt, kSeqBits);
    pl.max_tuples = static_cast<int>(mt);
    pl.need_cap = (mt >= c->hard_cap) ? 1 : 0;
    pl.maxcand = static_cast<int>(std::min<int64_t>(mt, static_cast<int64_t>(c->hard_cap) - 1 + pl.S));
    // B1 inserts every tuple (also those behind a HARD_CAP cut): size for max_tuples, load factor <= 0.5
    {   // load factor: <= 0.8 by default (2 workgroups per CU at BASELINE config #2); knob_ht_x4 -> <= 0.5
        const int64_t want = c->knob_ht_x4 ? static_cast<int64_t>(pl.max_tuples) * 2 : static_cast<int64_t>(pl.max_tuples) + pl.max_tuples / 4;
        pl.ht_size = std::max(64, next_pow2(want));
    }
    pl.ht_shift = 32 - __builtin_ctz(pl.ht_size);
    pl.nbins = c->bits + 1;
    // threads per workgroup of the full select: 512 for the short lists of the headline shape; a long list (thousands of entries, one
    // workgroup per CU) is a string of latency-bound passes over the tuples, where 1024 threads simply halve the trips
    // (1024 queries at SIFT_P10_HIGH: 5.6 -> 4.6 ms, SIFT_P4_FAST: 643 -> 516 us)
    pl.threads = (c->knob_threads == 1024 || c->knob_threads == 512) ? c->knob_threads
                 : (std::min<int64_t>(limit, std::min<int64_t>(mt, static_cast<int64_t>(c->hard_cap) - 1 + pl.S)) > kRankSortMax - 128 ? 1024 : 512);
    const int full_sort = next_pow2(std::max(pl.maxcand, 1));
    pl.sort_cap = std::min(full_sort, 1024);
    const size_t TP = static_cast<size_t>(c->TD) * pl.P;
    const size_t small = static_cast<size_t>(c->TD) * 8 + TP * 16 + 4096 + kDupListMax * 4 + TP * 4 + static_cast<size_t>(c->TD) * 8 + 64;
    auto arena = [&](int sort_cap) {
        return static_cast<size_t>(sort_cap) * 8 + static_cast<size_t>(pl.ht_size) * 4 + static_cast<size_t>(pl.max_tuples) * 4 +
               ((static_cast<size_t>(pl.max_tuples) * 2 + 15) & ~size_t(15));
    };
    pl.small_bytes = small;
    const size_t lds_cap = (c->knob_route_lds_kb > 0) ? std::min<size_t>(c->lds_limit, static_cast<size_t>(c->knob_route_lds_kb) * 1024) : static_cast<size_t>(c->lds_limit);
    const size_t budget = lds_cap - 1024;  // static __shared__ + margin
    if (small + 8192 > budget) return fail(FSPANN_E_RANGE, "route: T*D*probes = %zu probe slots do not fit in LDS", TP);
    pl.arena_bytes = (arena(pl.sort_cap) + 255) & ~size_t(255);
    pl.lds_mode = (pl.arena_bytes + small <= budget) ? 1 : 0;
    if (pl.lds_mode && limit > kRankSortMax) {
        // long result lists: grow the LDS sort buffer while it fits (avoids the global sort fallback)
        while (pl.sort_cap < full_sort && ((arena(pl.sort_cap * 2) + 255) & ~size_t(255)) + small <= budget) pl.sort_cap *= 2;
        pl.arena_bytes = (arena(pl.sort_cap) + 255) & ~size_t(255);
    }
    // long lists are ordered score group by score group in LDS (route.hip.h, phase C): in LDS mode the hash table's space is
    // reused, in global mode 64 KB behind the small arrays are reserved for it
    pl.lds_sort_words = 0;
    pl.long_lists = std::min<int64_t>(limit, pl.maxcand) > kRankSortMax - 128;
    if (!pl.lds_mode && small + 65536 + 64 <= budget) {
        // room for every sub-key of the longest possible list + eight wave slices + the cursors when the budget allows (one workgroup
        // per CU in this mode anyway), 64 KB otherwise (the sub-keys then go through global memory)
        const size_t words_max = (budget - small - 64) / 4;
        size_t want = static_cast<size_t>(pl.maxcand) + 8 * 512 + 1024;
        if (static_cast<size_t>(pl.ht_size) <= words_max) want = std::max<size_t>(want, pl.ht_size);     // ... and the hash table itself, if it fits (route.hip.h)
        pl.lds_sort_words = static_cast<int>(std::min(words_max, std::max<size_t>(16384, want)));
    }
    // Global-arena mode: the hash is built slice by slice of the id space in that LDS region (route.hip.h, B1): as many slices as keep
    // a slice's table at most ~5/8 full (SIFT_P10_HIGH: 35 840 tuples, two slices of a 32 768-slot table; SIFT_P4_FAST: one)
    pl.slice_bits = 0; pl.slice_ht = 0;
    if (!pl.lds_mode && pl.lds_sort_words >= 4096 && pl.lds_sort_words < pl.ht_size && c->knob_slice) {   // (a table that fits the region is built there as it is)
        int hs = 4096;
        while (hs * 2 <= pl.lds_sort_words) hs <<= 1;
        int kb = 0;
        while (kb < 3 && (static_cast<int64_t>(pl.max_tuples) >> kb) * 8 > static_cast<int64_t>(hs) * 5) kb++;
        if ((static_cast<int64_t>(pl.max_tuples) >> kb) * 8 <= static_cast<int64_t>(hs) * 5 || kb > 0) { pl.slice_bits = kb; pl.slice_ht = hs; }
    }
    if (pl.slice_ht > 0) pl.arena_bytes = (arena(pl.sort_cap) + static_cast<size_t>(pl.max_tuples) * 4 + 255) & ~size_t(255);   // + fseq (route.hip.h)
    pl.lds_bytes = pl.lds_mode ? pl.arena_bytes + small : small + static_cast<size_t>(pl.lds_sort_words) * 4 + 16;
    const int per_cu = std::max<int>(1, static_cast<int>(static_cast<size_t>(c->lds_limit) / (pl.lds_bytes + 512)));
    int wgs_per_cu = std::min(per_cu, 4);
    if (!pl.lds_mode) {
        // every workgroup owns an arena slice (hash table, tuples) in global memory that it hits at random: keep the slices of
        // all resident workgroups inside the 256 MiB Infinity Cache (1024 workgroups x 470 KB at SIFT_P10_HIGH thrashed HBM)
        wgs_per_cu = (pl.arena_bytes * static_cast<size_t>(c->num_cus) * 2 <= (size_t(128) << 20)) ? std::min(wgs_per_cu, 2) : 1;
    }
    if (c->knob_route_wgs > 0) wgs_per_cu = std::min(per_cu, c->knob_route_wgs);
    pl.grid = static_cast<int>(std::min<int64_t>(nq, static_cast<int64_t>(c->num_cus) * wgs_per_cu));
    pl.g_sort_stride = (pl.sort_cap < full_sort) ? full_sort : 0;
    // ---- bounded select: legal when the first `limit` entries do not depend on how many ids exist ----------------
    pl.lazy = 0;
    const bool cap_fixed = java_final_cap_host(c->cap0, mt) == c->cap0;       // HashMap never resizes
    // ... and a treeified bin of bestScore (nine distinct ids in one bin, PIS:619 + HashMap.TREEIFY_THRESHOLD) cannot go unnoticed: the
    // bounded select loads only the partitions that decide the first `limit` entries, so a bin that fills through ids it never loads
    // is seen only by its exact check over bin16 (route_lazy.hip.h, step 0).  Opaque ids (caller-supplied String.hashCode) always run
    // it; decimal ordinals — whose hashCodes spread ~4 500 ids over 32 768 bins like random draws, nine in one bin ~2e-9 per query —
    // run it when FSPANN_ROUTE_BINCHECK=1 asks for it.  No bin16 where it is needed: the full select.
    pl.bincheck = (c->knob_bincheck == 1) || (c->knob_bincheck < 0 && !c->decimal_ids);
    const bool check_ok = !pl.bincheck || c->d_bin16 != nullptr;
    const bool legal = c->d_inv && c->d_ids_bk && c->bk_epoch == c->meta_epoch && !pl.need_cap && cap_fixed && !want_counters && limit <= (for_tick ? 512 : 1024) && pl.lds_mode && check_ok;
    if (legal && c->route_mode != 1 && (c->route_mode == 2 || static_cast<int64_t>(limit) * 4 <= mt)) {
        const int cap_env = c->knob_lazy_cap;   // tests: distinct ids one query may hold before it is handed back
        // size class: 512 entries (19.6 KB, 6 workgroups per CU) when limit <= 256 and the probe's scratch fits the smaller key
        // array; the tick kernel keeps the large class (its redo runs the full select over the same LDS)
        // (the 512-entry kernel is built WITHOUT the exact treeify check: its loads live across the ordering step and would spill at 80
        // registers — and any scratch use costs every dispatch of the stream; a checked Route takes the 1024-entry class, 4 per CU)
        const bool small_cls = c->knob_lazy_small && !for_tick && !pl.bincheck && limit <= 256 && (kLzThreads / 16) * (2 * pl.P - 1) * 12 <= 512 * 4;
        const int kent = small_cls ? 512 : (limit <= 512 ? kLzEntriesMax : 2048);
        const size_t lds = lz_lds_bytes(kent, c->TD, pl.P);
        if (TP < 32768 && lds <= budget && (small <= lds || !for_tick)) {   // small <= lds: a handed-over query runs the full select over this LDS
            pl.lazy = 1;
            pl.lz_entries = kent;
            pl.lazy_cap = (cap_env > 0) ? std::min(cap_env, kent) : kent;
            pl.lz_ht_size = lz_ht_size(kent);
            pl.lz_lds_bytes = lds;
            const int lz_per_cu = std::max<int>(1, std::min<int>(8, static_cast<int>(static_cast<size_t>(c->lds_limit) / (lds + 256))));
            pl.lz_grid = static_cast<int>(std::min<int64_t>(nq, static_cast<int64_t>(c->num_cus) * lz_per_cu));
        }
    }
    return FSPANN_OK;
}


}  // namespace

extern "C" {

// ---- route --------------------------------------------------------------------------
int fspann_effective_probes(fspann_ctx* c, int probe_override) { return c ? effective_probes(c, probe_override) : FSPANN_E_NULL; }

int64_t fspann_route_max_candidates(fspann_ctx* c, int probe_override) {
    if (!c) return FSPANN_E_NULL;
    const int64_t mt = static_cast<int64_t>(c->TD) * effective_probes(c, probe_override) * c->cfg.block_size;
    return std::min<int64_t>(mt, static_cast<int64_t>(c->hard_cap) - 1 + c->cfg.block_size);
}


}  // extern "C"


namespace {

// Argument checks + plan + kernel parameters of one Route call (shared by fspann_route_dev and fspann_tick_dev).
int prepare_route(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit, int64_t cap, int32_t* ids_dev,
                  int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev, int32_t* raw_seen_dev, RoutePlan* plan_out, RouteParams* prm_out,
                  bool* fused_out, bool for_tick = false, bool launches_lazy = true) {
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");  // PIS:594
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (!codes_dev) return fail(FSPANN_E_STATE, "MSANNP violation: QueryToken missing BitSet codes");  // PIS:602
    if (!ids_dev || !count_dev) return fail(FSPANN_E_NULL, "output buffer is null");
    if (limit <= 0) return fail(FSPANN_E_ARG, "limit must be > 0");
    RoutePlan pl;
    int rc = plan_route(c, probe_override, nq, limit, pl, kept_dev != nullptr || raw_seen_dev != nullptr, for_tick);
    if (rc) return rc;
    const int64_t need = std::min<int64_t>(limit, pl.maxcand);
    if (cap < need) return fail(FSPANN_E_RANGE, "cap %lld < min(limit, worst case) = %lld", (long long)cap, (long long)need);
    // global arenas of the full select: per-workgroup scratch when it does not fit LDS, the sort buffer of long lists
    const size_t ar_g = pl.lds_mode ? 0 : static_cast<size_t>(pl.grid) * pl.arena_bytes;
    const size_t so_g = static_cast<size_t>(pl.grid) * pl.g_sort_stride * 8;
    const size_t su_g = pl.long_lists ? static_cast<size_t>(pl.grid) * static_cast<size_t>(pl.maxcand) * 4 : 0;     // sub-keys grouped by score
    if (ar_g + so_g + su_g && (rc = ensure(c, c->ws_route, ar_g + so_g + su_g + 1024))) return rc;
    RouteParams p{};
    p.codes = codes_dev; p.tables = c->d_tables; p.recs = c->d_recs; p.rec_words = c->rec_words; p.ids = c->d_ids;
    p.dir = c->knob_probe_dir ? c->d_dir : nullptr; p.dir_bits = c->dir_bits;
    p.java_hash = c->d_java_hash; p.deleted_bits = index_owner(c)->d_deleted_bits.load(std::memory_order_acquire);
    p.nq = nq; p.TD = c->TD; p.W = c->W; p.P = pl.P; p.S = pl.S; p.S_shift = pl.S_shift;
    p.hard_cap = c->hard_cap; p.cap0 = c->cap0; p.limit = limit; p.need_cap = pl.need_cap; p.nbins = pl.nbins;
    p.seq_bits = 1; while ((1 << p.seq_bits) < pl.max_tuples) p.seq_bits++;
    p.ht_size = pl.ht_size; p.ht_shift = pl.ht_shift; p.sort_cap = pl.sort_cap; p.max_tuples = pl.max_tuples;
    p.g_sort = so_g ? static_cast<uint64_t*>(c->ws_route.p) : nullptr;
    p.g_sort_stride = pl.g_sort_stride;
    p.g_scratch = ar_g ? static_cast<unsigned char*>(c->ws_route.p) + ((so_g + 255) & ~size_t(255)) : nullptr;
    p.g_stride = static_cast<int64_t>(pl.arena_bytes);
    p.g_sub = su_g ? reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->ws_route.p) + ((so_g + 255) & ~size_t(255)) + ((ar_g + 255) & ~size_t(255))) : nullptr;
    p.g_sub_stride = pl.maxcand;
    p.lds_sort_words = pl.lds_sort_words;
    p.slice_bits = pl.slice_bits; p.slice_ht = pl.slice_ht; p.dev_flags = c->knob_devflags;
    p.wave_sort = c->knob_wave_sort > 0 ? 1 : 0;
    if (c->knob_wave_sort < 0) p.g_sub = nullptr;
    p.dbg = c->dbg_route;
    p.unmodelled = c->d_unmodelled;
    p.decimal_ids = c->decimal_ids ? 1 : 0;
    p.out_cap = cap; p.out_ids = ids_dev; p.out_score = score_dev; p.out_count = count_dev; p.out_kept = kept_dev; p.out_raw = raw_seen_dev;
    // probe lists in global memory: route_probe_kernel's output, and where the bounded select puts a query it hands over
    const size_t TPn = static_cast<size_t>(c->TD) * pl.P;
    const size_t probe_bytes = static_cast<size_t>(nq) * TPn * 16, np_bytes = static_cast<size_t>(nq) * c->TD * 4;
    if ((rc = ensure(c, c->ws_probe, probe_bytes + np_bytes + 256))) return rc;
    p.probe_g = static_cast<int4*>(c->ws_probe.p);
    p.nprobe_g = reinterpret_cast<int32_t*>(static_cast<char*>(c->ws_probe.p) + ((probe_bytes + 255) & ~size_t(255)));
    bool fused = false;
    if (pl.lazy) {
        if ((rc = ensure(c, c->ws_ovf, static_cast<size_t>(nq) * 4 + 256))) return rc;
        if (c->ovf_gen_seen != c->ws_ovf.gen) {    // fresh allocation: both overflow counters start at zero
            FSP_HIP(hipMemsetAsync(c->ws_ovf.p, 0, 256, c->stream));
            c->ovf_gen_seen = c->ws_ovf.gen;
        }
        // The overflow counters alternate: the bounded select of THIS call counts in one and zeroes the other for the next
        // call.  Only a call that really launches a bounded select may take its turn — parameters prepared for a redo
        // (tick: the full select of PENDING queries) leave the turn alone, or the next call would start on a counter
        // nobody zeroed and hand its full select a list with another batch's queries in front.
        if (launches_lazy) c->ovf_flip ^= 1;
        p.bin16 = pl.bincheck ? c->d_bin16 : nullptr; p.bin16_shift = c->bin16_shift;
        p.inv = c->d_inv; p.ids_bk = c->d_ids_bk; p.n_ids = c->n_ids; p.lazy_cap = pl.lazy_cap; p.lz_ht_size = pl.lz_ht_size;
        p.lz_ht_shift = 32 - __builtin_ctz(pl.lz_ht_size);
        p.ovf_count = static_cast<int32_t*>(c->ws_ovf.p) + 16 * c->ovf_flip;          // this call's counter ...
        p.ovf_next = static_cast<int32_t*>(c->ws_ovf.p) + 16 * (c->ovf_flip ^ 1);      // ... the next call's is zeroed meanwhile
        p.ovf_list = static_cast<int32_t*>(c->ws_ovf.p) + 64;
        // the probe runs inside the bounded select when its scratch fits the arrays it borrows there
        fused = c->knob_fused_probe && (kLzThreads / 16) * (2 * pl.P - 1) * 12 <= 4096 && c->TD <= 512;
        p.probe_G = fused ? 16 : 0;
    }
    *plan_out = pl;
    *prm_out = p;
    *fused_out = fused;
    return FSPANN_OK;
}

// kernel 1 of the unfused route: search + probe order, one lane group per (query, table)
int launch_route_probe(fspann_ctx* c, const RouteParams& p, const RoutePlan& pl) {
    int G = 64;
    while (G > 2 && G / 2 >= 2 * pl.P - 1 && G / 2 >= 16) G >>= 1;  // >= 16 lanes per table: 3-4 search rounds
    const int gpb = kProbeThreads / G;
    const int64_t nitems = p.nq * c->TD;
    const unsigned grid1 = static_cast<unsigned>((nitems + gpb - 1) / gpb);
    const size_t lds1 = static_cast<size_t>(gpb) * (2 * pl.P - 1) * 12;
    hipLaunchKernelGGL(route_probe_kernel, dim3(grid1), dim3(kProbeThreads), lds1, c->stream, p, p.probe_g, p.nprobe_g, G);
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
}

// the full select (every query, or — qcount / qlist set — the queries the bounded select handed over)
int launch_full_select(fspann_ctx* c, const RoutePlan& pl, const RouteParams& p) {
#define FSP_LAUNCH_SEL(LDS, THR)                                                                                         \
    do {                                                                                                                 \
        auto kern = route_select_kernel<LDS, THR>;                                                                       \
        const unsigned abit = 1u << ((LDS ? 0 : 2) + (THR == 1024 ? 1 : 0));                                             \
        if (!(c->attr_mask & abit)) {   /* once per context: the attribute is the ceiling, not the launch size */        \
            FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        159 * 1024));                                                                    \
            c->attr_mask |= abit;                                                                                        \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(THR), pl.lds_bytes, c->stream, p, p.probe_g, p.nprobe_g);           \
    } while (0)
    if (pl.lds_mode) { if (pl.threads == 1024) FSP_LAUNCH_SEL(true, 1024); else FSP_LAUNCH_SEL(true, 512); }
    else { if (pl.threads == 1024) FSP_LAUNCH_SEL(false, 1024); else FSP_LAUNCH_SEL(false, 512); }
#undef FSP_LAUNCH_SEL
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
}

// fspann_route_dev.  deferred (optional): when the bounded select runs, the launch that finishes the queries it hands over — none,
// normally — is NOT issued; its plan and parameters are returned instead and the caller issues it (launch_full_select) once it has
// seen a PENDING count (fspann_route's calls of a handful of queries: one launch less in front of the synchronisation).
int route_dev_impl(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit,
                   int64_t cap, int32_t* ids_dev, int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev,
                   int32_t* raw_seen_dev, RoutePlan* deferred_pl = nullptr, RouteParams* deferred_p = nullptr, bool* deferred = nullptr) {
    if (deferred) *deferred = false;
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");  // PIS:594
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    RoutePlan pl;
    RouteParams p{};
    bool fused = false;
    int rc = prepare_route(c, nq, codes_dev, probe_override, limit, cap, ids_dev, score_dev, count_dev, kept_dev, raw_seen_dev, &pl, &p, &fused);
    if (rc) return rc;
    if (!fused && (rc = launch_route_probe(c, p, pl))) return rc;
    c->last_route_lazy = pl.lazy;
    if (pl.lazy) {
        if (pl.lz_entries == 512) {
            // (the shape of BASELINE configs #2 / #3 — 16 tables x 5 probes, blocks of 64 — has a build with the shape as constants)
            if (c->knob_shape_spec && c->TD == 16 && pl.P == 5 && pl.S == 64 && c->W == 1 && c->rec_words == 4 && (p.probe_G == 16 || p.probe_G == 0))
                hipLaunchKernelGGL((route_select_lazy_kernel<kLzThreads, 512, false, 16, 5>), dim3(pl.lz_grid), dim3(kLzThreads), pl.lz_lds_bytes, c->stream, p);
            else
                hipLaunchKernelGGL((route_select_lazy_kernel<kLzThreads, 512, false>), dim3(pl.lz_grid), dim3(kLzThreads), pl.lz_lds_bytes, c->stream, p);
        } else if (pl.lz_entries == 2048) {
            auto lk = route_select_lazy_kernel<kLzThreads, 2048, true>;
            if (!(c->attr_mask & 1024u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 1024u;
            }
            hipLaunchKernelGGL(lk, dim3(pl.lz_grid), dim3(kLzThreads), pl.lz_lds_bytes, c->stream, p);
        } else {
            auto lk = route_select_lazy_kernel<kLzThreads, kLzEntriesMax, true>;
            if (!(c->attr_mask & 16u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 16u;
            }
            hipLaunchKernelGGL(lk, dim3(pl.lz_grid), dim3(kLzThreads), pl.lz_lds_bytes, c->stream, p);
        }
        FSP_HIP(hipGetLastError());
        // queries the bounded select handed over (none, normally): the full select over the overflow list
        p.qcount = p.ovf_count; p.qlist = p.ovf_list;
        pl.grid = std::min(pl.grid, 32);    // normally nothing to do: keep the launch small
        if (deferred_pl && deferred_p && deferred) { *deferred_pl = pl; *deferred_p = p; *deferred = true; return FSPANN_OK; }
    }
    return launch_full_select(c, pl, p);
}

}  // namespace

extern "C" {

int fspann_route_dev(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit,
                     int64_t cap, int32_t* ids_dev, int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev,
                     int32_t* raw_seen_dev) {
    CHECK_CTX(c);
    return route_dev_impl(c, nq, codes_dev, probe_override, limit, cap, ids_dev, score_dev, count_dev, kept_dev, raw_seen_dev);
}

int fspann_route(fspann_ctx* c, int64_t nq, const uint64_t* codes, int probe_override, int32_t limit, int64_t cap,
                 int32_t* ids, int32_t* score, int32_t* count, int32_t* kept, int32_t* raw_seen) {
    CHECK_CTX(c);
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    if (!codes) return fail(FSPANN_E_STATE, "MSANNP violation: QueryToken missing BitSet codes");
    if (!ids || !count) return fail(FSPANN_E_NULL, "output buffer is null");
    const size_t cb = static_cast<size_t>(nq) * c->TD * c->W * 8;
    const size_t ob = static_cast<size_t>(nq) * cap * 4;
    int rc;
    // Small calls (QueryService.search is one token per call, ForwardSecureANNSystem.java:636): codes go up and lists, scores and
    // counts come down through ONE pinned block — one asynchronous copy each way and one synchronisation, where the general path
    // below pays a synchronous pageable copy per argument and a separate look at the counts (bench.py operator_surface).
    const size_t ob_a = (ob + 15) & ~size_t(15), cnt_a = (static_cast<size_t>(nq) * 12 + 15) & ~size_t(15);
    if (std::max(cb, 2 * ob_a + cnt_a) <= kPinBytes && pin_block(c)) {
        unsigned char* hp = static_cast<unsigned char*>(c->h_pin);
        const size_t cb_a = (cb + 15) & ~size_t(15);
        // a handful of queries: the kernels read the codes from the mapped pinned block and write lists and counts into it (no copy
        // commands: zero_copy_ok); else one copy each way through device buffers
        const bool zc = cb_a + 2 * ob_a + cnt_a <= kPinBytes && zero_copy_ok(c, nq);
        if (!zc) {
            if ((rc = ensure(c, c->ws_io[0], cb))) return rc;
            if ((rc = ensure(c, c->ws_io[1], 2 * ob_a + cnt_a))) return rc;       // ids | scores | count, kept, rawSeen: one block
        }
        unsigned char* hres = zc ? hp + cb_a : hp;                                  // where the results end up on the host
        unsigned char* dv = zc ? static_cast<unsigned char*>(c->d_pin) + cb_a : static_cast<unsigned char*>(c->ws_io[1].p);
        const uint64_t* codes_d = zc ? static_cast<const uint64_t*>(c->d_pin) : static_cast<const uint64_t*>(c->ws_io[0].p);
        int32_t* ids_d = reinterpret_cast<int32_t*>(dv), *sc_d = reinterpret_cast<int32_t*>(dv + ob_a), *cnt_d = reinterpret_cast<int32_t*>(dv + 2 * ob_a);
        std::memcpy(hp, codes, cb);
        if (!zc) FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, hp, cb, hipMemcpyHostToDevice, c->stream));
        RoutePlan def_pl;
        RouteParams def_p{};
        bool deferred = false;
        rc = route_dev_impl(c, nq, codes_d, probe_override, limit, cap, ids_d, sc_d, cnt_d,
                            kept ? cnt_d + nq : nullptr, raw_seen ? cnt_d + 2 * nq : nullptr, zc ? &def_pl : nullptr, zc ? &def_p : nullptr, zc ? &deferred : nullptr);
        if (rc) return rc;
        if (deferred) {
            // zero-copy call, bounded select: the counts are readable right after the synchronisation — the launch for handed-over queries
            // is issued only if one of them says PENDING (it was ~5 us of every call, for a list that is almost always empty)
            FSP_HIP(hipStreamSynchronize(c->stream));
            const int32_t* cnt_p = reinterpret_cast<const int32_t*>(hres + 2 * ob_a);
            bool pending = false;
            for (int64_t i = 0; i < nq; i++) pending = pending || cnt_p[i] == kRoutePending;
            if (pending && (rc = launch_full_select(c, def_pl, def_p))) return rc;
        }
        for (int pass = 0; pass < 2; pass++) {
            if (!zc) FSP_HIP(hipMemcpyAsync(hp, dv, 2 * ob_a + cnt_a, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipStreamSynchronize(c->stream));
            const int32_t* cnt_h = reinterpret_cast<const int32_t*>(hres + 2 * ob_a);
            bool flagged = false;
            for (int64_t i = 0; i < nq; i++) flagged = flagged || cnt_h[i] < 0;
            if (!flagged || pass == 1) break;
            // (rare) a bestScore map treeified a bin: finished by the literal JDK model on the host, then fetched again
            rc = guarded([&]() -> int {
                return resolve_unmodelled(c, nq, codes_d, probe_override, limit, cap, ids_d, sc_d, cnt_d,
                                          kept ? cnt_d + nq : nullptr, raw_seen ? cnt_d + 2 * nq : nullptr, nullptr, nullptr);
            });
            if (rc) return rc;
        }
        const int32_t* cnt_h = reinterpret_cast<const int32_t*>(hres + 2 * ob_a);
        std::memcpy(ids, hres, ob);
        if (score) std::memcpy(score, hres + ob_a, ob);
        std::memcpy(count, cnt_h, static_cast<size_t>(nq) * 4);
        if (kept) std::memcpy(kept, cnt_h + nq, static_cast<size_t>(nq) * 4);
        if (raw_seen) std::memcpy(raw_seen, cnt_h + 2 * nq, static_cast<size_t>(nq) * 4);
        for (int64_t i = 0; i < nq; i++)
            if (count[i] < 0)
                return fail(FSPANN_E_STATE, "query %lld: a treeified HashMap bin of bestScore orders different ids with equal String.hashCode by "
                            "String.compareTo, which the library cannot evaluate for non-decimal ids: not modelled, its count is -1", (long long)i);
        return FSPANN_OK;
    }
    if ((rc = ensure(c, c->ws_io[0], cb))) return rc;
    if ((rc = ensure(c, c->ws_io[1], ob))) return rc;
    if ((rc = ensure(c, c->ws_io[2], ob))) return rc;
    if ((rc = ensure(c, c->ws_io[3], static_cast<size_t>(nq) * 12))) return rc;
    int32_t* cnt = static_cast<int32_t*>(c->ws_io[3].p);
    FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, codes, cb, hipMemcpyHostToDevice, c->stream));
    rc = fspann_route_dev(c, nq, static_cast<const uint64_t*>(c->ws_io[0].p), probe_override, limit, cap,
                          static_cast<int32_t*>(c->ws_io[1].p), static_cast<int32_t*>(c->ws_io[2].p), cnt, kept ? cnt + nq : nullptr,
                          raw_seen ? cnt + 2 * nq : nullptr);
    if (rc) return rc;
    // a query whose HashMap would have treeified a bin (count = -1) is finished by the literal JDK model on the host (rare path)
    rc = guarded([&]() -> int {
        return resolve_unmodelled(c, nq, static_cast<const uint64_t*>(c->ws_io[0].p), probe_override, limit, cap, static_cast<int32_t*>(c->ws_io[1].p),
                                  static_cast<int32_t*>(c->ws_io[2].p), cnt, kept ? cnt + nq : nullptr, raw_seen ? cnt + 2 * nq : nullptr, nullptr, nullptr);
    });
    if (rc) return rc;
    FSP_HIP(hipMemcpyAsync(ids, c->ws_io[1].p, ob, hipMemcpyDeviceToHost, c->stream));
    if (score) FSP_HIP(hipMemcpyAsync(score, c->ws_io[2].p, ob, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(count, cnt, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, c->stream));
    if (kept) FSP_HIP(hipMemcpyAsync(kept, cnt + nq, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, c->stream));
    if (raw_seen) FSP_HIP(hipMemcpyAsync(raw_seen, cnt + 2 * nq, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    // all outputs are in place; what is still flagged could not be finished by the host model either: a treeified bin holds
    // different ids with EQUAL String.hashCode and the ids are not decimal ordinals, so their String.compareTo order is unknown here
    for (int64_t i = 0; i < nq; i++)
        if (count[i] < 0)
            return fail(FSPANN_E_STATE, "query %lld: a treeified HashMap bin of bestScore orders different ids with equal String.hashCode by "
                        "String.compareTo, which the library cannot evaluate for non-decimal ids: not modelled, its count is -1", (long long)i);
    return FSPANN_OK;
}

// Route select path: 0 = auto, 1 = always the full select (route_select_kernel), 2 = the bounded select whenever its
// preconditions hold (route_lazy.hip.h).  All modes return identical lists.
int fspann_set_route_mode(fspann_ctx* c, int mode) {
    if (!c) return fail(FSPANN_E_NULL, "ctx is null");
    if (mode < 0 || mode > 2) return fail(FSPANN_E_ARG, "route mode must be 0, 1 or 2");
    c->route_mode = mode;
    return FSPANN_OK;
}
// Queries flagged "unmodelled" (a java.util.HashMap bin would have been treeified; their count is -1) by Route calls of
// this context since the last reset.  Synchronises the stream.
int fspann_unmodelled_queries(fspann_ctx* c, int64_t* total, int reset) {
    CHECK_CTX(c);
    FSP_HIP(hipStreamSynchronize(c->stream));
    int32_t v = 0;
    FSP_HIP(hipMemcpy(&v, c->d_unmodelled, 4, hipMemcpyDeviceToHost));
    if (total) *total = v;
    if (reset && v) FSP_HIP(hipMemset(c->d_unmodelled, 0, 4));
    return FSPANN_OK;
}
// Which select the last fspann_route[_dev] ran: *lazy = 1 for the bounded select; *overflowed = queries it handed back
// to the full select (synchronises the stream).
int fspann_last_route_info(fspann_ctx* c, int* lazy, int* overflowed) {
    CHECK_CTX(c);
    if (lazy) *lazy = c->last_route_lazy;
    if (overflowed) {
        *overflowed = 0;
        if (c->last_route_lazy && c->ws_ovf.p) {
            FSP_HIP(hipStreamSynchronize(c->stream));
            int32_t v = 0;
            FSP_HIP(hipMemcpy(&v, static_cast<int32_t*>(c->ws_ovf.p) + 16 * c->ovf_flip, 4, hipMemcpyDeviceToHost));
            *overflowed = v;
        }
    }
    return FSPANN_OK;
}
#ifdef FSPANN_DEBUG_STAMPS
// debug builds only (tools/route_stamps.py): per-block phase stamps of the route kernels (dev pointer to [grid][16] int64)
int fspann_debug_route_stamps(fspann_ctx* c, void* dev_ptr) {
    if (!c) return FSPANN_E_NULL;
    c->dbg_route = static_cast<long long*>(dev_ptr);
    return FSPANN_OK;
}
#endif


}  // extern "C"
