// api_setup.hip.h — context lifecycle and Setup by import: GFunctions, frozen index upload (probe directory, bounded-select arrays), id metadata, index file
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

namespace {

int upload_index(fspann_ctx* c) {
    const int TD = c->TD, W = c->W;
    for (int td = 0; td < TD; td++)
        if (!c->h_table_set[td]) return fail(FSPANN_E_STATE, "table %d was never set (fspann_set_index)", td);
    // Every handle of every table must lie in [0, n_ids) — the kernels index java_hash / deleted_bits / the store with it —
    // and occur at most once per table (a division's HashMap holds an id once, PIS:331-346; the select kernels rely on it).
    // Checked here rather than in fspann_set_index because the documented import order sets the tables before the id metadata.
    {
        std::atomic<int> bad_td{-1}, bad_kind{0};
        std::atomic<long long> bad_id{0};
        std::atomic<bool> oom{false};
        const int nthv = std::max(1, std::min<int>(TD, static_cast<int>(std::thread::hardware_concurrency())));
        std::vector<std::thread> thv;
        for (int w = 0; w < nthv; w++)
            thv.emplace_back([&, w] {
                try {
                    std::vector<uint64_t> seen(static_cast<size_t>((c->n_ids + 63) / 64));
                    for (int td = w; td < TD && bad_td.load() < 0; td += nthv) {
                        std::fill(seen.begin(), seen.end(), 0ull);
                        for (const int32_t id : c->h_ids[td]) {
                            int kind = 0;
                            if (id < 0 || id >= c->n_ids) kind = 1;
                            else if ((seen[static_cast<size_t>(id) >> 6] >> (id & 63)) & 1ull) kind = 2;
                            if (kind) { bad_td = td; bad_kind = kind; bad_id = id; return; }
                            seen[static_cast<size_t>(id) >> 6] |= 1ull << (id & 63);
                        }
                    }
                } catch (...) { oom = true; }
            });
        for (auto& t : thv) t.join();
        if (oom) return fail(FSPANN_E_NOMEM, "out of host memory");
        if (bad_td.load() >= 0)
            return bad_kind.load() == 1
                       ? fail(FSPANN_E_ARG, "table %d: id handle %lld out of range [0,%lld)", bad_td.load(), bad_id.load(), (long long)c->n_ids)
                       : fail(FSPANN_E_ARG, "table %d holds id handle %lld twice", bad_td.load(), bad_id.load());
    }
    c->h_tables.assign(TD, RouteTable{});
    int64_t parts = 0, offs = 0, ids = 0;
    for (int td = 0; td < TD; td++) {
        RouteTable& t = c->h_tables[td];
        t.part_base = parts;
        t.off_base = offs;
        t.ids_base = ids;
        t.nparts = static_cast<int32_t>(c->h_min[td].size());
        t.dir_base = 0;
        parts += t.nparts;
        offs += t.nparts + 1;
        ids += static_cast<int64_t>(c->h_ids[td].size());
    }
    c->total_parts = parts;
    c->total_ids = ids;
    free_devt(c->d_tables); free_devt(c->d_recs); free_devt(c->d_ids); free_devt(c->d_dir);
    // Radix directory of the probe (route.hip.h, route_probe_table): for every table and every value p of the key's top
    // dir_bits bits, the first partition with maxKey >= p << s and the first with minKey >= p << s.  It needs what the
    // reference's own binary search needs, key ranges in ascending order; an imported index without that keeps the plain search.
    std::vector<int2> dir;
    c->dir_bits = 0;
    {
        int maxp = 0;
        bool mono = true;
        for (int td = 0; td < TD && mono; td++) {
            const auto& mn = c->h_min[td]; const auto& mx = c->h_max[td];
            maxp = std::max<int>(maxp, static_cast<int>(mn.size()));
            for (size_t i = 0; i < mn.size() && mono; i++)
                mono = mn[i] >= 0 && mn[i] <= mx[i] && (i == 0 || (mn[i] >= mn[i - 1] && mx[i] >= mx[i - 1]));
        }
        int bits = 1;
        while (bits < 16 && (4 << bits) < maxp) bits++;      // about four partitions per directory entry
        // ... and up to six bits more while the whole directory stays within 64 MB: keys are skewed (the most popular 12-bit prefix of
        // BASELINE config #2 covers 1 974 of 15 625 partitions), every extra bit halves the brackets the search starts from, and a
        // search round is a dependent load (step 44.9 -> 43.9 us at 18 bits = 33 MB; FSPANN_ROUTE_DIR_EXTRA_BITS overrides)
        int extra = c->knob_dir_extra_bits;
        if (extra == kDirBitsAuto) {
            extra = 0;
            while (extra < 6 && bits + extra + 1 <= 20 && static_cast<size_t>(TD) * ((size_t(1) << (bits + extra + 1)) + 1) * sizeof(int2) <= (size_t(64) << 20)) extra++;
        }
        bits = std::min(20, std::max(1, bits + extra));
        const size_t D = size_t(1) << bits;
        if (mono && maxp > 0 && static_cast<size_t>(TD) * (D + 1) < (size_t(1) << 30)) {
            dir.resize(static_cast<size_t>(TD) * (D + 1));
            const int sh = 63 - bits;
            for (int td = 0; td < TD; td++) {
                const auto& mn = c->h_min[td]; const auto& mx = c->h_max[td];
                const int np = static_cast<int>(mn.size());
                c->h_tables[td].dir_base = static_cast<int32_t>(static_cast<size_t>(td) * (D + 1));
                int2* dd = dir.data() + static_cast<size_t>(td) * (D + 1);
                int ia = 0, ie = 0;
                for (size_t pfx = 0; pfx < D; pfx++) {
                    const int64_t bound = static_cast<int64_t>(pfx) << sh;
                    while (ia < np && mx[ia] < bound) ia++;
                    while (ie < np && mn[ie] < bound) ie++;
                    dd[pfx] = make_int2(ia, ie);
                }
                dd[D] = make_int2(np, np);
            }
            c->dir_bits = bits;
        }
    }
    // One RECORD per partition with everything the probe reads about it — {minKey, maxKey, rep[W], id offset | size << 32},
    // padded to an even number of 8-byte words: the last rounds of the search, the gap rule and the Hamming round then touch
    // the same one or two cache lines instead of three arrays (the probe is bound by the latency of cold lines).
    const int rec_words = (3 + W + 1) & ~1;
    c->rec_words = rec_words;
    std::vector<int64_t> recs(static_cast<size_t>(std::max<int64_t>(parts, 1)) * rec_words, 0);
    std::vector<int32_t> idv(static_cast<size_t>(std::max<int64_t>(ids, 1)));
    for (int td = 0; td < TD; td++) {
        const RouteTable& t = c->h_tables[td];
        for (int p = 0; p < t.nparts; p++) {
            int64_t* r = recs.data() + static_cast<size_t>(t.part_base + p) * rec_words;
            r[0] = c->h_min[td][p];
            r[1] = c->h_max[td][p];
            for (int w = 0; w < W; w++) r[2 + w] = static_cast<int64_t>(c->h_rep[td][static_cast<size_t>(p) * W + w]);
            const uint64_t b0 = static_cast<uint32_t>(c->h_off[td][p]), sz = static_cast<uint32_t>(c->h_off[td][p + 1] - c->h_off[td][p]);
            r[2 + W] = static_cast<int64_t>(b0 | (sz << 32));
        }
        std::copy(c->h_ids[td].begin(), c->h_ids[td].end(), idv.begin() + t.ids_base);
    }
    FSP_HIP(hipMalloc(&c->d_tables, sizeof(RouteTable) * TD));
    FSP_HIP(hipMalloc(&c->d_recs, recs.size() * 8));
    FSP_HIP(hipMalloc(&c->d_ids, idv.size() * 4));
    FSP_HIP(hipMemcpy(c->d_tables, c->h_tables.data(), sizeof(RouteTable) * TD, hipMemcpyHostToDevice));
    FSP_HIP(hipMemcpy(c->d_recs, recs.data(), recs.size() * 8, hipMemcpyHostToDevice));
    FSP_HIP(hipMemcpy(c->d_ids, idv.data(), idv.size() * 4, hipMemcpyHostToDevice));
    if (!dir.empty()) {
        FSP_HIP(hipMalloc(&c->d_dir, dir.size() * sizeof(int2)));
        FSP_HIP(hipMemcpy(c->d_dir, dir.data(), dir.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    // For the bounded select (route_lazy.hip.h):
    //   inv[td][id]  position of id in table td's id list (a table holding an id twice cannot be inverted: select stays off)
    //   ids_bk       every partition's ids once more, as (id << 32 | bucket field at the initial HashMap capacity),
    //                sorted by bucket within the partition, so the ids with the smallest buckets are a prefix
    free_devt(c->d_inv); free_devt(c->d_ids_bk); free_devt(c->d_bin16);
    c->bk_epoch = -1;
    if (c->n_ids > 0 && static_cast<int64_t>(TD) * c->n_ids < (1LL << 33) && c->cap0 <= (1 << kBucketBits) && c->cfg.block_size <= 4096) {
        std::vector<int32_t> inv(static_cast<size_t>(TD) * static_cast<size_t>(c->n_ids), -1);
        std::vector<uint64_t> bk(static_cast<size_t>(std::max<int64_t>(ids, 1)));
        const int capbits = 31 - __builtin_clz(static_cast<unsigned>(c->cap0));
        const uint32_t bmask = static_cast<uint32_t>(c->cap0 - 1);
        const int bshift = kBucketBits - capbits;
        std::atomic<bool> ok{true};
        std::vector<std::thread> th;
        const int nth = std::max(1, std::min<int>(TD, static_cast<int>(std::thread::hardware_concurrency())));
        for (int w = 0; w < nth; w++)
            th.emplace_back([&, w] {
              try {
                std::vector<uint64_t> tmp;
                for (int td = w; td < TD; td += nth) {
                    int32_t* row = inv.data() + static_cast<size_t>(td) * static_cast<size_t>(c->n_ids);
                    const std::vector<int32_t>& v = c->h_ids[td];
                    for (size_t i = 0; i < v.size(); i++) {
                        if (v[i] < 0 || v[i] >= c->n_ids || row[v[i]] != -1) { ok = false; break; }
                        row[v[i]] = static_cast<int32_t>(i);
                    }
                    if (!ok) return;
                    const RouteTable& t = c->h_tables[td];
                    for (int p = 0; p < t.nparts; p++) {
                        const int64_t b0 = c->h_off[td][p], b1 = c->h_off[td][p + 1];
                        tmp.clear();
                        for (int64_t i = b0; i < b1; i++) {
                            uint32_t h = static_cast<uint32_t>(c->h_java_hash[static_cast<size_t>(v[i])]);
                            h ^= (h >> 16);   // HashMap.hash()
                            const uint64_t bf = static_cast<uint64_t>((h & bmask) << bshift);
                            tmp.push_back((bf << 44) | (static_cast<uint64_t>(i - b0) << 32) | static_cast<uint32_t>(v[i]));   // sort key: bucket, position
                        }
                        std::sort(tmp.begin(), tmp.end());
                        for (size_t j = 0; j < tmp.size(); j++)
                            bk[static_cast<size_t>(t.ids_base + b0) + j] = (static_cast<uint64_t>(static_cast<uint32_t>(tmp[j])) << 32) | (tmp[j] >> 44);
                    }
                }
              } catch (...) { ok = false; }   // out of host memory: the bounded select stays off
            });
        for (auto& t : th) t.join();
        // bin16: the HashMap bin (table length cap0 <= 65536) of every id in partition order, one padded row of 1 << bin16_shift
        // entries per partition — the bounded select's exact treeify check counts ALL ids of the probed partitions per bin with
        // it (route_lazy.hip.h, step 0).  Built when that check can be asked for: opaque ids (caller-supplied hashCodes), or forced.
        const bool want_bin16 = (c->knob_bincheck == 1) || (c->knob_bincheck < 0 && !c->decimal_ids);
        std::vector<uint16_t> b16;
        int b16_shift = 0;
        if (ok && want_bin16 && c->cap0 <= 65536 && parts > 0) {
            int64_t maxsz = 4;
            for (int td = 0; td < TD; td++)
                for (int p = 0; p < c->h_tables[td].nparts; p++) maxsz = std::max<int64_t>(maxsz, c->h_off[td][p + 1] - c->h_off[td][p]);
            while ((int64_t(1) << b16_shift) < maxsz) b16_shift++;
            if (b16_shift <= 12 && (static_cast<uint64_t>(parts) << b16_shift) < (uint64_t(1) << 33)) {
                try {
                    b16.assign(static_cast<size_t>(parts) << b16_shift, 0xFFFFu);
                    for (int td = 0; td < TD; td++) {
                        const RouteTable& t = c->h_tables[td];
                        const std::vector<int32_t>& v = c->h_ids[td];
                        for (int p = 0; p < t.nparts; p++) {
                            uint16_t* row = b16.data() + (static_cast<size_t>(t.part_base + p) << b16_shift);
                            const int64_t b0 = c->h_off[td][p], b1 = c->h_off[td][p + 1];
                            for (int64_t i = b0; i < b1; i++) {
                                uint32_t h = static_cast<uint32_t>(c->h_java_hash[static_cast<size_t>(v[i])]);
                                h ^= (h >> 16);   // HashMap.hash()
                                row[i - b0] = static_cast<uint16_t>(h & bmask);
                            }
                        }
                    }
                } catch (...) { b16.clear(); }   // out of host memory: no check -> the bounded select stays off for opaque ids
            }
        }
        if (ok) {
            if (!b16.empty()) {
                FSP_HIP(hipMalloc(&c->d_bin16, b16.size() * 2 + 64));
                FSP_HIP(hipMemcpy(c->d_bin16, b16.data(), b16.size() * 2, hipMemcpyHostToDevice));
                c->bin16_shift = b16_shift;
            }
            FSP_HIP(hipMalloc(&c->d_inv, inv.size() * 4));
            FSP_HIP(hipMemcpy(c->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
            FSP_HIP(hipMalloc(&c->d_ids_bk, bk.size() * 8));
            FSP_HIP(hipMemcpy(c->d_ids_bk, bk.data(), bk.size() * 8, hipMemcpyHostToDevice));
            c->bk_epoch = c->meta_epoch;
        }
    }
    c->dev_index_dirty = false;
    return FSPANN_OK;
}


}  // namespace

extern "C" {

const char* fspann_last_error(void) { return last_error_ref().c_str(); }
const char* fspann_version(void) { return "fspann-hip 0.1 (gfx950)"; }

int fspann_ctx_create(int device, const fspann_cfg* cfg, fspann_ctx** out) {
    if (!cfg || !out) return fail(FSPANN_E_NULL, "cfg/out is null");
    *out = nullptr;
    fspann_cfg g = *cfg;
    if (g.block_size <= 0) g.block_size = 64;
    if (g.default_probes <= 0) g.default_probes = 5;
    if (g.max_global_candidates <= 0) g.max_global_candidates = 20000;
    if (g.refinement_limit <= 0) g.refinement_limit = 20000;
    if (g.tables <= 0 || g.divisions <= 0 || g.m <= 0 || g.lambda <= 0 || g.dim <= 0)
        return fail(FSPANN_E_ARG, "tables, divisions, m, lambda, dim must be > 0");
    if (g.lambda > 32) return fail(FSPANN_E_ARG, "lambda > 32 is not supported (h_j is an int32)");
    if (g.m > kEncThreads) return fail(FSPANN_E_ARG, "m > %d is not supported", kEncThreads);
    if (g.block_size > 1024) return fail(FSPANN_E_ARG, "block_size > 1024 is not supported");
    const int64_t bits = static_cast<int64_t>(g.m) * g.lambda;
    if (bits >= (1 << kScoreBits)) return fail(FSPANN_E_ARG, "m*lambda = %lld exceeds %d code bits", (long long)bits, (1 << kScoreBits) - 1);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(FSPANN_E_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(FSPANN_E_ARG, "device %d out of range [0,%d)", device, ndev);
    FSP_HIP(hipSetDevice(device));
    fspann_ctx* c = new (std::nothrow) fspann_ctx();
    if (!c) return fail(FSPANN_E_NOMEM, "out of host memory");
    c->device = device;
    c->cfg = g;
    c->TD = g.tables * g.divisions;
    c->bits = static_cast<int>(bits);
    c->W = (c->bits + 63) / 64;
    c->P_total = c->TD * g.m;
    c->hard_cap = std::max(g.max_global_candidates, g.refinement_limit);  // PIS:612-615
    c->cap0 = table_size_for(std::min(c->hard_cap, 1 << 16));             // PIS:619
    if (c->hard_cap > 700000) {
        delete c;
        return fail(FSPANN_E_ARG, "max(maxGlobalCandidates, refinementLimit) > 700000 exceeds the %d-bit bucket field", kBucketBits);
    }
    if (c->cap0 < 64) {
        delete c;
        return fail(FSPANN_E_ARG, "max(maxGlobalCandidates, refinementLimit) < 33: HashMap order with a table shorter "
                                  "than MIN_TREEIFY_CAPACITY is not modelled");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (prop.sharedMemPerBlock > 0) c->lds_limit = static_cast<int>(std::min<size_t>(prop.sharedMemPerBlock, 160 * 1024));
        if (prop.maxSharedMemoryPerMultiProcessor > 0)
            c->lds_limit = static_cast<int>(std::min<size_t>(std::max<size_t>(prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor), 160 * 1024));
    }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(FSPANN_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    {   // tuning / test knobs: read once per context, never on the call path
        auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; };
        c->knob_ht_x4 = env_int("FSPANN_ROUTE_HT_X4", 0) == 1;
        c->knob_threads = env_int("FSPANN_ROUTE_THREADS", 0);       // 0: 512, and 1024 for long lists (the shipped profiles); 512 / 1024 force
        c->knob_lazy_cap = std::max(0, env_int("FSPANN_ROUTE_LAZY_CAP", 0));
        c->knob_fused_probe = env_int("FSPANN_ROUTE_FUSED_PROBE", 1) != 0;
        c->knob_probe_dir = env_int("FSPANN_ROUTE_DIR", 1) != 0;
        c->knob_lazy_small = env_int("FSPANN_ROUTE_LAZY_SMALL", 1) != 0;
        c->knob_bincheck = env_int("FSPANN_ROUTE_BINCHECK", -1);
        c->knob_slice = env_int("FSPANN_ROUTE_SLICE", 1) != 0;
        c->knob_shape_spec = env_int("FSPANN_ROUTE_SHAPE_SPEC", 1) != 0;
        c->knob_route_lds_kb = env_int("FSPANN_ROUTE_LDS_KB", 0);
        c->knob_zero_copy = env_int("FSPANN_ZERO_COPY", 1) != 0;
        c->knob_encode_qb = env_int("FSPANN_ENCODE_QB", 0);
        c->knob_mfma_tile = env_int("FSPANN_ENCODE_MFMA_TILE", 0);
        c->knob_route_wgs = env_int("FSPANN_ROUTE_WGS", 0);
        c->knob_refine_run = env_int("FSPANN_REFINE_RUN", 1) != 0;
        c->knob_devflags = env_int("FSPANN_ROUTE_DEVFLAGS", 0);
        c->knob_dir_extra_bits = env_int("FSPANN_ROUTE_DIR_EXTRA_BITS", kDirBitsAuto);   // unset: as many as fit 64 MB (at most six)
        c->knob_refine_dc = env_int("FSPANN_REFINE_DC", 0);
        c->knob_refine_stream = std::min(4, std::max(-1, env_int("FSPANN_REFINE_STREAM", -1)));   // -1: 4 per CU dense, 3 per CU gather
        c->knob_tick_refine = std::min(4, std::max(1, env_int("FSPANN_TICK_REFINE", 1)));
        c->knob_gpu_cut = env_int("FSPANN_GPU_CUT", 1) != 0;
        c->knob_tick_fuse = env_int("FSPANN_TICK_FUSE", 1) != 0;
        c->knob_wave_sort = env_int("FSPANN_ROUTE_WAVE_SORT", 1);      // 1: per-wave group sorts, 0: whole-workgroup group sorts, -1: general sort only
        c->knob_tick_front = std::min(100, std::max(0, env_int("FSPANN_TICK_FRONT", 100)));
    }
    c->h_min.resize(c->TD); c->h_max.resize(c->TD); c->h_off.resize(c->TD); c->h_rep.resize(c->TD); c->h_ids.resize(c->TD);
    c->h_table_set.assign(c->TD, 0);
    if (hipMalloc(&c->d_unmodelled, 256) != hipSuccess || hipMemset(c->d_unmodelled, 0, 256) != hipSuccess) {
        fspann_ctx_destroy(c);
        return fail(FSPANN_E_NOMEM, "hipMalloc failed");
    }
    *out = c;
    return FSPANN_OK;
}

void fspann_ctx_destroy(fspann_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_refs.load() > 0) {
        // a communicator still launches on this context's stream (fspann_allgather_topk_dev): the context goes with the last
        // of them (fspann_comm_destroy) instead of leaving it a dangling pointer
        c->destroy_deferred.store(true);
        if (c->comm_refs.load() > 0) return;
        if (!c->destroy_deferred.exchange(false)) return;      // the communicator went in between and took the destroy with it
    }
    fspann_ctx* parent = c->share_parent;
    {
        // the family's bookkeeping (clones alive, owner gone) changes under the OWNER's lock: clones are driven — and destroyed —
        // from different threads
        std::unique_lock<std::recursive_mutex> fam((parent ? parent : c)->mu);
        if (!parent && c->share_children.load() > 0 && !c->zombie) {   // clones still read this context's arrays: keep them until the last clone goes
            c->zombie = true;
            return;
        }
    }
    if (parent) {                                   // a clone owns none of the shared arrays
        c->d_alphaT = nullptr; c->d_r = nullptr; c->d_omega = nullptr; c->d_alphaT32 = nullptr; c->d_alpha_rows = nullptr;
        c->d_tables = nullptr; c->d_recs = nullptr; c->d_ids = nullptr; c->d_dir = nullptr; c->d_inv = nullptr; c->d_ids_bk = nullptr; c->d_bin16 = nullptr;
        c->d_java_hash = nullptr; c->d_deleted_bits = nullptr;
        if (!c->store_owned) c->d_store = nullptr;
    }
    free_devt(c->d_alphaT); free_devt(c->d_r); free_devt(c->d_omega); free_devt(c->d_alphaT32); free_devt(c->d_alpha_rows); free_dev(c->ws_fix.p);
    free_devt(c->d_tables); free_devt(c->d_recs); free_devt(c->d_ids); free_devt(c->d_dir);
    free_devt(c->d_java_hash); free_devt(c->d_unmodelled);
    if (uint32_t* db = c->d_deleted_bits.exchange(nullptr)) (void)hipFree(db);
    if (c->store_owned) free_dev(c->d_store);
    free_dev(c->ws_tickfix.p); free_dev(c->d_fixparams); free_dev(c->ws_gt.p); free_dev(c->bld_codes.p);
    free_dev(c->ws_route.p); free_dev(c->ws_refine.p); free_dev(c->ws_probe.p); free_dev(c->ws_ovf.p); free_dev(c->ws_search.p); free_devt(c->d_inv); free_devt(c->d_ids_bk); free_devt(c->d_bin16);
    for (hipEvent_t e : c->rt_events) (void)hipEventDestroy(e);
    for (auto& b : c->ws_io) free_dev(b.p);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_rows) (void)hipHostFree(c->h_rows);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    if (parent) {
        bool last_of_zombie;
        {
            std::unique_lock<std::recursive_mutex> fam(parent->mu);
            last_of_zombie = (--parent->share_children == 0) && parent->zombie;
        }
        if (last_of_zombie) fspann_ctx_destroy(parent);   // exactly one clone sees the transition to zero
    }
}

// A context that shares src's frozen state (include/fspann.h).
int fspann_ctx_clone(fspann_ctx* src, fspann_ctx** out) {
    CHECK_CTX_NOLOCK(src);
    if (!out) return fail(FSPANN_E_NULL, "out is null");
    fspann_ctx* root = src->share_parent ? src->share_parent : src;      // clones of clones share the same owner
    std::lock_guard<std::recursive_mutex> fam(root->mu);                 // the owner's state is read (and its clone count raised) under its lock
    if (!src->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (src->zombie || root->zombie) return fail(FSPANN_E_STATE, "context was destroyed");
    fspann_ctx* c = nullptr;
    int rc = fspann_ctx_create(src->device, &src->cfg, &c);
    if (rc) return rc;
    c->have_g = root->have_g; c->alpha_norm_max = root->alpha_norm_max; c->encode_mode = src->encode_mode;
    c->h_alpha = root->h_alpha; c->h_r = root->h_r; c->h_omega = root->h_omega;
    c->d_alphaT = root->d_alphaT; c->d_r = root->d_r; c->d_omega = root->d_omega; c->d_alphaT32 = root->d_alphaT32; c->d_alpha_rows = root->d_alpha_rows;
    c->h_tables = root->h_tables;
    c->h_table_set.assign(c->TD, 1);
    c->d_tables = root->d_tables; c->d_recs = root->d_recs; c->rec_words = root->rec_words; c->d_dir = root->d_dir; c->dir_bits = root->dir_bits;
    c->d_ids = root->d_ids; c->d_inv = root->d_inv; c->d_ids_bk = root->d_ids_bk; c->d_bin16 = root->d_bin16; c->bin16_shift = root->bin16_shift;
    c->meta_epoch = root->meta_epoch; c->bk_epoch = root->bk_epoch; c->route_mode = src->route_mode;
    c->total_parts = root->total_parts; c->total_ids = root->total_ids;
    c->n_ids = root->n_ids; c->d_java_hash = root->d_java_hash; c->decimal_ids = root->decimal_ids;   // (deleted bits: read from the owner at every call)
    c->d_store = root->d_store; c->store_owned = false; c->store_dtype = root->store_dtype; c->store_n = root->store_n;
    c->dev_index_dirty = false;
    c->frozen = true;
    c->share_parent = root;
    root->share_children++;
    *out = c;
    return FSPANN_OK;
}

void* fspann_ctx_stream(fspann_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int fspann_sync(fspann_ctx* c) {
    CHECK_CTX(c);
    FSP_HIP(hipStreamSynchronize(c->stream));
    return FSPANN_OK;
}

int fspann_set_gfunctions(fspann_ctx* c, const double* alpha, const double* r, const double* omega) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (!alpha || !r || !omega) return fail(FSPANN_E_NULL, "alpha/r/omega is null");
    const int P = c->P_total, d = c->cfg.dim;
    for (int p = 0; p < P; p++)
        if (!(omega[p] > 0.0)) return fail(FSPANN_E_ARG, "omega_j <= 0");  // Coding.java:84-86
    return guarded([&]() -> int {
    std::vector<double> aT(static_cast<size_t>(d) * P);
    for (int p = 0; p < P; p++)
        for (int i = 0; i < d; i++) aT[static_cast<size_t>(i) * P + p] = alpha[static_cast<size_t>(p) * d + i];
    free_devt(c->d_alphaT); free_devt(c->d_r); free_devt(c->d_omega); free_devt(c->d_alphaT32); free_devt(c->d_alpha_rows);
    {
        std::vector<float> aT32(aT.size());
        for (size_t i = 0; i < aT.size(); i++) aT32[i] = static_cast<float>(aT[i]);
        double nmax = 0.0;
        for (int pp = 0; pp < P; pp++) {
            double s2 = 0.0;
            for (int i = 0; i < d; i++) s2 += alpha[static_cast<size_t>(pp) * d + i] * alpha[static_cast<size_t>(pp) * d + i];
            nmax = std::max(nmax, std::sqrt(s2));
        }
        c->alpha_norm_max = nmax * (1.0 + 1e-12);
        FSP_HIP(hipMalloc(&c->d_alphaT32, aT32.size() * 4));
        FSP_HIP(hipMemcpy(c->d_alphaT32, aT32.data(), aT32.size() * 4, hipMemcpyHostToDevice));
    }
    FSP_HIP(hipMalloc(&c->d_alphaT, aT.size() * 8));
    FSP_HIP(hipMalloc(&c->d_r, static_cast<size_t>(P) * 8));
    FSP_HIP(hipMalloc(&c->d_omega, static_cast<size_t>(P) * 8));
    FSP_HIP(hipMemcpy(c->d_alphaT, aT.data(), aT.size() * 8, hipMemcpyHostToDevice));
    FSP_HIP(hipMalloc(&c->d_alpha_rows, aT.size() * 8));
    FSP_HIP(hipMemcpy(c->d_alpha_rows, alpha, aT.size() * 8, hipMemcpyHostToDevice));
    FSP_HIP(hipMemcpy(c->d_r, r, static_cast<size_t>(P) * 8, hipMemcpyHostToDevice));
    FSP_HIP(hipMemcpy(c->d_omega, omega, static_cast<size_t>(P) * 8, hipMemcpyHostToDevice));
    if (c->h_alpha.data() != alpha) { c->h_alpha.assign(alpha, alpha + static_cast<size_t>(P) * d); c->h_r.assign(r, r + P); c->h_omega.assign(omega, omega + P); }
    c->have_g = true;
    return FSPANN_OK;
    });
}

// GFunctionRegistry.initialize (idx/GFunctionRegistry.java:63-147) = T*D x Coding.buildFromSample
// (idx/Coding.java:184-241).  Host: SplittableRandom + Box-Muller rows (glibc log/cos — like any
// non-JVM generator NOT bit-portable to HotSpot); device: the sample's projections
// y = dot(v, alpha_j) with the exact fp64 kernel, from which omega_j = max(1e-6, max-min)/2.5.
int fspann_registry_initialize(fspann_ctx* c, const double* sample, int64_t ns, int64_t base_seed) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (!sample) return fail(FSPANN_E_NULL, "sample");
    if (ns <= 0) return fail(FSPANN_E_ARG, "Sample vectors cannot be empty");
    const int TD = c->TD, m = c->cfg.m, d = c->cfg.dim, P = c->P_total, D = c->cfg.divisions;
    return guarded([&]() -> int {
    struct Rng {
        uint64_t s;
        uint64_t nextLong() {
            s += 0x9E3779B97F4A7C15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            return z ^ (z >> 31);
        }
        double nextDouble() { return static_cast<double>(nextLong() >> 11) * 0x1.0p-53; }
    };
    std::vector<double> alpha(static_cast<size_t>(P) * d), r(P, 0.0), w(P, 1.0);
    std::vector<Rng> rngs(TD);
    for (int td = 0; td < TD; td++) {
        const int t = td / D, dv = td % D;
        Rng& g = rngs[td];
        g.s = static_cast<uint64_t>(base_seed + static_cast<int64_t>(t) * 1000003LL + dv);  // computeSeed :291-293
        for (int j = 0; j < m; j++) {
            double* row = alpha.data() + (static_cast<size_t>(td) * m + j) * d;
            double norm = 0.0;
            for (int i = 0; i < d; i++) {
                const double u1 = std::max(4.9e-324, g.nextDouble());
                const double u2 = g.nextDouble();
                const double mag = std::sqrt(-2.0 * std::log(u1));
                const double v = mag * std::cos(2.0 * M_PI * u2);
                row[i] = v;
                norm += v * v;
            }
            norm = std::sqrt(std::max(1e-12, norm));
            for (int i = 0; i < d; i++) row[i] /= norm;
        }
    }
    int rc = fspann_set_gfunctions(c, alpha.data(), r.data(), w.data());
    if (rc) return rc;
    // projections of the sample on the device (sequential fp64 == Coding.dot)
    const size_t sb = static_cast<size_t>(ns) * d * 8, pb = static_cast<size_t>(ns) * P * 8;
    if ((rc = ensure(c, c->ws_io[0], sb))) return rc;
    if ((rc = ensure(c, c->ws_io[1], static_cast<size_t>(ns) * TD * c->W * 8))) return rc;
    if ((rc = ensure(c, c->ws_io[2], static_cast<size_t>(ns) * 4))) return rc;
    if ((rc = ensure(c, c->ws_io[3], pb))) return rc;
    FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, sample, sb, hipMemcpyHostToDevice, c->stream));
    rc = launch_encode<double>(c, ns, static_cast<const double*>(c->ws_io[0].p), static_cast<uint64_t*>(c->ws_io[1].p), nullptr,
                               static_cast<int32_t*>(c->ws_io[2].p), static_cast<double*>(c->ws_io[3].p));
    if (rc) return rc;
    std::vector<double> proj(static_cast<size_t>(ns) * P);
    std::vector<int32_t> bad(static_cast<size_t>(ns));
    FSP_HIP(hipMemcpyAsync(proj.data(), c->ws_io[3].p, pb, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(bad.data(), c->ws_io[2].p, static_cast<size_t>(ns) * 4, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < ns; i++)
        if (bad[i]) { c->have_g = false; return fail(FSPANN_E_ARG, "Vector contains NaN/Inf (sample %lld)", (long long)i); }
    for (int p = 0; p < P; p++) {
        double mn = INFINITY, mx = -INFINITY;
        for (int64_t s = 0; s < ns; s++) {
            const double y = proj[static_cast<size_t>(s) * P + p];
            if (y < mn) mn = y;
            if (y > mx) mx = y;
        }
        const double range = std::max(1e-6, mx - mn);
        double omega = range / 2.5;  // OMEGA_DIVISOR
        if (!(omega > 0)) omega = 1e-3;
        w[p] = omega;
    }
    for (int td = 0; td < TD; td++)
        for (int j = 0; j < m; j++) r[td * m + j] = rngs[td].nextDouble() * w[td * m + j];  // one draw per j, after all alpha
    c->h_alpha = alpha; c->h_r = r; c->h_omega = w;
    rc = fspann_set_gfunctions(c, alpha.data(), r.data(), w.data());
    return rc;
    });
}

int fspann_get_gfunctions(fspann_ctx* c, double* alpha, double* r, double* omega) {
    CHECK_CTX(c);
    if (!c->have_g || c->h_alpha.empty()) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized");
    if (alpha) std::copy(c->h_alpha.begin(), c->h_alpha.end(), alpha);
    if (r) std::copy(c->h_r.begin(), c->h_r.end(), r);
    if (omega) std::copy(c->h_omega.begin(), c->h_omega.end(), omega);
    return FSPANN_OK;
}

int fspann_set_index(fspann_ctx* c, int td, int64_t n_parts, const int64_t* min_key, const int64_t* max_key,
                     const uint64_t* rep, const int64_t* id_off, const int32_t* ids) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (td < 0 || td >= c->TD) return fail(FSPANN_E_ARG, "td %d out of range [0,%d)", td, c->TD);
    if (n_parts < 0) return fail(FSPANN_E_ARG, "n_parts < 0");
    if (n_parts > 0 && (!min_key || !max_key || !rep || !id_off || !ids)) return fail(FSPANN_E_NULL, "index array is null");
    if (n_parts > 0 && id_off[0] != 0) return fail(FSPANN_E_ARG, "id_off[0] must be 0");
    for (int64_t p = 0; p < n_parts; p++) {
        const int64_t sz = id_off[p + 1] - id_off[p];
        if (sz < 0 || sz > c->cfg.block_size)
            return fail(FSPANN_E_ARG, "partition %lld of table %d has %lld ids (block_size %d)", (long long)p, td, (long long)sz, c->cfg.block_size);
    }
    const int64_t nid = n_parts > 0 ? id_off[n_parts] : 0;
    if (nid >= (1LL << 31)) return fail(FSPANN_E_RANGE, "table has >= 2^31 ids");
    // id handles are validated against n_ids by fspann_finalize (the id metadata may arrive after the tables)
    c->frozen = false;
    return guarded([&]() -> int {
    c->h_min[td].assign(min_key, min_key + n_parts);
    c->h_max[td].assign(max_key, max_key + n_parts);
    c->h_rep[td].assign(rep, rep + n_parts * c->W);
    if (n_parts > 0) c->h_off[td].assign(id_off, id_off + n_parts + 1); else c->h_off[td].assign(1, 0);
    c->h_ids[td].assign(ids, ids + nid);
    c->h_table_set[td] = 1;
    c->dev_index_dirty = true;
    return FSPANN_OK;
    });
}

int fspann_set_id_meta(fspann_ctx* c, int64_t n_ids, const int32_t* java_hash, const uint8_t* deleted) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (n_ids <= 0 || n_ids >= (1LL << 31)) return fail(FSPANN_E_ARG, "n_ids out of range");
    c->frozen = false;           // Route stays off until the next successful fspann_finalize re-validates every table
    return guarded([&]() -> int {
    c->h_java_hash.resize(static_cast<size_t>(n_ids));
    c->decimal_ids = (java_hash == nullptr);
    if (java_hash) std::copy(java_hash, java_hash + n_ids, c->h_java_hash.begin());
    else for (int64_t i = 0; i < n_ids; i++) c->h_java_hash[i] = decimal_string_hash(i);
    free_devt(c->d_java_hash);
    if (uint32_t* db = c->d_deleted_bits.exchange(nullptr)) (void)hipFree(db);
    FSP_HIP(hipMalloc(&c->d_java_hash, static_cast<size_t>(n_ids) * 4));
    FSP_HIP(hipMemcpy(c->d_java_hash, c->h_java_hash.data(), static_cast<size_t>(n_ids) * 4, hipMemcpyHostToDevice));
    {
        std::lock_guard<std::mutex> dl(c->deleted_mu);
        c->h_deleted_bits.assign(static_cast<size_t>((n_ids + 31) / 32), 0u);
        bool any = false;
        if (deleted)
            for (int64_t i = 0; i < n_ids; i++)
                if (deleted[i]) { c->h_deleted_bits[i >> 5] |= (1u << (i & 31)); any = true; }
        if (any) {      // (none deleted: the kernels skip the lookup until the first fspann_set_deleted)
            uint32_t* db = nullptr;
            FSP_HIP(hipMalloc(&db, c->h_deleted_bits.size() * 4));
            if (hipMemcpy(db, c->h_deleted_bits.data(), c->h_deleted_bits.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(db); return fail(FSPANN_E_DEVICE, "hipMemcpy failed"); }
            c->d_deleted_bits.store(db, std::memory_order_release);
        }
    }
    c->meta_epoch++;             // d_inv / d_ids_bk were built for the previous hashes: the bounded select waits for the next finalize
    c->dev_index_dirty = true;
    c->n_ids = n_ids;
    return FSPANN_OK;
    });
}

int fspann_finalize(fspann_ctx* c) {
    CHECK_CTX(c);
    if (c->share_parent) return FSPANN_OK;      // a clone is frozen with its parent's state
    if (c->share_children.load() > 0) return fail(FSPANN_E_STATE, "the index is shared with %d clone(s): destroy them first", c->share_children.load());
    if (!c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized");
    if (c->n_ids <= 0) return fail(FSPANN_E_STATE, "id metadata not set (fspann_set_id_meta)");
    if (c->dev_index_dirty) {
        c->frozen = false;
        int rc = guarded([&]() -> int { return upload_index(c); });
        if (rc) return rc;
    }
    c->frozen = true;
    return FSPANN_OK;
}

// ---- frozen-index file: the reference never persists routing state and rebuilds it by decrypting
// every point (ForwardSecureANNSystem.java:926-948).  Flat little-endian SoA, versioned:
//   magic "FSPANNIX" | u32 version=1 | cfg {tables,divisions,m,lambda,dim,block_size} | i64 n_ids | u8 decimal_ids
//   | alpha[TD*m*dim] r[TD*m] omega[TD*m] f64 | java_hash[n_ids] i32 | deleted[n_ids] u8
//   | per td: i64 n_parts, i64 n_ids_td, min[n_parts] max[n_parts] i64, rep[n_parts*W] u64, off[n_parts+1] i64, ids i32
}  // extern "C"
namespace {
template <typename T> bool wr(FILE* f, const T* p, size_t n) { return n == 0 || std::fwrite(p, sizeof(T), n, f) == n; }
template <typename T> bool rd(FILE* f, T* p, size_t n) { return n == 0 || std::fread(p, sizeof(T), n, f) == n; }
}  // namespace
extern "C" {

int fspann_index_save(fspann_ctx* c, const char* path) {
    CHECK_CTX(c);
    if (c->share_parent) c = c->share_parent;   // the host mirror of a shared index lives in its owner
    if (!path) return fail(FSPANN_E_NULL, "path is null");
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(FSPANN_E_ARG, "cannot open %s for writing", path);
    struct Closer { FILE*& f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    return guarded([&]() -> int {
    bool ok = true;
    const char magic[8] = {'F', 'S', 'P', 'A', 'N', 'N', 'I', 'X'};
    const uint32_t ver = 1;
    const int32_t hdr[6] = {c->cfg.tables, c->cfg.divisions, c->cfg.m, c->cfg.lambda, c->cfg.dim, c->cfg.block_size};
    const uint8_t dec = c->decimal_ids ? 1 : 0;
    ok = ok && wr(f, magic, 8) && wr(f, &ver, 1) && wr(f, hdr, 6) && wr(f, &c->n_ids, 1) && wr(f, &dec, 1);
    ok = ok && wr(f, c->h_alpha.data(), c->h_alpha.size()) && wr(f, c->h_r.data(), c->h_r.size()) && wr(f, c->h_omega.data(), c->h_omega.size());
    ok = ok && wr(f, c->h_java_hash.data(), c->h_java_hash.size());
    std::vector<uint8_t> del(static_cast<size_t>(c->n_ids), 0);
    {
        std::lock_guard<std::mutex> dl(c->deleted_mu);
        if (!c->h_deleted_bits.empty())
            for (int64_t i = 0; i < c->n_ids; i++) del[i] = (c->h_deleted_bits[i >> 5] >> (i & 31)) & 1u;
    }
    ok = ok && wr(f, del.data(), del.size());
    for (int td = 0; td < c->TD && ok; td++) {
        const int64_t np = static_cast<int64_t>(c->h_min[td].size()), ni = static_cast<int64_t>(c->h_ids[td].size());
        ok = ok && wr(f, &np, 1) && wr(f, &ni, 1) && wr(f, c->h_min[td].data(), np) && wr(f, c->h_max[td].data(), np) &&
             wr(f, c->h_rep[td].data(), c->h_rep[td].size()) && wr(f, c->h_off[td].data(), c->h_off[td].size()) && wr(f, c->h_ids[td].data(), ni);
    }
    ok = (std::fclose(f) == 0) && ok;
    f = nullptr;
    return ok ? FSPANN_OK : fail(FSPANN_E_ARG, "short write to %s", path);
    });
}

int fspann_index_load(fspann_ctx* c, const char* path) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (!path) return fail(FSPANN_E_NULL, "path is null");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(FSPANN_E_ARG, "cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    // whatever happens below, the context serves no Route until a finalize has succeeded on the new state
    c->frozen = false;
    return guarded([&]() -> int {
    // every count read from the file is checked against the bytes the file still holds BEFORE anything is sized from it
    if (std::fseek(f, 0, SEEK_END) != 0) return fail(FSPANN_E_ARG, "cannot seek in %s", path);
    const long long fsize = std::ftell(f);
    std::rewind(f);
    auto left = [&]() -> long long { return fsize - std::ftell(f); };
    char magic[8];
    uint32_t ver = 0;
    int32_t hdr[6];
    int64_t n_ids = 0;
    uint8_t dec = 0;
    if (!rd(f, magic, 8) || std::memcmp(magic, "FSPANNIX", 8) != 0 || !rd(f, &ver, 1) || ver != 1)
        return fail(FSPANN_E_ARG, "%s is not a version-1 fspann index file", path);
    if (!rd(f, hdr, 6) || !rd(f, &n_ids, 1) || !rd(f, &dec, 1)) return fail(FSPANN_E_ARG, "truncated header in %s", path);
    if (hdr[0] != c->cfg.tables || hdr[1] != c->cfg.divisions || hdr[2] != c->cfg.m || hdr[3] != c->cfg.lambda || hdr[4] != c->cfg.dim ||
        hdr[5] != c->cfg.block_size)
        return fail(FSPANN_E_STATE, "index file was built for tables=%d divisions=%d m=%d lambda=%d dim=%d (context differs)", hdr[0], hdr[1],
                    hdr[2], hdr[3], hdr[4]);
    if (n_ids <= 0 || n_ids >= (1LL << 31)) return fail(FSPANN_E_ARG, "bad n_ids in %s", path);
    const size_t P = static_cast<size_t>(c->P_total), d = static_cast<size_t>(c->cfg.dim);
    if (static_cast<long long>((P * d + 2 * P) * 8) + n_ids * 5 > left()) return fail(FSPANN_E_ARG, "truncated file %s", path);
    std::vector<double> alpha(P * d), r(P), w(P);
    std::vector<int32_t> jh(static_cast<size_t>(n_ids));
    std::vector<uint8_t> del(static_cast<size_t>(n_ids));
    if (!rd(f, alpha.data(), alpha.size()) || !rd(f, r.data(), P) || !rd(f, w.data(), P) || !rd(f, jh.data(), jh.size()) || !rd(f, del.data(), del.size()))
        return fail(FSPANN_E_ARG, "truncated file %s", path);
    int rc = fspann_set_gfunctions(c, alpha.data(), r.data(), w.data());
    if (rc) return rc;
    if ((rc = fspann_set_id_meta(c, n_ids, dec ? nullptr : jh.data(), del.data()))) return rc;
    for (int td = 0; td < c->TD; td++) {
        int64_t np = 0, ni = 0;
        if (!rd(f, &np, 1) || !rd(f, &ni, 1) || np < 0 || ni < 0 || ni > n_ids)
            return fail(FSPANN_E_ARG, "bad table header %d in %s", td, path);
        if (np * (16 + 8 * static_cast<long long>(c->W)) + (np + 1) * 8 + ni * 4 > left()) return fail(FSPANN_E_ARG, "truncated table %d in %s", td, path);
        std::vector<int64_t> mn(np), mx(np), off(np + 1);
        std::vector<uint64_t> rep(static_cast<size_t>(np) * c->W);
        std::vector<int32_t> ids(ni);
        if (!rd(f, mn.data(), np) || !rd(f, mx.data(), np) || !rd(f, rep.data(), rep.size()) || !rd(f, off.data(), np + 1) || !rd(f, ids.data(), ni))
            return fail(FSPANN_E_ARG, "truncated table %d in %s", td, path);
        if (off[0] != 0 || off[np] != ni) return fail(FSPANN_E_ARG, "bad id offsets in table %d of %s", td, path);
        if ((rc = fspann_set_index(c, td, np, mn.data(), mx.data(), rep.data(), off.data(), ids.data()))) return rc;
    }
    return fspann_finalize(c);
    });
}

int fspann_index_dims(fspann_ctx* c, int td, int64_t* n_parts, int64_t* n_ids) {
    CHECK_CTX(c);
    if (c->share_parent) c = c->share_parent;   // the host mirror of a shared index lives in its owner
    if (td < 0 || td >= c->TD) return fail(FSPANN_E_ARG, "td out of range");
    if (!c->h_table_set[td]) return fail(FSPANN_E_STATE, "table %d not set", td);
    if (n_parts) *n_parts = static_cast<int64_t>(c->h_min[td].size());
    if (n_ids) *n_ids = static_cast<int64_t>(c->h_ids[td].size());
    return FSPANN_OK;
}

int fspann_get_index(fspann_ctx* c, int td, int64_t* min_key, int64_t* max_key, uint64_t* rep, int64_t* id_off,
                     int32_t* ids) {
    CHECK_CTX(c);
    if (c->share_parent) c = c->share_parent;   // the host mirror of a shared index lives in its owner
    if (td < 0 || td >= c->TD) return fail(FSPANN_E_ARG, "td out of range");
    if (!c->h_table_set[td]) return fail(FSPANN_E_STATE, "table %d not set", td);
    std::copy(c->h_min[td].begin(), c->h_min[td].end(), min_key);
    std::copy(c->h_max[td].begin(), c->h_max[td].end(), max_key);
    std::copy(c->h_rep[td].begin(), c->h_rep[td].end(), rep);
    std::copy(c->h_off[td].begin(), c->h_off[td].end(), id_off);
    std::copy(c->h_ids[td].begin(), c->h_ids[td].end(), ids);
    return FSPANN_OK;
}


}  // extern "C"
