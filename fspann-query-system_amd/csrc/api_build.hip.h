// api_build.hip.h — native Setup: code all vectors on the GPU, cut the partitions (PIS:331-346,372-434; idx/GreedyPartitioner.java:37-76)
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

// ---- native Setup: code all vectors on the GPU, cut partitions -----------------------------
// Replaces PIS.insert's coding loop (PIS:331-346) + PIS.build (PIS:372-434) +
// GreedyPartitioner.build (idx/GreedyPartitioner.java:37-76).  The reference iterates a
// HashMap<String,BitSet>(staged.size()) and stable-sorts by key, so elements with equal keys keep
// HashMap iteration order = (bucket at the final capacity, insertion order) — the closed form used
// here (valid while no bin treeifies; DESIGN.md "Java order key").
namespace {
// Incremental Setup: rows arrive in chunks (IndexService.insert is one vector at a time, common/.../IndexService.java:19; a JVM
// hands over direct buffers of at most 2 GB), are coded on arrival — MFMA pre-filter + exact re-check for chunks >= 4096 rows,
// bit-identical codes either way — and only their codes stay in HBM until the cut.
int build_begin_impl(fspann_ctx* c, int64_t n) {
    const size_t need = static_cast<size_t>(n) * c->TD * c->W * 8;
    int rc = ensure(c, c->bld_codes, need);
    if (rc) return rc;
    c->bld_n = n;               // capacity in rows (grown by append when the hint was too small)
    c->bld_done = 0;
    c->frozen = false;
    return FSPANN_OK;
}
int build_append_impl(fspann_ctx* c, int64_t nrows, const void* rows, int dtype) {
    const int d = c->cfg.dim, TD = c->TD, W = c->W;
    const size_t esz = dtype == FSPANN_F64 ? 8 : 4;
    const int64_t chunk = 1 << 18;
    int rc;
    if (c->bld_done + nrows > c->bld_n) {       // more rows than the hint: grow the code buffer, keep what is coded
        const int64_t cap = std::max<int64_t>(c->bld_done + nrows, c->bld_n + c->bld_n / 2);
        if (cap >= (1LL << 31)) return fail(FSPANN_E_RANGE, "more than 2^31 - 1 rows");
        const size_t row = static_cast<size_t>(TD) * W * 8;
        void* bigger = nullptr;
        FSP_HIP(hipMalloc(&bigger, static_cast<size_t>(cap) * row + 256));
        if (hipMemcpyAsync(bigger, c->bld_codes.p, static_cast<size_t>(c->bld_done) * row, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipFree(bigger); return fail(FSPANN_E_DEVICE, "copy of the coded rows failed"); }
        (void)hipFree(c->bld_codes.p);
        c->bld_codes.p = bigger; c->bld_codes.bytes = static_cast<size_t>(cap) * row + 256; c->bld_codes.gen++;
        c->bld_n = cap;
    }
    if ((rc = ensure(c, c->ws_io[0], static_cast<size_t>(std::min(chunk, nrows)) * d * esz))) return rc;
    if ((rc = ensure(c, c->ws_io[2], static_cast<size_t>(std::min(chunk, nrows)) * 4))) return rc;
    uint64_t* codes_all = static_cast<uint64_t*>(c->bld_codes.p);
    std::vector<int32_t> bad(static_cast<size_t>(std::min(chunk, nrows)));
    for (int64_t s = 0; s < nrows; s += chunk) {
        const int64_t cn = std::min(chunk, nrows - s);
        FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, static_cast<const char*>(rows) + static_cast<size_t>(s) * d * esz,
                               static_cast<size_t>(cn) * d * esz, hipMemcpyHostToDevice, c->stream));
        uint64_t* cdst = codes_all + static_cast<size_t>(c->bld_done + s) * TD * W;
        rc = fspann_encode_dev(c, cn, c->ws_io[0].p, dtype, cdst, nullptr, static_cast<int32_t*>(c->ws_io[2].p));
        if (rc) return rc;
        FSP_HIP(hipMemcpyAsync(bad.data(), c->ws_io[2].p, static_cast<size_t>(cn) * 4, hipMemcpyDeviceToHost, c->stream));
        FSP_HIP(hipStreamSynchronize(c->stream));
        for (int64_t i = 0; i < cn; i++)
            if (bad[i]) { const long long hb = static_cast<long long>(c->bld_done + s + i); c->bld_done = -1; return fail(FSPANN_E_ARG, "Vector contains NaN/Inf (handle %lld)", hb); }
    }
    c->bld_done += nrows;
    return FSPANN_OK;
}
int build_finish_impl(fspann_ctx* c, const int32_t* order);
}  // namespace
extern "C" {

int fspann_build_index(fspann_ctx* c, int64_t n, const void* vectors, int dtype, const int32_t* order) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (!c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized");
    if (!vectors) return fail(FSPANN_E_NULL, "vector cannot be null");
    if (n <= 0) return fail(FSPANN_E_ARG, "n <= 0");
    if (c->n_ids < n) return fail(FSPANN_E_STATE, "set id metadata for at least n handles first");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    return guarded([&]() -> int {
        int rc = build_begin_impl(c, n);
        if (!rc) rc = build_append_impl(c, n, vectors, dtype);
        if (!rc) rc = build_finish_impl(c, order);
        c->bld_done = -1;
        return rc;
    });
}

// The same Setup with the rows handed over in pieces (include/fspann.h).
int fspann_build_begin(fspann_ctx* c, int64_t n_total) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (!c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized");
    if (n_total <= 0 || n_total >= (1LL << 31)) return fail(FSPANN_E_ARG, "n_hint out of range");
    return guarded([&]() -> int { return build_begin_impl(c, n_total); });
}
int fspann_build_append(fspann_ctx* c, int64_t n_rows, const void* rows, int dtype) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (c->bld_done < 0) return fail(FSPANN_E_STATE, "no build in progress (fspann_build_begin)");
    if (n_rows < 0) return fail(FSPANN_E_ARG, "n_rows < 0");
    if (n_rows == 0) return FSPANN_OK;
    if (!rows) return fail(FSPANN_E_NULL, "vector cannot be null");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    return guarded([&]() -> int { return build_append_impl(c, n_rows, rows, dtype); });
}
int fspann_build_finish(fspann_ctx* c, const int32_t* order) {
    CHECK_CTX(c);
    CHECK_UNSHARED(c);
    if (c->bld_done < 0) return fail(FSPANN_E_STATE, "no build in progress (fspann_build_begin)");
    if (c->bld_done == 0) return fail(FSPANN_E_STATE, "no rows appended");
    if (c->n_ids < c->bld_done) return fail(FSPANN_E_STATE, "%lld rows appended but id metadata covers %lld handles (fspann_set_id_meta)", (long long)c->bld_done, (long long)c->n_ids);
    return guarded([&]() -> int {
        c->bld_n = c->bld_done;     // the rows appended are the index
        const int rc = build_finish_impl(c, order);
        c->bld_done = -1;
        return rc;
    });
}

}  // extern "C"
namespace {
int build_finish_impl(fspann_ctx* c, const int32_t* order) {
    const int64_t n = c->bld_n;
    const int TD = c->TD, W = c->W, S = c->cfg.block_size;
    int rc;
    std::vector<int32_t> ord(static_cast<size_t>(n));
    if (order) {
        // order[] is a permutation of the n handles whose rows were appended: a handle >= n has no row (and no code),
        // a repeated handle would put an id twice into every table
        std::copy(order, order + n, ord.begin());
        std::vector<uint64_t> seen(static_cast<size_t>((n + 63) / 64), 0ull);
        for (int64_t i = 0; i < n; i++) {
            const int32_t h = ord[i];
            if (h < 0 || h >= n) return fail(FSPANN_E_ARG, "order[%lld] = %d is not a handle in [0,%lld)", (long long)i, h, (long long)n);
            if ((seen[static_cast<size_t>(h) >> 6] >> (h & 63)) & 1ull) return fail(FSPANN_E_ARG, "order[] holds handle %d twice", h);
            seen[static_cast<size_t>(h) >> 6] |= 1ull << (h & 63);
        }
    } else {  // first MIN_SAMPLE_SIZE-1 inserts are parked and flushed at finalize
        const int64_t ms = 1000;
        int64_t k = 0;
        if (n < ms) { for (int64_t i = 0; i < n; i++) ord[k++] = static_cast<int32_t>(i); }
        else {
            for (int64_t i = ms - 1; i < n; i++) ord[k++] = static_cast<int32_t>(i);
            for (int64_t i = 0; i < ms - 1; i++) ord[k++] = static_cast<int32_t>(i);
        }
    }
    // 1) the codes of every handle are in HBM (codes_all[h][td][w]); the host cut wants them on the host
    const bool gpu_cut = c->knob_gpu_cut != 0;
    uint64_t* codes_all = static_cast<uint64_t*>(c->bld_codes.p);
    std::vector<uint64_t> codes(gpu_cut ? 0 : static_cast<size_t>(n) * TD * W);
    if (!gpu_cut) {
        FSP_HIP(hipMemcpyAsync(codes.data(), codes_all, codes.size() * 8, hipMemcpyDeviceToHost, c->stream));
        FSP_HIP(hipStreamSynchronize(c->stream));
    }
    // 2) per table: order by (key, HashMap bucket, insertion position), cut blocks of S
    const int capf = java_final_cap_host(table_size_for(static_cast<int>(std::min<int64_t>(n, 1 << 30))), n);
    std::vector<uint32_t> bucket(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++) {
        uint32_t h = static_cast<uint32_t>(c->h_java_hash[ord[i]]);
        h ^= (h >> 16);
        bucket[i] = h & static_cast<uint32_t>(capf - 1);
    }
    // The closed form "iteration order = (bucket at the final capacity, insertion order)" holds only while no bin of
    // HashMap<String,BitSet>(staged.size()) (PIS:413, idx/GreedyPartitioner.java:45-48) is treeified: a put that finds 8
    // nodes in its bin (table >= 64) turns the bin into a red-black tree whose iteration order is not insertion order.
    // Replay the bin occupancy put by put, capacity stage by capacity stage; when a bin does treeify, the iteration order
    // of the staging map comes from the literal JDK model (host/java_hashmap.hpp) instead of the closed form.
    bool tree_bins = false;
    {
        int cap = table_size_for(static_cast<int>(std::min<int64_t>(n, 1 << 30)));
        int64_t thr = static_cast<int64_t>(static_cast<float>(cap) * 0.75f);
        std::vector<uint8_t> occ(static_cast<size_t>(cap), 0);
        for (int64_t i = 0; i < n && !tree_bins; i++) {
            uint32_t h = static_cast<uint32_t>(c->h_java_hash[ord[i]]);
            h ^= (h >> 16);
            uint8_t& o = occ[h & static_cast<uint32_t>(cap - 1)];
            if (o >= 8 && cap >= 64) { tree_bins = true; break; }
            if (o < 255) o++;
            if (i + 1 > thr && cap < (1 << 30)) {      // ++size > threshold -> resize(): every bin splits in two
                const int oldCap = cap;
                cap <<= 1;
                thr = (oldCap >= 16) ? (thr << 1) : static_cast<int64_t>(static_cast<float>(cap) * 0.75f);
                occ.assign(static_cast<size_t>(cap), 0);
                for (int64_t j = 0; j <= i; j++) {
                    uint32_t hj = static_cast<uint32_t>(c->h_java_hash[ord[j]]);
                    hj ^= (hj >> 16);
                    uint8_t& oj = occ[hj & static_cast<uint32_t>(cap - 1)];
                    if (oj < 255) oj++;
                }
            }
        }
    }
    std::vector<uint32_t> iter_pos;        // tree_bins: staged positions in the map's iteration order
    if (tree_bins) {
        jdk::HashMapModel<replay::KeyOrderView> mp(static_cast<int32_t>(std::min<int64_t>(n, INT32_MAX)), replay::KeyOrderView{c->decimal_ids});
        mp.reserve(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; i++) mp.put(ord[i], c->h_java_hash[ord[i]], i);
        if (mp.unmodelled)
            return fail(FSPANN_E_STATE, "a treeified HashMap bin of the staging map holds different ids with EQUAL String.hashCode and the ids are not "
                        "decimal ordinals: their String.compareTo order is unknown to the library, import the partitions with fspann_set_index instead");
        iter_pos.reserve(static_cast<size_t>(n));
        mp.for_each([&](int32_t, int64_t pos) { iter_pos.push_back(static_cast<uint32_t>(pos)); });
        // the host cut orders by (key, bucket, position): give it the iteration RANK as the "bucket" and it needs nothing else
        if (!gpu_cut) for (int64_t k = 0; k < n; k++) bucket[iter_pos[static_cast<size_t>(k)]] = static_cast<uint32_t>(k);
    }
    if (gpu_cut) {
        // ---- the cut on the GPU (build.hip.h): (bin, position) order once, then per table a stable radix sort by key + cut ----
        const int nblocks = static_cast<int>((n + kRsTile - 1) / kRsTile);
        const int64_t np = (n + S - 1) / S;
        const size_t kb = static_cast<size_t>(n) * 8, pb = static_cast<size_t>(n) * 4;
        auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
        // scratch: ord, bucket, perm0, 2 x keys, 2 x payload, hist, per-table outputs
        const size_t need = al(pb) * 3 + al(kb) * 2 + al(pb) * 2 + al(static_cast<size_t>(256) * nblocks * 4) + al(2 * 256 * 4) + al(np * 8) * 2 + al(np * W * 8) + al((np + 1) * 8) + al(pb);
        if ((rc = ensure(c, c->ws_io[3], need))) return rc;
        char* w = static_cast<char*>(c->ws_io[3].p);
        auto take = [&](size_t bytes) { char* q = w; w += al(bytes); return q; };
        int32_t* d_ord = reinterpret_cast<int32_t*>(take(pb));
        uint32_t* d_bucket = reinterpret_cast<uint32_t*>(take(pb));
        uint32_t* d_perm0 = reinterpret_cast<uint32_t*>(take(pb));
        uint64_t* d_key[2] = {reinterpret_cast<uint64_t*>(take(kb)), reinterpret_cast<uint64_t*>(take(kb))};
        uint32_t* d_pay[2] = {reinterpret_cast<uint32_t*>(take(pb)), reinterpret_cast<uint32_t*>(take(pb))};
        uint32_t* d_hist = reinterpret_cast<uint32_t*>(take(static_cast<size_t>(256) * nblocks * 4));
        uint32_t* d_tot = reinterpret_cast<uint32_t*>(take(2 * 256 * 4));      // digit totals of the radix passes, two arrays in turn
        int pass_no = 0;
        FSP_HIP(hipMemsetAsync(d_tot, 0, 2 * 256 * 4, c->stream));
        int64_t* d_min = reinterpret_cast<int64_t*>(take(np * 8));
        int64_t* d_max = reinterpret_cast<int64_t*>(take(np * 8));
        uint64_t* d_repo = reinterpret_cast<uint64_t*>(take(np * W * 8));
        int64_t* d_offo = reinterpret_cast<int64_t*>(take((np + 1) * 8));
        int32_t* d_idso = reinterpret_cast<int32_t*>(take(pb));
        FSP_HIP(hipMemcpyAsync(d_ord, ord.data(), pb, hipMemcpyHostToDevice, c->stream));
        FSP_HIP(hipMemcpyAsync(d_bucket, bucket.data(), pb, hipMemcpyHostToDevice, c->stream));
        const unsigned eg = static_cast<unsigned>((n + 255) / 256);
        // stable LSD radix sort of (key, payload) on the byte digits [p_lo, p_hi]; returns the buffer index holding the result
        auto radix = [&](int cur, int p_lo, int p_hi) -> int {
            for (int p = p_lo; p <= p_hi; p++) {
                uint32_t* tcur = d_tot + 256 * (pass_no & 1);
                uint32_t* tnext = d_tot + 256 * ((pass_no & 1) ^ 1);
                pass_no++;
                hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(kRsThreads), 0, c->stream, d_key[cur], n, 8 * p, d_hist, nblocks, tcur);
                hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, c->stream, d_hist, nblocks, tcur, tnext);
                hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(kRsThreads), 0, c->stream, d_key[cur], d_pay[cur], n, 8 * p, d_hist, nblocks,
                                   d_key[cur ^ 1], d_pay[cur ^ 1]);
                cur ^= 1;
            }
            return cur;
        };
        // (a) staged positions ordered by (bin at the map's final table length, position)
        int cur = 0;
        if (tree_bins) {                // a bin treeified: the iteration order of the staging map was computed by the JDK model
            FSP_HIP(hipMemcpyAsync(d_perm0, iter_pos.data(), pb, hipMemcpyHostToDevice, c->stream));
        } else {
            hipLaunchKernelGGL(build_bin_keys_kernel, dim3(eg), dim3(256), 0, c->stream, d_bucket, n, d_key[0], d_pay[0]);
            int capbits = 0;
            while ((1 << capbits) < capf) capbits++;
            cur = radix(0, 0, std::max(0, (capbits + 7) / 8 - 1));
            FSP_HIP(hipMemcpyAsync(d_perm0, d_pay[cur], pb, hipMemcpyDeviceToDevice, c->stream));
        }
        FSP_HIP(hipGetLastError());
        // (b) per table: keys of that sequence, stable sort by key over the bytes that can differ, cut
        const int sig = std::min(63, c->bits);                  // key bits [63 - sig, 62] carry code bits
        const int p_lo = (63 - sig) / 8, p_hi = 7;
        for (int td = 0; td < TD; td++) {
            hipLaunchKernelGGL(build_table_keys_kernel, dim3(eg), dim3(256), 0, c->stream, codes_all, TD, W, td, d_ord, d_perm0, n, d_key[0], d_pay[0]);
            cur = radix(0, p_lo, p_hi);
            hipLaunchKernelGGL(build_cut_kernel, dim3(eg), dim3(256), 0, c->stream, d_key[cur], d_pay[cur], d_ord, codes_all, TD, W, td, n, S, d_min, d_max,
                               d_repo, d_offo, d_idso);
            FSP_HIP(hipGetLastError());
            auto& mn = c->h_min[td]; auto& mx = c->h_max[td]; auto& rp = c->h_rep[td]; auto& of = c->h_off[td]; auto& ii = c->h_ids[td];
            mn.resize(np); mx.resize(np); rp.resize(static_cast<size_t>(np) * W); of.resize(np + 1); ii.resize(n);
            FSP_HIP(hipMemcpyAsync(mn.data(), d_min, np * 8, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipMemcpyAsync(mx.data(), d_max, np * 8, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipMemcpyAsync(rp.data(), d_repo, static_cast<size_t>(np) * W * 8, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipMemcpyAsync(of.data(), d_offo, (np + 1) * 8, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipMemcpyAsync(ii.data(), d_idso, pb, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipStreamSynchronize(c->stream));       // the outputs of this table are on the host before the scratch is reused
            c->h_table_set[td] = 1;
        }
        c->dev_index_dirty = true;
        return fspann_finalize(c);
    }
    struct Ent { int64_t key; uint32_t bucket; int32_t pos; };
    // host cut (FSPANN_GPU_CUT=0): one host thread per table
    const unsigned hw = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 16u));
    std::atomic<int> next_td{0};
    std::atomic<bool> worker_oom{false};
    auto worker = [&]() {
     try {
      std::vector<Ent> ents(static_cast<size_t>(n));
      for (int td = next_td.fetch_add(1); td < TD; td = next_td.fetch_add(1)) {
        for (int64_t i = 0; i < n; i++) {
            const uint64_t w0 = codes[(static_cast<size_t>(ord[i]) * TD + td) * W];
            // computeKey: code bit i -> key bit 62-i for i < 63
            uint64_t rev = 0;
            uint64_t x = w0;
            for (int b = 0; b < 64; b++) { rev = (rev << 1) | (x & 1); x >>= 1; }
            ents[i] = {static_cast<int64_t>(rev >> 1), bucket[i], static_cast<int32_t>(i)};
        }
        std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) {
            if (a.key != b.key) return a.key < b.key;
            if (a.bucket != b.bucket) return a.bucket < b.bucket;
            return a.pos < b.pos;
        });
        const int64_t np = (n + S - 1) / S;
        auto& mn = c->h_min[td]; auto& mx = c->h_max[td]; auto& rp = c->h_rep[td]; auto& of = c->h_off[td]; auto& ii = c->h_ids[td];
        mn.resize(np); mx.resize(np); rp.resize(static_cast<size_t>(np) * W); of.resize(np + 1); ii.resize(n);
        for (int64_t p = 0; p < np; p++) {
            const int64_t i0 = p * S, i1 = std::min<int64_t>(i0 + S, n);
            mn[p] = ents[i0].key;
            mx[p] = ents[i1 - 1].key;
            const int64_t mid = i0 + ((i1 - i0 - 1) >> 1);
            const int32_t rh = ord[ents[mid].pos];
            for (int w = 0; w < W; w++) rp[static_cast<size_t>(p) * W + w] = codes[(static_cast<size_t>(rh) * TD + td) * W + w];
            of[p] = i0;
            for (int64_t i = i0; i < i1; i++) ii[i] = ord[ents[i].pos];
        }
        of[np] = n;
        c->h_table_set[td] = 1;
      }
     } catch (...) { worker_oom = true; }
    };
    {
        std::vector<std::thread> pool;
        const unsigned nt = std::min<unsigned>(hw, static_cast<unsigned>(TD));
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    if (worker_oom) return fail(FSPANN_E_NOMEM, "out of host memory while cutting partitions");
    c->dev_index_dirty = true;
    return fspann_finalize(c);
}
}  // namespace

