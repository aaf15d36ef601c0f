// api_ext.hip.h — entry points of include/fspann.h added in round 3 (included at the end of fspann_api.hip: one translation unit).
//   * fspann_route_resolve_dev / fspann_search_store_finish_dev: queries whose HashMap<String,Long> bestScore would treeify a bin
//     (count = -1 after Route) are finished by the literal JDK model on the host (host/route_replay.hpp) — the reference answers
//     every query (PIS:619,690-693), so does the library;
//   * fspann_set_deleted: live mirror of metadata.isDeleted (PIS:739), no un-freeze, works while clones are alive.
// Product code: nothing here references oracle/.
#pragma once

namespace {

__global__ void set_deleted_bits_kernel(const int32_t* __restrict__ handles, int64_t n, int flag, uint32_t* __restrict__ bits) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t h = handles[i];
    if (flag) atomicOr(&bits[h >> 5], 1u << (h & 31));
    else atomicAnd(&bits[h >> 5], ~(1u << (h & 31)));
}

// Finish the flagged queries (count == -1) of a Route call on the host.  Synchronises the context's stream.
int resolve_unmodelled(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit, int64_t cap, int32_t* ids_dev,
                       int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev, int32_t* raw_dev, int64_t* resolved_out, int64_t* left_out) {
    if (resolved_out) *resolved_out = 0;
    if (left_out) *left_out = 0;
    FSP_HIP(hipStreamSynchronize(c->stream));
    std::vector<int32_t> cnt(static_cast<size_t>(nq));
    FSP_HIP(hipMemcpy(cnt.data(), count_dev, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> todo;
    for (int64_t i = 0; i < nq; i++)
        if (cnt[i] == kRouteUnmodelled) todo.push_back(i);
    if (todo.empty()) return FSPANN_OK;
    fspann_ctx* root = index_owner(c);
    const int TD = c->TD, W = c->W;
    const size_t cw = static_cast<size_t>(TD) * W;
    std::vector<uint64_t> codes(todo.size() * cw);
    for (size_t t = 0; t < todo.size(); t++)
        FSP_HIP(hipMemcpy(codes.data() + t * cw, codes_dev + static_cast<size_t>(todo[t]) * cw, cw * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> del;
    {
        std::lock_guard<std::mutex> dl(root->deleted_mu);
        if (root->d_deleted_bits.load(std::memory_order_acquire)) del = root->h_deleted_bits;
    }
    replay::IndexView v;
    v.TD = TD; v.W = W; v.S = c->cfg.block_size;
    v.min_key = &root->h_min; v.max_key = &root->h_max; v.rep = &root->h_rep; v.id_off = &root->h_off; v.ids = &root->h_ids;
    v.java_hash = root->h_java_hash.data(); v.decimal_ids = root->decimal_ids; v.deleted_bits = del.empty() ? nullptr : del.data();
    const int probes = effective_probes(c, probe_override);
    std::vector<replay::Result> res(todo.size());
    {
        std::atomic<size_t> next{0};
        std::atomic<bool> oom{false};
        auto work = [&]() {
            try {
                for (size_t t = next.fetch_add(1); t < todo.size(); t = next.fetch_add(1)) res[t] = replay::route_query(v, codes.data() + t * cw, probes, c->hard_cap);
            } catch (...) { oom = true; }
        };
        const unsigned nth = static_cast<unsigned>(std::min<size_t>(todo.size(), std::max(1u, std::min(16u, std::thread::hardware_concurrency()))));
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nth; i++) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        if (oom) return fail(FSPANN_E_NOMEM, "out of host memory");
    }
    int64_t resolved = 0, left = 0;
    for (size_t t = 0; t < todo.size(); t++) {
        const replay::Result& r = res[t];
        if (r.unmodelled) { left++; continue; }     // equal hashCodes of non-decimal ids inside a tree bin: String.compareTo unknown here
        const int64_t qi = todo[t];
        const int32_t n = static_cast<int32_t>(r.ids.size());
        const int32_t nout = static_cast<int32_t>(std::min<int64_t>(std::min<int64_t>(limit, n), cap));
        if (nout > 0) {
            FSP_HIP(hipMemcpy(ids_dev + qi * cap, r.ids.data(), static_cast<size_t>(nout) * 4, hipMemcpyHostToDevice));
            if (score_dev) FSP_HIP(hipMemcpy(score_dev + qi * cap, r.score.data(), static_cast<size_t>(nout) * 4, hipMemcpyHostToDevice));
        }
        FSP_HIP(hipMemcpy(count_dev + qi, &nout, 4, hipMemcpyHostToDevice));
        if (kept_dev) FSP_HIP(hipMemcpy(kept_dev + qi, &n, 4, hipMemcpyHostToDevice));
        if (raw_dev) FSP_HIP(hipMemcpy(raw_dev + qi, &r.raw_seen, 4, hipMemcpyHostToDevice));
        resolved++;
    }
    if (resolved) {       // they are no longer "unmodelled": take them off the context's counter
        int32_t v0 = 0;
        FSP_HIP(hipMemcpy(&v0, c->d_unmodelled, 4, hipMemcpyDeviceToHost));
        v0 = static_cast<int32_t>(std::max<int64_t>(0, static_cast<int64_t>(v0) - resolved));
        FSP_HIP(hipMemcpy(c->d_unmodelled, &v0, 4, hipMemcpyHostToDevice));
    }
    if (resolved_out) *resolved_out = resolved;
    if (left_out) *left_out = left;
    return FSPANN_OK;
}

}  // namespace

extern "C" {

int fspann_route_resolve_dev(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit, int64_t cap, int32_t* ids_dev,
                             int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev, int32_t* raw_seen_dev, int64_t* resolved) {
    CHECK_CTX(c);
    if (resolved) *resolved = 0;
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    if (!codes_dev) return fail(FSPANN_E_STATE, "MSANNP violation: QueryToken missing BitSet codes");
    if (!ids_dev || !count_dev) return fail(FSPANN_E_NULL, "output buffer is null");
    if (limit <= 0 || cap <= 0) return fail(FSPANN_E_ARG, "limit and cap must be > 0");
    return guarded([&]() -> int {
        int64_t left = 0;
        return resolve_unmodelled(c, nq, codes_dev, probe_override, limit, cap, ids_dev, score_dev, count_dev, kept_dev, raw_seen_dev, resolved, &left);
    });
}

int fspann_search_store_finish_dev(fspann_ctx* c, int64_t nq, const void* q_dev, int q_dtype, int probe_override, int64_t B, int k,
                                   int32_t* out_ids_dev, double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev,
                                   int32_t* sel_ids_dev, int32_t* sel_count_dev, int64_t* resolved) {
    CHECK_CTX(c);
    if (resolved) *resolved = 0;
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (!c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (nq < 0 || B <= 0 || B > INT32_MAX) return fail(FSPANN_E_ARG, "nq < 0 or B out of range");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");
    if (nq == 0) return FSPANN_OK;
    // the work area of the fspann_search_store_dev call this one completes (codes, and F_q when the caller did not ask for it)
    const size_t cb = (static_cast<size_t>(nq) * c->TD * c->W * 8 + 255) & ~size_t(255);
    const size_t ib = (static_cast<size_t>(nq) * B * 4 + 255) & ~size_t(255);
    const size_t nb = (static_cast<size_t>(nq) * 4 + 255) & ~size_t(255);
    if (!c->ws_search.p || c->ws_search.bytes < cb + ib + 2 * nb) return fail(FSPANN_E_STATE, "no fspann_search_store_dev call of this size precedes");
    char* w = static_cast<char*>(c->ws_search.p);
    const uint64_t* codes = reinterpret_cast<const uint64_t*>(w);
    int32_t* sel = sel_ids_dev ? sel_ids_dev : reinterpret_cast<int32_t*>(w + cb);
    int32_t* cnt = sel_count_dev ? sel_count_dev : reinterpret_cast<int32_t*>(w + cb + ib);
    return guarded([&]() -> int {
        int64_t done = 0, left = 0;
        int rc = resolve_unmodelled(c, nq, codes, probe_override, static_cast<int32_t>(B), B, sel, nullptr, cnt, nullptr, nullptr, &done, &left);
        if (rc) return rc;
        if (resolved) *resolved = done;
        if (done == 0) return FSPANN_OK;
        // the completed queries now have an F_q: score the batch again (rare path: the whole batch, same results for the others)
        return fspann_refine_store_dev(c, nq, q_dev, q_dtype, B, sel, cnt, k, out_ids_dev, out_dist_dev, out_count_dev, scored_dev);
    });
}

int fspann_set_deleted(fspann_ctx* c, const int32_t* handles, int64_t n, int flag) {
    CHECK_CTX(c);
    if (n < 0) return fail(FSPANN_E_ARG, "n < 0");
    if (n == 0) return FSPANN_OK;
    if (!handles) return fail(FSPANN_E_NULL, "handles is null");
    fspann_ctx* root = index_owner(c);
    if (root->n_ids <= 0) return fail(FSPANN_E_STATE, "id metadata not set (fspann_set_id_meta)");
    for (int64_t i = 0; i < n; i++)
        if (handles[i] < 0 || handles[i] >= root->n_ids) return fail(FSPANN_E_ARG, "handle %d out of range [0,%lld)", handles[i], (long long)root->n_ids);
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> dl(root->deleted_mu);
        const size_t words = static_cast<size_t>((root->n_ids + 31) / 32);
        if (root->h_deleted_bits.size() != words) root->h_deleted_bits.assign(words, 0u);
        uint32_t* bits = root->d_deleted_bits.load(std::memory_order_acquire);
        if (!bits) {
            if (!flag) return FSPANN_OK;                        // nothing is deleted: nothing to undelete
            // first deletion on this index: allocate the mirror (all clear) and publish it; kernels enqueued from now on read it
            FSP_HIP(hipMalloc(&bits, words * 4));
            if (hipMemset(bits, 0, words * 4) != hipSuccess) { (void)hipFree(bits); return fail(FSPANN_E_DEVICE, "hipMemset failed"); }
            root->d_deleted_bits.store(bits, std::memory_order_release);
        }
        for (int64_t i = 0; i < n; i++) {
            const int32_t h = handles[i];
            if (flag) root->h_deleted_bits[h >> 5] |= 1u << (h & 31);
            else root->h_deleted_bits[h >> 5] &= ~(1u << (h & 31));
        }
        // On THIS context's stream, then waited for: Route calls enqueued (on any context of the family) after this call
        // returns see the change; calls already in flight see the old or the new flag — as a JVM query racing a delete does.
        int rc = ensure(c, c->ws_io[6], static_cast<size_t>(n) * 4);
        if (rc) return rc;
        FSP_HIP(hipMemcpyAsync(c->ws_io[6].p, handles, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(set_deleted_bits_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, c->stream,
                           static_cast<const int32_t*>(c->ws_io[6].p), n, flag ? 1 : 0, bits);
        FSP_HIP(hipGetLastError());
        FSP_HIP(hipStreamSynchronize(c->stream));
        return FSPANN_OK;
    });
}

}  // extern "C"
