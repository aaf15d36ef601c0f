// api_refine.hip.h — Refine entry points: fspann_refine[_dev], the resident plaintext store, the one-call search, scan timing (QSI:238-316,364-372)
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

namespace {

template <typename TC, typename TQ, int DC, bool GATHER>
int launch_refine_dc(fspann_ctx* c, int64_t nq, const TQ* q, const TC* cand, int64_t B, const int32_t* cand_ids,
                     const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist, int32_t* out_count,
                     int32_t* scored) {
    constexpr int VN = VecOf<TC>::N;
    const int d = c->cfg.dim;
    const int nchunks = static_cast<int>((B + kRefRows - 1) / kRefRows);
    RefinePartial* partial = nullptr;
    int32_t* pcnt = nullptr;
    // Long candidate lists (B in the thousands, k = 100: the reference's shipped profiles): a workgroup walks a RUN of consecutive
    // chunks of one query and keeps its best k in LDS (refine_topk_running) — one list per run instead of one per chunk.  With at
    // least half a grid of queries a run is the whole query (no merge kernel at all); fewer queries are cut into as many runs as
    // fill the grid (one query: one chunk per workgroup, as before).
    const int stream_wgs_m = (c->knob_refine_stream >= 0) ? std::min(c->knob_refine_stream, 4) : 4;
    int npieces = 0, cpp = 0;
    if (!GATHER && nchunks > 1 && k > kRefFilterMaxK && k <= kRunMaxK && c->knob_refine_run && stream_wgs_m > 0 && DC * sizeof(TC) == 128 &&
        (d % VN == 0) && ((reinterpret_cast<uintptr_t>(cand) & 15) == 0) && nq * nchunks < (int64_t(1) << 31)) {   // (= the streaming scan will run)
        const int64_t slots = static_cast<int64_t>(c->num_cus) * stream_wgs_m;
        int np = (nq * 2 >= slots) ? 1 : static_cast<int>(std::min<int64_t>(nchunks, (slots + nq - 1) / std::max<int64_t>(nq, 1)));
        cpp = (nchunks + np - 1) / np;
        npieces = (nchunks + cpp - 1) / cpp;
        if (nq * npieces >= (int64_t(1) << 31)) { npieces = 0; cpp = 0; }
    }
    if (npieces > 1) {
        const size_t pb = static_cast<size_t>(nq) * npieces * k * sizeof(RefinePartial);
        const size_t cb = static_cast<size_t>(nq) * npieces * 2 * 4;
        int rc = ensure(c, c->ws_refine, pb + cb + 64);
        if (rc) return rc;
        partial = static_cast<RefinePartial*>(c->ws_refine.p);
        pcnt = reinterpret_cast<int32_t*>(static_cast<char*>(c->ws_refine.p) + ((pb + 15) & ~size_t(15)));
    } else if (nchunks > 1 && npieces == 0) {
        const size_t pb = static_cast<size_t>(nq) * nchunks * k * sizeof(RefinePartial);
        const size_t cb = static_cast<size_t>(nq) * nchunks * 2 * 4;
        int rc = ensure(c, c->ws_refine, pb + cb + 64);
        if (rc) return rc;
        partial = static_cast<RefinePartial*>(c->ws_refine.p);
        pcnt = reinterpret_cast<int32_t*>(static_cast<char*>(c->ws_refine.p) + ((pb + 15) & ~size_t(15)));
    }
    const bool vec = (d % VN == 0) && ((reinterpret_cast<uintptr_t>(cand) & 15) == 0);
    const size_t lds = std::max<size_t>(static_cast<size_t>(kRefRows) * (vec ? DC + VN : DC + 1) * sizeof(TC), static_cast<size_t>(kRefRows) * 16);
    const unsigned grid = static_cast<unsigned>(nq * nchunks);
    const RefineArgs<TC, TQ> ra{q, cand, GATHER ? c->store_n : 0, B, d, cand_ids, cand_count, k, nchunks, out_ids, out_dist, out_count, scored, partial, pcnt, npieces, cpp, c->dbg_route};
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        if (c->rt_on && (c->rt_seen++ % c->rt_every) == 0 && c->rt_used + 2 <= c->rt_events.size()) {   // start/stop events attached to this very dispatch
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kRefRows), lds, c->stream, c->rt_events[c->rt_used], c->rt_events[c->rt_used + 1], 0, ra);
            c->rt_used += 2;
        } else {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kRefRows), lds, c->stream, ra);
        }
        return FSPANN_OK;
    };
    int lrc = FSPANN_OK;
    bool streamed = false;
    // workgroups per CU of the streaming scan: dense blocks run at 128 registers (4 per CU: a 1024-query batch is exactly one
    // unit per workgroup on 256 CUs), the store gather at 3 per CU; FSPANN_REFINE_STREAM overrides, 0 = per-query scan
    const int stream_wgs = (c->knob_refine_stream >= 0) ? std::min(c->knob_refine_stream, GATHER ? 3 : 4) : (GATHER ? 3 : 4);
    if constexpr (DC * sizeof(TC) == 128) if (vec && stream_wgs > 0 && nq * nchunks < (int64_t(1) << 31)) {
        // the scan as a stream: knob_refine_stream workgroups per CU, each walking several (query, chunk) units with the loads
        // of the next tile in flight across unit boundaries (refine_stream_run)
        const int64_t units = npieces > 0 ? nq * npieces : nq * nchunks;
        const unsigned sgrid = static_cast<unsigned>(std::min<int64_t>(units, static_cast<int64_t>(c->num_cus) * stream_wgs));
        const bool timed = c->rt_on && (c->rt_seen++ % c->rt_every) == 0 && c->rt_used + 2 <= c->rt_events.size();
        hipEvent_t ev0 = timed ? c->rt_events[c->rt_used] : nullptr, ev1 = timed ? c->rt_events[c->rt_used + 1] : nullptr;
        if (timed) c->rt_used += 2;
        bool fixed = false;
        if constexpr (std::is_same<TC, float>::value && std::is_same<TQ, float>::value && DC == 32) {
            if (c->refine_fix_dev && nchunks == 1) {
                // the batch's Route ran with a hand-over buffer: the scan's workgroups finish its PENDING queries first (tick.hip.h)
                auto fk = refine_stream_fix_kernel<GATHER>;
                const size_t flds = std::max(lds, c->refine_fix_lds);
                const unsigned abit = GATHER ? 4096u : 8192u;
                if (!(c->attr_mask & abit)) {
                    FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                    c->attr_mask |= abit;
                }
                if (timed) hipExtLaunchKernelGGL(fk, dim3(sgrid), dim3(kRefRows), flds, c->stream, ev0, ev1, 0, ra, nq, static_cast<const RouteParams*>(c->refine_fix_dev));
                else hipLaunchKernelGGL(fk, dim3(sgrid), dim3(kRefRows), flds, c->stream, ra, nq, static_cast<const RouteParams*>(c->refine_fix_dev));
                fixed = true;
                c->refine_fix_used = true;
            }
        }
        if (!fixed) {
            if constexpr (!GATHER) {
                if (npieces > 0) {
                    auto kern = refine_stream_kernel<TC, TQ, DC, false, true>;
                    if (timed) hipExtLaunchKernelGGL(kern, dim3(sgrid), dim3(kRefRows), lds, c->stream, ev0, ev1, 0, ra, nq);
                    else hipLaunchKernelGGL(kern, dim3(sgrid), dim3(kRefRows), lds, c->stream, ra, nq);
                    fixed = true;
                }
            }
        }
        if (!fixed) {
            auto kern = refine_stream_kernel<TC, TQ, DC, GATHER>;
            if (timed) hipExtLaunchKernelGGL(kern, dim3(sgrid), dim3(kRefRows), lds, c->stream, ev0, ev1, 0, ra, nq);
            else hipLaunchKernelGGL(kern, dim3(sgrid), dim3(kRefRows), lds, c->stream, ra, nq);
        }
        streamed = true;
    }
    if (!streamed) lrc = vec ? launch(refine_scan_kernel<TC, TQ, DC, true, GATHER>) : launch(refine_scan_kernel<TC, TQ, DC, false, GATHER>);
    if (lrc) return lrc;
    FSP_HIP(hipGetLastError());
    if (npieces > 0 && !streamed) return fail(FSPANN_E_STATE, "refine: the running top-k was planned but the streaming scan did not run");
    const int nlists = npieces > 0 ? npieces : nchunks;      // partial lists per query (runs of chunks, or chunks)
    if (nlists > 1) {
        const int nchunks = nlists;                          // (the merge below: one list per run)
        // all keys of a query's partial lists in LDS when they fit (two workgroups per CU at least)
        const size_t mlds = static_cast<size_t>(nchunks) * k * 8 + static_cast<size_t>(nchunks) * 4 + 16;
        if (mlds <= 72 * 1024) {
            auto mk = refine_merge_kernel<true>;
            if (mlds > 64 * 1024 && !(c->attr_mask & 16384u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mk), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
                c->attr_mask |= 16384u;
            }
            hipLaunchKernelGGL(mk, dim3(static_cast<unsigned>(nq)), dim3(256), mlds, c->stream, partial, pcnt,
                               nchunks, k, out_ids, out_dist, out_count, scored);
        } else {
            hipLaunchKernelGGL(refine_merge_kernel<false>, dim3(static_cast<unsigned>(nq)), dim3(256), static_cast<size_t>(nchunks) * 4 + 16, c->stream, partial, pcnt,
                               nchunks, k, out_ids, out_dist, out_count, scored);
        }
        FSP_HIP(hipGetLastError());
    }
    return FSPANN_OK;
}

template <typename TC, typename TQ, bool GATHER>
int launch_refine_t(fspann_ctx* c, int64_t nq, const TQ* q, const TC* cand, int64_t B, const int32_t* cand_ids,
                    const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist, int32_t* out_count,
                    int32_t* scored) {
    constexpr int DC0 = (sizeof(TC) == 4) ? 32 : 16;
    const int dc_env = c->knob_refine_dc;
    if (dc_env == DC0 * 2) return launch_refine_dc<TC, TQ, DC0 * 2, GATHER>(c, nq, q, cand, B, cand_ids, cand_count, k, out_ids, out_dist, out_count, scored);
    if (dc_env == DC0 * 4) return launch_refine_dc<TC, TQ, DC0 * 4, GATHER>(c, nq, q, cand, B, cand_ids, cand_count, k, out_ids, out_dist, out_count, scored);
    return launch_refine_dc<TC, TQ, DC0, GATHER>(c, nq, q, cand, B, cand_ids, cand_count, k, out_ids, out_dist, out_count, scored);
}


}  // namespace

extern "C" {

// ---- refine ---------------------------------------------------------------------------
int fspann_refine_dev(fspann_ctx* c, int64_t nq, const void* q_dev, int q_dtype, const void* cand_dev, int cand_dtype,
                      int64_t B, const int32_t* cand_ids_dev, const int32_t* cand_count_dev, int k, int32_t* out_ids_dev,
                      double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev) {
    CHECK_CTX(c);
    if (nq < 0 || B <= 0) return fail(FSPANN_E_ARG, "nq < 0 or B <= 0");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");  // QueryTokenFactory.java:65
    if (nq == 0) return FSPANN_OK;
    if (!q_dev || !cand_dev || !cand_ids_dev || !cand_count_dev || !out_ids_dev || !out_dist_dev || !out_count_dev)
        return fail(FSPANN_E_NULL, "refine buffer is null");
#define FSP_REF(TC, TQ)                                                                                          \
    return launch_refine_t<TC, TQ, false>(c, nq, static_cast<const TQ*>(q_dev), static_cast<const TC*>(cand_dev), B, \
                                   cand_ids_dev, cand_count_dev, k, out_ids_dev, out_dist_dev, out_count_dev,    \
                                   scored_dev)
    if (cand_dtype == FSPANN_F32 && q_dtype == FSPANN_F32) FSP_REF(float, float);
    if (cand_dtype == FSPANN_F32 && q_dtype == FSPANN_F64) FSP_REF(float, double);
    if (cand_dtype == FSPANN_F64 && q_dtype == FSPANN_F32) FSP_REF(double, float);
    if (cand_dtype == FSPANN_F64 && q_dtype == FSPANN_F64) FSP_REF(double, double);
#undef FSP_REF
    return fail(FSPANN_E_ARG, "unknown dtype");
}

int fspann_refine(fspann_ctx* c, int64_t nq, const void* q, const void* cand, int dtype, int64_t B,
                  const int32_t* cand_ids, const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist,
                  int32_t* out_count, int32_t* scored) {
    CHECK_CTX(c);
    if (nq < 0 || B <= 0) return fail(FSPANN_E_ARG, "nq < 0 or B <= 0");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");
    if (nq == 0) return FSPANN_OK;
    if (!q || !cand || !cand_ids || !cand_count || !out_ids || !out_dist || !out_count) return fail(FSPANN_E_NULL, "refine buffer is null");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    const size_t esz = dtype == FSPANN_F64 ? 8 : 4;
    const int d = c->cfg.dim;
    const size_t qb = static_cast<size_t>(nq) * d * esz, cb = static_cast<size_t>(nq) * B * d * esz;
    const size_t ib = static_cast<size_t>(nq) * B * 4, nb = static_cast<size_t>(nq) * 4;
    const size_t ob_i = static_cast<size_t>(nq) * k * 4, ob_d = static_cast<size_t>(nq) * k * 8;
    int rc;
    // Small calls: query, ids and counts go up in ONE pinned block and every output comes down in one (the candidate rows keep
    // their own copy straight from the caller's buffer): three transfers and one synchronisation instead of eight and one.
    {
        auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
        const size_t up = al(qb) + al(ib) + al(nb), down = al(ob_d) + al(ob_i) + al(2 * nb);
        if (std::max(up, down) <= kPinBytes && pin_block(c)) {
            unsigned char* hp = static_cast<unsigned char*>(c->h_pin);
            // a handful of queries: query, ids and counts are read from the mapped pinned block and the results written into it by the
            // kernel itself (zero_copy_ok); the rows always travel by a copy command (a workgroup reading 256 KB over the bus is slow)
            // (short lists only: with thousands of candidate ids every chunk's workgroup would fetch its ids over the bus first — SIFT_P4_FAST,
            // B = 8 000: 439 against 405 us per call)
            const bool zc = up <= 8192 && al(up) + down <= kPinBytes && zero_copy_ok(c, nq);
            if (!zc) {
                if ((rc = ensure(c, c->ws_io[0], up))) return rc;
                if ((rc = ensure(c, c->ws_io[4], down))) return rc;
            }
            if ((rc = ensure(c, c->ws_io[1], cb))) return rc;
            unsigned char* du = zc ? static_cast<unsigned char*>(c->d_pin) : static_cast<unsigned char*>(c->ws_io[0].p);
            unsigned char* dd = zc ? static_cast<unsigned char*>(c->d_pin) + al(up) : static_cast<unsigned char*>(c->ws_io[4].p);
            unsigned char* hres = zc ? hp + al(up) : hp;
            std::memcpy(hp, q, qb);
            std::memcpy(hp + al(qb), cand_ids, ib);
            std::memcpy(hp + al(qb) + al(ib), cand_count, nb);
            if (!zc) FSP_HIP(hipMemcpyAsync(du, hp, up, hipMemcpyHostToDevice, c->stream));
            FSP_HIP(hipMemcpyAsync(c->ws_io[1].p, cand, cb, hipMemcpyHostToDevice, c->stream));
            int32_t* cnt_out = reinterpret_cast<int32_t*>(dd + al(ob_d) + al(ob_i));
            rc = fspann_refine_dev(c, nq, du, dtype, c->ws_io[1].p, dtype, B, reinterpret_cast<int32_t*>(du + al(qb)),
                                   reinterpret_cast<int32_t*>(du + al(qb) + al(ib)), k, reinterpret_cast<int32_t*>(dd + al(ob_d)),
                                   reinterpret_cast<double*>(dd), cnt_out, cnt_out + nq);
            if (rc) return rc;
            if (!zc) FSP_HIP(hipMemcpyAsync(hp, dd, down, hipMemcpyDeviceToHost, c->stream));   // (stream order: the way up has been read by then)
            FSP_HIP(hipStreamSynchronize(c->stream));
            std::memcpy(out_dist, hres, ob_d);
            std::memcpy(out_ids, hres + al(ob_d), ob_i);
            std::memcpy(out_count, hres + al(ob_d) + al(ob_i), nb);
            if (scored) std::memcpy(scored, hres + al(ob_d) + al(ob_i) + nb, nb);
            return FSPANN_OK;
        }
    }
    if ((rc = ensure(c, c->ws_io[0], qb))) return rc;
    if ((rc = ensure(c, c->ws_io[1], cb))) return rc;
    if ((rc = ensure(c, c->ws_io[2], ib))) return rc;
    if ((rc = ensure(c, c->ws_io[3], nb * 3))) return rc;
    if ((rc = ensure(c, c->ws_io[4], ob_i))) return rc;
    if ((rc = ensure(c, c->ws_io[5], ob_d))) return rc;
    int32_t* cnts = static_cast<int32_t*>(c->ws_io[3].p);
    FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, q, qb, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipMemcpyAsync(c->ws_io[1].p, cand, cb, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipMemcpyAsync(c->ws_io[2].p, cand_ids, ib, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipMemcpyAsync(cnts, cand_count, nb, hipMemcpyHostToDevice, c->stream));
    rc = fspann_refine_dev(c, nq, c->ws_io[0].p, dtype, c->ws_io[1].p, dtype, B, static_cast<int32_t*>(c->ws_io[2].p), cnts, k,
                           static_cast<int32_t*>(c->ws_io[4].p), static_cast<double*>(c->ws_io[5].p), cnts + nq, cnts + 2 * nq);
    if (rc) return rc;
    FSP_HIP(hipMemcpyAsync(out_ids, c->ws_io[4].p, ob_i, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(out_dist, c->ws_io[5].p, ob_d, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(out_count, cnts + nq, nb, hipMemcpyDeviceToHost, c->stream));
    if (scored) FSP_HIP(hipMemcpyAsync(scored, cnts + 2 * nq, nb, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    return FSPANN_OK;
}

void* fspann_host_buffer(fspann_ctx* c, size_t bytes) {
    if (!c) { fail(FSPANN_E_NULL, "ctx is null"); return nullptr; }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); fail(FSPANN_E_DEVICE, "hipSetDevice(%d) failed", c->device); return nullptr; }
    if (bytes == 0) bytes = 16;
    if (bytes > c->h_rows_bytes) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();      // nothing in flight may still read the old block
        if (c->h_rows) (void)hipHostFree(c->h_rows);
        c->h_rows = nullptr; c->h_rows_bytes = 0;
        const size_t want = (bytes + 4095) & ~size_t(4095);
        if (hipHostMalloc(&c->h_rows, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_rows = nullptr;
            fail(FSPANN_E_NOMEM, "cannot pin %zu bytes of host memory", want);
            return nullptr;
        }
        c->h_rows_bytes = want;
    }
    return c->h_rows;
}

// ---- plaintext store (test / bench harness) ----------------------------------------------
int fspann_store_set(fspann_ctx* c, int64_t n, const void* vectors, int dtype) {
    CHECK_CTX(c);
    if (c->share_children.load() > 0) return fail(FSPANN_E_STATE, "the store is shared with %d clone(s): destroy them first", c->share_children.load());
    if (!vectors) return fail(FSPANN_E_NULL, "vectors is null");
    if (n <= 0) return fail(FSPANN_E_ARG, "n <= 0");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    const size_t bytes = static_cast<size_t>(n) * c->cfg.dim * (dtype == FSPANN_F64 ? 8 : 4);
    FSP_HIP(hipStreamSynchronize(c->stream));
    if (c->store_owned) free_dev(c->d_store);
    c->d_store = nullptr;
    c->store_n = 0;
    FSP_HIP(hipMalloc(&c->d_store, bytes));
    c->store_owned = true;
    FSP_HIP(hipMemcpy(c->d_store, vectors, bytes, hipMemcpyHostToDevice));
    c->store_dtype = dtype;
    c->store_n = n;
    return FSPANN_OK;
}

// The same store over rows that already live in HBM (caller-owned, e.g. a tensor): no copy; the caller keeps the
// memory alive and unchanged while the context refers to it (until the next store_set / store_attach / ctx_destroy).
int fspann_store_attach_dev(fspann_ctx* c, int64_t n, const void* vectors_dev, int dtype) {
    CHECK_CTX(c);
    if (c->share_children.load() > 0) return fail(FSPANN_E_STATE, "the store is shared with %d clone(s): destroy them first", c->share_children.load());
    if (!vectors_dev) return fail(FSPANN_E_NULL, "vectors is null");
    if (n <= 0) return fail(FSPANN_E_ARG, "n <= 0");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    if (reinterpret_cast<uintptr_t>(vectors_dev) & 15) return fail(FSPANN_E_ARG, "store rows must be 16-byte aligned");
    FSP_HIP(hipStreamSynchronize(c->stream));
    if (c->store_owned) free_dev(c->d_store);
    c->d_store = const_cast<void*>(vectors_dev);
    c->store_owned = false;
    c->store_dtype = dtype;
    c->store_n = n;
    return FSPANN_OK;
}

// Refine straight from the resident store: row j of query qi is store[cand_ids[qi*B + j]].  Same kernel as
// fspann_refine_dev with the row address taken from the id (no [nq][B][dim] staging copy).
int fspann_refine_store_dev(fspann_ctx* c, int64_t nq, const void* q_dev, int q_dtype, int64_t B,
                            const int32_t* cand_ids_dev, const int32_t* cand_count_dev, int k, int32_t* out_ids_dev,
                            double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev) {
    CHECK_CTX(c);
    if (!c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (nq < 0 || B <= 0) return fail(FSPANN_E_ARG, "nq < 0 or B <= 0");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");  // QueryTokenFactory.java:65
    if (nq == 0) return FSPANN_OK;
    if (!q_dev || !cand_ids_dev || !cand_count_dev || !out_ids_dev || !out_dist_dev || !out_count_dev)
        return fail(FSPANN_E_NULL, "refine buffer is null");
#define FSP_REF(TC, TQ)                                                                                            \
    return launch_refine_t<TC, TQ, true>(c, nq, static_cast<const TQ*>(q_dev), static_cast<const TC*>(c->d_store), B, \
                                         cand_ids_dev, cand_count_dev, k, out_ids_dev, out_dist_dev, out_count_dev, \
                                         scored_dev)
    if (c->store_dtype == FSPANN_F32 && q_dtype == FSPANN_F32) FSP_REF(float, float);
    if (c->store_dtype == FSPANN_F32 && q_dtype == FSPANN_F64) FSP_REF(float, double);
    if (c->store_dtype == FSPANN_F64 && q_dtype == FSPANN_F32) FSP_REF(double, float);
    if (c->store_dtype == FSPANN_F64 && q_dtype == FSPANN_F64) FSP_REF(double, double);
#undef FSP_REF
    return fail(FSPANN_E_ARG, "unknown dtype");
}

int fspann_refine_store(fspann_ctx* c, int64_t nq, const void* q, int q_dtype, int64_t B, const int32_t* cand_ids,
                        const int32_t* cand_count, int k, int32_t* out_ids, double* out_dist, int32_t* out_count,
                        int32_t* scored) {
    CHECK_CTX(c);
    if (!c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (nq < 0 || B <= 0) return fail(FSPANN_E_ARG, "nq < 0 or B <= 0");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");
    if (nq == 0) return FSPANN_OK;
    if (!q || !cand_ids || !cand_count || !out_ids || !out_dist || !out_count) return fail(FSPANN_E_NULL, "refine buffer is null");
    if (q_dtype != FSPANN_F32 && q_dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", q_dtype);
    const size_t qb = static_cast<size_t>(nq) * c->cfg.dim * (q_dtype == FSPANN_F64 ? 8 : 4);
    const size_t ib = static_cast<size_t>(nq) * B * 4, nb = static_cast<size_t>(nq) * 4;
    const size_t ob_i = static_cast<size_t>(nq) * k * 4, ob_d = static_cast<size_t>(nq) * k * 8;
    int rc;
    if ((rc = ensure(c, c->ws_io[0], qb))) return rc;
    if ((rc = ensure(c, c->ws_io[2], ib))) return rc;
    if ((rc = ensure(c, c->ws_io[3], nb * 3))) return rc;
    if ((rc = ensure(c, c->ws_io[4], ob_i))) return rc;
    if ((rc = ensure(c, c->ws_io[5], ob_d))) return rc;
    int32_t* cnts = static_cast<int32_t*>(c->ws_io[3].p);
    FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, q, qb, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipMemcpyAsync(c->ws_io[2].p, cand_ids, ib, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipMemcpyAsync(cnts, cand_count, nb, hipMemcpyHostToDevice, c->stream));
    rc = fspann_refine_store_dev(c, nq, c->ws_io[0].p, q_dtype, B, static_cast<int32_t*>(c->ws_io[2].p), cnts, k,
                                 static_cast<int32_t*>(c->ws_io[4].p), static_cast<double*>(c->ws_io[5].p), cnts + nq, cnts + 2 * nq);
    if (rc) return rc;
    FSP_HIP(hipMemcpyAsync(out_ids, c->ws_io[4].p, ob_i, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(out_dist, c->ws_io[5].p, ob_d, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(out_count, cnts + nq, nb, hipMemcpyDeviceToHost, c->stream));
    if (scored) FSP_HIP(hipMemcpyAsync(scored, cnts + 2 * nq, nb, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    return FSPANN_OK;
}

// QueryServiceImpl.search for a batch, all three stages in stream order with one call: TokenGen codes (encode),
// Route with limit = B (stage A.5; counters not produced, so the bounded select may run), Refine from the resident store.
// The adaptive retry (QSI:327-337) stays with the caller: out_count / scored tell it when to call again with
// probe_override = 10.  sel_ids_dev / sel_count_dev (optional) receive F_q.
int fspann_search_store_dev(fspann_ctx* c, int64_t nq, const void* q_dev, int q_dtype, int probe_override, int64_t B, int k,
                            int32_t* out_ids_dev, double* out_dist_dev, int32_t* out_count_dev, int32_t* scored_dev,
                            int32_t* sel_ids_dev, int32_t* sel_count_dev, int32_t* bad_dev) {
    CHECK_CTX(c);
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (!c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (nq < 0 || B <= 0 || B > INT32_MAX) return fail(FSPANN_E_ARG, "nq < 0 or B out of range");
    if (k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");
    if (nq == 0) return FSPANN_OK;
    const size_t cb = (static_cast<size_t>(nq) * c->TD * c->W * 8 + 255) & ~size_t(255);
    const size_t ib = (static_cast<size_t>(nq) * B * 4 + 255) & ~size_t(255);
    const size_t nb = (static_cast<size_t>(nq) * 4 + 255) & ~size_t(255);
    int rc;
    if ((rc = ensure(c, c->ws_search, cb + ib + 2 * nb))) return rc;
    char* w = static_cast<char*>(c->ws_search.p);
    uint64_t* codes = reinterpret_cast<uint64_t*>(w);
    int32_t* sel = sel_ids_dev ? sel_ids_dev : reinterpret_cast<int32_t*>(w + cb);
    int32_t* cnt = sel_count_dev ? sel_count_dev : reinterpret_cast<int32_t*>(w + cb + ib);
    int32_t* bad = bad_dev ? bad_dev : reinterpret_cast<int32_t*>(w + cb + ib + nb);
    if ((rc = fspann_encode_dev(c, nq, q_dev, q_dtype, codes, nullptr, bad))) return rc;
    if ((rc = fspann_route_dev(c, nq, codes, probe_override, static_cast<int32_t>(B), B, sel, nullptr, cnt, nullptr, nullptr))) return rc;
    return fspann_refine_store_dev(c, nq, q_dev, q_dtype, B, sel, cnt, k, out_ids_dev, out_dist_dev, out_count_dev, scored_dev);
}

const void* fspann_store_dev_ptr(fspann_ctx* c, int* dtype) {
    if (!c) return nullptr;
    if (dtype) *dtype = c->store_dtype;
    return c->d_store;
}

int fspann_store_gather_dev(fspann_ctx* c, int64_t nq, const int32_t* sel_ids_dev, const int32_t* sel_count_dev, int64_t B,
                            void* cand_dev) {
    CHECK_CTX(c);
    if (!c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (!sel_ids_dev || !sel_count_dev || !cand_dev) return fail(FSPANN_E_NULL, "gather buffer is null");
    if (nq <= 0 || B <= 0) return FSPANN_OK;
    const int d = c->cfg.dim;
    const int64_t rows = nq * B;
    const unsigned grid = static_cast<unsigned>((rows + 7) / 8);
    if (c->store_dtype == FSPANN_F32) {
        const int vec_ok = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(cand_dev) & 15) == 0);
        hipLaunchKernelGGL(store_gather_kernel<float>, dim3(grid), dim3(256), 0, c->stream, static_cast<const float*>(c->d_store), d,
                           sel_ids_dev, sel_count_dev, B, nq, static_cast<float*>(cand_dev), vec_ok);
    } else {
        const int vec_ok = (d % 2 == 0) && ((reinterpret_cast<uintptr_t>(cand_dev) & 15) == 0);
        hipLaunchKernelGGL(store_gather_kernel<double>, dim3(grid), dim3(256), 0, c->stream, static_cast<const double*>(c->d_store), d,
                           sel_ids_dev, sel_count_dev, B, nq, static_cast<double*>(cand_dev), vec_ok);
    }
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
}

// Kernel-attached timing of the refinement scan: between _begin and _end every refine_scan_kernel dispatch of this
// context carries its own start/stop HIP events (hipExtLaunchKernel), i.e. the duration of the kernel itself on the
// context's stream — what a rocprofv3 kernel trace reports — without the gaps a record-before / record-after bracket adds.
// A dispatch with attached events costs a few microseconds of extra stream time, hence `every`: only every n-th one is timed.
int fspann_refine_timing_begin(fspann_ctx* c, int max_launches, int every) {
    CHECK_CTX(c);
    if (max_launches <= 0 || every <= 0) return fail(FSPANN_E_ARG, "max_launches <= 0 or every <= 0");
    c->rt_every = every;
    c->rt_seen = 0;
    while (c->rt_events.size() < static_cast<size_t>(max_launches) * 2) {
        hipEvent_t e;
        FSP_HIP(hipEventCreate(&e));
        c->rt_events.push_back(e);
    }
    c->rt_used = 0;
    c->rt_on = true;
    return FSPANN_OK;
}
int fspann_refine_timing_end(fspann_ctx* c, int* launches, double* total_ms) {
    CHECK_CTX(c);
    c->rt_on = false;
    FSP_HIP(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < c->rt_used; i += 2) {
        float ms = 0.f;
        FSP_HIP(hipEventElapsedTime(&ms, c->rt_events[i], c->rt_events[i + 1]));
        tot += ms;
    }
    if (launches) *launches = static_cast<int>(c->rt_used / 2);
    if (total_ms) *total_ms = tot;
    c->rt_used = 0;
    return FSPANN_OK;
}


}  // extern "C"
