// api_hostpipe.hip.h — the host candidate pipeline: packed point store, AES-GCM thread pool, three-stage pipeline
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

extern "C" {

// ---- host candidate pipeline (hostpipe.hip.h) ------------------------------------------------------------------------------
int fspann_pointstore_create(int64_t n, int dim, fspann_pointstore** out) {
    if (!out) return fail(FSPANN_E_NULL, "out is null");
    *out = nullptr;
    if (n <= 0 || n >= (1LL << 31) || dim <= 0 || dim > (1 << 20)) return fail(FSPANN_E_ARG, "n or dim out of range");
    if (!crypto_api()) return fail(FSPANN_E_STATE, "libcrypto (OpenSSL 3) not found: set FSPANN_CRYPTO_LIB");
    return guarded([&]() -> int {
        fspann_pointstore* ps = new fspann_pointstore();
        ps->n = n;
        ps->dim = dim;
        ps->stride = pointstore_stride(dim);
        ps->mem.assign(static_cast<size_t>(n) * ps->stride, 0);
        *out = ps;
        return FSPANN_OK;
    });
}
void fspann_pointstore_destroy(fspann_pointstore* ps) { delete ps; }

int fspann_pointstore_set_master_key(fspann_pointstore* ps, const uint8_t* key32) {
    if (!ps || !key32) return fail(FSPANN_E_NULL, "point store / key is null");
    std::lock_guard<std::mutex> lk(ps->key_mu);
    std::memcpy(ps->master, key32, 32);
    ps->have_master = true;
    for (auto& k : ps->keys) if (!k.empty()) cleanse(k.data(), k.size());
    ps->keys.clear();
    return FSPANN_OK;
}
int fspann_pointstore_current_version(fspann_pointstore* ps) { return ps ? ps->current_version.load() : FSPANN_E_NULL; }
// KeyRotationServiceImpl.rotateKeyOnly (:292-305): a new current version, no record is touched
int fspann_pointstore_rotate(fspann_pointstore* ps, int* new_version) {
    if (!ps) return fail(FSPANN_E_NULL, "point store is null");
    const int v = ps->current_version.fetch_add(1) + 1;
    if (new_version) *new_version = v;
    return FSPANN_OK;
}
// KeyManager retire (:274-317): K_v can no longer be derived; records still sealed with it become unreadable
int fspann_pointstore_retire(fspann_pointstore* ps, int version) {
    if (!ps) return fail(FSPANN_E_NULL, "point store is null");
    if (version <= 0) return fail(FSPANN_E_ARG, "version <= 0");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(ps->key_mu);
        if (static_cast<size_t>(version) >= ps->retired.size()) ps->retired.resize(version + 1, 0);
        ps->retired[version] = 1;
        if (static_cast<size_t>(version) < ps->keys.size() && !ps->keys[version].empty()) {
            cleanse(ps->keys[version].data(), ps->keys[version].size());
            ps->keys[version].clear();
        }
        return FSPANN_OK;
    });
}
int fspann_pointstore_encrypt(fspann_pointstore* ps, int64_t h0, int64_t cnt, const void* vectors, int dtype, int threads) {
    if (!ps || !vectors) return fail(FSPANN_E_NULL, "point store / vectors is null");
    if (h0 < 0 || cnt < 0 || h0 + cnt > ps->n) return fail(FSPANN_E_ARG, "handles [%lld, %lld) outside the store", (long long)h0, (long long)(h0 + cnt));
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    if (!ps->have_master) return fail(FSPANN_E_STATE, "Master key is not initialized");
    return guarded([&]() -> int {
        std::atomic<long long> bad{0};
        const int rc = dtype == FSPANN_F32 ? pointstore_encrypt<float>(ps, h0, cnt, static_cast<const float*>(vectors), threads, &bad)
                                           : pointstore_encrypt<double>(ps, h0, cnt, static_cast<const double*>(vectors), threads, &bad);
        if (rc == -2) return fail(FSPANN_E_DEVICE, "RAND_bytes failed");
        if (rc) return fail(FSPANN_E_STATE, "current key version is not derivable (retired?)");
        if (bad.load()) return fail(FSPANN_E_DEVICE, "AES-GCM seal failed for %lld records", bad.load());
        return FSPANN_OK;
    });
}
int fspann_pointstore_delete(fspann_pointstore* ps, int64_t h) {
    if (!ps) return fail(FSPANN_E_NULL, "point store is null");
    if (h < 0 || h >= ps->n) return fail(FSPANN_E_ARG, "handle out of range");
    (void)acquire_record(ps, h);                       // not under a writer's feet
    ps->ver(h)->store(0, std::memory_order_release);
    return FSPANN_OK;
}
// KeyRotationServiceImpl.reencryptTouched (:215-289): records older than the current version are opened with THEIR key and
// sealed again with the current one under a fresh IV (and the new version in the AAD); failures are skipped silently (:274-276).
int fspann_pointstore_reencrypt(fspann_pointstore* ps, const int32_t* handles, int64_t cnt, int threads, int64_t* reencrypted) {
    if (!ps || (cnt > 0 && !handles)) return fail(FSPANN_E_NULL, "point store / handles is null");
    if (cnt < 0 || cnt > (1LL << 26)) return fail(FSPANN_E_ARG, "cnt out of range (at most 2^26 handles per call)");
    return guarded([&]() -> int {
        long long done = 0;
        const int rc = pointstore_reencrypt(ps, handles, cnt, threads, &done);
        if (rc == -1) return fail(FSPANN_E_STATE, "current key version is not derivable");
        if (rc == -2) return fail(FSPANN_E_DEVICE, "RAND_bytes failed");
        if (reencrypted) *reencrypted = done;
        return FSPANN_OK;
    });
}
int fspann_pointstore_open_batch(fspann_pointstore* ps, int64_t nq, int64_t B, const int32_t* ids, const int32_t* count, void* dst, int dst_dtype,
                                 int32_t* out_ids, int32_t* out_count, int threads) {
    if (!ps || !ids || !count || !dst || !out_ids || !out_count) return fail(FSPANN_E_NULL, "point store / buffer is null");
    if (nq < 0 || B <= 0) return fail(FSPANN_E_ARG, "nq < 0 or B <= 0");
    if (dst_dtype != FSPANN_F32 && dst_dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dst_dtype);
    return guarded([&]() -> int {
        if (dst_dtype == FSPANN_F32) pointstore_open_batch<float>(ps, nq, B, ids, count, static_cast<float*>(dst), out_ids, out_count, threads);
        else pointstore_open_batch<double>(ps, nq, B, ids, count, static_cast<double*>(dst), out_ids, out_count, threads);
        return FSPANN_OK;
    });
}
// One record as stored (interop / tests): version (0: none), iv[12], ct[8*dim + 16].
int fspann_pointstore_get_record(fspann_pointstore* ps, int64_t h, int32_t* version, uint8_t* iv12, uint8_t* ct) {
    if (!ps) return fail(FSPANN_E_NULL, "point store is null");
    if (h < 0 || h >= ps->n) return fail(FSPANN_E_ARG, "handle out of range");
    const int v = ps->ver(h)->load(std::memory_order_acquire);
    if (version) *version = v;
    if (iv12) copy_from_shared(iv12, ps->rec(h) + kRecHeader, kIvBytes);
    if (ct) copy_from_shared(ct, ps->rec(h) + kRecHeader + kIvBytes, 8 * static_cast<size_t>(ps->dim) + kTagBytes);
    return FSPANN_OK;
}
// Import a record sealed elsewhere (the JVM's EncryptedPoint: keyVersion, iv, ciphertext || tag).
int fspann_pointstore_put_record(fspann_pointstore* ps, int64_t h, int32_t version, const uint8_t* iv12, const uint8_t* ct) {
    if (!ps || !iv12 || !ct) return fail(FSPANN_E_NULL, "point store / record is null");
    if (h < 0 || h >= ps->n) return fail(FSPANN_E_ARG, "handle out of range");
    if (version <= 0) return fail(FSPANN_E_ARG, "version <= 0");
    (void)acquire_record(ps, h);
    copy_to_shared(ps->rec(h) + kRecHeader, iv12, kIvBytes);
    copy_to_shared(ps->rec(h) + kRecHeader + kIvBytes, ct, 8 * static_cast<size_t>(ps->dim) + kTagBytes);
    ps->ver(h)->store(version, std::memory_order_release);
    return FSPANN_OK;
}
int fspann_pointstore_stats(fspann_pointstore* ps, int64_t* opened, int64_t* failed) {
    if (!ps) return fail(FSPANN_E_NULL, "point store is null");
    if (opened) *opened = ps->opened.load();
    if (failed) *failed = ps->failed.load();
    return FSPANN_OK;
}

}  // extern "C"
namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void pipeline_stage_a(fspann_pipeline* p) {
    for (;;) {
        int si;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->qa.empty(); });
            if (p->qa.empty()) return;
            si = p->qa.front(); p->qa.pop_front();
        }
        fspann_pipeline::Slot& s = p->slot[si];
        const double t0 = now_ms();
        {
            std::lock_guard<std::mutex> g(p->gpu_mu);
            fspann_ctx* c = p->ctx;
            const int d = c->cfg.dim;
            int rc = hipSetDevice(c->device) == hipSuccess ? 0 : FSPANN_E_DEVICE;
            if (!rc && hipMemcpyAsync(s.q_dev, s.q_pin, static_cast<size_t>(s.nq) * d * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = FSPANN_E_DEVICE;
            if (!rc) rc = fspann_encode_dev(c, s.nq, s.q_dev, FSPANN_F32, static_cast<uint64_t*>(s.codes_dev), nullptr, static_cast<int32_t*>(s.bad_dev));
            if (!rc) rc = fspann_route_dev(c, s.nq, static_cast<const uint64_t*>(s.codes_dev), -1, static_cast<int32_t>(p->B), p->B, static_cast<int32_t*>(s.sel_dev),
                                           nullptr, static_cast<int32_t*>(s.cnt_dev), nullptr, nullptr);
            if (!rc) {      // (rare) queries whose bestScore map treeifies a bin are finished by the host model before F_q leaves the device
                try {
                    rc = resolve_unmodelled(c, s.nq, static_cast<const uint64_t*>(s.codes_dev), -1, static_cast<int32_t>(p->B), p->B, static_cast<int32_t*>(s.sel_dev),
                                            nullptr, static_cast<int32_t*>(s.cnt_dev), nullptr, nullptr, nullptr, &s.unmodelled);
                } catch (...) { rc = FSPANN_E_NOMEM; }
            }
            if (!rc && (hipMemcpyAsync(s.sel_pin, s.sel_dev, static_cast<size_t>(s.nq) * p->B * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipMemcpyAsync(s.cnt_pin, s.cnt_dev, static_cast<size_t>(s.nq) * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess)) rc = FSPANN_E_DEVICE;
            s.rc = rc;
        }
        s.t_route_ms = now_ms() - t0;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->qb.push_back(si);
        }
        p->cv.notify_all();
    }
}
void pipeline_stage_b(fspann_pipeline* p) {
    for (;;) {
        int si;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->qb.empty(); });
            if (p->qb.empty()) return;
            si = p->qb.front(); p->qb.pop_front();
        }
        fspann_pipeline::Slot& s = p->slot[si];
        const double t0 = now_ms();
        if (!s.rc) {
            try {
                pointstore_open_batch<float>(p->ps, s.nq, p->B, s.sel_pin, s.cnt_pin, s.cand_pin, s.ids_pin, s.kcnt_pin, p->threads);
            } catch (...) { s.rc = FSPANN_E_NOMEM; }
        }
        s.t_decrypt_ms = now_ms() - t0;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->qc.push_back(si);
        }
        p->cv.notify_all();
    }
}
void pipeline_stage_c(fspann_pipeline* p) {
    for (;;) {
        int si;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->qc.empty(); });
            if (p->qc.empty()) return;
            si = p->qc.front(); p->qc.pop_front();
        }
        fspann_pipeline::Slot& s = p->slot[si];
        const double t0 = now_ms();
        if (!s.rc) {
            std::lock_guard<std::mutex> g(p->gpu_mu);
            fspann_ctx* c = p->ctx;
            const int d = c->cfg.dim;
            int rc = hipSetDevice(c->device) == hipSuccess ? 0 : FSPANN_E_DEVICE;
            const size_t rows = static_cast<size_t>(s.nq) * p->B;
            if (!rc && (hipMemcpyAsync(s.cand_dev, s.cand_pin, rows * d * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                        hipMemcpyAsync(s.ids_dev, s.ids_pin, rows * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                        hipMemcpyAsync(s.kcnt_dev, s.kcnt_pin, static_cast<size_t>(s.nq) * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess)) rc = FSPANN_E_DEVICE;
            if (!rc) rc = fspann_refine_dev(c, s.nq, s.q_dev, FSPANN_F32, s.cand_dev, FSPANN_F32, p->B, static_cast<int32_t*>(s.ids_dev), static_cast<int32_t*>(s.kcnt_dev),
                                            p->k, static_cast<int32_t*>(s.oi_dev), static_cast<double*>(s.od_dev), static_cast<int32_t*>(s.oc_dev), nullptr);
            if (!rc && (hipMemcpyAsync(s.out_ids_pin, s.oi_dev, static_cast<size_t>(s.nq) * p->k * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipMemcpyAsync(s.out_dist_pin, s.od_dev, static_cast<size_t>(s.nq) * p->k * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipMemcpyAsync(s.out_cnt_pin, s.oc_dev, static_cast<size_t>(s.nq) * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        hipStreamSynchronize(c->stream) != hipSuccess)) rc = FSPANN_E_DEVICE;
            s.rc = rc;
        }
        s.t_refine_ms = now_ms() - t0;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->sum_route_ms += s.t_route_ms; p->sum_decrypt_ms += s.t_decrypt_ms; p->sum_refine_ms += s.t_refine_ms; p->batches++;
            p->done_q.push_back(si);
        }
        p->cv.notify_all();
    }
}
}  // namespace
extern "C" {

void fspann_pipeline_destroy(fspann_pipeline* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    if (p->ta.joinable()) p->ta.join();
    if (p->tb.joinable()) p->tb.join();
    if (p->tc.joinable()) p->tc.join();
    if (p->ctx) { (void)hipSetDevice(p->ctx->device); (void)hipStreamSynchronize(p->ctx->stream); }
    for (auto& s : p->slot) {
        void* pins[] = {s.q_pin, s.sel_pin, s.cnt_pin, s.cand_pin, s.ids_pin, s.kcnt_pin, s.out_ids_pin, s.out_dist_pin, s.out_cnt_pin};
        for (void* x : pins) if (x) (void)hipHostFree(x);
        void* devs[] = {s.q_dev, s.codes_dev, s.sel_dev, s.cnt_dev, s.cand_dev, s.ids_dev, s.kcnt_dev, s.oi_dev, s.od_dev, s.oc_dev, s.bad_dev};
        for (void* x : devs) if (x) (void)hipFree(x);
    }
    delete p;
}

int fspann_pipeline_create(fspann_ctx* c, fspann_pointstore* ps, int64_t nq_max, int64_t B, int k, int host_threads, fspann_pipeline** out) {
    CHECK_CTX(c);
    if (!ps || !out) return fail(FSPANN_E_NULL, "point store / out is null");
    *out = nullptr;
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (nq_max <= 0 || B <= 0 || k <= 0 || B > INT32_MAX) return fail(FSPANN_E_ARG, "nq_max, B, k must be > 0");
    if (ps->dim != c->cfg.dim) return fail(FSPANN_E_ARG, "point store dimension %d != context dimension %d", ps->dim, c->cfg.dim);
    return guarded([&]() -> int {
        fspann_pipeline* p = new fspann_pipeline();
        p->ctx = c; p->ps = ps; p->nq_max = nq_max; p->B = B; p->k = k; p->threads = std::max(1, host_threads);
        const size_t d = c->cfg.dim, rows = static_cast<size_t>(nq_max) * B;
        bool ok = true;
        auto pin = [&](auto** ptr, size_t bytes) { if (ok && hipHostMalloc(reinterpret_cast<void**>(ptr), bytes, hipHostMallocDefault) != hipSuccess) ok = false; };
        auto dev = [&](void** ptr, size_t bytes) { if (ok && hipMalloc(ptr, bytes) != hipSuccess) ok = false; };
        for (auto& s : p->slot) {
            pin(&s.q_pin, nq_max * d * 4); pin(&s.sel_pin, rows * 4); pin(&s.cnt_pin, nq_max * 4); pin(&s.cand_pin, rows * d * 4); pin(&s.ids_pin, rows * 4);
            pin(&s.kcnt_pin, nq_max * 4); pin(&s.out_ids_pin, nq_max * k * 4); pin(&s.out_dist_pin, nq_max * k * 8); pin(&s.out_cnt_pin, nq_max * 4);
            dev(&s.q_dev, nq_max * d * 4); dev(&s.codes_dev, static_cast<size_t>(nq_max) * c->TD * c->W * 8); dev(&s.sel_dev, rows * 4); dev(&s.cnt_dev, nq_max * 4);
            dev(&s.cand_dev, rows * d * 4); dev(&s.ids_dev, rows * 4); dev(&s.kcnt_dev, nq_max * 4); dev(&s.oi_dev, nq_max * k * 4); dev(&s.od_dev, nq_max * k * 8);
            dev(&s.oc_dev, nq_max * 4); dev(&s.bad_dev, nq_max * 4);
        }
        if (!ok) { fspann_pipeline_destroy(p); return fail(FSPANN_E_NOMEM, "pinned / device staging buffers: allocation failed"); }
        for (int i = 0; i < fspann_pipeline::kSlots; i++) p->free_q.push_back(i);
        p->ta = std::thread(pipeline_stage_a, p);
        p->tb = std::thread(pipeline_stage_b, p);
        p->tc = std::thread(pipeline_stage_c, p);
        *out = p;
        return FSPANN_OK;
    });
}

// Hand a batch to the pipeline (copied into pinned memory before the call returns).  Blocks while every slot is in use:
// collect finished batches (in submission order) to make room.
int fspann_pipeline_submit(fspann_pipeline* p, int64_t nq, const float* q_host, uint64_t* ticket) {
    if (!p || !q_host) return fail(FSPANN_E_NULL, "pipeline / queries is null");
    if (nq <= 0 || nq > p->nq_max) return fail(FSPANN_E_ARG, "nq outside (0, nq_max]");
    int si;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv.wait(lk, [&] { return p->stop || !p->free_q.empty(); });
        if (p->stop) return fail(FSPANN_E_STATE, "pipeline is shutting down");
        si = p->free_q.front(); p->free_q.pop_front();
    }
    fspann_pipeline::Slot& s = p->slot[si];
    s.nq = nq; s.rc = 0; s.unmodelled = 0;
    std::memcpy(s.q_pin, q_host, static_cast<size_t>(nq) * p->ctx->cfg.dim * 4);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        s.ticket = p->next_ticket++;
        if (ticket) *ticket = s.ticket;
        p->qa.push_back(si);
    }
    p->cv.notify_all();
    return FSPANN_OK;
}

// The oldest finished batch: out_ids / out_dist = [nq][k], out_count [nq].  Blocks until one is done.
int fspann_pipeline_collect(fspann_pipeline* p, uint64_t* ticket, int64_t* nq, int32_t* out_ids, double* out_dist, int32_t* out_count) {
    if (!p) return fail(FSPANN_E_NULL, "pipeline is null");
    int si;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv.wait(lk, [&] { return p->stop || !p->done_q.empty(); });
        if (p->done_q.empty()) return fail(FSPANN_E_STATE, "pipeline is shutting down");
        si = p->done_q.front(); p->done_q.pop_front();
    }
    fspann_pipeline::Slot& s = p->slot[si];
    const int rc = s.rc;
    const uint64_t tk = s.ticket;            // the slot goes back to the free list below: nothing of it is read afterwards
    const int64_t unm = s.unmodelled;
    if (ticket) *ticket = tk;
    if (nq) *nq = s.nq;
    if (!rc) {
        if (out_ids) std::memcpy(out_ids, s.out_ids_pin, static_cast<size_t>(s.nq) * p->k * 4);
        if (out_dist) std::memcpy(out_dist, s.out_dist_pin, static_cast<size_t>(s.nq) * p->k * 8);
        if (out_count) std::memcpy(out_count, s.out_cnt_pin, static_cast<size_t>(s.nq) * 4);
    }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->free_q.push_back(si);
    }
    p->cv.notify_all();
    if (rc) return fail(rc, "a pipeline stage failed for ticket %llu", (unsigned long long)tk);
    if (unm) return fail(FSPANN_E_STATE, "ticket %llu: %lld queries need String.compareTo of non-decimal ids inside a treeified HashMap bin "
                         "(not modelled): their results are empty, the others are complete", (unsigned long long)tk, (long long)unm);
    return FSPANN_OK;
}

int fspann_pipeline_stats(fspann_pipeline* p, double* route_ms, double* decrypt_ms, double* refine_ms, int64_t* batches) {
    if (!p) return fail(FSPANN_E_NULL, "pipeline is null");
    std::lock_guard<std::mutex> lk(p->mu);
    const double n = std::max<long long>(1, p->batches);
    if (route_ms) *route_ms = p->sum_route_ms / n;
    if (decrypt_ms) *decrypt_ms = p->sum_decrypt_ms / n;
    if (refine_ms) *refine_ms = p->sum_refine_ms / n;
    if (batches) *batches = p->batches;
    return FSPANN_OK;
}


}  // extern "C"
