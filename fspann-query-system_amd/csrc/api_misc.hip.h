// api_misc.hip.h — ground truth + metrics, the multi-GPU top-k merge, HBM read-peak probes, device memory helpers
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

extern "C" {

// ---- exact ground truth + evaluation metrics (groundtruth.hip.h) -------------------------------------------------------------
int fspann_groundtruth_dev(fspann_ctx* c, int64_t n, const float* base_dev, int64_t nq, const float* q_dev, int dim, int k, int32_t* out_ids_dev,
                           double* out_d2_dev) {
    CHECK_CTX(c);
    if (!base_dev || !q_dev || !out_ids_dev) return fail(FSPANN_E_NULL, "ground truth buffer is null");
    if (n <= 0 || n >= (1LL << 31) || nq < 0 || dim <= 0) return fail(FSPANN_E_ARG, "Empty or malformed vector files (zero records).");
    if (k <= 0 || k > kGtMaxK) return fail(FSPANN_E_ARG, "k must be in [1, %d]", kGtMaxK);
    if (nq == 0) return FSPANN_OK;
    // the [chunk x n] fp64 distance matrix lives in scratch: at most ~8 GB at a time
    const int64_t chunk = std::max<int64_t>(kGtQT, std::min<int64_t>(nq, ((1LL << 33) / (n * 8)) / kGtQT * kGtQT));
    int rc = ensure(c, c->ws_gt, static_cast<size_t>(chunk) * n * 8);
    if (rc) return rc;
    double* dist = static_cast<double*>(c->ws_gt.p);
    for (int64_t s = 0; s < nq; s += chunk) {
        const int64_t cq = std::min(chunk, nq - s);
        dim3 grid(static_cast<unsigned>((n + kGtRows - 1) / kGtRows), static_cast<unsigned>((cq + kGtQT - 1) / kGtQT));
        hipLaunchKernelGGL(gt_dist_kernel, grid, dim3(kGtRows), 0, c->stream, base_dev, n, q_dev + s * dim, cq, dim, dist);
        FSP_HIP(hipGetLastError());
        hipLaunchKernelGGL(gt_select_kernel, dim3(static_cast<unsigned>(cq)), dim3(kGtSelThreads), 0, c->stream, dist, n, k, out_ids_dev + s * k,
                           out_d2_dev ? out_d2_dev + s * k : nullptr);
        FSP_HIP(hipGetLastError());
    }
    return FSPANN_OK;
}

int fspann_eval_metrics_dev(fspann_ctx* c, int64_t n, const float* base_dev, int64_t nq, const float* q_dev, int dim, int k, const int32_t* ann_ids_dev,
                            int64_t ann_stride, const int32_t* ann_count_dev, const int32_t* gt_ids_dev, int64_t gt_stride, double* recall_dev,
                            double* ratio_dev) {
    CHECK_CTX(c);
    if (!base_dev || !q_dev || !ann_ids_dev || !gt_ids_dev || !recall_dev || !ratio_dev) return fail(FSPANN_E_NULL, "metrics buffer is null");
    if (n <= 0 || nq < 0 || dim <= 0 || k <= 0 || k > kGtMaxK || gt_stride < k || ann_stride <= 0) return fail(FSPANN_E_ARG, "k must be in [1, %d] and gt must hold >= k ids per query", kGtMaxK);
    if (nq == 0) return FSPANN_OK;
    hipLaunchKernelGGL(gt_metrics_kernel, dim3(static_cast<unsigned>(nq)), dim3(64), 0, c->stream, base_dev, n, q_dev, dim, k, ann_ids_dev, ann_stride,
                       ann_count_dev, gt_ids_dev, gt_stride, recall_dev, ratio_dev);
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
}

// ---- multi-GPU merge: one RCCL all-gather of the packed per-rank top-k -------------------------------
size_t fspann_topk_bytes(int64_t nq, int k) {
    if (nq < 0 || k <= 0) return 0;
    const size_t idb = (static_cast<size_t>(nq) * k * 4 + 7) & ~size_t(7);     // keeps the fp64 part 8-byte aligned
    return idb + static_cast<size_t>(nq) * k * 8;
}
size_t fspann_topk_dist_offset(int64_t nq, int k) {
    if (nq < 0 || k <= 0) return 0;
    return (static_cast<size_t>(nq) * k * 4 + 7) & ~size_t(7);
}

int fspann_comm_available(void) { return rccl_api() ? 1 : 0; }

int fspann_comm_unique_id(void* id_out) {
    if (!id_out) return fail(FSPANN_E_NULL, "id_out is null");
    RcclApi* a = rccl_api();
    if (!a) return fail(FSPANN_E_STATE, "librccl not found (set FSPANN_RCCL_LIB): %s", dlerror() ? dlerror() : "no candidate loaded");
    RcclApi::UniqueId id;
    const int rc = a->GetUniqueId(&id);
    if (rc != 0) return fail(FSPANN_E_DEVICE, "ncclGetUniqueId: %s", rccl_err(a, rc));
    std::memcpy(id_out, &id, sizeof(id));
    return FSPANN_OK;
}

int fspann_comm_create(fspann_ctx* c, const void* unique_id, int world, int rank, fspann_comm** out) {
    CHECK_CTX(c);
    if (!unique_id || !out) return fail(FSPANN_E_NULL, "unique_id/out is null");
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world) return fail(FSPANN_E_ARG, "bad world %d / rank %d", world, rank);
    RcclApi* a = rccl_api();
    if (!a) return fail(FSPANN_E_STATE, "librccl not found (set FSPANN_RCCL_LIB)");
    RcclApi::UniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    fspann_comm* m = new (std::nothrow) fspann_comm();
    if (!m) return fail(FSPANN_E_NOMEM, "out of host memory");
    const int rc = a->CommInitRank(&m->nccl, world, id, rank);     // on the context's device (CHECK_CTX made it current)
    if (rc != 0) {
        delete m;
        return fail(FSPANN_E_DEVICE, "ncclCommInitRank(world %d, rank %d): %s", world, rank, rccl_err(a, rc));
    }
    m->ctx = c; m->world = world; m->rank = rank;
    c->comm_refs.fetch_add(1);
    *out = m;
    return FSPANN_OK;
}

int fspann_comm_destroy(fspann_comm* m) {
    if (!m) return FSPANN_OK;
    RcclApi* a = rccl_api();
    if (a && m->nccl) {
        if (m->ctx) { (void)hipSetDevice(m->ctx->device); (void)hipStreamSynchronize(m->ctx->stream); }
        (void)a->CommDestroy(m->nccl);
    }
    fspann_ctx* c = m->ctx;
    delete m;
    // the context was destroyed while this communicator held it: the last holder finishes that destroy
    if (c && c->comm_refs.fetch_sub(1) == 1 && c->destroy_deferred.exchange(false)) fspann_ctx_destroy(c);
    return FSPANN_OK;
}

int fspann_comm_info(fspann_comm* m, int* world, int* rank, const char** library) {
    if (!m) return fail(FSPANN_E_NULL, "comm is null");
    if (world) *world = m->world;
    if (rank) *rank = m->rank;
    if (library) { RcclApi* a = rccl_api(); *library = a ? a->path.c_str() : ""; }
    return FSPANN_OK;
}

// gathered_dev = world x fspann_topk_bytes(nq_local, k), in rank order = global query order when the batch was cut into
// contiguous equal shards (the last one padded with id -1 / +inf, which Refine writes for missing results anyway).
int fspann_allgather_topk_dev(fspann_comm* m, int64_t nq_local, int k, const void* local_packed_dev, void* gathered_dev) {
    if (!m || !m->ctx) return fail(FSPANN_E_NULL, "comm is null");
    CHECK_CTX(m->ctx);
    if (!local_packed_dev || !gathered_dev) return fail(FSPANN_E_NULL, "top-k buffer is null");
    const size_t nb = fspann_topk_bytes(nq_local, k);
    if (nb == 0) return fail(FSPANN_E_ARG, "nq_local < 0 or k <= 0");
    RcclApi* a = rccl_api();
    if (!a) return fail(FSPANN_E_STATE, "librccl not found");
    const int rc = a->AllGather(local_packed_dev, gathered_dev, nb, 0 /* ncclInt8 */, m->nccl, m->ctx->stream);
    if (rc != 0) return fail(FSPANN_E_DEVICE, "ncclAllGather: %s", rccl_err(a, rc));
    return FSPANN_OK;
}

// Measurement aid (bench.py `roofline.peak_measured`): the rate at which THIS device streams `bytes` of HBM through a
// pure 16-byte-load kernel (buffer owned by the library, larger than the 256 MiB Infinity Cache when bytes says so).
}  // extern "C"
namespace {
typedef unsigned int hbm_u32x4 __attribute__((ext_vector_type(4)));
template <bool kNT>   // kNT: the loads carry the nt policy (read-once data, as the refinement scan's row stream)
__global__ __launch_bounds__(256) void hbm_read_kernel(const hbm_u32x4* __restrict__ p, size_t n16, unsigned long long* __restrict__ sink) {
    hbm_u32x4 acc = {0, 0, 0, 0};
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const hbm_u32x4 v = kNT ? __builtin_nontemporal_load(p + i) : p[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1ull);   // keeps the loads alive; practically never taken
}
}  // namespace
extern "C" {
int fspann_hbm_read_peak(fspann_ctx* c, size_t bytes, int reps, double* gb_per_s) {
    CHECK_CTX(c);
    if (!gb_per_s || reps <= 0 || bytes < (1u << 20)) return fail(FSPANN_E_ARG, "bytes < 1 MiB, reps <= 0 or null output");
    void* buf = nullptr;
    unsigned long long* sink = nullptr;
    FSP_HIP(hipMalloc(&buf, bytes));
    if (hipMalloc(&sink, 8) != hipSuccess) { (void)hipFree(buf); return fail(FSPANN_E_NOMEM, "hipMalloc failed"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = FSPANN_OK;
    do {
        if (hipMemsetAsync(buf, 0x5A, bytes, c->stream) != hipSuccess || hipMemsetAsync(sink, 0, 8, c->stream) != hipSuccess ||
            hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = fail(FSPANN_E_DEVICE, "setup failed"); break; }
        const unsigned grid = static_cast<unsigned>(c->num_cus) * 8;
        const hbm_u32x4* src = static_cast<const hbm_u32x4*>(buf);
        double best = 0.0;
        for (int nt = 0; nt < 2 && rc == FSPANN_OK; nt++) {      // default cache policy and nt: the ceiling is the better of the two
            for (int r = -1; r < reps; r++) {                    // r = -1: warm-up
                (void)hipEventRecord(e0, c->stream);
                if (nt) hipLaunchKernelGGL(hbm_read_kernel<true>, dim3(grid), dim3(256), 0, c->stream, src, bytes / 16, sink);
                else hipLaunchKernelGGL(hbm_read_kernel<false>, dim3(grid), dim3(256), 0, c->stream, src, bytes / 16, sink);
                (void)hipEventRecord(e1, c->stream);
                if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(FSPANN_E_DEVICE, "hbm_read_kernel failed"); break; }
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 0 && ms > 0.f) best = std::max(best, static_cast<double>(bytes) / (ms * 1e-3) / 1e9);
            }
        }
        *gb_per_s = best;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    (void)hipFree(sink);
    return rc;
}

int fspann_hbm_read_window(fspann_ctx* c, size_t bytes, size_t window, int reps, double* gb_per_s) {
    CHECK_CTX(c);
    if (!gb_per_s || reps <= 0 || window < (1u << 20) || bytes < 2 * window || (window & 15))
        return fail(FSPANN_E_ARG, "window < 1 MiB or not a multiple of 16, bytes < 2 windows, reps <= 0 or null output");
    void* buf = nullptr;
    unsigned long long* sink = nullptr;
    FSP_HIP(hipMalloc(&buf, bytes));
    if (hipMalloc(&sink, 8) != hipSuccess) { (void)hipFree(buf); return fail(FSPANN_E_NOMEM, "hipMalloc failed"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = FSPANN_OK;
    do {
        if (hipMemsetAsync(buf, 0x5A, bytes, c->stream) != hipSuccess || hipMemsetAsync(sink, 0, 8, c->stream) != hipSuccess ||
            hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = fail(FSPANN_E_DEVICE, "setup failed"); break; }
        const unsigned grid = static_cast<unsigned>(c->num_cus) * 8;
        const size_t nwin = bytes / window;
        double best = 0.0;
        size_t wi = 0;
        for (int nt = 0; nt < 2 && rc == FSPANN_OK; nt++) {      // default cache policy and nt: the ceiling is the better of the two
            double total_ms = 0.0;
            int done = 0;
            for (int r = -1; r < reps; r++) {                    // r = -1: warm-up
                const hbm_u32x4* w = reinterpret_cast<const hbm_u32x4*>(static_cast<const char*>(buf) + (++wi % nwin) * window);
                if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(FSPANN_E_DEVICE, "hbm_read_kernel failed"); break; }
                if (nt) hipExtLaunchKernelGGL(hbm_read_kernel<true>, dim3(grid), dim3(256), 0, c->stream, e0, e1, 0, w, window / 16, sink);
                else hipExtLaunchKernelGGL(hbm_read_kernel<false>, dim3(grid), dim3(256), 0, c->stream, e0, e1, 0, w, window / 16, sink);
                if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(FSPANN_E_DEVICE, "hbm_read_kernel failed"); break; }
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= 0) { total_ms += ms; done++; }
            }
            if (rc == FSPANN_OK && total_ms > 0.0) best = std::max(best, static_cast<double>(window) * done / (total_ms * 1e-3) / 1e9);
        }
        if (rc == FSPANN_OK) *gb_per_s = best;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    (void)hipFree(sink);
    return rc;
}

// ---- device memory helpers -----------------------------------------------------------------
int fspann_dev_alloc(fspann_ctx* c, size_t bytes, void** out) {
    CHECK_CTX(c);
    if (!out) return fail(FSPANN_E_NULL, "out is null");
    FSP_HIP(hipMalloc(out, bytes ? bytes : 1));
    return FSPANN_OK;
}
int fspann_dev_free(fspann_ctx* c, void* p) {
    CHECK_CTX(c);
    if (p) {
        FSP_HIP(hipStreamSynchronize(c->stream));
        FSP_HIP(hipFree(p));
    }
    return FSPANN_OK;
}
int fspann_h2d(fspann_ctx* c, void* dst_dev, const void* src, size_t bytes) {
    CHECK_CTX(c);
    FSP_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    return FSPANN_OK;
}
int fspann_d2h(fspann_ctx* c, void* dst, const void* src_dev, size_t bytes) {
    CHECK_CTX(c);
    FSP_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    return FSPANN_OK;
}


}  // extern "C"
