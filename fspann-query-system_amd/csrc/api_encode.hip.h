// api_encode.hip.h — TokenGen / Setup coding entry points: fspann_encode[_dev], the encode path switch (Coding.H / Coding.C, idx/Coding.java:250-301)
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

namespace {

template <typename TIn>
int launch_encode_mfma(fspann_ctx* c, int64_t nq, const TIn* q_dev, uint64_t* codes_dev, int32_t* hashes_dev, int32_t* bad_dev) {
    const int P = c->P_total, d = c->cfg.dim;
    const int64_t cap = std::max<int64_t>(65536, nq * P / 16);
    int rc = ensure(c, c->ws_fix, 256 + static_cast<size_t>(cap) * 8);
    if (rc) return rc;
    unsigned long long* cnt = static_cast<unsigned long long*>(c->ws_fix.p);
    int64_t* list = reinterpret_cast<int64_t*>(static_cast<char*>(c->ws_fix.p) + 256);
    FSP_HIP(hipMemsetAsync(cnt, 0, 8, c->stream));
    // the code words start clear: the MFMA epilogue ORs in the bits of the pairs it can decide, encode_fix_kernel those of the rest
    FSP_HIP(hipMemsetAsync(codes_dev, 0, static_cast<size_t>(nq) * c->TD * c->W * 8, c->stream));
    unsigned long long* cw = reinterpret_cast<unsigned long long*>(codes_dev);
    // block tile: 64 x 256 for bulk coding (Setup: hundreds of thousands of rows per call); 32 x 128 with tiles in flight for batches
    // of up to a few thousand blocks of the large tile (8 192 x 1 024 x 768: 273 against 327 us; equal at 262 144 rows)
    const int64_t big_blocks = ((nq + 63) / 64) * ((P + 255) / 256);
    if (big_blocks >= 16 * static_cast<int64_t>(c->num_cus) && c->knob_mfma_tile != 1) {
        dim3 grid(static_cast<unsigned>((nq + mfma_tile_q(2) - 1) / mfma_tile_q(2)), static_cast<unsigned>((P + mfma_tile_p(2) - 1) / mfma_tile_p(2)));
        hipLaunchKernelGGL((encode_mfma_kernel<TIn, 2, 2>), grid, dim3(256), 0, c->stream, q_dev, nq, d, c->d_alphaT32, c->d_r, c->d_omega, P, c->cfg.m,
                           c->cfg.lambda, c->W, c->TD, hashes_dev, cw, bad_dev, list, cap, cnt, c->alpha_norm_max);
    } else {
        dim3 grid(static_cast<unsigned>((nq + mfma_tile_q(1) - 1) / mfma_tile_q(1)), static_cast<unsigned>((P + mfma_tile_p(1) - 1) / mfma_tile_p(1)));
        hipLaunchKernelGGL((encode_mfma_kernel<TIn, 1, 1>), grid, dim3(256), 0, c->stream, q_dev, nq, d, c->d_alphaT32, c->d_r, c->d_omega, P, c->cfg.m,
                           c->cfg.lambda, c->W, c->TD, hashes_dev, cw, bad_dev, list, cap, cnt, c->alpha_norm_max);
    }
    FSP_HIP(hipGetLastError());
    // (a wave per kFixG pairs: as many blocks as the list can need, at most four per CU — the waves stride over the tasks)
    const int64_t fix_blocks = std::min<int64_t>(static_cast<int64_t>(c->num_cus) * 4, (cap + kFixG * 4 - 1) / (kFixG * 4));
    hipLaunchKernelGGL((encode_fix_kernel<TIn>), dim3(static_cast<unsigned>(std::max<int64_t>(1, fix_blocks))), dim3(256), 0, c->stream,
                       q_dev, d, c->d_alpha_rows, c->d_r, c->d_omega, P, c->cfg.m, c->cfg.lambda, c->W, c->TD, list, cnt, cap, hashes_dev, cw);
    FSP_HIP(hipGetLastError());
    c->fix_cap_last = static_cast<unsigned long long>(cap);
    return 1;  // caller enqueues the exact kernel guarded by (count > cap): it only runs if the list overflowed
}

template <typename TIn>
int launch_encode(fspann_ctx* c, int64_t nq, const TIn* q_dev, uint64_t* codes_dev, int32_t* hashes_dev,
                  int32_t* bad_dev, double* proj_dev = nullptr) {
    const int m = c->cfg.m;
    const unsigned long long* guard = nullptr;
    unsigned long long guard_cap = 0;
    // auto: the MFMA pre-filter pays from ~5e8 multiply-adds per call (its fixed part: clearing the code words, the exact re-check of the
    // pairs on a bucket edge, four launches).  tools/encode_bench.py, exact vs MFMA: 1 024 x 256 x 128 12 / 31 us, 512 x 256 x 960
    // 48 / 75, 1 024 x 1 024 x 768 132 / 77, 4 096 x 256 x 960 169 / 94, 8 192 x 1 024 x 768 758 / 273, 262 144 x 256 x 128 1 080 / 502.
    const bool want_mfma = (c->encode_mode == 2) ||
                           (c->encode_mode == 0 && static_cast<double>(nq) * c->P_total * c->cfg.dim >= 5.0e8);
    if (want_mfma && !proj_dev && c->d_alphaT32 && c->W <= 3) {      // (the fused bit-pack epilogue carries three code words: <= 192 bits)
        int rc = launch_encode_mfma<TIn>(c, nq, q_dev, codes_dev, hashes_dev, bad_dev);
        if (rc <= 0) return rc;  // error
        // the list can only overflow when almost every pair sits on a bucket boundary (degenerate omega): the
        // exact kernel below is enqueued with a device-side guard and returns immediately otherwise.
        guard = static_cast<const unsigned long long*>(c->ws_fix.p);
        guard_cap = c->fix_cap_last;
        c->mfma_last = true;
    } else {
        c->mfma_last = false;
    }
    const int tdPerBlock = std::max(1, kEncThreads / m);
    const int gy = (c->TD + tdPerBlock - 1) / tdPerBlock;
    const EncodeArgs<TIn> ea{q_dev, nq, c->cfg.dim, c->d_alphaT, c->d_r, c->d_omega, c->P_total, m, c->cfg.lambda, c->W, c->TD, tdPerBlock,
                             codes_dev, hashes_dev, bad_dev, proj_dev, guard, guard_cap, c->dbg_route};
    // QB queries per block: 8 for bulk coding (index build), 4 for query batches so that
    // a 1024-query batch still fills 256 CUs.
    bool launched = false;
    if constexpr (sizeof(TIn) == 4) {
        // two rows per workgroup when four would leave CUs without one (BASELINE config #3's shard: 512 queries x 256 projections = 128
        // workgroups of four rows: 59.8 -> 48.4 us); FSPANN_ENCODE_QB forces 1 / 2 / 4 / 8 (dev A/B)
        int qb_env = c->knob_encode_qb;
        if (qb_env == 0 && nq >= 2 && ((nq + 3) / 4) * static_cast<int64_t>(gy) < c->num_cus) qb_env = 2;
        if (qb_env == 1 || qb_env == 2) {
            if (qb_env == 1) hipLaunchKernelGGL((encode_exact_kernel<TIn, 1>), dim3(static_cast<unsigned>(nq), gy), dim3(kEncThreads), 0, c->stream, ea);
            else hipLaunchKernelGGL((encode_exact_kernel<TIn, 2>), dim3(static_cast<unsigned>((nq + 1) / 2), gy), dim3(kEncThreads), 0, c->stream, ea);
            launched = true;
        }
    }
    if constexpr (sizeof(TIn) == 4) {       // (8 fp64 query rows per block do not fit the register budget)
        if (!launched && (nq >= 8192 || c->knob_encode_qb == 8)) {
            constexpr int QB = 8;
            hipLaunchKernelGGL((encode_exact_kernel<TIn, QB>), dim3(static_cast<unsigned>((nq + QB - 1) / QB), gy), dim3(kEncThreads), 0, c->stream, ea);
            launched = true;
        }
    }
    if (!launched) {
        // 4 rows per workgroup = 256 workgroups for a 1024-query batch.  Every workgroup reads all of alpha (256 KB) from L2:
        // fewer rows per workgroup (2: 12.8 us, 1: 20 us) cost more in that traffic than the extra waves per SIMD buy, and the
        // loop itself is bound by one dependent fp64 instruction per ~10 cycles of a lone wave (7.8 of 11.4 us, tools/encode_stamps.py).
        constexpr int QB = 4;
        hipLaunchKernelGGL((encode_exact_kernel<TIn, QB>), dim3(static_cast<unsigned>((nq + QB - 1) / QB), gy), dim3(kEncThreads), 0, c->stream, ea);
    }
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
}


}  // namespace

extern "C" {

// ---- encode -----------------------------------------------------------------------
int fspann_encode_dev(fspann_ctx* c, int64_t nq, const void* q_dev, int dtype, uint64_t* codes_dev,
                      int32_t* hashes_dev, int32_t* bad_dev) {
    CHECK_CTX(c);
    if (!c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized. Build index first.");
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    if (!q_dev || !codes_dev) return fail(FSPANN_E_NULL, "query vector is null");
    if (dtype == FSPANN_F64) return launch_encode<double>(c, nq, static_cast<const double*>(q_dev), codes_dev, hashes_dev, bad_dev);
    if (dtype == FSPANN_F32) return launch_encode<float>(c, nq, static_cast<const float*>(q_dev), codes_dev, hashes_dev, bad_dev);
    return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
}

int fspann_encode(fspann_ctx* c, int64_t nq, const void* q, int dtype, uint64_t* codes, int32_t* hashes) {
    CHECK_CTX(c);
    if (!c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized. Build index first.");
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    if (!q || !codes) return fail(FSPANN_E_NULL, "query vector is null");
    if (dtype != FSPANN_F32 && dtype != FSPANN_F64) return fail(FSPANN_E_ARG, "unknown dtype %d", dtype);
    const size_t esz = dtype == FSPANN_F64 ? 8 : 4;
    const size_t qb = static_cast<size_t>(nq) * c->cfg.dim * esz;
    const size_t cb = static_cast<size_t>(nq) * c->TD * c->W * 8;
    const size_t hb = hashes ? static_cast<size_t>(nq) * c->P_total * 4 : 0;
    return guarded([&]() -> int {
    int rc;
    {   // a handful of queries (QueryTokenFactory.create is one vector per call): through the mapped pinned block, no copy commands
        auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
        const size_t o_c = al(qb), o_b = o_c + al(cb), o_h = o_b + al(static_cast<size_t>(nq) * 4), tot = o_h + al(hb);
        if (tot <= kPinBytes && zero_copy_ok(c, nq)) {
            unsigned char* hp = static_cast<unsigned char*>(c->h_pin), *dp = static_cast<unsigned char*>(c->d_pin);
            std::memcpy(hp, q, qb);
            rc = fspann_encode_dev(c, nq, dp, dtype, reinterpret_cast<uint64_t*>(dp + o_c), hashes ? reinterpret_cast<int32_t*>(dp + o_h) : nullptr,
                                   reinterpret_cast<int32_t*>(dp + o_b));
            if (rc) return rc;
            FSP_HIP(hipStreamSynchronize(c->stream));
            const int32_t* bad = reinterpret_cast<const int32_t*>(hp + o_b);
            for (int64_t i = 0; i < nq; i++)
                if (bad[i]) return fail(FSPANN_E_ARG, "Vector contains NaN/Inf (query %lld)", (long long)i);  // Coding.java:360
            std::memcpy(codes, hp + o_c, cb);
            if (hashes) std::memcpy(hashes, hp + o_h, hb);
            return FSPANN_OK;
        }
    }
    if ((rc = ensure(c, c->ws_io[0], qb))) return rc;
    if ((rc = ensure(c, c->ws_io[1], cb))) return rc;
    if ((rc = ensure(c, c->ws_io[2], static_cast<size_t>(nq) * 4))) return rc;
    if (hashes && (rc = ensure(c, c->ws_io[3], hb))) return rc;
    FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, q, qb, hipMemcpyHostToDevice, c->stream));
    rc = fspann_encode_dev(c, nq, c->ws_io[0].p, dtype, static_cast<uint64_t*>(c->ws_io[1].p),
                           hashes ? static_cast<int32_t*>(c->ws_io[3].p) : nullptr, static_cast<int32_t*>(c->ws_io[2].p));
    if (rc) return rc;
    std::vector<int32_t> bad(static_cast<size_t>(nq));
    FSP_HIP(hipMemcpyAsync(codes, c->ws_io[1].p, cb, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipMemcpyAsync(bad.data(), c->ws_io[2].p, static_cast<size_t>(nq) * 4, hipMemcpyDeviceToHost, c->stream));
    if (hashes) FSP_HIP(hipMemcpyAsync(hashes, c->ws_io[3].p, hb, hipMemcpyDeviceToHost, c->stream));
    FSP_HIP(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < nq; i++)
        if (bad[i]) return fail(FSPANN_E_ARG, "Vector contains NaN/Inf (query %lld)", (long long)i);  // Coding.java:360
    return FSPANN_OK;
    });
}

// Encode path selection: 0 = auto (MFMA pre-filter for nq >= 4096, exact fp64 otherwise), 1 = exact fp64 VALU only,
// 2 = always MFMA fp32 GEMM + exact re-check.  All modes produce bit-identical hashes and codes.
int fspann_set_encode_mode(fspann_ctx* c, int mode) {
    if (!c) return fail(FSPANN_E_NULL, "ctx is null");
    if (mode < 0 || mode > 2) return fail(FSPANN_E_ARG, "encode mode must be 0, 1 or 2");
    c->encode_mode = mode;
    return FSPANN_OK;
}
// (query, projection) pairs the last MFMA-path encode re-checked with the exact kernel (0 for the exact path).
int64_t fspann_last_encode_rechecked(fspann_ctx* c) {
    if (!c) return FSPANN_E_NULL;
    if (!c->mfma_last || !c->ws_fix.p) return 0;
    unsigned long long n = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(&n, c->ws_fix.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return FSPANN_E_DEVICE;
    return static_cast<int64_t>(n);
}


}  // extern "C"
