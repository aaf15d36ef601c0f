// api_tick.hip.h — fspann_tick_dev: stages of different batches in one launch (tick.hip.h)
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

extern "C" {

// ---- one launch for encode / Route / Refine of three batches in flight (tick.hip.h) ---------------------------------------
size_t fspann_route_handover_bytes(fspann_ctx* c, int64_t nq, int probe_override) {
    if (!c || nq <= 0) return 0;
    const size_t TP = static_cast<size_t>(c->TD) * effective_probes(c, probe_override);
    return ((static_cast<size_t>(nq) * TP * 16 + 255) & ~size_t(255)) + static_cast<size_t>(nq) * c->TD * 4 + 256;
}
int fspann_last_tick_fused(fspann_ctx* c) { return c ? c->last_tick_fused : 0; }

}  // extern "C"
namespace {
void handover_ptrs(fspann_ctx* c, void* buf, int64_t nq, int P, int4** probe, int32_t** nprobe) {
    const size_t pb = (static_cast<size_t>(nq) * c->TD * P * 16 + 255) & ~size_t(255);
    *probe = static_cast<int4*>(buf);
    *nprobe = reinterpret_cast<int32_t*>(static_cast<char*>(buf) + pb);
}
}  // namespace
extern "C" {

int fspann_tick_dev(fspann_ctx* c, const fspann_tick* t) {
    CHECK_CTX(c);
    if (!t) return fail(FSPANN_E_NULL, "tick is null");
    if (t->nq_encode < 0 || t->nq_route < 0 || t->nq_refine < 0) return fail(FSPANN_E_ARG, "nq < 0");
    const bool E = t->nq_encode > 0, R = t->nq_route > 0, F = t->nq_refine > 0;
    if (!E && !R && !F) return FSPANN_OK;
    if ((R || (F && t->ref_handover_dev)) && !c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");
    if (E && !c->have_g) return fail(FSPANN_E_STATE, "GFunctionRegistry not initialized. Build index first.");
    if (E && (!t->enc_q_dev || !t->enc_codes_dev)) return fail(FSPANN_E_NULL, "query vector is null");
    if (R && !t->route_codes_dev) return fail(FSPANN_E_STATE, "MSANNP violation: QueryToken missing BitSet codes");
    if (R && (!t->route_ids_dev || !t->route_count_dev)) return fail(FSPANN_E_NULL, "output buffer is null");
    if (R && t->route_limit <= 0) return fail(FSPANN_E_ARG, "limit must be > 0");
    if (F && (t->ref_B <= 0 || t->ref_B > INT32_MAX)) return fail(FSPANN_E_ARG, "B out of range");
    if (F && t->k <= 0) return fail(FSPANN_E_ARG, "topK must be > 0");
    if (F && (!t->ref_q_dev || !t->ref_ids_dev || !t->ref_count_dev || !t->out_ids_dev || !t->out_dist_dev || !t->out_count_dev))
        return fail(FSPANN_E_NULL, "refine buffer is null");
    if (F && !t->ref_cand_dev && !c->d_store) return fail(FSPANN_E_STATE, "plaintext store not set");
    if (F && ((t->ref_handover_dev != nullptr) != (t->ref_codes_dev != nullptr)))
        return fail(FSPANN_E_ARG, "ref_handover_dev and ref_codes_dev go together (the batch's codes and the buffer its Route wrote)");
    const bool gather = F && !t->ref_cand_dev;
    const int d = c->cfg.dim;
    int rc;

    // ---- Route of the batch being routed; Route parameters of the batch being refined (to finish its PENDING queries)
    RoutePlan plR{}, plX{};
    RouteParams pR{}, pX{};
    bool fusedR = false, fusedX = false;
    // encode + Route without a Refine part: the front kernel (tick.hip.h), which may use the bounded select's small classes
    const bool front = E && R && !F && t->route_limit <= 512 && t->enc_dtype == FSPANN_F32 && c->knob_tick_fuse != 0;
    if (R) {
        if ((rc = prepare_route(c, t->nq_route, t->route_codes_dev, t->route_probe_override, t->route_limit, t->route_limit, t->route_ids_dev,
                                nullptr, t->route_count_dev, nullptr, nullptr, &plR, &pR, &fusedR, !front))) return rc;
        if (t->route_handover_dev) handover_ptrs(c, t->route_handover_dev, t->nq_route, plR.P, &pR.probe_g, &pR.nprobe_g);
    }
    const bool fix = F && t->ref_handover_dev != nullptr;
    if (fix) {
        if ((rc = prepare_route(c, t->nq_refine, t->ref_codes_dev, t->ref_probe_override, static_cast<int32_t>(t->ref_B), t->ref_B, t->ref_ids_dev,
                                nullptr, t->ref_count_dev, nullptr, nullptr, &plX, &pX, &fusedX, true, false))) return rc;
        handover_ptrs(c, t->ref_handover_dev, t->nq_refine, plX.P, &pX.probe_g, &pX.nprobe_g);
        // the redo runs with its arena in global memory: one slice (+ sort buffer for degenerate tie groups) per refine workgroup
        const int full_sort = next_pow2(std::max(plX.maxcand, 1));
        pX.sort_cap = std::min(full_sort, 1024);
        const size_t arena = ((static_cast<size_t>(pX.sort_cap) * 8 + static_cast<size_t>(plX.ht_size) * 4 + static_cast<size_t>(plX.max_tuples) * 4 +
                               ((static_cast<size_t>(plX.max_tuples) * 2 + 15) & ~size_t(15))) + 255) & ~size_t(255);
        const int64_t gstride = (pX.sort_cap < full_sort) ? full_sort : 0;
        const int64_t fix_wgs = t->nq_refine;   // one slice per refine workgroup
        const size_t so = static_cast<size_t>(fix_wgs) * gstride * 8;
        if ((rc = ensure(c, c->ws_tickfix, static_cast<size_t>(fix_wgs) * arena + so + 512))) return rc;
        pX.g_sort = so ? static_cast<uint64_t*>(c->ws_tickfix.p) : nullptr;
        pX.g_sort_stride = gstride;
        pX.g_scratch = static_cast<unsigned char*>(c->ws_tickfix.p) + ((so + 255) & ~size_t(255));
        pX.g_stride = static_cast<int64_t>(arena);
        pX.qcount = nullptr; pX.qlist = nullptr;
        pX.g_sub = nullptr; pX.lds_sort_words = 0;      // (limit <= 512 here: the long-list ordering is never reached)
        pX.slice_ht = 0; pX.slice_bits = 0;              // (no LDS region behind the small arrays here: the arena table)
    }

    // ---- can the three roles share one kernel?
    const int nchunks = F ? static_cast<int>((t->ref_B + kRefRows - 1) / kRefRows) : 1;
    const void* rows = gather ? c->d_store : t->ref_cand_dev;
    const int rows_dtype = gather ? c->store_dtype : t->ref_cand_dtype;
    bool fuse = c->knob_tick_fuse != 0;
    if (E) fuse = fuse && t->enc_dtype == FSPANN_F32;
    if (R) fuse = fuse && plR.lazy && fusedR;
    if (F) fuse = fuse && t->ref_q_dtype == FSPANN_F32 && rows_dtype == FSPANN_F32 && nchunks == 1 && (d % 4 == 0) &&
                  ((reinterpret_cast<uintptr_t>(rows) & 15) == 0);
    const size_t lds_ref = static_cast<size_t>(kRefRows) * (32 + 4) * sizeof(float);
    const size_t lds_enc = static_cast<size_t>(kTickEncQB * kEncThreads + kTickEncQB) * 4;
    size_t lds = 0;
    if (E) lds = std::max(lds, lds_enc);
    if (R) lds = std::max(lds, plR.lz_lds_bytes);
    if (F) lds = std::max(lds, lds_ref);
    if (fix) lds = std::max(lds, plX.small_bytes);
    fuse = fuse && lds + 1024 <= static_cast<size_t>(c->lds_limit);
    c->last_tick_fused = fuse ? 1 : 0;

    // the redo's parameters live in device memory (tick.hip.h): a small cache of recently used parameter blocks, so a serving
    // loop that cycles through a few buffer sets uploads each block once
    auto upload_fix = [&](const RouteParams& fixT, const RouteParams** out) -> int {
        if (!c->d_fixparams) {
            FSP_HIP(hipMalloc(&c->d_fixparams, sizeof(RouteParams) * fspann_ctx::kFixSlots));
            c->h_fixparams.assign(sizeof(RouteParams) * fspann_ctx::kFixSlots, 0);
            c->fix_valid = 0;
        }
        int slot = -1;
        for (int i = 0; i < fspann_ctx::kFixSlots; i++)
            if (((c->fix_valid >> i) & 1u) && std::memcmp(c->h_fixparams.data() + sizeof(RouteParams) * i, &fixT, sizeof(RouteParams)) == 0) { slot = i; break; }
        if (slot < 0) {
            slot = c->fix_next;
            c->fix_next = (c->fix_next + 1) % fspann_ctx::kFixSlots;
            std::memcpy(c->h_fixparams.data() + sizeof(RouteParams) * slot, &fixT, sizeof(RouteParams));
            // stream-ordered: ticks already enqueued that read this slot run before the copy
            FSP_HIP(hipMemcpyAsync(static_cast<char*>(c->d_fixparams) + sizeof(RouteParams) * slot, &fixT, sizeof(RouteParams), hipMemcpyHostToDevice, c->stream));
            c->fix_valid |= 1u << slot;
        }
        *out = reinterpret_cast<const RouteParams*>(static_cast<char*>(c->d_fixparams) + sizeof(RouteParams) * slot);
        return FSPANN_OK;
    };

    if (F && !E && !R) {
        // A tick with only a Refine part is the stand-alone scan.  With the batch's hand-over buffer the scan's own workgroups
        // finish the PENDING queries first (refine_stream_fix_kernel): Route and Refine as separate launches, no hand-back launch.
        // (the same conditions launch_refine_dc takes its refine_stream_fix_kernel under — incl. the default tile width: with
        // FSPANN_REFINE_DC = 64 / 128 the plain scan would run first and read PENDING as "no rows")
        const bool stream_ok = (c->knob_refine_dc == 0 || c->knob_refine_dc == 32) &&
                               t->ref_q_dtype == FSPANN_F32 && rows_dtype == FSPANN_F32 && nchunks == 1 && (d % 4 == 0) &&
                               ((reinterpret_cast<uintptr_t>(rows) & 15) == 0) && c->knob_refine_stream != 0 && c->knob_tick_fuse != 0;
        const RouteParams* fdev = nullptr;
        if (fix && stream_ok) {
            pX.dbg = nullptr;
            if ((rc = upload_fix(pX, &fdev))) return rc;
        } else if (fix) {
            auto fk = tick_fix_kernel;
            if (!(c->attr_mask & 64u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 64u;
            }
            hipLaunchKernelGGL(fk, dim3(static_cast<unsigned>(t->nq_refine)), dim3(kTickThreads), plX.small_bytes, c->stream, pX);
            FSP_HIP(hipGetLastError());
        }
        c->refine_fix_dev = fdev;
        c->refine_fix_lds = fdev ? plX.small_bytes : 0;
        c->refine_fix_used = false;
        rc = gather ? fspann_refine_store_dev(c, t->nq_refine, t->ref_q_dev, t->ref_q_dtype, t->ref_B, t->ref_ids_dev, t->ref_count_dev, t->k,
                                              t->out_ids_dev, t->out_dist_dev, t->out_count_dev, t->scored_dev)
                    : fspann_refine_dev(c, t->nq_refine, t->ref_q_dev, t->ref_q_dtype, t->ref_cand_dev, t->ref_cand_dtype, t->ref_B, t->ref_ids_dev,
                                        t->ref_count_dev, t->k, t->out_ids_dev, t->out_dist_dev, t->out_count_dev, t->scored_dev);
        const bool used = c->refine_fix_used;
        c->refine_fix_dev = nullptr;
        if (rc) return rc;
        if (fdev && !used) return fail(FSPANN_E_STATE, "tick: the scan did not take the streaming kernel that finishes PENDING queries");
        c->last_tick_fused = (c->knob_tick_fuse != 0 && nchunks == 1 && (!fix || fdev)) ? 1 : 0;
        return FSPANN_OK;
    }

    if (!fuse) {   // stand-alone kernels in stream order: same results
        // prepare_route above took the overflow counters' turn for a bounded select this call will not launch itself: the
        // stand-alone fspann_route_dev below takes its own.  Hand the turn back, or consecutive fall-back ticks would all
        // count into the counter nobody zeroes (stale overflow lists, then writes past the nq-sized list).
        if (R && plR.lazy) c->ovf_flip ^= 1;
        if (fix) {
            auto fk = tick_fix_kernel;
            if (!(c->attr_mask & 64u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 64u;
            }
            hipLaunchKernelGGL(fk, dim3(static_cast<unsigned>(t->nq_refine)), dim3(kTickThreads), plX.small_bytes, c->stream, pX);
            FSP_HIP(hipGetLastError());
        }
        if (F) {
            rc = gather ? fspann_refine_store_dev(c, t->nq_refine, t->ref_q_dev, t->ref_q_dtype, t->ref_B, t->ref_ids_dev, t->ref_count_dev, t->k,
                                                  t->out_ids_dev, t->out_dist_dev, t->out_count_dev, t->scored_dev)
                        : fspann_refine_dev(c, t->nq_refine, t->ref_q_dev, t->ref_q_dtype, t->ref_cand_dev, t->ref_cand_dtype, t->ref_B, t->ref_ids_dev,
                                            t->ref_count_dev, t->k, t->out_ids_dev, t->out_dist_dev, t->out_count_dev, t->scored_dev);
            if (rc) return rc;
        }
        if (R) {
            // the stand-alone call finishes handed-over queries itself (second launch); a hand-over buffer then stays unused
            if ((rc = fspann_route_dev(c, t->nq_route, t->route_codes_dev, t->route_probe_override, t->route_limit, t->route_limit, t->route_ids_dev,
                                       nullptr, t->route_count_dev, nullptr, nullptr))) return rc;
        }
        if (E && (rc = fspann_encode_dev(c, t->nq_encode, t->enc_q_dev, t->enc_dtype, t->enc_codes_dev, nullptr, t->enc_bad_dev))) return rc;
        return FSPANN_OK;
    }

    TickHead p{};
    EncodeArgs<float> eaT{};
    RouteParams routeT{}, fixT{};
    RefineArgs<float, float> raT{};
    if (E) {
        const int m = c->cfg.m;
        const int tdPerBlock = std::max(1, kEncThreads / m);
        const int gy = (c->TD + tdPerBlock - 1) / tdPerBlock;
        p.enc_gx = static_cast<int>((t->nq_encode + kTickEncQB - 1) / kTickEncQB);
        p.n_enc = p.enc_gx * gy;
        eaT = EncodeArgs<float>{static_cast<const float*>(t->enc_q_dev), t->nq_encode, d, c->d_alphaT, c->d_r, c->d_omega, c->P_total, m, c->cfg.lambda,
                                  c->W, c->TD, tdPerBlock, t->enc_codes_dev, nullptr, t->enc_bad_dev, nullptr, nullptr, 0};
        c->mfma_last = false;
    } else p.enc_gx = 1;
    if (R) {
        p.n_route = static_cast<int>(std::min<int64_t>(t->nq_route, 1 << 24));   // one query per workgroup
        routeT = pR;
        c->last_route_lazy = 1;
    }
    if (F) {
        p.n_refine = static_cast<int>(t->nq_refine);          // one workgroup per query (nchunks == 1)
        p.nq_refine = t->nq_refine;
        raT = RefineArgs<float, float>{static_cast<const float*>(t->ref_q_dev), static_cast<const float*>(rows), gather ? c->store_n : 0, t->ref_B, d,
                                         t->ref_ids_dev, t->ref_count_dev, t->k, 1, t->out_ids_dev, t->out_dist_dev, t->out_count_dev, t->scored_dev,
                                         nullptr, nullptr};
        p.has_fix = fix ? 1 : 0;
        if (fix) fixT = pX;
    }
    const RouteParams* fix_dev = nullptr;
    if (fix) { if ((rc = upload_fix(fixT, &fix_dev))) return rc; }
    // long jobs first: a share of the Route workgroups heads the grid, the rest is spread evenly between the others
    p.route_front = F ? static_cast<int>(static_cast<int64_t>(p.n_route) * c->knob_tick_front / 100) : p.n_route;
    p.dbg = c->dbg_route;           // debug builds: the tick's own per-workgroup stamps (the roles' phase stamps stay off)
    routeT.dbg = nullptr;
    fixT.dbg = nullptr;
    const int64_t total = static_cast<int64_t>(p.n_enc) + p.n_route + p.n_refine;
    if (total > INT32_MAX) return fail(FSPANN_E_RANGE, "too many workgroups in one tick");
    auto launch = [&](auto kern, unsigned abit) -> int {
        if (!(c->attr_mask & abit)) {
            FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            c->attr_mask |= abit;
        }
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(total)), dim3(kTickThreads), lds, c->stream, p, eaT, routeT, fix_dev, raT);
        FSP_HIP(hipGetLastError());
        return FSPANN_OK;
    };
    if (front && (plR.lz_entries == 512 || plR.lz_entries == kLzEntriesMax)) {
        if (plR.lz_entries == 512) {
            if (c->knob_shape_spec && c->TD == 16 && plR.P == 5 && plR.S == 64 && c->W == 1 && c->rec_words == 4 && (routeT.probe_G == 16 || routeT.probe_G == 0))
                hipLaunchKernelGGL((front_kernel<512, false, 16, 5>), dim3(static_cast<unsigned>(total)), dim3(kTickThreads), lds, c->stream, p, eaT, routeT);
            else
                hipLaunchKernelGGL((front_kernel<512, false>), dim3(static_cast<unsigned>(total)), dim3(kTickThreads), lds, c->stream, p, eaT, routeT);
        } else {
            auto fk = front_kernel<kLzEntriesMax, true>;
            if (!(c->attr_mask & 2048u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 2048u;
            }
            hipLaunchKernelGGL(fk, dim3(static_cast<unsigned>(total)), dim3(kTickThreads), lds, c->stream, p, eaT, routeT);
        }
        FSP_HIP(hipGetLastError());
    } else if ((rc = gather ? launch(tick_kernel<true>, 128u) : launch(tick_kernel<false>, 256u))) return rc;
    if (R && !t->route_handover_dev) {
        // no buffer travels with the batch: queries the bounded select handed over are finished now (normally none)
        RouteParams q2 = pR;
        q2.qcount = pR.ovf_count; q2.qlist = pR.ovf_list;
        const int g2 = std::min(plR.grid, 32);
        auto kern = route_select_kernel<true, 512>;
        if (plR.lds_mode && plR.threads == 512) {
            if (!(c->attr_mask & 1u)) {
                FSP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                c->attr_mask |= 1u;
            }
            hipLaunchKernelGGL(kern, dim3(g2), dim3(512), plR.lds_bytes, c->stream, q2, q2.probe_g, q2.nprobe_g);
            FSP_HIP(hipGetLastError());
        } else {
            return fail(FSPANN_E_STATE, "tick: hand-over buffer required for this configuration");
        }
    }
    return FSPANN_OK;
}


}  // extern "C"
