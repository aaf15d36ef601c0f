// fspann_api.hip — implementation of include/fspann.h for gfx950 (MI355X).
// Product code.  No CPU fallback exists: every compute entry point launches a
// HIP kernel or fails with FSPANN_E_DEVICE.  Nothing here references oracle/.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <type_traits>
#include <mutex>
#include <new>
#include <numeric>

#include "comm.hip.h"
#include "build.hip.h"
#include "encode.hip.h"
#include "groundtruth.hip.h"
#include "hostpipe.hip.h"
#include "refine.hip.h"
#include "route.hip.h"
#include "route_lazy.hip.h"
#include "tick.hip.h"
#include "../host/route_replay.hpp"

using namespace fspann;

namespace {

int next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return static_cast<int>(p);
}

int effective_probes(const fspann_ctx* c, int override_) {  // PIS:880-888
    if (override_ > 0) return override_;
    if (c->cfg.probe_override > 0) return c->cfg.probe_override;
    return c->cfg.default_probes;
}

int java_final_cap_host(int cap0, int64_t n) {
    int cap = cap0;
    int64_t thr = static_cast<int64_t>(static_cast<float>(cap) * 0.75f);
    while (n > thr && cap < (1 << 30)) {
        const int oldCap = cap;
        cap <<= 1;
        thr = (oldCap >= 16) ? (thr << 1) : static_cast<int64_t>(static_cast<float>(cap) * 0.75f);
    }
    return cap;
}

void free_dev(void*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}
template <typename T> void free_devt(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

constexpr size_t kPinBytes = size_t(1) << 20;
// the context's pinned block (allocated at the first small host-pointer call; false: none, the general path runs)
bool pin_block(fspann_ctx* c) {
    if (!c->h_pin && hipHostMalloc(&c->h_pin, kPinBytes, hipHostMallocDefault) != hipSuccess) { c->h_pin = nullptr; (void)hipGetLastError(); }
    return c->h_pin != nullptr;
}

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
    dim3 grid(static_cast<unsigned>((nq + kMfmaTileQ - 1) / kMfmaTileQ), static_cast<unsigned>((P + kMfmaTileP - 1) / kMfmaTileP));
    unsigned long long* cw = reinterpret_cast<unsigned long long*>(codes_dev);
    hipLaunchKernelGGL((encode_mfma_kernel<TIn>), grid, dim3(256), 0, c->stream, q_dev, nq, d, c->d_alphaT32, c->d_r, c->d_omega, P, c->cfg.m, c->cfg.lambda,
                       c->W, c->TD, hashes_dev, cw, bad_dev, list, cap, cnt, c->alpha_norm_max);
    FSP_HIP(hipGetLastError());
    hipLaunchKernelGGL((encode_fix_kernel<TIn>), dim3(static_cast<unsigned>(std::min<int64_t>(1024, (cap + 255) / 256))), dim3(256), 0, c->stream,
                       q_dev, d, c->d_alphaT, c->d_r, c->d_omega, P, c->cfg.m, c->cfg.lambda, c->W, c->TD, list, cnt, cap, hashes_dev, cw);
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
    const bool want_mfma = (c->encode_mode == 2) || (c->encode_mode == 0 && nq >= 4096);
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
    if constexpr (sizeof(TIn) == 4) {       // (8 fp64 query rows per block do not fit the register budget)
        if (nq >= 8192) {
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
    const size_t budget = static_cast<size_t>(c->lds_limit) - 1024;  // static __shared__ + margin
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

int resolve_unmodelled(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit, int64_t cap, int32_t* ids_dev,
                       int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev, int32_t* raw_dev, int64_t* resolved_out, int64_t* left_out);   // api_ext.hip.h

// No C++ exception crosses the C ABI (include/fspann.h): every entry point that allocates host memory or starts
// threads runs its body through guarded(); worker threads catch on their own and report through a flag.
template <class F> int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(FSPANN_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(FSPANN_E_ARG, "C++ exception: %s", e.what());
    } catch (...) {
        return fail(FSPANN_E_ARG, "unknown C++ exception");
    }
}

#define CHECK_CTX_NOLOCK(c)                                               \
    do {                                                                  \
        if (!(c)) return fail(FSPANN_E_NULL, "ctx is null");              \
        hipError_t _e = hipSetDevice((c)->device);                        \
        if (_e != hipSuccess) return fail(FSPANN_E_DEVICE, "hipSetDevice(%d): %s", (c)->device, hipGetErrorString(_e)); \
    } while (0)
// ... and the context's lock for the rest of the entry point (calls on one context are serialised inside the library)
#define CHECK_CTX(c)         \
    CHECK_CTX_NOLOCK(c);     \
    std::lock_guard<std::recursive_mutex> _ctx_lock((c)->mu)

// the deleted-id mirror of the index this context serves (its own, or its owner's when it is a clone)
inline fspann_ctx* index_owner(fspann_ctx* c) { return c->share_parent ? c->share_parent : c; }

// State shared through fspann_ctx_clone is read-only: a clone cannot change it, its owner cannot while clones are alive.
#define CHECK_UNSHARED(c)                                                                                               \
    do {                                                                                                                \
        if ((c)->share_parent) return fail(FSPANN_E_STATE, "a clone reads its parent's index: it cannot be changed here"); \
        if ((c)->share_children.load() > 0) return fail(FSPANN_E_STATE, "the index is shared with %d clone(s): destroy them first", (c)->share_children.load()); \
    } while (0)


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
        c->d_alphaT = nullptr; c->d_r = nullptr; c->d_omega = nullptr; c->d_alphaT32 = nullptr;
        c->d_tables = nullptr; c->d_recs = nullptr; c->d_ids = nullptr; c->d_dir = nullptr; c->d_inv = nullptr; c->d_ids_bk = nullptr; c->d_bin16 = nullptr;
        c->d_java_hash = nullptr; c->d_deleted_bits = nullptr;
        if (!c->store_owned) c->d_store = nullptr;
    }
    free_devt(c->d_alphaT); free_devt(c->d_r); free_devt(c->d_omega); free_devt(c->d_alphaT32); free_dev(c->ws_fix.p);
    free_devt(c->d_tables); free_devt(c->d_recs); free_devt(c->d_ids); free_devt(c->d_dir);
    free_devt(c->d_java_hash); free_devt(c->d_unmodelled);
    if (uint32_t* db = c->d_deleted_bits.exchange(nullptr)) (void)hipFree(db);
    if (c->store_owned) free_dev(c->d_store);
    free_dev(c->ws_tickfix.p); free_dev(c->d_fixparams); free_dev(c->ws_gt.p); free_dev(c->bld_codes.p);
    free_dev(c->ws_route.p); free_dev(c->ws_refine.p); free_dev(c->ws_probe.p); free_dev(c->ws_ovf.p); free_dev(c->ws_search.p); free_devt(c->d_inv); free_devt(c->d_ids_bk); free_devt(c->d_bin16);
    for (hipEvent_t e : c->rt_events) (void)hipEventDestroy(e);
    for (auto& b : c->ws_io) free_dev(b.p);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
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
    c->d_alphaT = root->d_alphaT; c->d_r = root->d_r; c->d_omega = root->d_omega; c->d_alphaT32 = root->d_alphaT32;
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
    free_devt(c->d_alphaT); free_devt(c->d_r); free_devt(c->d_omega); free_devt(c->d_alphaT32);
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

}  // namespace

extern "C" {

int fspann_route_dev(fspann_ctx* c, int64_t nq, const uint64_t* codes_dev, int probe_override, int32_t limit,
                     int64_t cap, int32_t* ids_dev, int32_t* score_dev, int32_t* count_dev, int32_t* kept_dev,
                     int32_t* raw_seen_dev) {
    CHECK_CTX(c);
    if (!c->frozen) return fail(FSPANN_E_STATE, "Index not finalized");  // PIS:594
    if (nq < 0) return fail(FSPANN_E_ARG, "nq < 0");
    if (nq == 0) return FSPANN_OK;
    RoutePlan pl;
    RouteParams p{};
    bool fused = false;
    int rc = prepare_route(c, nq, codes_dev, probe_override, limit, cap, ids_dev, score_dev, count_dev, kept_dev, raw_seen_dev, &pl, &p, &fused);
    if (rc) return rc;
    if (!fused && (rc = launch_route_probe(c, p, pl))) return rc;
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
    c->last_route_lazy = pl.lazy;
    if (pl.lazy) {
        if (pl.lz_entries == 512) {
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
    }
    if (pl.lds_mode) { if (pl.threads == 1024) FSP_LAUNCH_SEL(true, 1024); else FSP_LAUNCH_SEL(true, 512); }
    else { if (pl.threads == 1024) FSP_LAUNCH_SEL(false, 1024); else FSP_LAUNCH_SEL(false, 512); }
#undef FSP_LAUNCH_SEL
    FSP_HIP(hipGetLastError());
    return FSPANN_OK;
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
        if ((rc = ensure(c, c->ws_io[0], cb))) return rc;
        if ((rc = ensure(c, c->ws_io[1], 2 * ob_a + cnt_a))) return rc;       // ids | scores | count, kept, rawSeen: one block
        unsigned char* dv = static_cast<unsigned char*>(c->ws_io[1].p);
        int32_t* ids_d = reinterpret_cast<int32_t*>(dv), *sc_d = reinterpret_cast<int32_t*>(dv + ob_a), *cnt_d = reinterpret_cast<int32_t*>(dv + 2 * ob_a);
        std::memcpy(hp, codes, cb);
        FSP_HIP(hipMemcpyAsync(c->ws_io[0].p, hp, cb, hipMemcpyHostToDevice, c->stream));
        rc = fspann_route_dev(c, nq, static_cast<const uint64_t*>(c->ws_io[0].p), probe_override, limit, cap, ids_d, sc_d, cnt_d,
                              kept ? cnt_d + nq : nullptr, raw_seen ? cnt_d + 2 * nq : nullptr);
        if (rc) return rc;
        for (int pass = 0; pass < 2; pass++) {
            FSP_HIP(hipMemcpyAsync(hp, dv, 2 * ob_a + cnt_a, hipMemcpyDeviceToHost, c->stream));
            FSP_HIP(hipStreamSynchronize(c->stream));
            const int32_t* cnt_h = reinterpret_cast<const int32_t*>(hp + 2 * ob_a);
            bool flagged = false;
            for (int64_t i = 0; i < nq; i++) flagged = flagged || cnt_h[i] < 0;
            if (!flagged || pass == 1) break;
            // (rare) a bestScore map treeified a bin: finished by the literal JDK model on the host, then fetched again
            rc = guarded([&]() -> int {
                return resolve_unmodelled(c, nq, static_cast<const uint64_t*>(c->ws_io[0].p), probe_override, limit, cap, ids_d, sc_d, cnt_d,
                                          kept ? cnt_d + nq : nullptr, raw_seen ? cnt_d + 2 * nq : nullptr, nullptr, nullptr);
            });
            if (rc) return rc;
        }
        const int32_t* cnt_h = reinterpret_cast<const int32_t*>(hp + 2 * ob_a);
        std::memcpy(ids, hp, ob);
        if (score) std::memcpy(score, hp + ob_a, ob);
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
            if ((rc = ensure(c, c->ws_io[0], up))) return rc;
            if ((rc = ensure(c, c->ws_io[1], cb))) return rc;
            if ((rc = ensure(c, c->ws_io[4], down))) return rc;
            unsigned char* du = static_cast<unsigned char*>(c->ws_io[0].p), *dd = static_cast<unsigned char*>(c->ws_io[4].p);
            std::memcpy(hp, q, qb);
            std::memcpy(hp + al(qb), cand_ids, ib);
            std::memcpy(hp + al(qb) + al(ib), cand_count, nb);
            FSP_HIP(hipMemcpyAsync(du, hp, up, hipMemcpyHostToDevice, c->stream));
            FSP_HIP(hipMemcpyAsync(c->ws_io[1].p, cand, cb, hipMemcpyHostToDevice, c->stream));
            int32_t* cnt_out = reinterpret_cast<int32_t*>(dd + al(ob_d) + al(ob_i));
            rc = fspann_refine_dev(c, nq, du, dtype, c->ws_io[1].p, dtype, B, reinterpret_cast<int32_t*>(du + al(qb)),
                                   reinterpret_cast<int32_t*>(du + al(qb) + al(ib)), k, reinterpret_cast<int32_t*>(dd + al(ob_d)),
                                   reinterpret_cast<double*>(dd), cnt_out, cnt_out + nq);
            if (rc) return rc;
            FSP_HIP(hipMemcpyAsync(hp, dd, down, hipMemcpyDeviceToHost, c->stream));   // (stream order: the way up has been read by then)
            FSP_HIP(hipStreamSynchronize(c->stream));
            std::memcpy(out_dist, hp, ob_d);
            std::memcpy(out_ids, hp + al(ob_d), ob_i);
            std::memcpy(out_count, hp + al(ob_d) + al(ob_i), nb);
            if (scored) std::memcpy(scored, hp + al(ob_d) + al(ob_i) + nb, nb);
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
        const bool stream_ok = t->ref_q_dtype == FSPANN_F32 && rows_dtype == FSPANN_F32 && nchunks == 1 && (d % 4 == 0) &&
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

#ifdef FSPANN_DEBUG_STAMPS
// debug builds only (tools/route_stamps.py): per-block phase stamps of the route kernels (dev pointer to [grid][16] int64)
int fspann_debug_route_stamps(fspann_ctx* c, void* dev_ptr) {
    if (!c) return FSPANN_E_NULL;
    c->dbg_route = static_cast<long long*>(dev_ptr);
    return FSPANN_OK;
}
#endif

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

// ---- native Setup: code all vectors on the GPU, cut partitions -----------------------------
// Replaces PIS.insert's coding loop (PIS:331-346) + PIS.build (PIS:372-434) +
// GreedyPartitioner.build (idx/GreedyPartitioner.java:37-76).  The reference iterates a
// HashMap<String,BitSet>(staged.size()) and stable-sorts by key, so elements with equal keys keep
// HashMap iteration order = (bucket at the final capacity, insertion order) — the closed form used
// here (valid while no bin treeifies; DESIGN.md "Java order key").
}  // extern "C"
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

#include "api_ext.hip.h"
