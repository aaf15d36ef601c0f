// api_common.hip.h — helpers shared by every entry point of include/fspann.h: sizes, the pinned block, guarded(), the context checks
// Part of the single translation unit fspann_api.hip (included there, in order); product code, no CPU fallback.
#pragma once

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
    if (!c->h_pin) {
        if (hipHostMalloc(&c->h_pin, kPinBytes, hipHostMallocDefault) != hipSuccess) { c->h_pin = nullptr; (void)hipGetLastError(); }
        else if (hipHostGetDevicePointer(&c->d_pin, c->h_pin, 0) != hipSuccess) { c->d_pin = nullptr; (void)hipGetLastError(); }
    }
    return c->h_pin != nullptr;
}
// Calls of a handful of queries (QueryService.search is one token per call, ForwardSecureANNSystem.java:636): the kernels read their
// arguments from the pinned block and write their results into it THROUGH THE BUS — no copy command either way, one launch sequence and
// one synchronisation (a copy command costs ~8 us of the stream's time whatever its size; a few KB written by the kernel itself cost
// less).  More queries than this go through the block with one copy each way: thousands of scattered 4-byte stores over PCIe do not.
constexpr int64_t kZeroCopyMaxQ = 4;
bool zero_copy_ok(fspann_ctx* c, int64_t nq) { return nq <= kZeroCopyMaxQ && c->knob_zero_copy && pin_block(c) && c->d_pin != nullptr; }

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
