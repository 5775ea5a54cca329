// hagrid/mem_manager.h -- MemManager and profile() with the reference's interface
// (src/mem_manager.h:34-119, src/common.h:15), as header-only shims over the C ABI.
//
// The reference's MemManager has seven out-of-line members implemented with the CUDA runtime
// (mem_manager.cu:6-75).  Here the class owns one hagrid_ctx (device + stream + buffer pool) and every
// member forwards to a hagrid_mem_* entry point of libhagrid_amd.so.  The manager created last is the
// "current" one: setup_traversal / traverse_grid / profile take no manager in the reference's API.
#ifndef HAGRID_MEM_MANAGER_H
#define HAGRID_MEM_MANAGER_H

#include <cstdlib>
#include <functional>
#include <iostream>

#include "common.h"
#include "../hagrid_amd.h"

namespace hagrid {

enum class Copy { HST_TO_DEV, DEV_TO_HST, DEV_TO_DEV };

namespace detail {
inline hagrid_ctx*& current_ctx() { static hagrid_ctx* ctx = nullptr; return ctx; }
/// Error convention of the reference: print "file(line): message", abort (common.h:103-108).
inline void check(hagrid_ctx* ctx, int rc) {
    if (rc < 0) { std::cerr << hagrid_last_error(ctx) << std::endl; std::abort(); }
}
} // namespace detail

class MemManager {
public:
    /// keep = retain freed buffers for later builds (faster re-builds, more memory).
    /// The device is HAGRID_DEVICE or LOCAL_RANK from the environment, else 0.
    explicit MemManager(bool keep = false) : ctx_(nullptr) {
        int device = 0;
        if (const char* e = std::getenv("HAGRID_DEVICE")) device = std::atoi(e);
        else if (const char* r = std::getenv("LOCAL_RANK")) device = std::atoi(r);
        if (hagrid_ctx_create(&ctx_, device, keep ? 1 : 0) != HAGRID_OK) {
            std::cerr << "hagrid: cannot create a context on device " << device << std::endl;
            std::abort();
        }
        detail::current_ctx() = ctx_;
    }
    ~MemManager() {
        if (detail::current_ctx() == ctx_) detail::current_ctx() = nullptr;
        hagrid_ctx_destroy(ctx_);
    }
    MemManager(const MemManager&) = delete;
    MemManager& operator=(const MemManager&) = delete;

    template <typename T> T* alloc(size_t n) {
        void* p = hagrid_mem_alloc(ctx_, n * sizeof(T));
        if (!p) detail::check(ctx_, HAGRID_ENOMEM);
        return static_cast<T*>(p);
    }
    template <typename T> void free(T* ptr) { detail::check(ctx_, hagrid_mem_free(ctx_, const_cast<void*>(static_cast<const void*>(ptr)))); }

    template <Copy type, typename T> void copy(T* dst, const T* src, size_t n) {
        const size_t bytes = sizeof(T) * n;
        if (type == Copy::DEV_TO_DEV) detail::check(ctx_, hagrid_mem_copy_d2d(ctx_, dst, src, bytes));
        else if (type == Copy::DEV_TO_HST) detail::check(ctx_, hagrid_mem_copy_d2h(ctx_, dst, src, bytes));
        else detail::check(ctx_, hagrid_mem_copy_h2d(ctx_, dst, src, bytes));
    }
    template <typename T> void zero(T* ptr, size_t n) { detail::check(ctx_, hagrid_mem_zero(ctx_, ptr, n * sizeof(T))); }
    template <typename T> void one(T* ptr, size_t n) { detail::check(ctx_, hagrid_mem_one(ctx_, ptr, n * sizeof(T))); }

    void debug_slots() const { hagrid_mem_debug_slots(ctx_); }
    size_t usage() const { return hagrid_mem_usage(ctx_); }
    size_t max_usage() const { return hagrid_mem_max_usage(ctx_); }

    /// The C ABI handle (for hagrid_ctx_set_stream and friends).
    hagrid_ctx* context() const { return ctx_; }
    /// Makes this manager the one the free functions (build_grid, traverse_grid, profile ...) work on; the last manager
    /// constructed is current by default.  Several managers = several contexts / streams (traverse.h: share_traversal).
    void make_current() { detail::current_ctx() = ctx_; }
    /// Waits for all work queued on this manager's stream.
    void synchronize() { detail::check(ctx_, hagrid_ctx_synchronize(ctx_)); }

private:
    hagrid_ctx* ctx_;
};

/// Device milliseconds spent while f runs: an event pair on the current manager's stream.
inline float profile(std::function<void()> f) {
    hagrid_ctx* ctx = detail::current_ctx();
    if (!ctx) { std::cerr << "hagrid: profile() needs a MemManager" << std::endl; std::abort(); }
    detail::check(ctx, hagrid_profile_begin(ctx));
    f();
    return hagrid_profile_end(ctx);
}

} // namespace hagrid

#endif // HAGRID_MEM_MANAGER_H
