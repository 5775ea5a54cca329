// ctx.h -- internal state behind the C ABI (include/hagrid_amd.h).  Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "hagrid_amd.h"

namespace hagrid_impl {

struct Slot {
    void* ptr = nullptr;
    size_t size = 0;
    bool in_use = false;
    int refs = 0;          // tracked pointers into this slot: 1 after alloc; an unpacked grid blob is one slot with four
};

} // namespace hagrid_impl

namespace hagrid_impl {

// Traversal image (trav_image.hip): one per context, derived from the grid of the last hagrid_setup_traversal call.
struct TravImageCache {
    void* table = nullptr;          // uint2 per top-level cell; general layout: the wide records (16 bytes each)
    size_t table_bytes = 0;
    int wide_records = 0;           // table and general layouts: cells whose bounds the records' bytes cannot hold (a 16-byte wide record each, behind the table)
    void* blocks = nullptr;         // the slim records, 16 bytes each (uniform / table layout: a block per top-level cell; general layout: one per voxel-map entry)
    size_t block_bytes = 0;
    bool valid = false;
    bool uniform = false;           // flat, every block at the full resolution: block T starts at T * (2^shift)^3 records
    bool general = false;           // flat, one slim record per voxel-map entry at the entry's index (links to child blocks, wide records): any depth
    int slim = 0;                   // bits per inline reference id of the slim records (20: four ids, 26: three)
    int vtop_k = 0;                 // general layout: levels between the voxel map's top level and the image's virtual top level (0: none, 1)
    uint32_t vtop_base = 0;         // general layout: index of the first record of the virtual top level (behind the records of the entries)
    bool detached = false;          // hagrid_grid_release_for_traversal freed entries and cells; the image stands for them
    bool borrowed = false;          // hagrid_share_traversal: table and blocks belong to another context's pool, never freed here
    // Set by the context that built the image, cleared when THAT context drops it (new setup, construction pass, a source array
    // freed, context destroyed).  Borrowers hold the same flag: a borrowed image whose flag is down is refused, not read.
    std::shared_ptr<std::atomic<bool>> alive;
    // identity of the source grid
    const void* entries = nullptr; const void* cells = nullptr; const void* refs = nullptr;
    int num_cells = 0, num_entries = 0, num_refs = 0, shift = 0, dims[3] = {0, 0, 0}, cell_bytes = 32;
    int max_ref = -1;               // largest primitive id the grid refers to
    // A uniform layout that costs more than a quarter more records than the table layout is built NEXT TO the table layout (trav_image.hip build_blocks): rays in image
    // order gather from the uniform one (the record of a voxel by arithmetic: -7 ... -13 % on grids of three levels), batches without coherence -- binned ones -- from the
    // compact one (a 977 MB image instead of 120 MB costs them a quarter more time).  alt_blocks == nullptr: one layout serves every batch.
    void* alt_blocks = nullptr; void* alt_table = nullptr; size_t alt_block_bytes = 0, alt_table_bytes = 0; int alt_wide_records = 0, alt_slim = 0;
};

} // namespace hagrid_impl

// Words of the context's pinned mailbox and of its device scratch that have a fixed owner (the first 256 mailbox words are the read-back area of the
// construction passes; every other use of dscratch is a pass-local counter named where it is used).
namespace hagrid_impl {
enum MailboxWord : int {
    kMbxRowLen = 300,          // + hint slot (4): row length found for a ray buffer (traverse.hip; copied behind the launch, polled)
    kMbxOrderStale = 304,      // + hint slot (4): epoch of a tile order whose buffer holds other rays now (written by the kernel's first wavefront, polled)
    kMbxReadBackEpoch = 310,   // ctx.hip read_back: the epoch the publishing wavefront stores behind the words
    kMbxHeadSuggest = 312,     // + hint slot (4): tiles the last sort suggests for the four-lanes-per-ray head (tile_order_kernel, polled)
    kMbxPriorStats = 316,      // + 4: the grid prior's statistics (trav_prior.hip)
};
enum ScratchWord : int {
    kScrRowLenBinned = 232,    // ray_order.hip: row length / flag of the binning pass
    kScrRowLen = 236,          // + hint slot (4): row length of a ray buffer (detect_ray_rows writes, the kernels read)
    kScrMaxRef = 240,          // trav_image.hip: largest reference id
    kScrBlobFlag = 250,        // blob.hip: validation flag
};
} // namespace hagrid_impl

struct hagrid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool keep = false;
    int num_cus = 0;

    // buffer pool (MemManager backend)
    std::vector<hagrid_impl::Slot> slots;
    std::unordered_map<void*, int> tracker;
    size_t usage = 0, max_usage = 0;

    // profile()
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;

    // pinned mailbox for scalar read-backs (build passes) -- 256 ints, + words 256.. for values the host only polls (named in MailboxWord above)
    int* mailbox = nullptr;
    // device scratch words (counters, scan totals) -- 256 ints, zeroed by the passes that use them
    int* dscratch = nullptr;
    unsigned long long* lb_state = nullptr;   // status words of the look-back scans (wave_prims.h), never cleared: epochs
    size_t lb_words = 0;
    unsigned lb_epoch = 0;
    int* row_scores = nullptr;           // row-length detection from origins: one score per candidate + a ticket counter
    int* bin_diff = nullptr;             // automatic ray binning: 64 partial counts of neighbouring rays in different bins

    // traversal options (hagrid_set_ray_binning, hagrid_set_option)
    int ray_binning = 0;
    int opt_variant = 0;        // 0 = the image kernel when the grid has an image, else v2; 1 / 2 / 4 = force the reference-shaped kernel / v2 / the image kernel
    int opt_lookback = 1;        // scans of the construction passes: single-pass decoupled look-back
    int opt_merge_inplace = 1;   // merge_grid: iterations in place (dirty cells only, one compaction at the end) once an iteration has merged less than half of its cells (merge.inplace_div)
    int opt_merge_inplace_iters = 0;   // tests: leave the in-place mode after this many iterations (0: only for lack of room), the next iteration compacts
    int opt_merge_inplace_div = 0;     // the mode is entered once an iteration merges less than 1 / this of its cells (0: the default, 2); tests enter earlier or later
    int opt_merge_inplace_room = 0;    // tests: the in-place mode may use the reference buffer up to this index only (0: all of it) -- the overflow path
    int opt_merge_narrow = 1;    // merge_grid: 16-byte working cell records between the passes (0: the 32-byte record throughout)
    int opt_expand_voxel_map = 1;     // expand_grid: the voxel map resolved into one word per voxel for the passes' look-ups (expand.hip); 0: the chain through the levels
    int opt_expand_subset_only = 1;   // expand_grid: 1 = the reference's compiled setting, 0 = precise (compute_overlap)
    int opt_image_width = 0;    // tile packets: -1 off, 0 detect the row length on the device, > 0 row length given by the caller
    int opt_band_rows = 0;      // tile packets: rows of super-tiles per band; 0 = as many as make the in-flight tiles a square block of the image
    int opt_super_log2 = 3;     // tile packets: 2^k x 2^k tiles per super-tile (Z order inside); round-3 sweep: 3 (profiles/dev_r3_tile_params.txt)
    int opt_xcd_chunk_log2 = -2; // tile packets: the XCDs take chunks of 2^k blocks in turn; -1 = one eighth of the block range each; -2 = by launch size (3 below twelve rounds of wavefronts, else 5)
    int opt_narrow = 1;         // v2: 32-bit offsets / 24-bit multiplies when the arrays allow it
    int opt_image_max_mb = 0;   // traversal image: size limit in MB (0 = max(1 GB, 8x the arrays it replaces)); an image beyond it is not built
    int opt_quad_head = 20;     // tail kernel: the tiles of a learned order that cost at least this many TENTHS of the median working tile -- if they are more than a twelfth of the tiles -- start with four lanes per ray, first (0: never)
    int opt_image_vtop = 1;     // traversal image, general layout: records of a virtual top level one level below the map's (0: look-ups start at the map's top level, rounds 1-5a)
    int opt_image_uniform = 1;  // traversal image: use the table-free uniform layout when it is not much bigger than the table layout (2: whatever it costs; 0: never)
    int opt_row_cache = 1;      // tile packets: the row length found for a ray buffer is reused by the next 15 calls with the same buffer and count
    // What the context remembers about a ray buffer it has traversed (traverse.hip): the row length found for it and the order of its tiles.  A few
    // buffers are remembered at once (a renderer that alternates between two or three ray buffers keeps the hints of each); the least recently used
    // slot is taken over by a new buffer.  Slot i owns the device word dscratch[kScrRowLen + i] and the pinned words mailbox[kMbxRowLen + i], [kMbxOrderStale + i], [kMbxHeadSuggest + i].
    struct RayHints {
        const void* key_rays = nullptr; int key_n = 0;       // the buffer the slot belongs to
        const void* rowlen_rays = nullptr; int rowlen_n = 0, rowlen_age = 0, rowlen_known = -1, rowlen_seen = 0; bool rows_from_origins = false /* the row length came from neighbouring origins alone: an image order without coherent directions (bounce rays) */;   // rowlen_known: the row length as the host has seen it (-1: not yet)
        hipEvent_t rowlen_evt = nullptr; bool rowlen_pending = false;
        // tile order of the tail kernel: cost | order, lpt_cap ints each; the order is valid for launches over (lpt_rays, lpt_n)
        int* lpt_buf = nullptr; int lpt_cap = 0; const void* lpt_rays = nullptr; int lpt_n = 0, lpt_blocks = 0, lpt_age = 0, lpt_period = 32, lpt_rot = 0 /* positions the stored order is rotated by: its last lpt_rot tiles are the longest */; bool rot_adopted = false /* the first suggestion of a sort was taken up by a sort of its own */;
        // the head share measures itself: launches in the learned order are timed (an event pair around the kernel, polled by later calls) without it and with it --
        // three samples each, the smaller ones compared -- and a share that does not pay is dropped for as long as the order lives
        hipEvent_t trial_evt[2] = {nullptr, nullptr}; bool trial_pending = false, trial_with_head = false, head_disabled = false;
        float t_base = 0.0f, t_head = 0.0f; int n_base = 0, n_head = 0, trial_opt = -1 /* the value of traverse.quad_head the trial belongs to */; unsigned head_serial = 0 /* ... and the traversal image (ctx->image_serial) */; bool lpt_valid = false;
        // The order is only as good as the rays it was learned on: the sort leaves a copy of one sample ray of the buffer behind the order
        // (lpt_buf + 2 * lpt_cap: 2 float4), the first wavefront of every launch compares them with the buffer's rays ON THE DEVICE, bit for bit, and when the
        // buffer holds other rays (refilled, recycled address, a camera that moved) reports the order's epoch in the pinned word mailbox[kMbxOrderStale + i],
        // which the host polls: that launch is the only one that follows the stale order, the order is learned again.  Orders that do not last (a camera that moves fast) are not learned for a while (cooldown).
        int lpt_epoch = 1 /* never 0: the pinned report word starts as 0 and is reset to -1 */, relearn_streak = 0, cooldown = 0, cooldown_len = 64; unsigned long long relearn_clock = 0;     // (cooldown_len: doubles with every give-up in a row, up to 1024 launches)
        // the share trial of launches in the default order (traverse.hip "traverse.share_trial"): candidates = the rule's share of tiles with four lanes per ray, a half, none, all
        // (sample k = candidate k % share_ncand, each with its own event pair: all may be in flight at once)
        hipEvent_t share_evt[12][2] = {}; int share_issued = 0, share_done = 0, share_choice = -1 /* index into share_cands; -1: being measured */, share_ncand = 0 /* candidates of the running trial */, share_shape_nc = 0 /* ... of the launch shape's first trial */, share_cands[4] = {0, 0, 0, 0} /* per cent */,
            share_last = -1 /* the share (per cent) the last trial chose */, share_launches = 0; float share_t[4] = {0.0f, 0.0f, 0.0f, 0.0f}; unsigned share_serial = 0 /* ctx->image_serial the answer belongs to */;
        int trial_kind = 0 /* the timed launch in flight: 0 the order alone, 1 with the head share, 2 with ALL tiles four lanes per ray */, n_all = 0, all_stage = 0 /* 0 not tried, 1 - 2 learning its own order, 3 timed, 4 decided */; float t_all = 0.0f; bool learned_all = false;
        bool learned_once = false;          // the first share trial of this launch shape has been decided and the order learned behind it
        unsigned order_serial = 0;          // ctx->image_serial the learned tile order belongs to
        bool cmp_pending = false, cmp_done = false; int n_conf = 0; float t_conf = 0.0f;     // the order against the default order: three default-order launches timed next to the order's own samples
        bool order_loses = false;           // the learned order's steady launches were not 3 % faster than the best default-order launch: not followed until the next trial
        int last_report = -1;               // the report word (mailbox[kMbxOrderStale + slot]) as last seen
        unsigned long long used = 0;                    // clock of the last call that used the slot
    };
    static constexpr int kRayHints = 4;
    RayHints hints[kRayHints];
    unsigned long long hint_clock = 0;
    int opt_mailbox = -1;       // tail kernel: per ray a mailbox of the last four triangles it was tested against (LDS); -1: chosen per launch
    int opt_tail = 1;           // table-free slim image, nearest hit: the kernel with the tail mode (four lanes per ray once a wavefront holds at most 16 live rays)
    int opt_lds_pad = 0;         // experiments: dynamic LDS bytes per block of the tail kernel
    int opt_tail_dual = -1;      // tail kernel, phase 1: two ids of an inline list per round trip (trav_kernels.h, test_list); -1: chosen per launch
    int opt_share_trial = 1;            // launches in the default order: the share of tiles that start with four lanes per ray measures itself (the rule's share against a half); 0: the rule
    int opt_order_gate = 1;             // ... and only while the buffer holds the rays it was learned on (0: the order is followed unseen -- A/B runs)
    int opt_tile_order = -1;     // tail kernel: tiles dispatched longest first, by the costs the previous launches over the same ray buffer left; -1: chosen per launch
    int opt_quad_tail = -1;      // per cent of the tiles (the last in dispatch order) that start with four lanes per ray; -1: chosen per launch
    int opt_image_slim = 1;     // traversal image: 1: reference ids packed in 20 bits where every id fits, else 26; 2: always 26 bits (tests)
    int opt_image_general = 1;  // traversal image: the general layout of slim records (a record per voxel-map entry) for grids deeper than three levels and for cells the block layouts' bound bytes cannot hold; 0: no general layout -- such grids are traversed in the construction format (tests); 2: for every grid (tests)
    int opt_image = 2;          // 0: none; 1 / 2: setup_traversal builds the traversal image (trav_image.hip; the two values are one since round 5) and traverse_grid uses it

    int opt_id_is_steps = 0;    // hagrid_traverse_grid writes the reference kernel's step count into Hit.id (traverse.cu:93) instead of the primitive id

    hagrid_impl::TravImageCache image;
    unsigned image_serial = 0;           // counts the traversal images this context has built (what is measured per launch shape belongs to one of them)
    int readback_epoch = 0;               // read_back (ctx.hip): the epoch the publishing wavefront leaves behind the words in the mailbox (word 310)
    int opt_fast_readback = 1;            // scalar read-backs through a publishing wavefront and a spinning host instead of hipMemcpyAsync + hipStreamSynchronize
    int build_arena_tris = 0;             // ... and the primitives of that construction
    size_t build_arena_hint = 0;          // bytes of temporaries the last build_grid of this context asked for (build.hip: one pool buffer for all of them)
    hagrid_build_counts counts = {};      // sizes of the last construction (hagrid_get_build_counts)

    std::string err;
};

namespace hagrid_impl {

inline int fail(hagrid_ctx* ctx, int code, const char* file, int line, const char* msg) {
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s(%d): %s", file, line, msg);
        ctx->err = buf;
    }
    return code;
}

#define HG_FAIL(ctx, code, msg) return ::hagrid_impl::fail((ctx), (code), __FILE__, __LINE__, (msg))
#define HG_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return ::hagrid_impl::fail((ctx), HAGRID_EHIP, __FILE__, __LINE__, hipGetErrorString(e_)); \
    } while (0)
#define HG_TRY(expr)                                                                              \
    do {                                                                                          \
        int rc_ = (expr);                                                                         \
        if (rc_ < 0) return rc_;                                                                  \
    } while (0)

// Debug build (HAGRID_DEBUG_SYNC=1 python hagrid_amd/build.py --force -> -DHAGRID_DEBUG_SYNC): after every kernel launch of a pass the
// stream is drained and the runtime's error state checked; a failure prints "file(line): message" and aborts -- the reference's
// DEBUG_SYNC of non-NDEBUG builds (common.h:95-108).  In a normal build the macro is empty.
void debug_sync(hagrid_ctx* ctx, const char* file, int line);
#ifdef HAGRID_DEBUG_SYNC
#define HG_DBG(ctx) ::hagrid_impl::debug_sync((ctx), __FILE__, __LINE__)
#else
#define HG_DBG(ctx) do { } while (0)
#endif

// pool access for the passes (typed convenience over hagrid_mem_alloc)
template <typename T>
inline T* pool_alloc(hagrid_ctx* ctx, size_t n) {
    return static_cast<T*>(hagrid_mem_alloc(ctx, (n ? n : 1) * sizeof(T)));
}

// An OPTIONAL buffer (a pass runs without it, only slower): a failed allocation leaves neither the runtime's sticky out-of-memory error nor a message in the
// context behind -- the pass's final hipGetLastError() would otherwise turn its success into HAGRID_EHIP.
template <typename T>
inline T* pool_try_alloc(hagrid_ctx* ctx, size_t n) {
    const std::string saved = ctx->err;
    T* p = pool_alloc<T>(ctx, n);
    if (!p) { (void)hipGetLastError(); ctx->err = saved; }
    return p;
}

// Pool buffers that are released on every exit path of a pass.
struct PoolTemps {
    hagrid_ctx* ctx;
    std::vector<void*> ptrs;
    explicit PoolTemps(hagrid_ctx* c) : ctx(c) {}
    PoolTemps(const PoolTemps&) = delete;
    PoolTemps& operator=(const PoolTemps&) = delete;
    template <typename T> T* get(size_t n) { T* p = pool_alloc<T>(ctx, n); if (p) ptrs.push_back(p); return p; }
    void drop(void* p) { for (auto& q : ptrs) if (q == p && p) { hagrid_mem_free(ctx, p); q = nullptr; } }
    void* keep(void* p) { for (auto& q : ptrs) if (q == p) q = nullptr; return p; }       // ownership passes to the caller
    ~PoolTemps() { for (void* p : ptrs) if (p) hagrid_mem_free(ctx, p); }
};

// Turns the pool buffer `base` into `n` separately freeable buffers parts[0..n) that lie inside it (a grid blob unpacked in
// place): `base` stops being a pool pointer, every part becomes one, the memory is released with the last of them.
int pool_split(hagrid_ctx* ctx, void* base, void* const* parts, int n);

// Reads `count` ints from device memory into host memory after draining the stream.
int read_back(hagrid_ctx* ctx, const void* dptr, void* hptr, size_t bytes);

inline int grid_blocks(long long n, int block) { return (int)((n + block - 1) / block); }

// Status words for one look-back scan of `tiles` tiles with `words_per_tile` words each, and a fresh epoch.
unsigned long long* lookback_state(hagrid_ctx* ctx, int tiles, int words_per_tile, unsigned* epoch);

// trav_image.hip
int trav_image_build(hagrid_ctx* ctx, const hagrid_grid* grid);       // leaves the image invalid for grids it does not cover
void trav_image_drop(hagrid_ctx* ctx);
bool trav_image_matches(const hagrid_ctx* ctx, const hagrid_grid* grid);
bool trav_image_stale(const hagrid_ctx* ctx);        // a borrowed image whose owner dropped it
// a pool buffer is freed or overwritten: the image goes if it was derived from that buffer
void trav_image_source_touched(hagrid_ctx* ctx, const void* ptr, size_t bytes);

} // namespace hagrid_impl
