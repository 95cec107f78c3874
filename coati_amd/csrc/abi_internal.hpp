// abi_internal.hpp -- what the translation units behind the C ABI of include/coati_hip.h share: the two handle types,
// the error channel, the model's workspace cache, how a batch is planned (BatchOpts / batch_create_impl).
//   abi.hip          handles, launches, result transfer, debug exports
//   plan.hip         batch_create_impl: validation, strip plans, kernel choice, workspace layout, upload
//   pipeline.hip     coati_hip_viterbi_batch: the streamed form (one persistent kernel) and the chunk pipeline
//   sample_host.hip  coati_hip_sampleback: the host loop of the speculative exact-stream sampler
#ifndef COATI_HIP_ABI_INTERNAL_HPP
#define COATI_HIP_ABI_INTERNAL_HPP
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace coati_hip_detail {
// sampleback.hip: the samples' ops, right-aligned in slots of la + lb bytes, packed back to back (offsets by an exclusive scan
// over the lengths; *total = bytes in `packed`)
hipError_t launch_ops_pack(const uint8_t* ops, const uint64_t* start, const uint32_t* len, uint64_t n, uint64_t* packed_off, uint64_t* total, uint8_t* packed,
                           hipStream_t stream);
}
using namespace coati_hip_detail;

/* A few helper threads that live as long as their model (started by the first call that wants them): the streamed call
 * plans its first chunks on them.  Creating a thread per task (std::async) cost ~40 us each on the calling thread --
 * as much as planning the chunk -- waking a sleeping one ~5 us. */
class HelperPool {
  public:
    explicit HelperPool(int n) {
        for(int i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
    }
    ~HelperPool() {
        {
            std::lock_guard<std::mutex> g(lock_);
            stop_ = true;
        }
        wake_.notify_all();
        for(std::thread& t : threads_) t.join();
    }
    HelperPool(const HelperPool&) = delete;
    HelperPool& operator=(const HelperPool&) = delete;
    template<class F>
    auto submit(F f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
        auto fut = task->get_future();
        {
            std::lock_guard<std::mutex> g(lock_);
            tasks_.emplace_back([task] { (*task)(); });
        }
        wake_.notify_one();
        return fut;
    }

  private:
    void run() {
        for(;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> g(lock_);
                wake_.wait(g, [this] { return stop_ || !tasks_.empty(); });
                if(tasks_.empty()) return;  // (stop_)
                job = std::move(tasks_.front());
                tasks_.pop_front();
            }
            job();
        }
    }
    std::mutex lock_;
    std::condition_variable wake_;
    std::deque<std::function<void()>> tasks_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

struct coati_hip_model {
    int device = 0;
    int gap_len = 1;
    GapConsts k{};
    uint32_t n_tables = 1;
    // COATI_HIP_OPT_FORWARD_MODE: how the log semiring's plus is evaluated by the Forward kernels of batches created from now on
    // (COATI_HIP_FORWARD_EXACT: glibc's expf / log1pf restated, the CPU's bits; COATI_HIP_FORWARD_TOLERANCE: hardware exp2 / log2)
    std::atomic<int> forward_mode{0};
    float* d_table = nullptr;  // n_tables * 183*15 floats
    hipStream_t stream = nullptr;
    // Workspaces of destroyed batches, kept for the next batch_create (hipMalloc of a multi-GB
    // workspace was measured at 0.4 ms when the driver still had the pages and 250-550 ms when it
    // did not).  At most kCachedArenas are kept (batch workspaces and the sampler's temporaries); coati_hip_model_trim / model_destroy free them.
    struct Arena {
        void* ptr;
        uint64_t bytes;
    };
    static constexpr size_t kCachedArenas = 4;
    static constexpr uint64_t kMaxCachedBytes = 20ull << 30;  // larger blocks are freed, not cached (a 48 000-pair chunk of the sharded job: 14-16 GB)
    std::vector<Arena> free_arenas;
    std::mutex arena_lock;
    // the handle itself + one per live batch: coati_hip_model_destroy while batches are alive only
    // marks the model; the last batch_destroy releases it (a batch keeps launching on m->stream)
    std::atomic<int> refs{1};
    // page-locked host staging for the sampler's per-round exchanges (candidate lists down, draw
    // counts up): pageable std::vectors made a round's copies cost between 0.1 and several ms
    // depending on where the process ran; grown on demand, freed with the model
    void* pinned = nullptr;
    uint64_t pinned_bytes = 0;
    // coati_hip_viterbi_batch pipelines its chunks through these slots: each has its own stream, its
    // own page-locked staging block and its own HBM workspace, all kept between calls
    struct Slot {
        hipStream_t stream = nullptr;
        void* pinned = nullptr;
        uint64_t pinned_bytes = 0;
        void* arena = nullptr;
        uint64_t arena_bytes = 0;
    };
    static constexpr int kSlots = 3;
    Slot slots[kSlots];
    // viterbi_ck_stream (one persistent launch per coati_hip_viterbi_batch call): its control block in HBM and
    // the page-locked words the host and the kernel talk through
    void* d_stream_ctl = nullptr;
    void* h_stream = nullptr;  // CkStreamHost
    struct StreamSlot {        // a chunk in flight: its workspace and its page-locked staging block
        void* arena = nullptr;
        size_t arena_bytes = 0;
        void* pinned = nullptr;
        size_t pinned_bytes = 0;
    };
    StreamSlot sslots[kCkStreamSlots];
    static constexpr int kStreamTails = 6;
    void* stream_tail_arena[kStreamTails] = {};  // workspaces of a call's last chunks (their pairs are cut into row parts and keep their checkpoints)
    size_t stream_tail_bytes = 0;
    void* d_stream_waves = nullptr;  // per-wavefront checkpoint slots + traceback scratch, shared by all chunks of a call
    size_t stream_waves_bytes = 0;
    hipEvent_t stream_events[2 * kCkStreamSlots] = {};  // [slot]: its download is done; [kCkStreamSlots + slot]: its upload is done
    std::mutex pipeline_lock;  // one pipelined call at a time per model
    uint32_t stream_calls = 0;     // streamed calls this model has served (the first one allocates lazily: a one-shot process pays for what it uses)
    bool stream_unusable = false;  // the persistent kernel's first upload did not arrive in time once (copies not on the copy engine): never again on this model
    uint32_t ck_band = 64;  // viterbi_ck: half width of the kept checkpoint band, kCkBandOff = keep everything (COATI_HIP_OPT_CK_BAND; default: ck_band_setting())
    bool stream_forbidden = false;  // coati_hip_model_set_option(COATI_HIP_OPT_PERSISTENT_CALL, 0): the embedder shares the GPU
    std::unique_ptr<HelperPool> helpers;  // the streamed call's planning helpers (pipeline.hip), made by its first use or by coati_hip_model_prepare
};

struct coati_hip_batch {
    coati_hip_model* model = nullptr;
    uint64_t n_pairs = 0;
    uint64_t cells = 0;
    uint64_t ops_total = 0;    // sum(la + lb)
    uint64_t flag_dwords = 0;  // dwords in the bit-plane arena
    uint64_t bnd_floats = 0;
    uint64_t mdi_floats = 0;   // floats the Forward M/D/I arena needs (allocated on first use)
    uint64_t device_bytes = 0;
    std::vector<PairDesc> desc;
    // device: one workspace allocation, everything below except d_mdi / d_final_mdi points into it
    void* arena = nullptr;
    uint64_t arena_bytes = 0;
    bool arena_owned = true;        // false: the workspace belongs to a pipeline slot of the model
    hipStream_t stream = nullptr;   // where this batch's Viterbi work runs (the model's stream, or a slot's)
    PairDesc* d_desc = nullptr;
    uint32_t* d_order = nullptr;   // pair indices, most cells first
    uint32_t* d_queue = nullptr;   // ticket counter of the persistent fill kernel
    WorkItem* d_items = nullptr;   // viterbi_l1 work list: (pair, strip), longest pairs first
    WorkItem* d_fwd_items = nullptr;  // forward_l1 work list (1024-column strips)
    uint32_t n_fwd_items = 0;
    bool long_pairs = false;     // decision-bit plan with 4-column strips throughout (a few long pairs): viterbi_lp runs it
    bool multi_strip = false;    // the Viterbi strip plan has a pair of more than one strip
    bool ck_keep_all = false;    // viterbi_ck keeps every checkpoint (no band): the debug export decodes every tile
    bool fwd_quad = false;       // the Forward items are quad strips (common.hpp: kFwdQuadCols)
    bool fwd_fast = false;       // the model's Forward mode when the batch was planned (the strip shapes depend on it)
    uint32_t fwd_wlog2_max = 4;  // widest Forward strip shape of the batch (forward_l1 has a leaner build for <= 8 columns per lane)
    uint32_t* d_progress = nullptr;
    uint32_t n_items = 0;
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_ops = nullptr;
    uint32_t* d_flags = nullptr;   // decision bits (viterbi_l1/_k, dp_generic) or checkpoints (viterbi_ck)
    uint32_t* d_wscratch = nullptr;  // viterbi_ck: traceback scratch of the persistent wavefronts
    uint64_t ck_slot_dwords = 0;     // viterbi_ck: per-wavefront checkpoint slots at the start of d_flags (0: none)
    uint32_t ck_split_items = 0;     // viterbi_ck: pairs cut into row parts (the last ones of the LPT order); 0: none
    bool ck_walk_items = false;      // ... and their tracebacks are work items of their own, behind the last row parts (resident batches, round 5)
    bool ck = false;                 // gap_len 1 runs viterbi_ck (checkpoint layout in d_flags)
    float *d_bnd = nullptr, *d_scores = nullptr;
    float *d_mdi = nullptr, *d_final_mdi = nullptr;  // Forward (parts of mdi_block)
    void* mdi_block = nullptr;
    uint64_t mdi_block_bytes = 0;
    bool forward_done = false;
    bool compact = false;  // Viterbi plan is the live-cell layout of viterbi_k (gap_len 2, 3)
    bool compact_narrow_only = false;  // ... and every strip has the narrow shape
    uint64_t* d_ops_start = nullptr;
    uint32_t* d_ops_len = nullptr;
    static constexpr int kTimingRing = 64;  // launches whose kernel times can still be read back
    hipEvent_t ev[kTimingRing][2] = {};  // around each launch
    uint64_t n_launches = 0;
    bool launched = false;
};

namespace coati_hip_abi {

// the error channel of the C ABI: formats into the calling thread's message, returns `code`
int fail(int code, const char* fmt, ...);
// is p inside a block coati_hip_host_alloc handed out (and coati_hip_host_free has not taken back)?
bool host_block_contains(const void* p);
#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if(e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while(0)

BatchDeviceView device_view(const coati_hip_batch* b);
hipError_t model_take_arena(coati_hip_model* m, uint64_t need, void** ptr, uint64_t* bytes);
void model_give_arena(coati_hip_model* m, void* ptr, uint64_t bytes);
hipError_t model_pinned(coati_hip_model* m, uint64_t bytes, void** out);

// fixed-capacity array in the model's page-locked staging block
template <typename T>
struct PinnedVec {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    void push_back(const T& v) { p[n++] = v; }  // (callers keep within cap: see sampleback_speculative)
    size_t size() const { return n; }
    void clear() { n = 0; }
    void resize(size_t k) { n = k; }
    T* data() { return p; }
    T& operator[](size_t i) { return p[i]; }
};
// Carves 256-byte aligned parts out of a block whose size is not known yet: first pass with
// base == nullptr to add up the need, second pass with the block.
struct Carver {
    char* base = nullptr;
    uint64_t used = 0;
    template <typename T>
    T* take(uint64_t count) {
        const uint64_t at = used;
        used += (std::max<uint64_t>(count * sizeof(T), 16) + 255) / 256 * 256;
        return base != nullptr ? reinterpret_cast<T*>(base + at) : nullptr;
    }
};

// How batch_create_impl places a batch: by default on the model's stream with its own workspace and
// blocking uploads; a pipeline slot passes its stream, its workspace and its page-locked staging block,
// and every upload becomes an asynchronous copy on that stream.
// hipMemcpyAsync moves this many bytes or fewer with a copy KERNEL, more with the copy engine (the runtime's
// GPU_FORCE_BLIT_COPY_SIZE, 16 KB by default)
constexpr uint64_t kMinDmaBytes = 16 * 1024;

// the streamed form of coati_hip_viterbi_batch takes pairs of at most this many cells (the checkpoints of a pair
// that does not use a wavefront slot must fit a stream slot's workspace: 1.1 bytes per cell and the narrow last strip)
constexpr uint64_t kStreamPairCells = 64ull << 20;

struct BatchOpts {
    hipStream_t stream = nullptr;
    void* arena = nullptr;
    uint64_t arena_bytes = 0;
    char* staging = nullptr;  // page-locked; descriptors (and sequences that are not page-locked themselves) pass through it
    uint64_t staging_bytes = 0;
    bool seqs_pinned = false;  // a_cat / b_cat are page-locked: copied straight from the caller's memory
    uint64_t* arena_need_out = nullptr;  // receives the workspace size of the plan (also when `arena` is too small)
    bool ck_per_pair = false;  // viterbi_ck: keep every pair's checkpoints (coati_hip_debug_viterbi_flags reads them afterwards)
    uint32_t force_w_main = 0;  // (debug re-run of one pair: the strip shape it had in its batch)
    bool force_ck = false;      // viterbi_ck whatever the planner's rule says (the chunks of a streamed call)
    // chunks of a streamed call: the per-wavefront checkpoint slots (this many dwords each) and the traceback
    // scratch live outside the chunk's workspace, shared by all chunks (viterbi_batch_stream)
    uint64_t wave_slot_dwords = 0;
    uint32_t tail_parts = 0;  // one of the call's LAST chunks: every pair that can be is cut into this many row parts
    bool device_validates = false;  // the kernel checks the sequence codes it loads (viterbi_ck_stream): do not read them here
};
int batch_create_impl(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat, const uint64_t* a_off,
                      const uint8_t* b_cat, const uint64_t* b_off, const uint32_t* table_index, const BatchOpts* opts,
                      coati_hip_batch_t** out);


}  // namespace coati_hip_abi
#endif
