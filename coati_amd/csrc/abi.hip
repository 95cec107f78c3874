// abi.hip -- host side of the C ABI of include/coati_hip.h: handles, validation,
// HBM arenas, launches, result transfer.  The kernels live in viterbi_l1.hip,
// dp_generic.hip and sampleback.hip.
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sched.h>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace coati_hip_detail;

namespace {

// Debug: decode one pair's bit-planes into the oracle's byte-per-cell encoding.
__global__ void decode_flags(const PairDesc* __restrict__ pairs, uint32_t pair,
                             const uint32_t* __restrict__ flags, uint8_t* __restrict__ out) {
    const PairDesc pd = pairs[pair];
    const uint64_t n = static_cast<uint64_t>(pd.la) * pd.lb;
    for(uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; idx < n;
        idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t bi = idx / pd.lb, bj = idx % pd.lb;
        if(pd.v_compact != 0 && (bi % pd.v_compact) != (bj % pd.v_compact)) {
            out[idx] = 0xffu;  // not a live cell: never stored (viterbi_k.hip)
            continue;
        }
        const CellAddr ca = cell_addr(pd, bi, bj);
        const uint32_t mm = pair_bits(flags, ca, 0), dd = pair_bits(flags, ca, 1), im = im_bit(flags, ca);
        // (bit1: the M argument is not the maximum, bit0: the D argument is not) -> 0 M, 1 D, 2 I
        const uint32_t fm = !(mm & 2u) ? 0u : ((mm & 1u) ? 2u : 1u), fd = !(dd & 2u) ? 0u : ((dd & 1u) ? 2u : 1u);
        out[idx] = static_cast<uint8_t>(fm | (fd << 2) | ((im ^ 1u) << 4));
    }
}


// Debug: the libm restatements element-wise (glibc_math.hpp).
__global__ void libm_kernel(int op, const float* __restrict__ in, uint64_t n, float* __restrict__ out) {
    __shared__ uint64_t exp_tab[32];
    load_exp_table(exp_tab, threadIdx.x);
    __syncthreads();
    for(uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < n;
        i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float x = in[i];
        out[i] = op == 0   ? libm::expf_nonpos(x, exp_tab)
                 : op == 1 ? libm::log1pf_unit(x)
                 : op == 2 ? libm::logf_pos(x)
                           : libm::log1pf_mid(x);  // op 3: the straight-line log1pf of [2^-29, 1]
    }
}

// Debug: gather one pair's Forward M/D/I into three row-major la x lb matrices.
__global__ void decode_mdi(const PairDesc* __restrict__ pairs, uint32_t pair, const float* __restrict__ mdi,
                           float* __restrict__ out) {
    const PairDesc pd = pairs[pair];
    const uint64_t n = static_cast<uint64_t>(pd.la) * pd.lb;
    for(uint64_t idx = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; idx < n;
        idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint32_t bi = idx / pd.lb, bj = idx % pd.lb;
        for(int mat = 0; mat < 3; ++mat)
            out[mat * n + idx] = mdi_stored(pd, bi, bj) ? mdi[mdi_index(pd, bi, bj, mat)] : kLowest;  // (not live: lowest)
    }
}

}  // namespace

namespace coati_hip_detail {
bool forward_fast_math() {
    static const bool fast = [] {
        const char* e = std::getenv("COATI_HIP_FORWARD_FAST");
        return e != nullptr && e[0] != '\0' && e[0] != '0';
    }();
    return fast;
}
}  // namespace coati_hip_detail

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if(e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while(0)

bool device_is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

struct coati_hip_model {
    int device = 0;
    int gap_len = 1;
    GapConsts k{};
    uint32_t n_tables = 1;
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
    static constexpr uint64_t kMaxCachedBytes = 16ull << 30;  // larger blocks are freed, not cached
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
    hipEvent_t stream_events[kCkStreamSlots + 1] = {};  // [slot]: its download is done; [last]: an upload is done
    std::mutex pipeline_lock;  // one pipelined call at a time per model
    uint32_t stream_calls = 0;     // streamed calls this model has served (the first one allocates lazily: a one-shot process pays for what it uses)
    bool stream_unusable = false;  // the persistent kernel's first upload did not arrive in time once (copies not on the copy engine): never again on this model
    bool stream_forbidden = false;  // coati_hip_model_set_option(COATI_HIP_OPT_PERSISTENT_CALL, 0): the embedder shares the GPU
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
    bool ck_keep_all = false;    // viterbi_ck keeps every checkpoint (no band): the debug export decodes every tile
    uint32_t fwd_wlog2_max = 4;  // widest Forward strip shape of the batch (forward_l1 has a leaner build for <= 8 columns per lane)
    uint32_t* d_progress = nullptr;
    uint32_t n_items = 0;
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_ops = nullptr;
    uint32_t* d_flags = nullptr;   // decision bits (viterbi_l1/_k, dp_generic) or checkpoints (viterbi_ck)
    uint32_t* d_wscratch = nullptr;  // viterbi_ck: traceback scratch of the persistent wavefronts
    uint64_t ck_slot_dwords = 0;     // viterbi_ck: per-wavefront checkpoint slots at the start of d_flags (0: none)
    uint32_t ck_split_items = 0;     // viterbi_ck: pairs cut into row parts (the last ones of the LPT order); 0: none
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
    hipEvent_t ev[kTimingRing][3] = {};
    uint64_t n_launches = 0;
    bool launched = false;
};

namespace {
BatchDeviceView device_view(const coati_hip_batch* b) {
    const coati_hip_model* m = b->model;
    return BatchDeviceView{m->d_table,  m->k,      static_cast<uint32_t>(m->gap_len),
                           b->d_desc,   b->d_order, static_cast<uint32_t>(b->n_pairs),
                           b->d_queue,  b->d_items, b->n_items, b->d_fwd_items, b->n_fwd_items, b->d_progress, b->d_a,    b->d_b,
                           b->d_flags,  b->d_bnd,  b->bnd_floats * sizeof(float), b->d_scores,
                           b->d_ops,    b->d_ops_start, b->d_ops_len, b->d_wscratch, b->ck_slot_dwords, b->ck_split_items,
                           b->d_mdi,    b->d_final_mdi, b->fwd_wlog2_max, b->ck_keep_all ? 1u : 0u};
}
}  // namespace


namespace {
// An HBM block of at least `need` bytes: one the model cached (not more than ~2x too large) or a
// fresh hipMalloc; when that fails for lack of memory the cache is emptied and it is tried again.
hipError_t model_take_arena(coati_hip_model* m, uint64_t need, void** ptr, uint64_t* bytes) {
    *ptr = nullptr;
    {
        std::lock_guard<std::mutex> hold(m->arena_lock);
        size_t best = m->free_arenas.size();
        for(size_t i = 0; i < m->free_arenas.size(); ++i) {
            const uint64_t have = m->free_arenas[i].bytes;
            if(have >= need && have <= 2 * need + (64ull << 20) &&
               (best == m->free_arenas.size() || have < m->free_arenas[best].bytes))
                best = i;
        }
        if(best != m->free_arenas.size()) {
            *ptr = m->free_arenas[best].ptr;
            *bytes = m->free_arenas[best].bytes;
            m->free_arenas.erase(m->free_arenas.begin() + static_cast<std::ptrdiff_t>(best));
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(ptr, need);
    if(e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        std::vector<coati_hip_model::Arena> drop;
        {
            std::lock_guard<std::mutex> hold(m->arena_lock);
            drop.swap(m->free_arenas);
        }
        for(const auto& a : drop) (void)hipFree(a.ptr);
        e = hipMalloc(ptr, need);
    }
    if(e != hipSuccess) {
        *ptr = nullptr;
        return e;
    }
    *bytes = need;
    return hipSuccess;
}
// Back to the cache (the caller made sure nothing on the stream can still touch the block); the
// smallest cached block goes when there are too many.
void model_give_arena(coati_hip_model* m, void* ptr, uint64_t bytes) {
    if(ptr == nullptr) return;
    if(bytes > coati_hip_model::kMaxCachedBytes) {  // too large to sit on: other users of the GPU need the memory
        (void)hipFree(ptr);
        return;
    }
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> hold(m->arena_lock);
        m->free_arenas.push_back({ptr, bytes});
        if(m->free_arenas.size() > coati_hip_model::kCachedArenas) {
            size_t k = 0;
            for(size_t i = 1; i < m->free_arenas.size(); ++i)
                if(m->free_arenas[i].bytes < m->free_arenas[k].bytes) k = i;
            drop = m->free_arenas[k].ptr;
            m->free_arenas.erase(m->free_arenas.begin() + static_cast<std::ptrdiff_t>(k));
        }
    }
    if(drop != nullptr) (void)hipFree(drop);
}
// Carves 256-byte aligned parts out of a block whose size is not known yet: first pass with
// base == nullptr to add up the need, second pass with the block.
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
hipError_t model_pinned(coati_hip_model* m, uint64_t bytes, void** out) {
    if(m->pinned_bytes < bytes) {
        if(m->pinned != nullptr) (void)hipHostFree(m->pinned);
        m->pinned = nullptr;
        m->pinned_bytes = 0;
        const hipError_t e = hipHostMalloc(&m->pinned, bytes, hipHostMallocDefault);
        if(e != hipSuccess) return e;
        m->pinned_bytes = bytes;
    }
    *out = m->pinned;
    return hipSuccess;
}
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
}  // namespace

extern "C" {

uint32_t coati_hip_version(void) { return (0u << 16) | 1u; }

const char* coati_hip_last_error(void) { return g_error.c_str(); }

int coati_hip_device_count(void) {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for(int d = 0; d < n; ++d) ok += device_is_gfx950(d) ? 1 : 0;
    return ok;
}

int coati_hip_model_create(const float* table, float no_gap, float gap_stop, float gap_open,
                           float gap_extend, int gap_len, int device, coati_hip_model_t** out) {
    return coati_hip_model_create_tables(table, 1, no_gap, gap_stop, gap_open, gap_extend, gap_len, device, out);
}

int coati_hip_model_create_tables(const float* table, uint32_t n_tables, float no_gap, float gap_stop, float gap_open,
                                  float gap_extend, int gap_len, int device, coati_hip_model_t** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "model_create: out is NULL");
    *out = nullptr;
    if(table == nullptr) return fail(COATI_HIP_EINVAL, "model_create: table is NULL");
    if(n_tables < 1 || n_tables > 65535)
        return fail(COATI_HIP_EINVAL, "model_create: n_tables must be in [1, 65535] (got %u)", n_tables);
    if(gap_len < 1) return fail(COATI_HIP_EINVAL, "model_create: gap_len must be >= 1 (got %d)", gap_len);
    if(gap_len > 8)
        return fail(COATI_HIP_EINVAL, "model_create: gap_len %d not supported by the GPU path (1..8)", gap_len);
    // log-probabilities: NaN or +inf in the table is never meaningful and turns scores into NaN (-inf, the log
    // of a zero probability in a --sub matrix, is let through as in the reference; the samplers bound their walks)
    for(uint64_t i = 0; i < static_cast<uint64_t>(n_tables) * kTabFloats; ++i)
        if(std::isnan(table[i]) || (std::isinf(table[i]) && table[i] > 0))
            return fail(COATI_HIP_EINVAL, "model_create: table %llu has a non-finite entry at row %llu column %llu",
                        static_cast<unsigned long long>(i / kTabFloats), static_cast<unsigned long long>(i % kTabFloats / kTabCols),
                        static_cast<unsigned long long>(i % kTabCols));
    for(const float c : {no_gap, gap_stop, gap_open, gap_extend})
        if(std::isnan(c) || (std::isinf(c) && c > 0)) return fail(COATI_HIP_EINVAL, "model_create: a gap constant is NaN or +inf");
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(COATI_HIP_ENODEVICE, "model_create: no HIP device available");
    if(device < 0 || device >= n)
        return fail(COATI_HIP_EINVAL, "model_create: device %d out of range [0,%d)", device, n);
    if(!device_is_gfx950(device))
        return fail(COATI_HIP_ENODEVICE, "model_create: device %d is not gfx950 (MI355X)", device);
    auto* m = new(std::nothrow) coati_hip_model;
    if(m == nullptr) return fail(COATI_HIP_ENOMEM, "model_create: host allocation failed");
    m->device = device;
    m->gap_len = gap_len;
    m->n_tables = n_tables;
    m->k = GapConsts{no_gap, gap_stop, gap_open, gap_extend};
    auto cleanup = [&](int rc) {
        coati_hip_model_destroy(m);
        return rc;
    };
    hipError_t e;
    if((e = hipSetDevice(device)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipSetDevice: %s", hipGetErrorString(e)));
    if((e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)));
    const size_t bytes = sizeof(float) * kTabFloats * n_tables;
    if((e = hipMalloc(&m->d_table, bytes)) != hipSuccess)
        return cleanup(fail(COATI_HIP_ENOMEM, "hipMalloc(table): %s", hipGetErrorString(e)));
    if((e = hipMemcpy(m->d_table, table, bytes, hipMemcpyHostToDevice)) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipMemcpy(table): %s", hipGetErrorString(e)));
    *out = m;
    return COATI_HIP_OK;
}

namespace {
void model_release(coati_hip_model* m) {
    if(m->refs.fetch_sub(1) != 1) return;  // batches (or the handle) still hold it
    (void)hipSetDevice(m->device);
    for(const auto& a : m->free_arenas) (void)hipFree(a.ptr);
    for(auto& sl : m->slots) {
        if(sl.arena != nullptr) (void)hipFree(sl.arena);
        if(sl.pinned != nullptr) (void)hipHostFree(sl.pinned);
        if(sl.stream != nullptr && sl.stream != m->stream) (void)hipStreamDestroy(sl.stream);
    }
    if(m->pinned != nullptr) (void)hipHostFree(m->pinned);
    if(m->d_stream_ctl != nullptr) (void)hipFree(m->d_stream_ctl);
    if(m->h_stream != nullptr) (void)hipHostFree(m->h_stream);
    if(m->d_stream_waves != nullptr) (void)hipFree(m->d_stream_waves);
    for(void* t : m->stream_tail_arena)
        if(t != nullptr) (void)hipFree(t);
    for(hipEvent_t e : m->stream_events)
        if(e != nullptr) (void)hipEventDestroy(e);
    for(auto& ss : m->sslots) {
        if(ss.arena != nullptr) (void)hipFree(ss.arena);
        if(ss.pinned != nullptr) (void)hipHostFree(ss.pinned);
    }
    if(m->d_table != nullptr) (void)hipFree(m->d_table);
    if(m->stream != nullptr) (void)hipStreamDestroy(m->stream);
    delete m;
}
}  // namespace

void coati_hip_model_destroy(coati_hip_model_t* m) {
    if(m == nullptr) return;
    model_release(m);
}

void coati_hip_batch_destroy(coati_hip_batch_t* b) {
    if(b == nullptr) return;
    coati_hip_model* m = b->model;
    if(m != nullptr) (void)hipSetDevice(m->device);
    if((b->arena != nullptr && b->arena_owned) || b->mdi_block != nullptr) {
        // the blocks go back to the model once nothing on the stream can still touch them
        bool idle = m != nullptr && hipStreamSynchronize(m->stream) == hipSuccess;
        if(idle && b->stream != nullptr && b->stream != m->stream) idle = hipStreamSynchronize(b->stream) == hipSuccess;
        for(auto blk : {std::pair<void*, uint64_t>{b->arena_owned ? b->arena : nullptr, b->arena_bytes}, std::pair<void*, uint64_t>{b->mdi_block, b->mdi_block_bytes}}) {
            if(blk.first == nullptr) continue;
            if(idle)
                model_give_arena(m, blk.first, blk.second);
            else
                (void)hipFree(blk.first);
        }
    }
    for(auto& trio : b->ev)
        for(hipEvent_t e : trio)
            if(e != nullptr) (void)hipEventDestroy(e);
    delete b;
    if(m != nullptr) model_release(m);
}

int coati_hip_model_trim(coati_hip_model_t* m) {
    if(m == nullptr) return fail(COATI_HIP_EINVAL, "model_trim: model is NULL");
    HIP_TRY(hipSetDevice(m->device));
    std::vector<coati_hip_model::Arena> drop;
    {
        std::lock_guard<std::mutex> hold(m->arena_lock);
        drop.swap(m->free_arenas);
    }
    for(const auto& a : drop) (void)hipFree(a.ptr);
    {
        std::lock_guard<std::mutex> hold(m->pipeline_lock);
        for(auto& sl : m->slots) {
            if(sl.stream != nullptr) (void)hipStreamSynchronize(sl.stream);
            if(sl.arena != nullptr) (void)hipFree(sl.arena);
            sl.arena = nullptr;
            sl.arena_bytes = 0;
        }
        for(auto& ss : m->sslots) {  // (no streamed call is running: pipeline_lock)
            if(ss.arena != nullptr) (void)hipFree(ss.arena);
            ss.arena = nullptr;
            ss.arena_bytes = 0;
        }
        if(m->d_stream_waves != nullptr) (void)hipFree(m->d_stream_waves);
        m->d_stream_waves = nullptr;
        m->stream_waves_bytes = 0;
        for(void*& t : m->stream_tail_arena) {
            if(t != nullptr) (void)hipFree(t);
            t = nullptr;
        }
        m->stream_tail_bytes = 0;
    }
    return COATI_HIP_OK;
}

int coati_hip_model_set_option(coati_hip_model_t* model, int option, int64_t value) {
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "model_set_option: model is NULL");
    if(option == COATI_HIP_OPT_PERSISTENT_CALL) {
        std::lock_guard<std::mutex> one_call(model->pipeline_lock);
        model->stream_forbidden = value == 0;
        return COATI_HIP_OK;
    }
    return fail(COATI_HIP_EINVAL, "model_set_option: unknown option %d", option);
}

int coati_hip_batch_create(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                           coati_hip_batch_t** out) {
    return coati_hip_batch_create_tables(model, n_pairs, a_cat, a_off, b_cat, b_off, nullptr, out);
}

namespace {
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
}

int coati_hip_batch_create_tables(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                                  const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                                  const uint32_t* table_index, coati_hip_batch_t** out) {
    try {  // no C++ exception may cross the C ABI (host-side vectors can throw bad_alloc)
        return batch_create_impl(model, n_pairs, a_cat, a_off, b_cat, b_off, table_index, nullptr, out);
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "batch_create: host allocation failed");
    } catch(const std::exception& ex) {
        return fail(COATI_HIP_EHIP, "batch_create: %s", ex.what());
    }
}

namespace {
inline uint8_t max_byte(const uint8_t* p, uint64_t n) {
    uint8_t m = 0;
    for(uint64_t i = 0; i < n; ++i) m = p[i] > m ? p[i] : m;
    return m;
}

// Pair indices, most cells first: exactly (equal pairs in input order), or -- `quick`, the chunks of a streamed
// call, where planning is on the critical path -- by a counting sort on the cell count's exponent and top six
// mantissa bits (1.6 % classes, input order inside a class).  The order only decides which wavefront takes which
// pair when; an exact sort of the 2 000 pairs of a streamed chunk was a third of its planning time (45 ns per
// pair), and costs a resident 10 000-pair launch 0.5 % if replaced by the classes (4.98 vs 5.01 ms).
void lpt_order(const std::vector<PairDesc>& desc, std::vector<uint32_t>& order, bool quick) {
    const size_t n = desc.size();
    auto cells_of = [&](size_t p) { return static_cast<uint64_t>(desc[p].la) * desc[p].lb; };
    if(n < 256 || !quick) {
        for(size_t p = 0; p < n; ++p) order[p] = static_cast<uint32_t>(p);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cells_of(x) > cells_of(y); });
        return;
    }
    constexpr uint32_t kClasses = 65 * 64;
    auto class_of = [&](size_t p) -> uint32_t {  // larger pairs -> smaller class number
        const uint64_t c = cells_of(p);
        if(c == 0) return kClasses - 1;
        const uint32_t e = 63u - static_cast<uint32_t>(__builtin_clzll(c));                          // exponent 0..63
        const uint32_t m = e >= 6 ? static_cast<uint32_t>((c >> (e - 6)) & 63u) : static_cast<uint32_t>((c << (6 - e)) & 63u);  // top six bits below the leading one
        return kClasses - 2 - (e * 64 + m);
    };
    std::vector<uint32_t> start(kClasses + 1, 0), cls(n);
    for(size_t p = 0; p < n; ++p) {
        cls[p] = class_of(p);
        ++start[cls[p] + 1];
    }
    for(uint32_t q = 0; q < kClasses; ++q) start[q + 1] += start[q];
    for(size_t p = 0; p < n; ++p) order[start[cls[p]]++] = static_cast<uint32_t>(p);
}

int batch_create_impl(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat, const uint64_t* a_off,
                      const uint8_t* b_cat, const uint64_t* b_off, const uint32_t* table_index, const BatchOpts* opts,
                      coati_hip_batch_t** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: out is NULL");
    *out = nullptr;
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: model is NULL");
    if(a_off == nullptr || b_off == nullptr) return fail(COATI_HIP_EINVAL, "batch_create: offsets are NULL");
    if(n_pairs > 0xffffffffull) return fail(COATI_HIP_EINVAL, "batch_create: too many pairs");
    const uint64_t a_total = a_off[n_pairs] - a_off[0], b_total = b_off[n_pairs] - b_off[0];
    if((a_total > 0 && a_cat == nullptr) || (b_total > 0 && b_cat == nullptr))
        return fail(COATI_HIP_EINVAL, "batch_create: sequence data is NULL");

    auto* b = new(std::nothrow) coati_hip_batch;
    if(b == nullptr) return fail(COATI_HIP_ENOMEM, "batch_create: host allocation failed");
    b->model = model;
    model->refs.fetch_add(1);  // released by coati_hip_batch_destroy
    b->n_pairs = n_pairs;
    b->stream = opts != nullptr && opts->stream != nullptr ? opts->stream : model->stream;
    // a chunk of a streamed Viterbi call never runs Forward: no Forward work items, no Forward boundary arrays
    const bool viterbi_only = opts != nullptr && opts->wave_slot_dwords != 0;
    struct Owner {  // destroys the half-built batch on every exit but the successful one
        coati_hip_batch* b;
        ~Owner() {
            if(b != nullptr) coati_hip_batch_destroy(b);
        }
    } owner{b};
    auto cleanup = [&](int rc) { return rc; };
    // COATI_HIP_TIMING=1: host-side stage times of this call on stderr
    static const bool timing = std::getenv("COATI_HIP_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if(!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "batch_create: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    b->desc.resize(n_pairs);
    const uint64_t L = static_cast<uint64_t>(model->gap_len);
    static const bool force_generic = std::getenv("COATI_HIP_FORCE_GENERIC") != nullptr;
    // Forward strip shape (forward_l1): 16 columns per lane, narrowed to 8 and 4 while the batch has
    // fewer strips than 1.5 rounds of the kernel's wavefront slots (3 per SIMD) -- a wavefront per
    // 1 024 columns leaves a small batch on a handful of SIMDs (16 pairs of 1 kb: 17.8 ms at W = 16,
    // 6.3 ms at W = 4), and just over one round of full-width strips wastes most of a second one
    // (3 000 pairs: 41.7 ms at W = 16, 35.9 ms at W = 8).  A Forward cell is ~440 instructions, so the per-step overhead of a narrow strip is
    // small, unlike in viterbi_l1.  COATI_HIP_FWD_W=<4|8|16> overrides.
    constexpr uint64_t kFwdSlots = 3 * 1024 * 3 / 2;
    // The bit-exact build (glibc's expf / log1pf restated: ~440 instructions per cell) starts from 8 columns per lane:
    // that shape fits 128 VGPRs without spills and runs 4 wavefronts per SIMD (forward_l1<false, true>), measured
    // 6 144 pairs of 1 kb: 16 columns (168 VGPRs, 81 spilled values, 3 per SIMD) 68.7 ms, 8 columns in the same build 63.0 ms.
    uint32_t fwd_wlog2 = 4;  // (dp_generic and forward_k lay their cells out for 16 columns per lane)
    if(L == 1 && !force_generic) {
        if(!forward_fast_math()) fwd_wlog2 = 3;
        auto count_strips = [&](uint32_t w) {
            uint64_t n = 0;
            for(uint64_t p = 0; p < n_pairs && n < kFwdSlots; ++p) {
                const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
                n += (la > 0 && lb > 0 && lb <= 0x7fffff00ull) ? fwd_strips_w(static_cast<uint32_t>(lb), w) : 1;
            }
            return n;
        };
        while(fwd_wlog2 > 2 && count_strips(1u << fwd_wlog2) < kFwdSlots) --fwd_wlog2;
        // a handful of pairs (`coati sample` works on ONE): 2 and 1 columns per lane put 8 and 16 wavefronts on a 1 kb
        // pair.  Measured, 1 kb pairs, 4 / 2 / 1 columns: 1 or 16 pairs 6.2 / 4.5 / 3.75 ms, 64 pairs 6.45 / 4.65 / 3.95,
        // 256 pairs 7.1 / 6.0 / 7.1, 1 024 pairs 15 / 16 / 19.6 -- i.e. while the strips still fit ~2 per SIMD.
        if(fwd_wlog2 == 2 && count_strips(2) <= 2304) fwd_wlog2 = 1;
        if(fwd_wlog2 == 1 && count_strips(1) <= 1536) fwd_wlog2 = 0;
        if(const char* e = std::getenv("COATI_HIP_FWD_W")) {
            const int w = std::atoi(e);
            if(w == 1 || w == 2 || w == 4 || w == 8 || w == 16) fwd_wlog2 = w == 1 ? 0u : w == 2 ? 1u : w == 4 ? 2u : (w == 8 ? 3u : 4u);
        }
    }
    for(uint64_t p = 0; p < n_pairs; ++p) {
        if(a_off[p + 1] < a_off[p] || b_off[p + 1] < b_off[p])
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: offsets of pair %llu decrease",
                                static_cast<unsigned long long>(p)));
        const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
        if(la > 0x7fffff00ull || lb > 0x7fffff00ull)
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: pair %llu too long",
                                static_cast<unsigned long long>(p)));
        // process_marginal, src/lib/utils.cc:822-835
        if(la % 3 != 0 || la % L != 0)
            return cleanup(fail(COATI_HIP_EINVAL,
                                "Length of reference sequence must be multiple of 3 and gap unit "
                                "length. (pair %llu)",
                                static_cast<unsigned long long>(p)));
        if(lb % L != 0)
            return cleanup(fail(COATI_HIP_EINVAL,
                                "Length of descendant sequence must be multiple of gap unit length. "
                                "(pair %llu)",
                                static_cast<unsigned long long>(p)));
        // code ranges: a branch-free max over the bytes (vectorises); the offender is only looked up on failure.
        // (Reading every byte once from DRAM is most of the planning time of a 1 kb pair: the chunks of a streamed
        // call leave the check to the kernel, which reports through the same error.)
        if(opts != nullptr && opts->device_validates) {
        } else if(max_byte(a_cat + a_off[p], la) >= kTabRows) {
            uint64_t q = a_off[p];
            while(a_cat[q] < kTabRows) ++q;
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: ancestor code %u out of range (pair %llu)", a_cat[q],
                                static_cast<unsigned long long>(p)));
        }
        if(!(opts != nullptr && opts->device_validates) && max_byte(b_cat + b_off[p], lb) >= kTabCols) {
            uint64_t q = b_off[p];
            while(b_cat[q] < kTabCols) ++q;
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: descendant code %u out of range (pair %llu)", b_cat[q],
                                static_cast<unsigned long long>(p)));
        }
        if(table_index != nullptr && table_index[p] >= model->n_tables)
            return cleanup(fail(COATI_HIP_EINVAL, "batch_create: table index %u of pair %llu out of range [0,%u)",
                                table_index[p], static_cast<unsigned long long>(p), model->n_tables));
        PairDesc& d = b->desc[p];
        d.table = static_cast<uint16_t>(table_index != nullptr ? table_index[p] : 0u);
        d.a_off = a_off[p] - a_off[0];
        d.b_off = b_off[p] - b_off[0];
        d.la = static_cast<uint32_t>(la);
        d.lb = static_cast<uint32_t>(lb);
        d.ops_off = b->ops_total;
        // Forward M/D/I arena: gap_len 2, 3 store the live cells only (forward_k.hip)
        const bool fwd_k = (L == 2 || L == 3) && !force_generic;
        d.f_compact = static_cast<uint16_t>(fwd_k ? L : 0u);
        d.f_wlog2 = static_cast<uint8_t>(fwd_wlog2);
        d.v_parts = 0;
        b->ops_total += la + lb;
        b->cells += la * lb;
    }

    stage("plan: checks + descriptors");
    // ---- Viterbi strip plan (common.hpp).  Full-speed strips are 16 columns per lane; the last
    // strip of a pair takes the narrowest shape that holds the remainder.  When the whole batch
    // has fewer strips than the GPU has SIMDs (a few long pairs), narrower strips everywhere put
    // more wavefronts to work on each pair.  dp_generic (gap_len > 1) writes 16-column strips only.
    uint32_t w_main = kW;
    const bool plan_l1 = L == 1 && std::getenv("COATI_HIP_FORCE_GENERIC") == nullptr;  // viterbi_ck / viterbi_l1 will run
    // COATI_HIP_VITERBI_BITS=1: the round-1 kernel (five decision bits per cell written by the fill), kept
    // as the A/B partner and second implementation of viterbi_ck
    b->ck = plan_l1 && std::getenv("COATI_HIP_VITERBI_BITS") == nullptr;
    const bool ck_shared = model->n_tables == 1;
    // longest-processing-time-first order for the dynamic queue
    std::vector<uint32_t> order(n_pairs);
    lpt_order(b->desc, order, opts != nullptr && opts->device_validates);
    stage("plan: longest-first order");
    // Ragged end of the queue.  The persistent kernel runs kFillSlots wavefronts, three per SIMD, and
    // the SIMD's issue arbitration favours the oldest: in the trace build one 1 kb item takes a
    // wavefront between 1.15 and 3.4 ms (mean 2.1; `make trace`, tools/trace_fill.py).  The SIMD as a
    // whole is work-conserving, but when the queue runs empty every SIMD still holds up to three
    // items in different states of progress and drains them alone -- a batch of equal pairs ends
    // raggedly however many rounds it has.  The last third of a round (1 024 pairs) of such a
    // batch (the end of the LPT order) therefore gets 8-column-per-lane strips -- twice as many,
    // half as long items that the early finishers pick up.  Measured (tools/ab_fill.py, 1 kb pairs):
    // 9 216 pairs +11 %, 6 644 +6 %, 10 000 +3.7 %, 20 000 and 40 000 +2.3 %, 12 000 and 125 000 +-0.5 %; narrowing more
    // than ~1 100 pairs (768 and 1 024 are within 1 %, 1 280 loses 3 %), or to 4 columns, loses (W = 8 runs at ~85 %, W = 4 at ~57 % of the W = 16
    // rate per cell).  A mixed bag needs none of it: its short pairs already end the queue.
    // COATI_HIP_TAIL_PAIRS=<n> overrides the count (0: off).
    std::vector<uint8_t> pair_w(n_pairs, 0);
    if(plan_l1) {
        auto items_of = [&](uint64_t p, uint32_t w) {
            uint32_t ns = 1, wl = w;
            if(b->desc[p].la > 0 && b->desc[p].lb > 0) viterbi_strip_plan(b->desc[p].lb, w, ns, wl);
            return static_cast<uint64_t>(ns);
        };
        auto count_items = [&](uint32_t w) {
            uint64_t items = 0;
            for(uint64_t p = 0; p < n_pairs; ++p) items += items_of(p, w);
            return items;
        };
        const uint64_t kSimds = 4ull * device_cu_count();
        while(w_main > 4 && count_items(w_main) < kSimds) w_main /= 2;
        if(const char* e = std::getenv("COATI_HIP_STRIP_W")) {
            const int w = std::atoi(e);
            if(w == 4 || w == 8 || w == 16) w_main = static_cast<uint32_t>(w);
        }
        if(opts != nullptr && opts->force_w_main != 0) w_main = opts->force_w_main;
        // Which gap_len-1 kernel.  viterbi_ck (lean fill + checkpoint traceback) wins where the fill
        // dominates; viterbi_l1 (decision bits written by the fill) keeps two regimes, both measured
        // (profiles/r02/kernel_choice.txt): batches of SHORT pairs, where a traceback round recomputes a
        // large share of the little matrix (150 nt pairs: 588 vs 440 GCUPS; from 300 nt on the two are level,
        // at 750 nt viterbi_ck leads by 16 %), and a few LONG pairs cut into narrow strips, where every
        // wavefront is alone on its SIMD and the 4x larger checkpoint stream of 4-column strips costs more
        // than the shorter cell saves (160 kb pair: 86 vs 104 ms).  COATI_HIP_VITERBI_CK=1 / _BITS=1 force one.
        if(b->ck && std::getenv("COATI_HIP_VITERBI_CK") == nullptr && !(opts != nullptr && (opts->force_w_main != 0 || opts->force_ck))) {
            long double cells = 0;
            uint64_t live = 0;
            for(uint64_t p = 0; p < n_pairs; ++p)
                if(b->desc[p].la > 0 && b->desc[p].lb > 0) {
                    cells += static_cast<long double>(b->desc[p].la) * b->desc[p].lb;
                    ++live;
                }
            constexpr long double kShortPair = 250.0L * 250.0L;
            if(w_main < kW || (live > 0 && cells / live < kShortPair)) b->ck = false;
        }
        const uint64_t kFillSlots = (b->ck && ck_shared ? 4 : 3) * kSimds;  // resident wavefronts of the persistent kernel
        uint64_t tail_pairs = 0;
        if(const char* tp = std::getenv("COATI_HIP_TAIL_PAIRS")) {
            tail_pairs = std::min<uint64_t>(n_pairs, std::strtoull(tp, nullptr, 10));
        } else if(w_main == kW && n_pairs > 0) {
            // "equal pairs": the smallest has at least half the cells of the largest (LPT order);
            // and the batch must be clearly longer than one round (3 500 pairs: -1 %, 6 644: +6 %)
            auto cells_of = [&](uint64_t p) { return static_cast<uint64_t>(b->desc[p].la) * b->desc[p].lb; };
            const bool homogeneous = cells_of(order[n_pairs - 1]) > 0 && cells_of(order[n_pairs - 1]) * 2 >= cells_of(order[0]);
            // (viterbi_ck: measured again with the lean fill, 10 000 pairs: 0 and 700 narrowed pairs within
            // noise of each other, 1 365 -7 %, 2 730 -10 % -- the narrow strips cost more than they balance)
            if(homogeneous && n_pairs > kFillSlots * 3 / 2 && !b->ck) tail_pairs = kFillSlots / 3;
        }
        for(uint64_t q = n_pairs - tail_pairs; q < n_pairs; ++q) pair_w[order[q]] = 8;
    }
    // the same remedy for forward_l1 (3 slots per SIMD as well): a batch of equal pairs that runs
    // full-width strips ends with its last quarter round in 8-column strips
    if(L == 1 && !force_generic && fwd_wlog2 == 4 && n_pairs > 3 * 1024 * 3 / 2 && std::getenv("COATI_HIP_FWD_W") == nullptr) {
        auto cells_of = [&](uint64_t p) { return static_cast<uint64_t>(b->desc[p].la) * b->desc[p].lb; };
        if(cells_of(order[n_pairs - 1]) > 0 && cells_of(order[n_pairs - 1]) * 2 >= cells_of(order[0]))
            for(uint64_t q = n_pairs - 3 * 1024 / 4; q < n_pairs; ++q) b->desc[order[q]].f_wlog2 = 3;
    }
    stage("plan: strip shapes");
    b->fwd_wlog2_max = 0;
    for(uint64_t p = 0; p < n_pairs; ++p) b->fwd_wlog2_max = std::max<uint32_t>(b->fwd_wlog2_max, b->desc[p].f_wlog2);
    // Forward M/D/I arena, now that every pair's strip shape is known
    for(uint64_t p = 0; p < n_pairs; ++p) {
        PairDesc& d = b->desc[p];
        d.mdi_off = b->mdi_floats;
        if(d.la > 0 && d.lb > 0)
            b->mdi_floats += d.f_compact != 0
                                 ? fwd_compact_strips(d.lb, d.f_compact) * fwd_compact_strip_floats(d.la, d.f_compact)
                                 : fwd_strips_w(d.lb, 1u << d.f_wlog2) * strip_mdi_floats_w(d.la, 1u << d.f_wlog2);
    }
    // gap_len 2 and 3: viterbi_k works on the live cells only, in block columns (lb / L), strips of
    // 16 block columns per lane and a narrow shape (6 for L = 3, 8 for L = 2) for the last strip
    const bool plan_k = (L == 2 || L == 3) && std::getenv("COATI_HIP_FORCE_GENERIC") == nullptr;
    b->compact = plan_k;
    uint32_t k_main = L == 3 ? 12u : 16u;  // (viterbi_k.hip: kWMain / kWNarrow)
    const uint32_t k_narrow = L == 3 ? 6u : 8u;
    if(plan_k) {  // few long pairs: the narrow shape everywhere puts more wavefronts on each pair
        uint64_t items_main = 0;
        for(uint64_t p = 0; p < n_pairs; ++p)
            items_main += (b->desc[p].la > 0 && b->desc[p].lb > 0) ? (b->desc[p].lb / L + kWave * k_main - 1) / (kWave * k_main) : 1;
        if(items_main < 1024) k_main = k_narrow;
    }
    bool all_narrow = plan_k;
    for(uint64_t p = 0; p < n_pairs; ++p) {
        PairDesc& d = b->desc[p];
        const uint64_t la = d.la;
        const uint32_t w_main_p = pair_w[p] != 0 ? std::min<uint32_t>(pair_w[p], w_main) : w_main;
        uint32_t w_main_q = w_main_p;  // columns per lane of every strip but the last
        uint32_t ns = 1, wl = w_main_p;
        d.v_compact = 0;
        if(plan_l1) {
            if(d.la > 0 && d.lb > 0) viterbi_strip_plan(d.lb, w_main_p, ns, wl);
        } else if(plan_k) {
            d.v_compact = static_cast<uint32_t>(L);
            const uint32_t cols_b = static_cast<uint32_t>(d.lb / L), narrow = k_narrow;
            w_main_q = k_main;
            wl = w_main_q;
            if(d.la > 0 && d.lb > 0) {
                const uint32_t full = kWave * w_main_q, whole = cols_b / full, rem = cols_b % full;
                ns = whole + (rem != 0 ? 1u : 0u);
                if(rem != 0 && rem <= kWave * narrow) wl = narrow;
            }
        } else {
            ns = std::max(1u, n_strips(d.lb));
        }
        d.v_strips = ns;
        d.v_wmain = static_cast<uint8_t>(w_main_q);
        d.v_wlast = static_cast<uint8_t>(wl);
        if(plan_k && d.la > 0 && d.lb > 0 && (wl != k_narrow || (ns > 1 && w_main_q != k_narrow))) all_narrow = false;
        d.flags_off = b->flag_dwords;
        d.bnd_off = b->bnd_floats;
        if(d.la > 0 && d.lb > 0)
            b->flag_dwords += plan_k  ? ns * compact_strip_dwords(d.la, static_cast<uint32_t>(L))
                              : b->ck ? (ns - 1) * ck_strip_dwords(d.la, w_main_p) + ck_strip_dwords(d.la, wl)
                                      : (ns - 1) * strip_dwords(d.la, w_main_p) + strip_dwords(d.la, wl);
        // strip-boundary arrays, 128-byte aligned so that no two waves ever share a cache line:
        // viterbi_l1 one 2(la+1) array per boundary of its plan, forward_l1 one 3(la+1) array per
        // boundary of 1024-column strips, dp_generic one (la+1)(3+2L) array
        const uint64_t nf = viterbi_only ? 1 : fwd_strips_w(d.lb, 1u << d.f_wlog2);
        const uint64_t need = std::max<uint64_t>({(ns - 1) * 2 * (la + 1), nf > 1 ? (nf - 1) * 3 * (la + 1) : 0,
                                                  nf > 1 ? (la + 1) * (3 + 2 * L) : 0,
                                                  plan_k ? (ns - 1) * ((la / L + 1) + la) : 0,
                                                  d.f_compact != 0 && d.lb > 0
                                                      ? (fwd_compact_strips(d.lb, static_cast<uint32_t>(L)) - 1) * 3 * L * (la / L + 1)
                                                      : 0});
        b->bnd_floats += (need + 31) / 32 * 32;
    }
    b->compact_narrow_only = plan_k && all_narrow;
    // viterbi_ck: checkpoints of single-strip pairs in per-wavefront slots instead of per pair, when that is
    // the smaller arena (a 1 kb pair needs 1.09 MB: 10 000 pairs 10.9 GB per pair, 4.5 GB in 4 096 slots; a
    // batch of a few pairs keeps per-pair storage).  Pairs above kSlotCap keep their own storage either way.
    if(b->ck && opts != nullptr && opts->wave_slot_dwords != 0) {
        // streamed chunk: every single-strip pair that fits the call's shared slots uses them; the workspace keeps the rest.
        // One of the call's last chunks: its pairs are cut into row parts (the ragged end, below) and keep their checkpoints.
        if(opts->tail_parts >= 2) {
            std::vector<uint32_t> whole, cut;
            for(const uint32_t p : order) {
                PairDesc& d = b->desc[p];
                if(d.la > 0 && d.lb > 0 && d.v_strips == 1 && d.v_wlast == kW && ck_strip_dwords(d.la, kW) <= (1ull << 20) &&
                   d.la + kWave >= 128 * opts->tail_parts) {
                    d.v_parts = static_cast<uint8_t>(opts->tail_parts);
                    cut.push_back(p);
                } else {
                    whole.push_back(p);
                }
            }
            if(!cut.empty()) {
                whole.insert(whole.end(), cut.begin(), cut.end());
                order.swap(whole);
                b->ck_split_items = static_cast<uint32_t>(cut.size());
            }
        }
        uint64_t at = 0;
        for(uint64_t p = 0; p < n_pairs; ++p) {
            PairDesc& d = b->desc[p];
            if(!(d.la > 0 && d.lb > 0)) {
                d.flags_off = at;
            } else if(d.v_parts >= 2) {
                d.flags_off = at;
                at += ck_strip_dwords(d.la, d.v_wlast) + kCkPartStateDwords;
            } else if(d.v_strips == 1 && ck_strip_dwords(d.la, d.v_wlast) <= opts->wave_slot_dwords) {
                d.flags_off = kCkWaveSlot;
            } else {
                d.flags_off = at;
                at += (d.v_strips - 1) * ck_strip_dwords(d.la, d.v_wmain) + ck_strip_dwords(d.la, d.v_wlast);
            }
        }
        b->flag_dwords = at;
        b->ck_slot_dwords = opts->wave_slot_dwords;
    } else if(b->ck && opts != nullptr && opts->ck_per_pair) {
        b->ck_keep_all = true;  // (the debug export: per-pair storage, every tile kept)
    } else if(b->ck && std::getenv("COATI_HIP_CK_PER_PAIR") == nullptr) {
        constexpr uint64_t kSlotCap = 1ull << 20;  // dwords (4 MB)
        uint64_t slot = 0, per_pair_total = 0;
        auto need_of = [&](const PairDesc& d) { return d.la > 0 && d.lb > 0 ? ck_strip_dwords(d.la, d.v_wlast) : 0; };
        for(uint64_t p = 0; p < n_pairs; ++p) {
            const PairDesc& d = b->desc[p];
            if(d.v_strips != 1) continue;
            const uint64_t nd = need_of(d);
            if(nd == 0 || nd > kSlotCap) continue;
            slot = std::max(slot, nd);
            per_pair_total += nd;
        }
        const uint64_t slots_total = slot * ck_scratch_waves();
        const bool use_slots = slot > 0 && slots_total < per_pair_total;
        // The ragged end: when the ticket queue runs dry every wavefront holds an item, and the launch lasts as long as
        // the SIMD with the most left (10 000 pairs of 1 kb: 0.5 ms of 5.5).  The last pairs of the LPT order are
        // therefore cut into ROW parts, each its own item at the end of the queue: a part leaves the lane state at a
        // 64-step boundary and whichever wavefront takes the next part continues there (viterbi_ck.hip).  Measured
        // (tools/split_ab.py, 10 000 pairs): 2 048 pairs in 3 parts 5.50 -> 4.95 ms; 2 parts 5.05; 4 parts 5.05;
        // 8 parts or 4 096 pairs lose again (hand-overs, waits for the predecessor).  6 000 pairs: 1 024 x 3 +4 %;
        // 40 000: +0.8 %.  Their checkpoints must outlive the wavefront that wrote them: own storage.
        // COATI_HIP_CK_SPLIT="pairs,parts" forces a plan (0 = off).
        uint64_t split_pairs = 0, parts = 3;
        // (round 3, with banded checkpoints -- a hand-over now writes back a fifth of the bytes -- and the band kept by
        // cut pairs too: 10 000 pairs 2 048 / 4 096 / 5 904 / 8 000 / all pairs cut in 3: 2 320 / 2 423 / 2 495 / 2 488 /
        // 2 404 GCUPS, in 2: 2 435 (5 904), in 4: 2 310 (all); 40 000 pairs 2 048 / 8 192 / 16 384 / all: 2 738 / 2 775 /
        // 2 719 / 2 542; 6 000 pairs 1 904 / all: 2 272 / 2 158.  So: every pair beyond the first round of wavefronts, up to 8 192.)
        if(use_slots && n_pairs > ck_scratch_waves()) {
            split_pairs = std::min<uint64_t>(2 * ck_scratch_waves(), n_pairs - ck_scratch_waves());
            if(split_pairs < 256) split_pairs = 0;
        }
        if(const char* e = std::getenv("COATI_HIP_CK_SPLIT")) {
            char* rest = nullptr;
            split_pairs = std::strtoull(e, &rest, 10);
            if(rest != nullptr && *rest == ',') parts = std::strtoull(rest + 1, nullptr, 10);
            if(parts < 2 || parts > 8) split_pairs = 0;
        }
        split_pairs = std::min<uint64_t>(split_pairs, n_pairs);
        std::vector<uint32_t> cut;  // in LPT order
        for(uint64_t q = n_pairs - split_pairs; q < n_pairs; ++q) {
            PairDesc& d = b->desc[order[q]];
            // (a part is at least two 64-step chunks; narrow last strips and multi-strip pairs stay whole)
            if(d.la > 0 && d.lb > 0 && d.v_strips == 1 && d.v_wlast == kW && need_of(d) <= kSlotCap && d.la + kWave >= 128 * parts) {
                d.v_parts = static_cast<uint8_t>(parts);
                cut.push_back(order[q]);
            }
        }
        if(!cut.empty()) {  // the cut pairs go to the end of the order, still longest first
            std::vector<uint32_t> whole;
            for(const uint32_t p : order)
                if(b->desc[p].v_parts < 2) whole.push_back(p);
            whole.insert(whole.end(), cut.begin(), cut.end());
            order.swap(whole);
            b->ck_split_items = static_cast<uint32_t>(cut.size());
        }
        if(use_slots || !cut.empty()) {
            // re-lay the arena: [wave slots | pairs that keep their own storage]
            uint64_t at = use_slots ? slots_total : 0;
            for(uint64_t p = 0; p < n_pairs; ++p) {
                PairDesc& d = b->desc[p];
                if(!(d.la > 0 && d.lb > 0)) {
                    d.flags_off = at;
                    continue;
                }
                if(d.v_parts >= 2) {
                    d.flags_off = at;
                    at += ck_strip_dwords(d.la, d.v_wlast) + kCkPartStateDwords;
                } else if(use_slots && d.v_strips == 1 && need_of(d) <= kSlotCap) {
                    d.flags_off = kCkWaveSlot;
                } else {
                    d.flags_off = at;
                    at += (d.v_strips - 1) * ck_strip_dwords(d.la, d.v_wmain) + ck_strip_dwords(d.la, d.v_wlast);
                }
            }
            b->flag_dwords = at;
            b->ck_slot_dwords = use_slots ? slot : 0;
        }
    }

    stage("plan: layout");
    if(hipSetDevice(model->device) != hipSuccess)
        return cleanup(fail(COATI_HIP_EHIP, "hipSetDevice failed"));
#define B_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if(e_ != hipSuccess)                                                                    \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,  \
                                "%s failed: %s", #expr, hipGetErrorString(e_)));                \
    } while(0)
    // work lists: one item per strip, pairs in LPT order
    // (a pair cut into row parts -- they are the last ones of `order` -- contributes its part 0 here; parts 1.. of
    // all of them follow in the same order, so that a part's predecessor is ck_split_items tickets before it)
    std::vector<WorkItem> items, fwd_items;
    for(const uint32_t p : order) {
        for(uint32_t st = 0; st < b->desc[p].v_strips; ++st) items.push_back(WorkItem{p, st});
        if(viterbi_only) continue;
        uint32_t nf = 1;
        if(b->desc[p].la > 0 && b->desc[p].lb > 0)
            nf = b->desc[p].f_compact != 0 ? fwd_compact_strips(b->desc[p].lb, b->desc[p].f_compact)
                                             : fwd_strips_w(b->desc[p].lb, 1u << b->desc[p].f_wlog2);
        for(uint32_t st = 0; st < nf; ++st) fwd_items.push_back(WorkItem{p, st});
    }
    if(b->ck_split_items > 0) {
        const uint32_t parts = b->desc[order[n_pairs - 1]].v_parts;
        for(uint32_t part = 1; part < parts; ++part)
            for(uint64_t q = n_pairs - b->ck_split_items; q < n_pairs; ++q) items.push_back(WorkItem{order[q], part << 16});
    }
    b->n_items = static_cast<uint32_t>(items.size());
    b->n_fwd_items = static_cast<uint32_t>(fwd_items.size());
    stage("work lists");
    // ONE workspace for everything but the Forward M/D/I arena, carved into 256-byte aligned parts
    uint64_t arena_need = 0;
    auto carve = [&](uint64_t bytes) {
        const uint64_t at = arena_need;
        arena_need += (std::max<uint64_t>(bytes, 16) + 255) / 256 * 256;
        return at;
    };
    // [what goes up: descriptors, order, queue word, work items, progress words, sequences | what comes back: scores,
    // ops offsets and lengths, ops | scratch]: each group contiguous, so that a pipeline slot moves it with ONE copy
    const uint64_t o_desc = carve(n_pairs * sizeof(PairDesc)), o_order = carve(n_pairs * sizeof(uint32_t)), o_queue = carve(sizeof(uint32_t)),
                   o_items = carve(items.size() * sizeof(WorkItem)), o_fwd = carve(fwd_items.size() * sizeof(WorkItem)),
                   o_progress = carve(std::max<size_t>(std::max(items.size(), fwd_items.size()), 4) * sizeof(uint32_t)),
                   o_a = carve(a_total), o_b = carve(b_total), o_up_end = arena_need,
                   o_scores = carve(n_pairs * sizeof(float)), o_start = carve(n_pairs * sizeof(uint64_t)),
                   o_len = carve(n_pairs * sizeof(uint32_t)), o_ops = carve(b->ops_total),
                   o_flags = carve(b->flag_dwords * sizeof(uint32_t)), o_bnd = carve(b->bnd_floats * sizeof(float)),
                   o_wscratch = carve(b->ck && !(opts != nullptr && opts->wave_slot_dwords != 0)
                                          ? ck_scratch_waves() * ck_scratch_dwords_per_wave() * sizeof(uint32_t) : 0);
    arena_need = std::max<uint64_t>(arena_need, 2 * kMinDmaBytes);
    if(opts != nullptr && opts->arena_need_out != nullptr) *opts->arena_need_out = arena_need;
    if(opts != nullptr && opts->arena != nullptr) {
        if(opts->arena_bytes < arena_need)
            return cleanup(fail(COATI_HIP_ENOMEM, "batch_create: the slot's workspace (%llu bytes) is smaller than the chunk needs (%llu)",
                                static_cast<unsigned long long>(opts->arena_bytes), static_cast<unsigned long long>(arena_need)));
        b->arena = opts->arena;
        b->arena_bytes = opts->arena_bytes;
        b->arena_owned = false;
    } else {
        // a workspace a destroyed batch of this model left behind, or a fresh one
        const hipError_t e = model_take_arena(model, arena_need, &b->arena, &b->arena_bytes);
        if(e != hipSuccess)
            return cleanup(fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,
                                "hipMalloc(workspace, %llu bytes) failed: %s", static_cast<unsigned long long>(arena_need),
                                hipGetErrorString(e)));
    }
    b->device_bytes += arena_need;
    auto at = [&](uint64_t off) { return static_cast<char*>(b->arena) + off; };
    b->d_desc = reinterpret_cast<PairDesc*>(at(o_desc));
    b->d_a = reinterpret_cast<uint8_t*>(at(o_a));
    b->d_b = reinterpret_cast<uint8_t*>(at(o_b));
    b->d_ops = reinterpret_cast<uint8_t*>(at(o_ops));
    b->d_flags = reinterpret_cast<uint32_t*>(at(o_flags));
    b->d_bnd = reinterpret_cast<float*>(at(o_bnd));
    b->d_scores = reinterpret_cast<float*>(at(o_scores));
    b->d_ops_start = reinterpret_cast<uint64_t*>(at(o_start));
    b->d_ops_len = reinterpret_cast<uint32_t*>(at(o_len));
    b->d_order = reinterpret_cast<uint32_t*>(at(o_order));
    b->d_queue = reinterpret_cast<uint32_t*>(at(o_queue));
    b->d_items = reinterpret_cast<WorkItem*>(at(o_items));
    b->d_fwd_items = reinterpret_cast<WorkItem*>(at(o_fwd));
    b->d_progress = reinterpret_cast<uint32_t*>(at(o_progress));
    b->d_wscratch = reinterpret_cast<uint32_t*>(at(o_wscratch));
    stage("workspace");
    // uploads: blocking copies by default; for a pipeline slot asynchronous copies on its stream, out of
    // page-locked memory (the slot's staging block, or the caller's arrays when those are page-locked)
    const bool seqs_pinned = opts != nullptr && opts->seqs_pinned;
    if(opts == nullptr || opts->staging == nullptr) {
        if(n_pairs > 0) {
            B_TRY(hipMemcpy(b->d_desc, b->desc.data(), n_pairs * sizeof(PairDesc), hipMemcpyHostToDevice));
            B_TRY(hipMemcpy(b->d_order, order.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice));
            if(!items.empty()) B_TRY(hipMemcpy(b->d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
            if(!fwd_items.empty()) B_TRY(hipMemcpy(b->d_fwd_items, fwd_items.data(), fwd_items.size() * sizeof(WorkItem), hipMemcpyHostToDevice));
        }
        stage("descriptors + work items upload");
        if(a_total > 0) B_TRY(hipMemcpy(b->d_a, a_cat + a_off[0], a_total, hipMemcpyHostToDevice));
        if(b_total > 0) B_TRY(hipMemcpy(b->d_b, b_cat + b_off[0], b_total, hipMemcpyHostToDevice));
    } else {
        // a pipeline slot: the group is laid out in the slot's page-locked block exactly as in the workspace and goes
        // up as one asynchronous copy (queue and progress words as zeros); page-locked caller sequences go directly.
        // Copies of kMinDmaBytes or less are done by a kernel, not by the copy engine -- which must not happen while
        // viterbi_ck_stream owns the chip -- so short groups are padded (what follows in the workspace is scratch).
        const bool stage_a = !seqs_pinned || a_total <= kMinDmaBytes, stage_b = !seqs_pinned || b_total <= kMinDmaBytes;
        const uint64_t group = stage_a ? (stage_b ? o_up_end : o_b) : o_a;
        const uint64_t sent = std::max<uint64_t>(group, kMinDmaBytes + 256);
        const uint64_t b_alone = stage_b && !stage_a ? std::min<uint64_t>(std::max<uint64_t>(b_total, kMinDmaBytes + 256), arena_need - o_b) : 0;
        if(std::max(sent, o_a + b_alone) > opts->staging_bytes) B_TRY(hipErrorOutOfMemory);
        char* st = opts->staging;
        std::memset(st + o_queue, 0, o_items - o_queue);
        std::memset(st + o_progress, 0, o_a - o_progress);
        if(n_pairs > 0) {
            std::memcpy(st + o_desc, b->desc.data(), n_pairs * sizeof(PairDesc));
            std::memcpy(st + o_order, order.data(), n_pairs * sizeof(uint32_t));
            if(!items.empty()) std::memcpy(st + o_items, items.data(), items.size() * sizeof(WorkItem));
            if(!fwd_items.empty()) std::memcpy(st + o_fwd, fwd_items.data(), fwd_items.size() * sizeof(WorkItem));
        }
        if(stage_a && a_total > 0) std::memcpy(st + o_a, a_cat + a_off[0], a_total);
        if(stage_a && stage_b && b_total > 0) std::memcpy(st + o_b, b_cat + b_off[0], b_total);
        B_TRY(hipMemcpyAsync(at(0), st, sent, hipMemcpyHostToDevice, b->stream));
        stage("descriptors + work items upload");
        if(!stage_a) B_TRY(hipMemcpyAsync(b->d_a, a_cat + a_off[0], a_total, hipMemcpyHostToDevice, b->stream));
        if(!stage_b) {
            B_TRY(hipMemcpyAsync(b->d_b, b_cat + b_off[0], b_total, hipMemcpyHostToDevice, b->stream));
        } else if(!stage_a && b_total > 0) {  // (short b beside long page-locked a: staged behind the group, after it has gone)
            std::memcpy(st + o_a, b_cat + b_off[0], b_total);
            B_TRY(hipMemcpyAsync(b->d_b, st + o_a, b_alone, hipMemcpyHostToDevice, b->stream));
        }
    }
    stage("sequences upload");
#undef B_TRY
    owner.b = nullptr;
    *out = b;
    return COATI_HIP_OK;
}
}  // namespace

uint64_t coati_hip_batch_pairs(const coati_hip_batch_t* b) { return b ? b->n_pairs : 0; }
uint64_t coati_hip_batch_device_bytes(const coati_hip_batch_t* b) { return b ? b->device_bytes : 0; }
uint64_t coati_hip_batch_cells(const coati_hip_batch_t* b) { return b ? b->cells : 0; }

int coati_hip_viterbi_launch(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_launch: batch is NULL");
    coati_hip_model* m = b->model;
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t n = static_cast<uint32_t>(b->n_pairs);
    hipEvent_t* ev = b->ev[b->n_launches % coati_hip_batch::kTimingRing];
    for(int q = 0; q < 3; ++q)
        if(ev[q] == nullptr) HIP_TRY(hipEventCreate(&ev[q]));  // (created on first use: 192 events per batch cost 0.3 ms)
    HIP_TRY(hipEventRecord(ev[0], b->stream));
    if(n > 0) {
        const BatchDeviceView v = device_view(b);
        static const bool force_generic = std::getenv("COATI_HIP_FORCE_GENERIC") != nullptr;
        if(b->ck)
            HIP_TRY(launch_viterbi_ck(v, m->n_tables == 1, b->stream));
        else if(m->gap_len == 1 && !force_generic)
            HIP_TRY(launch_viterbi_l1(v, b->stream));
        else if(b->compact)
            HIP_TRY(launch_viterbi_k(v, b->compact_narrow_only, b->stream));
        else
            HIP_TRY(launch_dp_generic(v, /*forward=*/false, b->stream));
    }
    HIP_TRY(hipEventRecord(ev[1], b->stream));
    HIP_TRY(hipEventRecord(ev[2], b->stream));  // (the traceback is fused into the fill kernel)
    b->n_launches += 1;
    b->launched = true;
    return COATI_HIP_OK;
}

int coati_hip_batch_sync(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "batch_sync: batch is NULL");
    HIP_TRY(hipSetDevice(b->model->device));
    HIP_TRY(hipStreamSynchronize(b->model->stream));
    if(b->stream != b->model->stream) HIP_TRY(hipStreamSynchronize(b->stream));
    return COATI_HIP_OK;
}

int coati_hip_viterbi_wait(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_wait: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_wait: nothing was launched");
    HIP_TRY(hipSetDevice(b->model->device));
    HIP_TRY(hipEventSynchronize(b->ev[(b->n_launches - 1) % coati_hip_batch::kTimingRing][2]));
    return COATI_HIP_OK;
}

int coati_hip_viterbi_fetch(coati_hip_batch_t* b, float* scores, uint8_t* ops, uint64_t ops_capacity,
                            uint64_t* ops_off, uint32_t* ops_len) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_fetch: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_fetch: nothing was launched");
    if(ops != nullptr && ops_capacity < b->ops_total)
        return fail(COATI_HIP_EINVAL, "viterbi_fetch: ops_capacity %llu < %llu",
                    static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(b->ops_total));
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    const uint64_t n = b->n_pairs;
    if(n == 0) return COATI_HIP_OK;
    if(scores != nullptr) HIP_TRY(hipMemcpy(scores, b->d_scores, n * sizeof(float), hipMemcpyDeviceToHost));
    if(ops != nullptr && b->ops_total > 0) HIP_TRY(hipMemcpy(ops, b->d_ops, b->ops_total, hipMemcpyDeviceToHost));
    if(ops_off != nullptr)
        HIP_TRY(hipMemcpy(ops_off, b->d_ops_start, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if(ops_len != nullptr)
        HIP_TRY(hipMemcpy(ops_len, b->d_ops_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return COATI_HIP_OK;
}

int coati_hip_batch_result_ptrs(coati_hip_batch_t* b, void** scores, void** ops, uint64_t* ops_bytes,
                                void** ops_off, void** ops_len) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "batch_result_ptrs: batch is NULL");
    if(scores != nullptr) *scores = b->d_scores;
    if(ops != nullptr) *ops = b->d_ops;
    if(ops_bytes != nullptr) *ops_bytes = b->ops_total;
    if(ops_off != nullptr) *ops_off = b->d_ops_start;
    if(ops_len != nullptr) *ops_len = b->d_ops_len;
    return COATI_HIP_OK;
}

int coati_hip_viterbi_timing(coati_hip_batch_t* b, uint32_t launches_back, float* fill_ms, float* walk_ms) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_timing: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_timing: nothing was launched");
    if(launches_back >= coati_hip_batch::kTimingRing || launches_back >= b->n_launches)
        return fail(COATI_HIP_EINVAL, "viterbi_timing: launch %u back is not recorded", launches_back);
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    hipEvent_t* ev = b->ev[(b->n_launches - 1 - launches_back) % coati_hip_batch::kTimingRing];
    float f = 0.f, w = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&w, ev[1], ev[2]));
    if(fill_ms != nullptr) *fill_ms = f;
    if(walk_ms != nullptr) *walk_ms = w;
    return COATI_HIP_OK;
}

int coati_hip_viterbi_last_timing(coati_hip_batch_t* b, float* fill_ms, float* walk_ms) {
    return coati_hip_viterbi_timing(b, 0, fill_ms, walk_ms);
}

int coati_hip_forward_launch(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "forward_launch: batch is NULL");
    coati_hip_model* m = b->model;
    HIP_TRY(hipSetDevice(m->device));
    if(b->d_mdi == nullptr) {  // the 12 B/cell arena is only reserved when Forward is actually used
        Carver sizing;
        auto carve = [&](Carver& cv) {
            b->d_final_mdi = cv.take<float>(b->n_pairs * 3);
            b->d_mdi = cv.take<float>(b->mdi_floats);
        };
        carve(sizing);
        const hipError_t e = model_take_arena(m, sizing.used, &b->mdi_block, &b->mdi_block_bytes);
        if(e != hipSuccess) {
            b->d_mdi = b->d_final_mdi = nullptr;
            return fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP,
                        "forward_launch: %llu bytes for the Forward matrices: %s", static_cast<unsigned long long>(sizing.used),
                        hipGetErrorString(e));
        }
        Carver cv{static_cast<char*>(b->mdi_block), 0};
        carve(cv);
        b->device_bytes += sizing.used;
    }
    if(b->n_pairs > 0) {
        static const bool force_generic = std::getenv("COATI_HIP_FORCE_GENERIC") != nullptr;
        if(m->gap_len == 1 && !force_generic)
            HIP_TRY(launch_forward_l1(device_view(b), m->n_tables == 1, m->stream));
        else if((m->gap_len == 2 || m->gap_len == 3) && !force_generic)
            HIP_TRY(launch_forward_k(device_view(b), m->stream));
        else
            HIP_TRY(launch_dp_generic(device_view(b), /*forward=*/true, m->stream));
    }
    b->forward_done = true;
    return COATI_HIP_OK;
}

int coati_hip_forward_final(coati_hip_batch_t* b, float* final_mdi) {
    if(b == nullptr || final_mdi == nullptr) return fail(COATI_HIP_EINVAL, "forward_final: NULL argument");
    if(!b->forward_done) return fail(COATI_HIP_ESTATE, "forward_final: forward was not launched");
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    if(b->n_pairs > 0)
        HIP_TRY(hipMemcpy(final_mdi, b->d_final_mdi, b->n_pairs * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return COATI_HIP_OK;
}

int coati_hip_debug_forward_matrices(coati_hip_batch_t* b, uint64_t pair, float* M, float* D, float* I,
                                     uint64_t capacity) {
    if(b == nullptr || M == nullptr || D == nullptr || I == nullptr)
        return fail(COATI_HIP_EINVAL, "debug_forward_matrices: NULL argument");
    if(!b->forward_done) return fail(COATI_HIP_ESTATE, "debug_forward_matrices: forward was not launched");
    if(pair >= b->n_pairs) return fail(COATI_HIP_EINVAL, "debug_forward_matrices: pair out of range");
    const uint64_t n = static_cast<uint64_t>(b->desc[pair].la) * b->desc[pair].lb;
    if(capacity < n) return fail(COATI_HIP_EINVAL, "debug_forward_matrices: capacity too small");
    if(n == 0) return COATI_HIP_OK;
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    float* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), 3 * n * sizeof(float)));
    hipLaunchKernelGGL(decode_mdi, dim3(static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 4096))), dim3(256), 0,
                       b->model->stream, b->d_desc, static_cast<uint32_t>(pair), b->d_mdi, d_out);
    hipError_t e = hipStreamSynchronize(b->model->stream);
    if(e == hipSuccess) e = hipMemcpy(M, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    if(e == hipSuccess) e = hipMemcpy(D, d_out + n, n * sizeof(float), hipMemcpyDeviceToHost);
    if(e == hipSuccess) e = hipMemcpy(I, d_out + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if(e != hipSuccess) return fail(COATI_HIP_EHIP, "debug_forward_matrices: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}

namespace {
// ---- exact-stream sampling in parallel (kernels and rationale: sampleback.hip) ------------------
// Per chunk and pair: sample j of the chunk is expected to start j * mean draws after the chunk
// origin; every offset within +-(z * sigma * sqrt(j) + 2), z = 2, of that is walked as a candidate.  The
// true chain origin -> sample 0 -> sample 1 ... is then followed through the candidates' draw
// counts; a sample whose true offset was not a candidate ends the chunk for its pair (it becomes
// sample 0 of the next chunk, whose offset is always a candidate), so the loop always advances.
using u128 = unsigned __int128;
constexpr uint64_t kLehmerMult = 0xda942042e4dd58b5ULL;  // contrib/random/random.hpp:95

u128 lehmer_pow(uint64_t n) {
    u128 r = 1, bpow = kLehmerMult;
    for(; n != 0; n >>= 1, bpow *= bpow)
        if(n & 1u) r *= bpow;
    return r;
}

hipError_t sampleback_speculative(coati_hip_batch* b, uint32_t n_samples, const uint64_t* rng_state,
                                  const std::vector<uint64_t>& base, uint8_t* d_ops, uint64_t* d_start, uint32_t* d_len,
                                  float* d_lw, uint64_t* states_out) {
    coati_hip_model* m = b->model;
    const uint64_t n = b->n_pairs;
    constexpr uint32_t kChunkMax = 512;
    static const uint32_t kMaxCands = [] {
        const char* e = std::getenv("COATI_HIP_SPEC_CANDS");
        const long v = e != nullptr ? std::atol(e) : 0;
        return v >= 1024 && v <= (1 << 22) ? static_cast<uint32_t>(v) : (1u << 17);  // measured best (tools/sample_bench.py)
    }();
    // half-width of a candidate window in standard deviations of the offset; too narrow only ends
    // a chunk early (COATI_HIP_SPEC_Z overrides, for tuning)
    static const double kZ = [] {
        const char* e = std::getenv("COATI_HIP_SPEC_Z");
        const double v = e != nullptr ? std::atof(e) : 0.0;
        return v >= 0.25 && v <= 10.0 ? v : 2.0;  // measured (16 x 1 000 samples of 1 kb pairs): z = 5: 38.8 ms, 3: 30.9, 2: 25.9, 1.5: 26.0, 1: 35.9
    }();
    size_t free_b = 0, total_b = 0;
    hipError_t e = hipMemGetInfo(&free_b, &total_b);
    if(e != hipSuccess) return e;
    {   // (cached blocks of this model count as free)
        std::lock_guard<std::mutex> hold(m->arena_lock);
        for(const auto& a : m->free_arenas) free_b += a.bytes;
    }
    // work arena for the candidates' ops: 2 GB, or a power of two below a quarter of the free HBM
    // (a stable size, so that repeated calls find their block in the cache)
    uint64_t tmp_budget = 2ull << 30;
    while(tmp_budget > (1ull << 20) && tmp_budget > free_b / 4) tmp_budget >>= 1;

    uint64_t dbg_rounds = 0, dbg_cands = 0;  // reported with COATI_HIP_TIMING=1
    struct PairState {
        u128 st0;
        uint64_t origin = 0;  // draws consumed by the samples resolved so far
        uint32_t done = 0, cnt = 0;
        double mean = 0.0, m2 = 0.0;
    };
    std::vector<PairState> ps(n);
    for(uint64_t p = 0; p < n; ++p) ps[p].st0 = (static_cast<u128>(rng_state[2 * p + 1]) << 64) | rng_state[2 * p];

    uint64_t mult_pow[64];
    {
        u128 bpow = kLehmerMult;
        for(int bit = 0; bit < 32; ++bit, bpow *= bpow) {
            mult_pow[2 * bit] = static_cast<uint64_t>(bpow);
            mult_pow[2 * bit + 1] = static_cast<uint64_t>(bpow >> 64);
        }
    }
    // all temporaries in ONE block from the model's workspace cache (a 2 GB hipMalloc per call costs
    // between 0.4 and several hundred ms, see coati_hip_model::free_arenas)
    uint64_t *d_origin = nullptr, *d_pow = nullptr, *d_cstart = nullptr;
    SpecCandidate* d_cands = nullptr;
    SpecCommit* d_commits = nullptr;
    uint8_t* d_tmp = nullptr;
    uint32_t *d_clen = nullptr, *d_cdraws = nullptr;
    float* d_clw = nullptr;
    void* block = nullptr;
    uint64_t block_bytes = 0;
    auto carve = [&](Carver& cv) {
        d_origin = cv.take<uint64_t>(2 * n);
        d_pow = cv.take<uint64_t>(64);
        d_cstart = cv.take<uint64_t>(kMaxCands);
        d_cands = cv.take<SpecCandidate>(kMaxCands);
        d_commits = cv.take<SpecCommit>(std::max<uint64_t>(std::min<uint64_t>(n * kChunkMax, kMaxCands), 1));
        d_clen = cv.take<uint32_t>(kMaxCands);
        d_cdraws = cv.take<uint32_t>(kMaxCands);
        d_clw = cv.take<float>(kMaxCands);
        d_tmp = cv.take<uint8_t>(tmp_budget);
    };
    auto release = [&]() {
        if(block == nullptr) return;
        if(hipStreamSynchronize(m->stream) == hipSuccess)
            model_give_arena(m, block, block_bytes);
        else
            (void)hipFree(block);
        block = nullptr;
    };
#define S_TRY(expr)                 \
    do {                            \
        e = (expr);                 \
        if(e != hipSuccess) {       \
            release();              \
            return e;               \
        }                           \
    } while(0)
    {
        Carver sizing;
        carve(sizing);
        S_TRY(model_take_arena(m, sizing.used, &block, &block_bytes));
        Carver cv{static_cast<char*>(block), 0};
        carve(cv);
    }
    S_TRY(hipMemcpyAsync(d_pow, mult_pow, sizeof(mult_pow), hipMemcpyHostToDevice, m->stream));

    struct Window {  // candidates of one (pair, sample-in-chunk)
        uint32_t first_cand, lo, hi;
    };
    std::vector<std::vector<Window>> windows(n);
    // host sides of the per-round copies, page-locked (a round adds at most one candidate per pair
    // beyond kMaxCands before the overflow check below)
    PinnedVec<SpecCandidate> cands;
    PinnedVec<uint64_t> origin_states;
    PinnedVec<uint32_t> draws;
    PinnedVec<SpecCommit> commits;
    {
        const uint64_t cap_c = static_cast<uint64_t>(kMaxCands) + n + 16, cap_m = std::min<uint64_t>(n * kChunkMax, cap_c) + 16;
        Carver sizing;
        auto carve_host = [&](Carver& cv) {
            cands.p = cv.take<SpecCandidate>(cap_c);
            origin_states.p = cv.take<uint64_t>(2 * n);
            draws.p = cv.take<uint32_t>(cap_c);
            commits.p = cv.take<SpecCommit>(cap_m);
        };
        carve_host(sizing);
        void* host_block = nullptr;
        S_TRY(model_pinned(m, sizing.used, &host_block));
        Carver cv{static_cast<char*>(host_block), 0};
        carve_host(cv);
        cands.cap = draws.cap = cap_c;
        origin_states.cap = origin_states.n = 2 * n;
        commits.cap = cap_m;
    }
    const BatchDeviceView view = device_view(b);
    try {
    for(;;) {
        cands.clear();
        uint64_t tmp_used = 0;
        bool any = false;
        uint64_t active = 0;
        for(uint64_t p = 0; p < n; ++p) active += ps[p].done < n_samples ? 1 : 0;
        // every unfinished pair gets an equal share of the candidate and work-arena budget; the chunk
        // of a pair is as long as its share allows (windows grow with sqrt(j))
        const uint64_t cand_share = kMaxCands / std::max<uint64_t>(active, 1), tmp_share = tmp_budget / std::max<uint64_t>(active, 1);
        for(uint64_t p = 0; p < n; ++p) {
            windows[p].clear();
            PairState& s = ps[p];
            if(s.done >= n_samples) continue;
            any = true;
            const uint64_t cand_begin = cands.size(), tmp_begin = tmp_used;
            if(cand_begin + 1 > kMaxCands ||
               tmp_used + static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb > tmp_budget)
                continue;  // more unfinished pairs than one round holds: this pair waits for the next round
            const uint64_t width = static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb;
            const uint32_t remaining = n_samples - s.done;
            const uint32_t want = s.cnt == 0 ? 1u : (s.cnt < 4 ? 4u : (s.cnt < 16 ? 16u : kChunkMax));
            const uint32_t chunk = std::min(remaining, want);
            // (few observations: widen, a window that is too narrow only ends the chunk early)
            const double sigma = s.cnt >= 2 ? std::sqrt(s.m2 / (s.cnt - 1)) * (1.0 + 4.0 / s.cnt) + 1.0
                                            : 0.02 * static_cast<double>(width) + 2.0;
            for(uint32_t j = 0; j < chunk; ++j) {
                const int64_t center = std::llround(j * s.mean);
                const int64_t half = j == 0 ? 0 : static_cast<int64_t>(std::ceil(kZ * sigma * std::sqrt(static_cast<double>(j)))) + 2;
                const int64_t lo = std::max<int64_t>(center - half, j), hi = std::max<int64_t>(center + half, lo);
                const uint64_t count = static_cast<uint64_t>(hi - lo + 1);
                if(j > 0 && (cands.size() - cand_begin + count > cand_share ||
                            tmp_used - tmp_begin + count * std::max<uint64_t>(width, 1) > tmp_share))
                    break;
                windows[p].push_back(Window{static_cast<uint32_t>(cands.size()), static_cast<uint32_t>(lo), static_cast<uint32_t>(hi)});
                for(int64_t off = lo; off <= hi; ++off) {
                    cands.push_back(SpecCandidate{static_cast<uint32_t>(p), static_cast<uint32_t>(off), tmp_used});
                    tmp_used += width;
                }
            }
            const u128 st = s.st0 * lehmer_pow(s.origin);
            origin_states[2 * p] = static_cast<uint64_t>(st);
            origin_states[2 * p + 1] = static_cast<uint64_t>(st >> 64);
        }
        if(!any) break;
        if(cands.size() > kMaxCands || tmp_used > tmp_budget) {  // a single sample does not fit the work arena
            release();
            return hipErrorOutOfMemory;
        }
        const uint32_t nc = static_cast<uint32_t>(cands.size());
        ++dbg_rounds;
        dbg_cands += nc;
        S_TRY(hipMemcpyAsync(d_origin, origin_states.data(), 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(hipMemcpyAsync(d_cands, cands.data(), nc * sizeof(SpecCandidate), hipMemcpyHostToDevice, m->stream));
        S_TRY(launch_spec_walk(view, d_origin, d_pow, d_cands, nc, d_tmp, d_cstart, d_clen, d_clw, d_cdraws, m->stream));
        draws.resize(nc);
        S_TRY(hipMemcpyAsync(draws.data(), d_cdraws, nc * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));
        // follow the chain of true offsets
        commits.clear();
        for(uint64_t p = 0; p < n; ++p) {
            PairState& s = ps[p];
            const uint64_t width = static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb;
            uint64_t off = 0;
            for(const Window& w : windows[p]) {
                if(off < w.lo || off > w.hi) break;  // not speculated: first sample of the next chunk
                const uint32_t cand = w.first_cand + static_cast<uint32_t>(off - w.lo);
                const uint64_t out_index = p * n_samples + s.done;
                commits.push_back(SpecCommit{cand, 0u, base[p] + (static_cast<uint64_t>(s.done) + 1) * width, out_index});
                const double x = static_cast<double>(draws[cand]);
                s.cnt += 1;  // Welford
                const double d1 = x - s.mean;
                s.mean += d1 / s.cnt;
                s.m2 += d1 * (x - s.mean);
                off += draws[cand];
                s.done += 1;
            }
            s.origin += off;
        }
        const uint32_t ncm = static_cast<uint32_t>(commits.size());
        S_TRY(hipMemcpyAsync(d_commits, commits.data(), ncm * sizeof(SpecCommit), hipMemcpyHostToDevice, m->stream));
        S_TRY(launch_spec_commit(d_commits, ncm, d_tmp, d_cstart, d_clen, d_clw, d_ops, d_start, d_len, d_lw, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));  // `commits`/`cands` are reused by the next round
    }
    } catch(...) {  // host-side allocation failure: free the device work areas, report at the ABI
        release();
        throw;
    }
#undef S_TRY
    if(std::getenv("COATI_HIP_TIMING") != nullptr)
        std::fprintf(stderr, "sampleback_speculative: %llu rounds, %llu candidate walks for %llu samples\n",
                     static_cast<unsigned long long>(dbg_rounds), static_cast<unsigned long long>(dbg_cands),
                     static_cast<unsigned long long>(n * n_samples));
    for(uint64_t p = 0; p < n; ++p) {
        const u128 st = ps[p].st0 * lehmer_pow(ps[p].origin);  // where n serial sampleback calls leave the stream
        states_out[2 * p] = static_cast<uint64_t>(st);
        states_out[2 * p + 1] = static_cast<uint64_t>(st >> 64);
    }
    release();
    return hipSuccess;
}
}  // namespace

namespace {
int sampleback_impl(coati_hip_batch_t* b, uint32_t n_samples, const uint64_t* rng_state, int independent_streams,
                    float* log_weights, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                    uint64_t* rng_state_out);
}

int coati_hip_sampleback(coati_hip_batch_t* b, uint32_t n_samples, const uint64_t* rng_state, int independent_streams,
                         float* log_weights, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                         uint64_t* rng_state_out) {
    try {
        return sampleback_impl(b, n_samples, rng_state, independent_streams, log_weights, ops, ops_capacity, ops_off, ops_len,
                               rng_state_out);
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "sampleback: host allocation failed");
    } catch(const std::exception& ex) {
        return fail(COATI_HIP_EHIP, "sampleback: %s", ex.what());
    }
}

namespace {
int sampleback_impl(coati_hip_batch_t* b, uint32_t n_samples, const uint64_t* rng_state, int independent_streams,
                    float* log_weights, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                    uint64_t* rng_state_out) {
    if(b == nullptr || rng_state == nullptr) return fail(COATI_HIP_EINVAL, "sampleback: NULL argument");
    if(!b->forward_done) return fail(COATI_HIP_ESTATE, "sampleback: forward was not launched");
    const uint64_t n = b->n_pairs;
    if(n == 0 || n_samples == 0) return COATI_HIP_OK;
    coati_hip_model* m = b->model;
    HIP_TRY(hipSetDevice(m->device));
    // ops slots: pair p, sample s at sample_base[p] + s * (la + lb)
    std::vector<uint64_t> base(n);
    uint64_t total = 0;
    for(uint64_t p = 0; p < n; ++p) {
        base[p] = total;
        total += static_cast<uint64_t>(n_samples) * (static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb);
    }
    if(ops != nullptr && ops_capacity < total)
        return fail(COATI_HIP_EINVAL, "sampleback: ops_capacity %llu < %llu", static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(total));
    const uint64_t walkers = independent_streams ? n * n_samples : n;
    std::vector<uint64_t> states(2 * walkers);
    if(independent_streams) {
        // sample s of pair p starts s * 2^32 draws into the pair's stream: state * (MULT^(2^32))^s mod 2^128
        using u128 = unsigned __int128;
        u128 jump = static_cast<u128>(0xda942042e4dd58b5ULL);
        for(int sq = 0; sq < 32; ++sq) jump *= jump;
        for(uint64_t p = 0; p < n; ++p) {
            u128 st = (static_cast<u128>(rng_state[2 * p + 1]) << 64) | rng_state[2 * p];
            for(uint32_t sidx = 0; sidx < n_samples; ++sidx) {
                states[2 * (p * n_samples + sidx)] = static_cast<uint64_t>(st);
                states[2 * (p * n_samples + sidx) + 1] = static_cast<uint64_t>(st >> 64);
                st *= jump;
            }
        }
    } else {
        std::memcpy(states.data(), rng_state, sizeof(uint64_t) * 2 * n);
    }
    const uint64_t n_out = n * n_samples;
    uint64_t *d_states = nullptr, *d_base = nullptr, *d_start = nullptr;
    uint8_t* d_ops = nullptr;
    uint32_t* d_len = nullptr;
    float* d_lw = nullptr;
    void* block = nullptr;
    uint64_t block_bytes = 0;
    auto carve = [&](Carver& cv) {
        d_states = cv.take<uint64_t>(states.size());
        d_base = cv.take<uint64_t>(n);
        d_start = cv.take<uint64_t>(n_out);
        d_len = cv.take<uint32_t>(n_out);
        d_lw = cv.take<float>(n_out);
        d_ops = cv.take<uint8_t>(std::max<uint64_t>(total, 16));
    };
    auto release = [&]() {
        if(block == nullptr) return;
        if(hipStreamSynchronize(m->stream) == hipSuccess)
            model_give_arena(m, block, block_bytes);
        else
            (void)hipFree(block);
        block = nullptr;
    };
    auto attempt = [&]() -> hipError_t {
        hipError_t e;
        {
            Carver sizing;
            carve(sizing);
            if((e = model_take_arena(m, sizing.used, &block, &block_bytes)) != hipSuccess) return e;
            Carver cv{static_cast<char*>(block), 0};
            carve(cv);
        }
        // exact stream with several samples per pair: walked in parallel by speculating the stream
        // offsets (identical results); COATI_HIP_SAMPLE_SEQUENTIAL=1 keeps the one-walker-per-pair loop
        static const bool sequential = std::getenv("COATI_HIP_SAMPLE_SEQUENTIAL") != nullptr;
        if(!independent_streams && n_samples >= 4 && !sequential) {
            if((e = sampleback_speculative(b, n_samples, rng_state, base, d_ops, d_start, d_len, d_lw, states.data())) != hipSuccess) return e;
            if((e = hipMemcpy(d_states, states.data(), 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
        } else {
            if((e = hipMemcpyAsync(d_states, states.data(), states.size() * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream)) != hipSuccess) return e;
            if((e = hipMemcpyAsync(d_base, base.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream)) != hipSuccess) return e;
            if((e = launch_sampleback(device_view(b), n_samples, independent_streams != 0, d_states, d_base, d_ops, d_start, d_len,
                                      d_lw, m->stream)) != hipSuccess) return e;
            if((e = hipStreamSynchronize(m->stream)) != hipSuccess) return e;
        }
        if(log_weights != nullptr && (e = hipMemcpy(log_weights, d_lw, n_out * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        if(ops != nullptr && total > 0 && (e = hipMemcpy(ops, d_ops, total, hipMemcpyDeviceToHost)) != hipSuccess) return e;
        if(ops_off != nullptr && (e = hipMemcpy(ops_off, d_start, n_out * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        if(ops_len != nullptr && (e = hipMemcpy(ops_len, d_len, n_out * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        if(rng_state_out != nullptr && !independent_streams &&
           (e = hipMemcpy(rng_state_out, d_states, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        return hipSuccess;
    };
    const hipError_t e = attempt();
    release();
    if(e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "sampleback: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}
}  // namespace

int coati_hip_debug_libm(coati_hip_model_t* model, int op, const float* in, uint64_t n, float* out) {
    if(model == nullptr || in == nullptr || out == nullptr) return fail(COATI_HIP_EINVAL, "debug_libm: NULL argument");
    if(op < 0 || op > 3) return fail(COATI_HIP_EINVAL, "debug_libm: op %d unknown", op);
    if(n == 0) return COATI_HIP_OK;
    HIP_TRY(hipSetDevice(model->device));
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), n * sizeof(float));
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), n * sizeof(float));
    if(e == hipSuccess) e = hipMemcpy(d_in, in, n * sizeof(float), hipMemcpyHostToDevice);
    if(e == hipSuccess) {
        hipLaunchKernelGGL(libm_kernel, dim3(2048), dim3(256), 0, model->stream, op, d_in, n, d_out);
        e = hipStreamSynchronize(model->stream);
    }
    if(e == hipSuccess) e = hipMemcpy(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    if(d_in != nullptr) (void)hipFree(d_in);
    if(d_out != nullptr) (void)hipFree(d_out);
    if(e != hipSuccess) return fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "debug_libm: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}

int coati_hip_debug_rng_f24(coati_hip_model_t* model, const uint64_t rng_state[2], uint32_t n, float* out) {
    if(model == nullptr || rng_state == nullptr || out == nullptr) return fail(COATI_HIP_EINVAL, "debug_rng_f24: NULL argument");
    HIP_TRY(hipSetDevice(model->device));
    float* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), std::max<uint32_t>(n, 1) * sizeof(float)));
    hipError_t e = launch_rng_f24(rng_state, n, d_out, model->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(model->stream);
    if(e == hipSuccess) e = hipMemcpy(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if(e != hipSuccess) return fail(COATI_HIP_EHIP, "debug_rng_f24: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}

namespace {
// Is `p` page-locked host memory HIP knows about (hipHostMalloc / hipHostRegister / coati_hip_host_alloc)?
bool is_pinned_host(const void* p) {
    if(p == nullptr) return false;
    hipPointerAttribute_t attr;
    if(hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// HBM workspace + page-locked staging a chunk of pairs [p0, p1) needs (upper bounds; the plan of
// batch_create_impl is authoritative and fails cleanly if a chunk does not fit after all)
struct ChunkNeed {
    uint64_t fixed = 0, ck_sum = 0, ck_max = 0, ck_own16 = 0, ck_cut16 = 0, pairs = 0, seq_bytes = 0, meta_bytes = 0, ops = 0, cells = 0;
    // checkpoints: per pair, or in per-wavefront slots when that is smaller (batch_create_impl decides the same way)
    // (+ the own storage of the pairs a large batch cuts into row parts: batch_create_impl, "the ragged end")
    uint64_t arena() const {
        const uint64_t cut = pairs > ck_scratch_waves() ? std::min<uint64_t>(2 * ck_scratch_waves(), pairs - ck_scratch_waves()) : 0;
        return fixed + std::min<uint64_t>(ck_sum, ck_max * (ck_scratch_waves() + cut) + ck_sum / 64);
    }
    // chunk of a streamed call: wavefront slots and traceback scratch are the call's, not the chunk's
    uint64_t arena_streamed(bool tail = false) const {
        return fixed - static_cast<uint64_t>(ck_scratch_waves()) * ck_scratch_dwords_per_wave() * sizeof(uint32_t) + ck_own16 + (tail ? ck_cut16 : 0);
    }
};
void chunk_need_add(ChunkNeed& nd, uint64_t la, uint64_t lb, uint32_t gap_len) {
    uint64_t w = 0;
    if(la > 0 && lb > 0) {
        if(gap_len == 1) {  // the plan may narrow the strips (4, 8 or 16 columns per lane): take the largest
            for(uint32_t cw = 4; cw <= 16; cw *= 2)
                w = std::max<uint64_t>(w, (lb + kWave * cw - 1) / (kWave * cw) * ck_strip_dwords(static_cast<uint32_t>(la), cw) * 4);
        } else
            w = static_cast<uint64_t>(n_strips(static_cast<uint32_t>(lb))) * strip_dwords(static_cast<uint32_t>(la)) * 4;
    }
    const uint64_t strips = std::max<uint64_t>(1, (lb + 255) / 256);  // (narrowest plan: 4 columns per lane)
    nd.ck_sum += w;
    if(gap_len == 1 && lb <= static_cast<uint64_t>(kWave) * kW && w <= (4ull << 20))
        nd.ck_max = std::max(nd.ck_max, w);  // slot-eligible
    if(gap_len == 1 && la > 0 && lb > 0) {
        // chunk of a streamed call (16 columns per lane always; batch_create_impl with wave_slot_dwords): pairs of
        // several strips, or too long for a wavefront slot, keep their checkpoints in the chunk's workspace
        uint32_t ns = 1, wl = kW;
        viterbi_strip_plan(static_cast<uint32_t>(lb), kW, ns, wl);
        const uint64_t last = ck_strip_dwords(static_cast<uint32_t>(la), wl);
        if(ns > 1 || last > (1ull << 20))
            nd.ck_own16 += ((ns - 1) * ck_strip_dwords(static_cast<uint32_t>(la), kW) + last) * 4;
        else if(wl == kW)
            nd.ck_cut16 += (last + kCkPartStateDwords) * 4;  // (what it keeps if the chunk is one of the call's last: row parts)
    }
    nd.fixed += 3 * (la + lb) + 16 * (la + 1) + sizeof(PairDesc) + 32 + strips * 24 + 7 * 12 + 1024;
    nd.pairs += 1;
    nd.seq_bytes += la + lb;
    nd.meta_bytes += sizeof(PairDesc) + 4 + strips * 24 + 16 + 7 * 12;  // (+ the items and progress words of up to 8 row parts)  // descriptor, order entry, work items (both lists), progress word
    nd.ops += la + lb;
    nd.cells += la * lb;
}

struct PipeChunk {
    uint64_t p0 = 0, p1 = 0, ops_base = 0, ops_bytes = 0;
};

// Streamed form (viterbi_batch_stream): fixed slot sizes -- nothing can grow while the persistent kernel runs, the
// chunks are cut to fit.  Workspace: room for the largest pair this form accepts with its own checkpoints; staging
// block [what goes up | short result arrays, and the ops when the caller's array is pageable].
constexpr uint64_t kStreamSlotArena = 192ull << 20, kStreamSlotStaging = 48ull << 20;
uint64_t stream_out_bytes(uint64_t n, uint64_t ops_bytes, bool out_pinned) {
    return 4 * 256 + 2 * kMinDmaBytes + n * (sizeof(float) + sizeof(uint64_t) + sizeof(uint32_t)) + (out_pinned ? uint64_t{0} : ops_bytes);
}
uint64_t stream_staging_bytes(const ChunkNeed& nd, uint64_t n, bool in_pinned, bool out_pinned) {
    return nd.meta_bytes + 8 * 256 + 2 * kMinDmaBytes + (in_pinned ? std::min<uint64_t>(nd.seq_bytes, 2 * kMinDmaBytes) : nd.seq_bytes) + 512 +
           stream_out_bytes(n, nd.ops, out_pinned) + 512;
}
uint64_t stream_chunk_fixed() { return static_cast<uint64_t>(ck_scratch_waves()) * ck_scratch_dwords_per_wave() * sizeof(uint32_t) + (64u << 10); }
// Does a chunk made of this ONE pair fit a stream slot (workspace and staging)?  The chunk cutter always accepts
// the first pair of a chunk, so every pair of a streamed call must pass this (a long-thin pair -- la = 50 M,
// lb = 1 -- has few cells but 19 bytes of workspace and 2 bytes of staging per ancestor position).
bool stream_pair_fits(uint64_t la, uint64_t lb, uint32_t gap_len, bool in_pinned, bool out_pinned) {
    ChunkNeed one;
    one.fixed = stream_chunk_fixed();
    chunk_need_add(one, la, lb, gap_len);
    const uint64_t arena = one.arena_streamed(true);
    return arena + arena / 8 + (1u << 20) <= kStreamSlotArena && stream_staging_bytes(one, 1, in_pinned, out_pinned) <= kStreamSlotStaging;
}
}  // namespace

namespace {
// The streamed form of coati_hip_viterbi_batch: viterbi_ck_stream runs for the whole call on the model's stream;
// the host plans chunk after chunk into kCkStreamSlots small workspaces, uploads on ONE in-order stream, tells the
// kernel how many work items exist through page-locked memory, and downloads a chunk (on a third stream) when the
// kernel has flagged it complete.  Only copy-ENGINE copies may be issued while the kernel owns every wavefront
// slot of the chip: no hipMemset, no copy of kMinDmaBytes or less (both are kernels), no hipMalloc / hipFree
// (they may wait for the device).  Everything is allocated before the launch; COATI_HIP_ESTATE = nothing usable
// happened (an allocation failed before the launch, or the kernel gave up waiting): the caller runs the chunk
// pipeline instead.
int viterbi_batch_stream(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat, const uint64_t* a_off, const uint8_t* b_cat,
                         const uint64_t* b_off, float* scores, uint8_t* ops, uint64_t* ops_off, uint32_t* ops_len, bool in_pinned,
                         bool out_pinned, long double total_cells, uint64_t longest_single, std::chrono::steady_clock::time_point t_call) {
    constexpr int kSlots = kCkStreamSlots;
    for(int q = 1; q <= 2; ++q)
        if(model->slots[q].stream == nullptr && hipStreamCreateWithFlags(&model->slots[q].stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            return COATI_HIP_ESTATE;
        }
    hipStream_t kernel_stream = model->stream, up_stream = model->slots[1].stream, down_stream = model->slots[2].stream;
    const uint64_t host_bytes = ck_stream_host_bytes();
    auto soft = [](hipError_t e) {  // an allocation that fails here is not an error of the call
        if(e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    if(model->d_stream_ctl == nullptr && !soft(hipMalloc(&model->d_stream_ctl, ck_stream_ctl_bytes()))) return COATI_HIP_ESTATE;
    if(model->h_stream == nullptr && !soft(hipHostMalloc(&model->h_stream, host_bytes, hipHostMallocCoherent | hipHostMallocMapped)))
        return COATI_HIP_ESTATE;
    // the wavefronts' checkpoint slots (as large as the longest single-strip pair of the call needs, at most 4 MB:
    // longer ones keep their checkpoints in their chunk's workspace) and traceback scratch
    uint64_t slot_dwords = 256;
    for(uint32_t cw = 4; cw <= 16; cw *= 2) slot_dwords = std::max<uint64_t>(slot_dwords, ck_strip_dwords(static_cast<uint32_t>(longest_single), cw));
    const uint64_t wave_slot_bytes = (std::min<uint64_t>(slot_dwords, 1ull << 20) * 4 + 255) / 256 * 256;
    const uint64_t scratch_bytes = ck_scratch_dwords_per_wave() * sizeof(uint32_t);
    const uint64_t waves_bytes = static_cast<uint64_t>(ck_scratch_waves()) * (wave_slot_bytes + scratch_bytes);
    if(model->stream_waves_bytes < waves_bytes) {
        if(model->d_stream_waves != nullptr) (void)hipFree(model->d_stream_waves);
        model->d_stream_waves = nullptr;
        model->stream_waves_bytes = 0;
        if(!soft(hipMalloc(&model->d_stream_waves, waves_bytes))) return COATI_HIP_ESTATE;
        model->stream_waves_bytes = waves_bytes;
    }
    // slots: fixed sizes (nothing can grow while the kernel runs; the chunks are cut to fit).  Workspace: room for
    // the largest pair this form accepts (kStreamPairCells) with its own checkpoints; staging block
    // [what goes up | short result arrays, and the ops when the caller's array is pageable]
    constexpr uint64_t kSlotArena = kStreamSlotArena, kSlotStaging = kStreamSlotStaging;
    auto out_bytes_of = [&](uint64_t n, uint64_t ops_bytes) { return stream_out_bytes(n, ops_bytes, out_pinned); };
    auto staging_of = [&](const ChunkNeed& nd, uint64_t n) { return stream_staging_bytes(nd, n, in_pinned, out_pinned); };
    // How many slots can this call use?  The chunk targets below in cells: 1/2, 1, 2, then 3 units, and 1 unit each
    // once four units are left.  A one-shot process (coati-alignpair --batch: ~0.1 ms per MB of fresh hipMalloc /
    // hipHostMalloc, 12 slots are 2.9 GB) allocates what its input needs; a second call on the model takes the rest.
    long double kUnitCells = 1000.0L * 1002 * 1002;
    if(const char* e = std::getenv("COATI_HIP_STREAM_UNIT")) {
        const long double forced = std::strtold(e, nullptr);
        if(forced >= 1.0L) kUnitCells = forced;
    }
    int n_slots = kSlots;
    if(model->stream_calls == 0) {
        int est = 0;
        for(long double done = 0; done < total_cells && est < kSlots; ++est)
            done += est == 0 ? kUnitCells / 2 : (est == 1 || total_cells - done <= 4 * kUnitCells) ? kUnitCells : est == 2 ? 2 * kUnitCells : 3 * kUnitCells;
        n_slots = std::min(kSlots, std::max(3, est + 1));  // (+1: a memory-bound cut may add a chunk; fewer slots than chunks only means reuse)
        for(int q = 0; q < kSlots; ++q)
            if(model->sslots[q].arena_bytes >= kSlotArena && model->sslots[q].pinned_bytes >= kSlotStaging) n_slots = std::max(n_slots, q + 1);
    }
    for(int q = 0; q < n_slots; ++q) {
        auto& ss = model->sslots[q];
        if(ss.arena_bytes < kSlotArena) {
            if(ss.arena != nullptr) (void)hipFree(ss.arena);
            ss.arena = nullptr;
            ss.arena_bytes = 0;
            if(!soft(hipMalloc(&ss.arena, kSlotArena))) return COATI_HIP_ESTATE;
            ss.arena_bytes = kSlotArena;
        }
        if(ss.pinned_bytes < kSlotStaging) {
            if(ss.pinned != nullptr) (void)hipHostFree(ss.pinned);
            ss.pinned = nullptr;
            ss.pinned_bytes = 0;
            if(!soft(hipHostMalloc(&ss.pinned, kSlotStaging, hipHostMallocDefault))) return COATI_HIP_ESTATE;
            ss.pinned_bytes = kSlotStaging;
        }
    }
    // the call's last chunks -- everything behind the first round of 4 096 wavefronts, up to ~7 500 pairs of 1 kb -- are
    // cut into row parts (finer items for the ragged end of the kernel, as a resident batch's later pairs are, abi.hip
    // "the ragged end"): their pairs keep their checkpoints, ~1.1 MB per 1 kb pair -- six larger workspaces of ~1 250 pairs
    const uint64_t tail_bytes = std::min<uint64_t>(3ull << 30, std::max<uint64_t>(kSlotArena, 1250 * (wave_slot_bytes + 4096) + (64ull << 20)));
    // (a model's FIRST call on a small input runs without them: 2 x 1.4 GB of fresh allocation cost a one-shot process
    // ~30 ms and buy its 10 000-pair kernel 0.5 ms)
    const bool want_tails = model->stream_calls > 0 || total_cells >= 30 * kUnitCells;
    if(want_tails && model->stream_tail_bytes < tail_bytes) {
        for(void*& t : model->stream_tail_arena) {
            if(t != nullptr) (void)hipFree(t);
            t = nullptr;
        }
        model->stream_tail_bytes = 0;
        bool ok = true;
        for(void*& t : model->stream_tail_arena) ok = ok && soft(hipMalloc(&t, tail_bytes));
        if(ok) model->stream_tail_bytes = tail_bytes;  // (else: no row parts in this call)
    }
    void* hs = model->h_stream;
    std::memset(hs, 0, host_bytes);
    ck_stream_host_set_slots(hs, static_cast<uint32_t>(n_slots));
    void* hs_dev = nullptr;
    if(!soft(hipHostGetDevicePointer(&hs_dev, hs, 0))) return COATI_HIP_ESTATE;
    if(model->stream_events[0] == nullptr) {
        bool events_ok = true;
        for(hipEvent_t& e : model->stream_events) events_ok = events_ok && soft(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if(!events_ok) {
            for(hipEvent_t& e : model->stream_events) {
                if(e != nullptr) (void)hipEventDestroy(e);
                e = nullptr;
            }
            return COATI_HIP_ESTATE;
        }
    }
    hipEvent_t up_done = model->stream_events[kSlots];
    hipEvent_t* copied = model->stream_events;
    // the control block starts zeroed (before the launch a fill kernel may run)
    uint32_t* wave_ck = static_cast<uint32_t*>(model->d_stream_waves);
    uint32_t* wave_scratch = reinterpret_cast<uint32_t*>(static_cast<char*>(model->d_stream_waves) + static_cast<uint64_t>(ck_scratch_waves()) * wave_slot_bytes);
    hipError_t e0 = hipMemsetAsync(model->d_stream_ctl, 0, ck_stream_ctl_bytes(), kernel_stream);
    if(e0 == hipSuccess)
        e0 = launch_viterbi_ck_stream(model->d_table, model->k, model->n_tables == 1, model->d_stream_ctl, hs_dev, wave_ck, wave_slot_bytes / 4,
                                      wave_scratch, kernel_stream);
    if(e0 != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e0));
    struct Closer {  // whatever happens below, the kernel is told to finish
        void* host;
        ~Closer() { ck_stream_host_close(host); }
    } closer{hs};

    struct InFlight {
        coati_hip_batch_t* batch = nullptr;
        PipeChunk chunk;
        uint32_t chunk_no = 0;
        bool d2h_submitted = false;
        char* out_stage = nullptr;
        uint64_t out_off = 0;
        bool ops_staged = false;
    };
    InFlight fl[kSlots];
    int rc = COATI_HIP_OK;
    const bool pipe_timing = std::getenv("COATI_HIP_PIPE_TIMING") != nullptr;
    auto t_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    if(pipe_timing) std::fprintf(stderr, "viterbi_batch[stream]: schedule made, kernel launched at %.2f ms\n", t_ms());
    constexpr int kGaveUp = -1000, kRedo = -1001;  // (private to this function)
    // (the kernel only ends after `closed`: early = it gave up; a stream in an error state is gone too -- never spin on it)
    auto kernel_gone = [&]() {
        const hipError_t q = hipStreamQuery(kernel_stream);
        if(q != hipSuccess && q != hipErrorNotReady) (void)hipGetLastError();
        return q != hipErrorNotReady;
    };

    // results: [scores | ops offsets | ops lengths | ops] are one contiguous group of the workspace.  Pageable
    // destination: one copy of the group into the slot's page-locked block.  Page-locked destination: the three
    // short arrays still go through the block (one copy, padded past kMinDmaBytes: shorter ones would be done by
    // a copy kernel, which cannot start under viterbi_ck_stream), the ops go straight to the caller's array.
    auto submit_d2h = [&](InFlight& f, int slot) -> hipError_t {
        f.d2h_submitted = true;
        coati_hip_model::StreamSlot& sl = model->sslots[slot];
        const PipeChunk& c = f.chunk;
        coati_hip_batch* b = f.batch;
        f.out_stage = static_cast<char*>(sl.pinned) + f.out_off;
        const uint64_t group = static_cast<uint64_t>(reinterpret_cast<char*>(b->d_ops) - reinterpret_cast<char*>(b->d_scores));  // (the three short arrays, each padded to 256 bytes)
        const bool ops_direct = out_pinned && ops != nullptr && c.ops_bytes > kMinDmaBytes;
        f.ops_staged = ops != nullptr && c.ops_bytes > 0 && !ops_direct;
        const uint64_t bytes = std::max<uint64_t>(group + (f.ops_staged ? c.ops_bytes : 0), kMinDmaBytes + 256);
        hipError_t e = hipMemcpyAsync(f.out_stage, b->d_scores, bytes, hipMemcpyDeviceToHost, down_stream);
        if(e == hipSuccess && ops_direct) e = hipMemcpyAsync(ops + c.ops_base, b->d_ops, c.ops_bytes, hipMemcpyDeviceToHost, down_stream);
        if(e == hipSuccess) e = hipEventRecord(copied[slot], down_stream);
        return e;
    };
    // non-blocking: submit the download of every chunk the kernel has flagged, retire every chunk whose download is done
    auto progress = [&]() -> int {
        for(int q = 0; q < kSlots; ++q) {
            InFlight& f = fl[q];
            if(f.batch == nullptr) continue;
            if(!f.d2h_submitted) {
                if(*ck_stream_host_done_flag(hs, q) != f.chunk_no + 1u) continue;
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                if(const unsigned long long bad = ck_stream_host_bad(hs, q)) {  // (the kernel's check of the codes: ck_report_bad)
                    const uint64_t pair = f.chunk.p0 + (bad & 0xffffffffull);
                    return fail(COATI_HIP_EINVAL, "batch_create: %s code %u out of range (pair %llu)", (bad >> 40) & 1 ? "descendant" : "ancestor",
                                static_cast<unsigned>((bad >> 32) & 0xff), static_cast<unsigned long long>(pair));
                }
                const hipError_t e = submit_d2h(f, q);
                if(e != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
            }
            const hipError_t qd = hipEventQuery(copied[q]);
            if(qd == hipErrorNotReady) continue;
            if(qd != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(qd));
            const PipeChunk& c = f.chunk;
            const uint64_t n = c.p1 - c.p0;
            {
                char* at = f.out_stage;
                if(scores != nullptr) std::memcpy(scores + c.p0, at, n * sizeof(float));
                at += (n * sizeof(float) + 255) / 256 * 256;
                if(ops_off != nullptr) std::memcpy(ops_off + c.p0, at, n * sizeof(uint64_t));
                at += (n * sizeof(uint64_t) + 255) / 256 * 256;
                if(ops_len != nullptr) std::memcpy(ops_len + c.p0, at, n * sizeof(uint32_t));
                at += (n * sizeof(uint32_t) + 255) / 256 * 256;
                if(f.ops_staged) std::memcpy(ops + c.ops_base, at, c.ops_bytes);
            }
            if(ops_off != nullptr)
                for(uint64_t p = c.p0; p < c.p1; ++p) ops_off[p] += c.ops_base;
            if(pipe_timing)
                std::fprintf(stderr, "viterbi_batch[stream]: chunk %u (%llu pairs) complete %.2f ms after the kernel started, on the host at %.2f ms\n",
                             f.chunk_no, static_cast<unsigned long long>(n), ck_stream_host_done_ms(hs, q), t_ms());
            coati_hip_batch_destroy(f.batch);
            f.batch = nullptr;
        }
        return COATI_HIP_OK;
    };
    // blocking: until slot q is free (or the kernel is gone without finishing it)
    auto wait_free = [&](int q) -> int {
        for(uint64_t spins = 0; fl[q].batch != nullptr; ++spins) {
            const int r = progress();
            if(r != COATI_HIP_OK) return r;
            if(fl[q].batch == nullptr) break;
            if((spins & 1023u) == 1023u && !fl[q].d2h_submitted && kernel_gone() && *ck_stream_host_done_flag(hs, q) != fl[q].chunk_no + 1u)
                return kGaveUp;  // (the kernel ended before this chunk was complete: it had given up waiting for the host)
            if(spins > 64) sched_yield();
        }
        return COATI_HIP_OK;
    };

    // chunks are cut as the call goes (the kernel is already waiting): 3 units of 10^9 cells (3 000 pairs of 1 kb),
    // the first ones and the last ones smaller (the GPU starts after ~0.1 ms of planning; the very last download is
    // the only one nothing hides); never more than fits a slot.  The persistent kernel takes chunks of any size at
    // full rate, but a chunk occupies its slot for as long as its SLOWEST pair takes -- measured: 5 to 6 ms for a
    // 1 kb pair on a fully shared SIMD, three times the mean, the four wavefronts of a SIMD do not advance evenly --
    // so the slots together must hold well over 6 ms of work (12 300 pairs of 1 kb) or the GPU runs dry
    const long double kUnit = kUnitCells;  // (COATI_HIP_STREAM_UNIT, tests: many small chunks out of a small input)
    const uint32_t gap_len = static_cast<uint32_t>(model->gap_len);
    uint32_t published = 0;
    uint64_t p0 = 0, ops_base = 0;
    long double cells_done = 0;
    int tails_used = 0;
    const char* const no_tail_parts = std::getenv("COATI_HIP_STREAM_NO_PARTS");  // (A/B)
    for(size_t ci = 0; p0 < n_pairs && rc == COATI_HIP_OK; ++ci) {
        const int q = static_cast<int>(ci % static_cast<size_t>(n_slots));
        coati_hip_model::StreamSlot& sl = model->sslots[q];
        InFlight& f = fl[q];
        const double t_begin = t_ms();
        rc = wait_free(q);
        if(rc != COATI_HIP_OK) break;
        const long double target = ci == 0 ? kUnit / 2 : (ci == 1 || total_cells - cells_done <= 4 * kUnit) ? kUnit : ci == 2 ? 2 * kUnit : 3 * kUnit;
        // row parts, in the large workspaces: the chunks behind the first 4 100 pairs' worth of cells, while at most
        // 8 300 pairs' worth are left
        const bool tail = ci >= 2 && cells_done >= 4.1L * kUnit && total_cells - cells_done <= 8.3L * kUnit &&
                          tails_used < coati_hip_model::kStreamTails && model->stream_tail_bytes != 0 && no_tail_parts == nullptr;
        void* const arena = tail ? model->stream_tail_arena[tails_used] : sl.arena;
        const uint64_t arena_bytes = tail ? model->stream_tail_bytes : sl.arena_bytes;
        ChunkNeed nd;
        nd.fixed = static_cast<uint64_t>(ck_scratch_waves()) * ck_scratch_dwords_per_wave() * sizeof(uint32_t) + (64u << 10);
        uint64_t p1 = p0;
        while(p1 < n_pairs) {
            ChunkNeed with = nd;
            chunk_need_add(with, a_off[p1 + 1] - a_off[p1], b_off[p1 + 1] - b_off[p1], gap_len);
            if(p1 > p0 && (static_cast<long double>(with.cells) > target || with.arena_streamed(tail) + with.arena_streamed(tail) / 8 + (1u << 20) > arena_bytes ||
                           staging_of(with, p1 + 1 - p0) > sl.pinned_bytes))
                break;
            nd = with;
            ++p1;
        }
        const PipeChunk c{p0, p1, ops_base, nd.ops};
        p0 = p1;
        ops_base += nd.ops;
        cells_done += static_cast<long double>(nd.cells);
        const uint64_t n = c.p1 - c.p0;
        // (the cutter takes the first pair of a chunk unseen: one that does not fit a slot ends the streamed form --
        // the kernel is closed below and the chunk pipeline, whose workspaces grow, does the call; never compute the
        // staging split from an unchecked subtraction)
        if(staging_of(nd, n) > sl.pinned_bytes || out_bytes_of(n, c.ops_bytes) > sl.pinned_bytes ||
           nd.arena_streamed(tail) + nd.arena_streamed(tail) / 8 + (1u << 20) > arena_bytes) {
            rc = kRedo;
            break;
        }
        const uint64_t out_off = (sl.pinned_bytes - out_bytes_of(n, c.ops_bytes)) / 256 * 256;
        BatchOpts bo;
        bo.stream = up_stream;
        bo.arena = arena;
        bo.arena_bytes = arena_bytes;
        bo.staging = static_cast<char*>(sl.pinned);
        bo.staging_bytes = out_off;
        bo.seqs_pinned = in_pinned;
        bo.force_ck = true;
        bo.force_w_main = kW;  // (a small chunk is not a small batch: no narrowed strips)
        bo.device_validates = true;
        if(tail) {
            bo.tail_parts = 3;
            ++tails_used;
        }
        bo.wave_slot_dwords = wave_slot_bytes / 4;
        rc = batch_create_impl(model, n, a_cat, a_off + c.p0, b_cat, b_off + c.p0, nullptr, &bo, &f.batch);
        if(rc != COATI_HIP_OK) {  // (ENOMEM: the slot's workspace cannot grow while the kernel runs)
            if(pipe_timing)
                std::fprintf(stderr, "viterbi_batch[stream]: chunk %zu of %llu pairs: estimate %llu bytes (fixed %llu, own checkpoints %llu)\n", ci,
                             static_cast<unsigned long long>(n), static_cast<unsigned long long>(nd.arena_streamed()),
                             static_cast<unsigned long long>(nd.fixed), static_cast<unsigned long long>(nd.ck_own16));
            break;
        }
        coati_hip_batch* b = f.batch;
        if(!b->ck) {
            rc = fail(COATI_HIP_ESTATE, "viterbi_batch: a streamed chunk was not planned for viterbi_ck");
            break;
        }
        f.chunk = c;
        f.chunk_no = static_cast<uint32_t>(ci);
        f.d2h_submitted = false;
        f.out_off = out_off;
        // the chunk's data (with its zeroed progress words) is on its way; once it is in HBM the kernel may know
        hipError_t e = hipEventRecord(up_done, up_stream);
        ck_stream_fill_chunk(hs, hs_dev, q, b->arena, device_view(b), static_cast<uint32_t>(n), published, static_cast<uint32_t>(ci));
        // Every copy under the persistent kernel must be done by the copy ENGINE: a copy the runtime does with a blit
        // kernel (HSA_ENABLE_SDMA=0, or its own choice) cannot start while viterbi_ck_stream holds every wavefront
        // slot.  So the wait is bounded -- 100 ms for the call's first chunk (a copy engine delivers it in well under
        // a millisecond), 5 s later on -- and a miss closes the kernel, hands the call to the chunk pipeline and is
        // remembered on the model (no later call tries the streamed form again).
        if(e == hipSuccess) {
            const auto t_up = std::chrono::steady_clock::now();
            const double bound_ms = ci == 0 ? 100.0 : 5000.0;
            for(uint64_t spins = 0;; ++spins) {
                e = hipEventQuery(up_done);
                if(e != hipErrorNotReady) break;
                if(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count() > bound_ms) break;
                if(spins > 256) sched_yield();
            }
            if(e == hipErrorNotReady) {
                (void)hipGetLastError();
                model->stream_unusable = true;
                if(pipe_timing) std::fprintf(stderr, "viterbi_batch[stream]: upload of chunk %zu not done after %.0f ms; falling back\n", ci, bound_ms);
                rc = kRedo;
                break;
            }
        }
        if(e != hipSuccess) {
            rc = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
            break;
        }
        published += b->n_items;
        ck_stream_host_announce(hs, static_cast<uint32_t>(ci) + 1u, published);
        if(pipe_timing)
            std::fprintf(stderr, "viterbi_batch[stream]: chunk %zu (%llu pairs, %u items, slot %d) planned + uploaded %.2f .. %.2f ms\n", ci,
                         static_cast<unsigned long long>(n), b->n_items, q, t_begin, t_ms());
        rc = progress();
    }
    ck_stream_host_close(hs);
    ++model->stream_calls;
    for(int q = 0; q < kSlots && rc == COATI_HIP_OK; ++q) rc = wait_free(q);
    // the kernel ends by itself once it has seen `closed`; then its verdict
    const hipError_t es = hipStreamSynchronize(kernel_stream);
    uint32_t dev_error = 0;
    if(es == hipSuccess) (void)hipMemcpy(&dev_error, static_cast<char*>(model->d_stream_ctl) + ck_stream_error_offset(), sizeof dev_error, hipMemcpyDeviceToHost);
    for(InFlight& f : fl) {
        if(f.batch != nullptr) {  // (only after an error)
            (void)hipStreamSynchronize(up_stream);
            (void)hipStreamSynchronize(down_stream);
            coati_hip_batch_destroy(f.batch);
            f.batch = nullptr;
        }
    }
    if(pipe_timing) std::fprintf(stderr, "viterbi_batch[stream]: done at %.2f ms\n", t_ms());
    if(rc == COATI_HIP_OK && es != hipSuccess) rc = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(es));
    if(es == hipSuccess && rc == kRedo) return COATI_HIP_ESTATE;  // (a pair or an upload the streamed form cannot serve: the chunk pipeline does the call)
    if(rc == kRedo) rc = COATI_HIP_OK;                            // (and the stream failed on top of it: reported just below)
    if(es == hipSuccess && (rc == kGaveUp || (rc == COATI_HIP_OK && dev_error != 0))) {
        // the kernel's waits are bounded (a host thread that was stopped for seconds must not hang the GPU): it gave
        // up, some chunks are incomplete.  Everything is quiet now; the chunk pipeline does the call again.
        if(pipe_timing) std::fprintf(stderr, "viterbi_batch[stream]: the kernel gave up waiting (code %u); falling back\n", dev_error);
        return COATI_HIP_ESTATE;
    }
    if(rc == kGaveUp) rc = fail(COATI_HIP_EHIP, "viterbi_batch: the streaming kernel ended early");
    return rc;
}
}  // namespace

/* One-shot Viterbi over any number of pairs, PIPELINED: the input is cut into chunks; chunk k's
 * upload and kernel run on one of three slots (stream + HBM workspace + page-locked staging, kept by
 * the model between calls) while chunk k-1's results travel back and the host plans chunk k+1; the
 * kernels of consecutive chunks overlap at their ragged ends.  The first chunks are small so that the
 * GPU starts early.  Arrays the caller allocated with coati_hip_host_alloc (or page-locked otherwise)
 * are copied from / into directly; pageable ones pass through the slot's staging block. */
int coati_hip_viterbi_batch(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                            const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                            float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off,
                            uint32_t* ops_len) {
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_batch: model is NULL");
    if(a_off == nullptr || b_off == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_batch: offsets are NULL");
    if(n_pairs == 0) return COATI_HIP_OK;
    try {
    const auto t_entry = std::chrono::steady_clock::now();
    std::lock_guard<std::mutex> one_call(model->pipeline_lock);
    HIP_TRY(hipSetDevice(model->device));
    const uint32_t gap_len = static_cast<uint32_t>(model->gap_len);
    // ---- the input once: valid offsets, cells, what decides the form of the call
    long double total_cells = 0;
    uint64_t widest = 0, max_pair_cells = 0, longest_single = 0, longest_a = 0;
    for(uint64_t p = 0; p < n_pairs; ++p) {
        if(a_off[p + 1] < a_off[p] || b_off[p + 1] < b_off[p])
            return fail(COATI_HIP_EINVAL, "viterbi_batch: offsets of pair %llu decrease", static_cast<unsigned long long>(p));
        const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
        if(la > 0xffffffffull || lb > 0xffffffffull)
            return fail(COATI_HIP_EINVAL, "viterbi_batch: pair %llu is longer than 2^32", static_cast<unsigned long long>(p));
        const uint64_t cells = la * lb;
        total_cells += static_cast<long double>(cells);
        widest = std::max(widest, lb);
        max_pair_cells = std::max(max_pair_cells, cells);
        longest_a = std::max(longest_a, la);
        if(lb > 0 && lb <= static_cast<uint64_t>(kStrip)) longest_single = std::max(longest_single, la);
    }
    const uint64_t ops_total = (a_off[n_pairs] - a_off[0]) + (b_off[n_pairs] - b_off[0]);
    if(ops != nullptr && ops_capacity < ops_total)
        return fail(COATI_HIP_EINVAL, "viterbi_batch: ops_capacity %llu < %llu", static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(ops_total));
    const bool in_pinned = is_pinned_host(a_cat) && is_pinned_host(b_cat);
    const bool out_pinned = (ops == nullptr || is_pinned_host(ops)) && (scores == nullptr || is_pinned_host(scores)) &&
                            (ops_off == nullptr || is_pinned_host(ops_off)) && (ops_len == nullptr || is_pinned_host(ops_len));
    // ---- which form: ONE persistent kernel fed chunk by chunk (viterbi_batch_stream) for many pairs of viterbi_ck's
    // kind (the planner's rule: not short pairs, not lone long ones; and no pair whose own checkpoints would not fit
    // a stream slot's workspace); else a launch per chunk (below).  COATI_HIP_PIPE=chunks|stream forces one.
    {
        const char* pipe_env = std::getenv("COATI_HIP_PIPE");
        bool streamed = gap_len == 1 && std::getenv("COATI_HIP_VITERBI_BITS") == nullptr && std::getenv("COATI_HIP_FORCE_GENERIC") == nullptr &&
                        !(pipe_env != nullptr && std::strcmp(pipe_env, "chunks") == 0) && widest <= 8 * kStrip && max_pair_cells <= kStreamPairCells;
        if(streamed && !(pipe_env != nullptr && std::strcmp(pipe_env, "stream") == 0))
            streamed = n_pairs >= 4096 && total_cells / n_pairs >= 250.0L * 250.0L;
        // the persistent kernel owns the GPU for the length of the call: not when the embedder said no
        // (coati_hip_model_set_option), not where it failed before, and not where copies are done by kernels
        if(streamed && (model->stream_forbidden || model->stream_unusable)) streamed = false;
        if(streamed) {
            const char* sdma = std::getenv("HSA_ENABLE_SDMA");
            if(sdma != nullptr && std::atoi(sdma) == 0) streamed = false;
        }
        // every pair must fit a stream slot on its own (the chunk cutter takes the first pair of a chunk unseen).
        // Ordinary pairs pass by two comparisons; the few long or wide ones are priced exactly.
        if(streamed && (longest_a > 32768 || widest > static_cast<uint64_t>(kStrip))) {
            for(uint64_t p = 0; p < n_pairs && streamed; ++p) {
                const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
                if((la > 32768 || lb > static_cast<uint64_t>(kStrip)) && !stream_pair_fits(la, lb, gap_len, in_pinned, out_pinned)) streamed = false;
            }
        }
        if(streamed) {
            const int rc_stream = viterbi_batch_stream(model, n_pairs, a_cat, a_off, b_cat, b_off, scores, ops, ops_off, ops_len, in_pinned,
                                                       out_pinned, total_cells, longest_single, t_entry);
            if(rc_stream != COATI_HIP_ESTATE) return rc_stream;  // (ESTATE: nothing was started; the chunk pipeline takes the call)
        }
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    {   // cached workspaces of this model count as free: they are reused or released on demand
        std::lock_guard<std::mutex> hold(model->arena_lock);
        for(const auto& a : model->free_arenas) free_b += a.bytes;
    }
    for(const auto& sl : model->slots) free_b += sl.arena_bytes;
    constexpr int kSlots = coati_hip_model::kSlots;
    // per-slot workspace budget: a third of 80 % of the free HBM, at most 16 GB (~14 000 pairs of 1 kb:
    // larger chunks gain nothing, the kernel is at its steady rate from ~10 000 pairs)
    uint64_t budget = std::min<uint64_t>(static_cast<uint64_t>(free_b * 0.8) / kSlots, 16ull << 30);
    if(const char* e = std::getenv("COATI_HIP_MEM_BUDGET")) {  // tests: force chunking with a small budget (bytes)
        const uint64_t forced = std::strtoull(e, nullptr, 10);
        if(forced > 0) budget = std::min(budget, forced);
    }
    // ---- chunk schedule.  Full chunks hold ~1.6e10 cells (16 000 pairs of 1 kb) or what the budget
    // allows; the first one is a sixth of that (planning it takes ~0.4 ms, then the GPU has work while the next is planned).
    constexpr uint64_t kFullCells = 16000ull * 1002 * 1002;
    std::vector<PipeChunk> chunks;
    ChunkNeed max_need;
    uint64_t max_arena = 0;
    {
        // targets: a sixth and a third of a full chunk to get the GPU going while the next chunks are planned
        // and uploaded, then equal chunks of at most kFullCells, the last of them cut 2:1 (the smaller part
        // fills the ragged end of the larger and its download, the only exposed one, is short)
        std::vector<uint64_t> targets;
        {
            const long double total = total_cells;
            long double left = total;
            for(const uint64_t ramp : {kFullCells / 6, kFullCells / 3}) {
                if(left <= 0) break;
                targets.push_back(ramp);
                left -= static_cast<long double>(ramp);
            }
            if(left > 0) {
                const uint64_t parts = static_cast<uint64_t>(left / kFullCells) + 1;
                const uint64_t each = static_cast<uint64_t>(left / parts) + 2 * 1002 * 1002;
                for(uint64_t q = 0; q + 1 < parts; ++q) targets.push_back(each);
                targets.push_back(each * 2 / 3);
                targets.push_back(each);  // (what is left)
            }
        }
        uint64_t p0 = 0, ops_base = 0;
        while(p0 < n_pairs) {
            const uint64_t target = chunks.size() < targets.size() ? targets[chunks.size()] : kFullCells;
            ChunkNeed nd;
            // per-batch fixed parts of the workspace: the traceback scratch of the persistent wavefronts
            // (viterbi_ck), queue words, alignment slack of the ~15 carved arrays
            nd.fixed = static_cast<uint64_t>(ck_scratch_waves()) * ck_scratch_dwords_per_wave() * sizeof(uint32_t) + (64u << 10);
            uint64_t p1 = p0;
            while(p1 < n_pairs) {
                const uint64_t la = a_off[p1 + 1] - a_off[p1], lb = b_off[p1 + 1] - b_off[p1];
                ChunkNeed with = nd;
                chunk_need_add(with, la, lb, gap_len);
                if(p1 > p0 && (with.arena() > budget || with.cells > target)) break;
                nd = with;
                ++p1;
            }
            chunks.push_back(PipeChunk{p0, p1, ops_base, nd.ops});
            max_arena = std::max(max_arena, nd.arena());
            max_need.seq_bytes = std::max(max_need.seq_bytes, nd.seq_bytes);
            max_need.meta_bytes = std::max(max_need.meta_bytes, nd.meta_bytes);
            ops_base += nd.ops;
            p0 = p1;
        }
    }
    // ---- slots: stream, staging, workspace (grown on demand, kept by the model)
    const int n_slots = static_cast<int>(std::min<uint64_t>(kSlots, chunks.size()));
    uint64_t max_pairs = 0;
    for(const PipeChunk& c : chunks) max_pairs = std::max(max_pairs, c.p1 - c.p0);
    // staging block of a slot: [descriptors + (pageable) sequences going up | (pageable) results coming back]
    auto out_bytes_of = [&](uint64_t n, uint64_t ops_bytes) {
        return 4 * 256 + 2 * kMinDmaBytes + n * (sizeof(float) + sizeof(uint64_t) + sizeof(uint32_t)) + (out_pinned ? uint64_t{0} : ops_bytes);
    };
    // (sequences: a chunk with short ones stages them even when the caller's arrays are page-locked)
    const uint64_t staging_need = max_need.meta_bytes + 8 * 256 + 2 * kMinDmaBytes +
                                  (in_pinned ? std::min<uint64_t>(max_need.seq_bytes, 2 * kMinDmaBytes) + 512 : max_need.seq_bytes + 512) +
                                  out_bytes_of(max_pairs, max_need.seq_bytes) + 512;
    for(int q = 0; q < n_slots; ++q) {
        coati_hip_model::Slot& sl = model->slots[q];
        // (slot 0 runs on the model's own stream: HIP multiplexes its streams onto a handful of hardware queues --
        // four by default -- and two slots that share one queue run strictly one after the other)
        if(sl.stream == nullptr) {
            if(q == 0)
                sl.stream = model->stream;
            else
                HIP_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        }
        if(sl.pinned_bytes < staging_need) {
            if(sl.pinned != nullptr) (void)hipHostFree(sl.pinned);
            sl.pinned = nullptr;
            sl.pinned_bytes = 0;
            HIP_TRY(hipHostMalloc(&sl.pinned, staging_need, hipHostMallocDefault));
            sl.pinned_bytes = staging_need;
        }
        if(sl.arena_bytes < max_arena) {
            HIP_TRY(hipStreamSynchronize(sl.stream));
            if(sl.arena != nullptr) (void)hipFree(sl.arena);
            sl.arena = nullptr;
            sl.arena_bytes = 0;
            hipError_t e = hipMalloc(&sl.arena, max_arena);
            if(e == hipErrorOutOfMemory) {  // give the model's cached blocks back and try again
                (void)hipGetLastError();
                std::vector<coati_hip_model::Arena> drop;
                {
                    std::lock_guard<std::mutex> hold(model->arena_lock);
                    drop.swap(model->free_arenas);
                }
                for(const auto& a : drop) (void)hipFree(a.ptr);
                e = hipMalloc(&sl.arena, max_arena);
            }
            if(e != hipSuccess)
                return fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "viterbi_batch: hipMalloc(%llu bytes of workspace): %s",
                            static_cast<unsigned long long>(max_arena), hipGetErrorString(e));
            sl.arena_bytes = max_arena;
        }
    }
    // ---- the pipeline
    struct InFlight {
        coati_hip_batch_t* batch = nullptr;
        const PipeChunk* chunk = nullptr;
        hipEvent_t kernel_done = nullptr, copied = nullptr;
        bool d2h_submitted = false;
        char* out_stage = nullptr;  // results in the slot's staging block (pageable destinations)
        uint64_t out_off = 0;
        int slot = 0;
    };
    InFlight fl[kSlots];
    int rc = COATI_HIP_OK;
    const bool pipe_timing = std::getenv("COATI_HIP_PIPE_TIMING") != nullptr;  // timeline of the call on stderr
    static const bool no_d2h = std::getenv("COATI_HIP_PIPE_NO_D2H") != nullptr;   // (timing experiment: results stay on the device)
    const auto t_call = std::chrono::steady_clock::now();
    auto t_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    hipEvent_t ev_base = nullptr;
    if(pipe_timing) {
        HIP_TRY(hipEventCreate(&ev_base));
        HIP_TRY(hipEventRecord(ev_base, model->slots[0].stream));
    }
    // The download of a chunk's results is only SUBMITTED once its kernel has finished: a copy that waits
    // for a kernel sits at the head of the copy engine's queue and holds up the uploads of the following
    // chunks behind it (measured: their kernels then started only after the waiting chunk's kernel had ended).
    auto submit_d2h = [&](InFlight& f) -> hipError_t {
        f.d2h_submitted = true;
        coati_hip_model::Slot& sl = model->slots[f.slot];
        const PipeChunk& c = *f.chunk;
        const uint64_t n = c.p1 - c.p0;
        coati_hip_batch* b = f.batch;
        hipError_t e = hipSuccess;
        if(no_d2h) {
            f.out_stage = nullptr;
        } else if(out_pinned) {
            f.out_stage = nullptr;
            if(scores != nullptr) e = hipMemcpyAsync(scores + c.p0, b->d_scores, n * sizeof(float), hipMemcpyDeviceToHost, sl.stream);
            if(e == hipSuccess && ops_off != nullptr) e = hipMemcpyAsync(ops_off + c.p0, b->d_ops_start, n * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream);
            if(e == hipSuccess && ops_len != nullptr) e = hipMemcpyAsync(ops_len + c.p0, b->d_ops_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.stream);
            if(e == hipSuccess && ops != nullptr && c.ops_bytes > 0) e = hipMemcpyAsync(ops + c.ops_base, b->d_ops, c.ops_bytes, hipMemcpyDeviceToHost, sl.stream);
        } else {
            f.out_stage = static_cast<char*>(sl.pinned) + f.out_off;
            char* at = f.out_stage;
            if(scores != nullptr) e = hipMemcpyAsync(at, b->d_scores, n * sizeof(float), hipMemcpyDeviceToHost, sl.stream);
            at += (n * sizeof(float) + 255) / 256 * 256;
            if(e == hipSuccess && ops_off != nullptr) e = hipMemcpyAsync(at, b->d_ops_start, n * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream);
            at += (n * sizeof(uint64_t) + 255) / 256 * 256;
            if(e == hipSuccess && ops_len != nullptr) e = hipMemcpyAsync(at, b->d_ops_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost, sl.stream);
            at += (n * sizeof(uint32_t) + 255) / 256 * 256;
            if(e == hipSuccess && ops != nullptr && c.ops_bytes > 0) e = hipMemcpyAsync(at, b->d_ops, c.ops_bytes, hipMemcpyDeviceToHost, sl.stream);
        }
        if(e == hipSuccess && f.copied == nullptr) e = hipEventCreateWithFlags(&f.copied, hipEventDisableTiming);
        if(e == hipSuccess) e = hipEventRecord(f.copied, sl.stream);
        return e;
    };
    // submit the download of every chunk whose kernel has finished by now (never blocks)
    auto drain_ready = [&]() -> int {
        for(InFlight& f : fl) {
            if(f.batch == nullptr || f.d2h_submitted) continue;
            const hipError_t q = hipEventQuery(f.kernel_done);
            if(q == hipErrorNotReady) continue;
            hipError_t e = q;
            if(e == hipSuccess) e = submit_d2h(f);
            if(e != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
        }
        return COATI_HIP_OK;
    };
    // wait for a slot's chunk, hand its results to the caller, free the slot
    auto finish = [&](InFlight& f) -> int {
        if(f.batch == nullptr) return COATI_HIP_OK;
        int r = COATI_HIP_OK;
        hipError_t e = hipSuccess;
        if(!f.d2h_submitted) {
            e = hipEventSynchronize(f.kernel_done);
            if(e == hipSuccess) e = submit_d2h(f);
        }
        if(e == hipSuccess) e = hipEventSynchronize(f.copied);
        if(e != hipSuccess) r = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
        if(pipe_timing && r == COATI_HIP_OK) {
            hipEvent_t* ev = f.batch->ev[(f.batch->n_launches - 1) % coati_hip_batch::kTimingRing];
            float k0 = 0, k1 = 0;
            (void)hipEventElapsedTime(&k0, ev_base, ev[0]);
            (void)hipEventElapsedTime(&k1, ev_base, ev[1]);
            std::fprintf(stderr, "viterbi_batch: chunk of %llu pairs: kernel on the GPU %.2f .. %.2f ms, results on the host at %.2f ms\n",
                         static_cast<unsigned long long>(f.chunk->p1 - f.chunk->p0), k0, k1, t_ms());
        }
        const PipeChunk& c = *f.chunk;
        const uint64_t n = c.p1 - c.p0;
        if(r == COATI_HIP_OK && f.out_stage != nullptr) {
            char* at = f.out_stage;
            if(scores != nullptr) std::memcpy(scores + c.p0, at, n * sizeof(float));
            at += (n * sizeof(float) + 255) / 256 * 256;
            if(ops_off != nullptr) std::memcpy(ops_off + c.p0, at, n * sizeof(uint64_t));
            at += (n * sizeof(uint64_t) + 255) / 256 * 256;
            if(ops_len != nullptr) std::memcpy(ops_len + c.p0, at, n * sizeof(uint32_t));
            at += (n * sizeof(uint32_t) + 255) / 256 * 256;
            if(ops != nullptr && c.ops_bytes > 0) std::memcpy(ops + c.ops_base, at, c.ops_bytes);
        }
        if(r == COATI_HIP_OK && ops_off != nullptr && !no_d2h)
            for(uint64_t p = c.p0; p < c.p1; ++p) ops_off[p] += c.ops_base;
        coati_hip_batch_destroy(f.batch);
        f.batch = nullptr;
        return r;
    };
    for(size_t ci = 0; ci < chunks.size() && rc == COATI_HIP_OK; ++ci) {
        const PipeChunk& c = chunks[ci];
        const double t_begin = t_ms();
        const int q = static_cast<int>(ci % static_cast<size_t>(n_slots));
        coati_hip_model::Slot& sl = model->slots[q];
        InFlight& f = fl[q];
        rc = drain_ready();
        if(rc == COATI_HIP_OK) rc = finish(f);
        if(rc != COATI_HIP_OK) break;
        const uint64_t n = c.p1 - c.p0;
        const uint64_t out_off = (sl.pinned_bytes - out_bytes_of(n, c.ops_bytes)) / 256 * 256;  // results land behind the uploads
        BatchOpts bo;
        bo.stream = sl.stream;
        bo.arena = sl.arena;
        bo.arena_bytes = sl.arena_bytes;
        bo.staging = static_cast<char*>(sl.pinned);
        bo.staging_bytes = out_off;
        bo.seqs_pinned = in_pinned;
        uint64_t plan_need = 0;
        bo.arena_need_out = &plan_need;
        rc = batch_create_impl(model, n, a_cat, a_off + c.p0, b_cat, b_off + c.p0, nullptr, &bo, &f.batch);
        if(rc == COATI_HIP_ENOMEM && plan_need > sl.arena_bytes) {
            // the estimate behind the slot's workspace was short of this chunk's plan: grow the slot, once
            HIP_TRY(hipStreamSynchronize(sl.stream));
            (void)hipFree(sl.arena);
            sl.arena = nullptr;
            sl.arena_bytes = 0;
            const uint64_t grown = plan_need + plan_need / 16;
            const hipError_t ge = hipMalloc(&sl.arena, grown);
            if(ge != hipSuccess) {
                rc = fail(ge == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "viterbi_batch: hipMalloc(%llu bytes of workspace): %s",
                          static_cast<unsigned long long>(grown), hipGetErrorString(ge));
                break;
            }
            sl.arena_bytes = grown;
            bo.arena = sl.arena;
            bo.arena_bytes = sl.arena_bytes;
            rc = batch_create_impl(model, n, a_cat, a_off + c.p0, b_cat, b_off + c.p0, nullptr, &bo, &f.batch);
        }
        if(rc != COATI_HIP_OK) break;
        f.chunk = &c;
        f.slot = q;
        f.out_off = out_off;
        f.d2h_submitted = false;
        rc = coati_hip_viterbi_launch(f.batch);
        if(rc != COATI_HIP_OK) break;
        hipError_t e = hipSuccess;
        if(f.kernel_done == nullptr) e = hipEventCreateWithFlags(&f.kernel_done, hipEventDisableTiming);
        if(e == hipSuccess) e = hipEventRecord(f.kernel_done, sl.stream);
        if(e != hipSuccess) rc = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
        if(rc == COATI_HIP_OK) rc = drain_ready();
        if(pipe_timing)
            std::fprintf(stderr, "viterbi_batch: chunk %zu (%llu pairs, slot %d) host work %.2f .. %.2f ms\n", ci,
                         static_cast<unsigned long long>(n), q, t_begin, t_ms());
    }
    if(pipe_timing) std::fprintf(stderr, "viterbi_batch: all chunks enqueued at %.2f ms\n", t_ms());
    // the rest in the order the kernels finish
    for(size_t k = 0; k < chunks.size() && k < static_cast<size_t>(n_slots); ++k) {
        const size_t ci = chunks.size() - std::min<size_t>(chunks.size(), static_cast<size_t>(n_slots)) + k;
        InFlight& f = fl[ci % static_cast<size_t>(n_slots)];
        const int r = finish(f);
        if(rc == COATI_HIP_OK) rc = r;
    }
    for(InFlight& f : fl) {
        if(f.batch != nullptr) {  // (only after an error above)
            (void)hipStreamSynchronize(model->slots[f.slot].stream);
            coati_hip_batch_destroy(f.batch);
            f.batch = nullptr;
        }
        if(f.kernel_done != nullptr) (void)hipEventDestroy(f.kernel_done);
        if(f.copied != nullptr) (void)hipEventDestroy(f.copied);
    }
    if(pipe_timing) std::fprintf(stderr, "viterbi_batch: done at %.2f ms\n", t_ms());
    if(ev_base != nullptr) (void)hipEventDestroy(ev_base);
    return rc;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "viterbi_batch: host allocation failed");
    } catch(const std::exception& ex) {
        return fail(COATI_HIP_EHIP, "viterbi_batch: %s", ex.what());
    }
}

int coati_hip_shard_bounds(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world, uint64_t* bounds) {
    if(world < 1 || bounds == nullptr || (n_pairs > 0 && (a_off == nullptr || b_off == nullptr)))
        return fail(COATI_HIP_EINVAL, "shard_bounds: bad argument");
    // weights as long double sums are overkill: cells fit 2^62 for any input the ABI accepts per pair,
    // but a sum over 2^32 pairs may not -- accumulate in unsigned __int128
    using u128 = unsigned __int128;
    u128 total = 0;
    for(uint64_t p = 0; p < n_pairs; ++p) {
        if(a_off[p + 1] < a_off[p] || b_off[p + 1] < b_off[p]) return fail(COATI_HIP_EINVAL, "shard_bounds: offsets of pair %llu decrease", static_cast<unsigned long long>(p));
        total += static_cast<u128>(a_off[p + 1] - a_off[p]) * (b_off[p + 1] - b_off[p]);
    }
    bounds[0] = 0;
    u128 run = 0;
    uint64_t p = 0;
    for(int r = 1; r < world; ++r) {
        // first index at which the cells of pairs [0, index) reach r/world of the total
        while(p < n_pairs && run * static_cast<u128>(world) < total * static_cast<u128>(r)) {
            run += static_cast<u128>(a_off[p + 1] - a_off[p]) * (b_off[p + 1] - b_off[p]);
            ++p;
        }
        bounds[r] = p;
    }
    bounds[world] = n_pairs;
    return COATI_HIP_OK;
}

/* Page-locked host memory for the arrays of coati_hip_viterbi_batch (inputs and outputs): copies to
 * and from such memory are asynchronous DMA transfers that overlap the kernels; pageable memory goes
 * through a staging copy. */
int coati_hip_host_alloc(uint64_t bytes, void** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "host_alloc: out is NULL");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    return COATI_HIP_OK;
}
void coati_hip_host_free(void* p) {
    if(p != nullptr) (void)hipHostFree(p);
}

int coati_hip_debug_viterbi_flags(coati_hip_batch_t* b, uint64_t pair, uint8_t* out, uint64_t capacity) {
    if(b == nullptr || out == nullptr) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: NULL argument");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "debug_viterbi_flags: nothing was launched");
    if(pair >= b->n_pairs) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: pair out of range");
    const uint64_t n = static_cast<uint64_t>(b->desc[pair].la) * b->desc[pair].lb;
    if(capacity < n) return fail(COATI_HIP_EINVAL, "debug_viterbi_flags: capacity too small");
    if(n == 0) return COATI_HIP_OK;
    int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    uint8_t* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), n));
    hipError_t e = hipSuccess;
    if(b->ck && b->desc[pair].flags_off == kCkWaveSlot) {
        // the pair's checkpoints lived in a wavefront's slot and are gone: run the pair again, alone, in a
        // batch that keeps them, and decode that
        const PairDesc& d = b->desc[pair];
        std::vector<uint8_t> ha(std::max<uint32_t>(d.la, 1)), hb(std::max<uint32_t>(d.lb, 1));
        e = hipMemcpy(ha.data(), b->d_a + d.a_off, d.la, hipMemcpyDeviceToHost);
        if(e == hipSuccess) e = hipMemcpy(hb.data(), b->d_b + d.b_off, d.lb, hipMemcpyDeviceToHost);
        if(e != hipSuccess) {
            (void)hipFree(d_out);
            return fail(COATI_HIP_EHIP, "debug_viterbi_flags: %s", hipGetErrorString(e));
        }
        const uint64_t ao[2] = {0, d.la}, bo[2] = {0, d.lb};
        const uint32_t ti = d.table;
        BatchOpts keep;
        keep.ck_per_pair = true;
        keep.force_w_main = d.v_wmain;
        coati_hip_batch_t* one = nullptr;
        int rc1 = batch_create_impl(b->model, 1, ha.data(), ao, hb.data(), bo, &ti, &keep, &one);
        if(rc1 == COATI_HIP_OK) rc1 = coati_hip_viterbi_launch(one);
        if(rc1 == COATI_HIP_OK) rc1 = coati_hip_debug_viterbi_flags(one, 0, out, capacity);
        if(one != nullptr) coati_hip_batch_destroy(one);
        (void)hipFree(d_out);
        return rc1;
    } else if(b->ck) {
        // no bits in memory: every tile recomputed from the checkpoints by the traceback's own routine
        constexpr uint32_t kWaves = 64;
        uint32_t* d_scratch = nullptr;
        e = hipMalloc(reinterpret_cast<void**>(&d_scratch), kWaves * ck_scratch_dwords_per_wave() * sizeof(uint32_t));
        if(e == hipSuccess) e = hipMemsetAsync(d_out, 0xff, n, b->model->stream);
        if(e == hipSuccess) e = launch_ck_all_flags(device_view(b), static_cast<uint32_t>(pair), d_scratch, kWaves, d_out, b->model->stream);
        if(e == hipSuccess) e = hipStreamSynchronize(b->model->stream);
        if(d_scratch != nullptr) (void)hipFree(d_scratch);
    } else {
        hipLaunchKernelGGL(decode_flags, dim3(static_cast<uint32_t>(std::min<uint64_t>((n + 255) / 256, 4096))),
                           dim3(256), 0, b->model->stream, b->d_desc, static_cast<uint32_t>(pair), b->d_flags, d_out);
        e = hipStreamSynchronize(b->model->stream);
    }
    if(e == hipSuccess) e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if(e != hipSuccess) return fail(COATI_HIP_EHIP, "debug_viterbi_flags: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}

}  // extern "C"
