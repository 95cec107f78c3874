// abi.hip -- host side of the C ABI of include/coati_hip.h: handles, launches, result transfer, debug exports.
// The planner is plan.hip, the one-shot pipelines pipeline.hip, the sampler's host loop sample_host.hip
// (shared declarations: abi_internal.hpp); the kernels live in viterbi_ck.hip, viterbi_l1.hip, viterbi_k.hip,
// forward_l1.hip, forward_k.hip, dp_generic.hip and sampleback.hip.
#include "abi_internal.hpp"
#include <map>

using namespace coati_hip_abi;


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
namespace {
std::atomic<const EnvOptions*> g_env{nullptr};
std::mutex g_env_lock;
const EnvOptions* read_env() {
    auto* o = new EnvOptions;  // (never freed: a reader may still hold the old one; a reload is a test-only event)
    auto set = [](const char* name) { return std::getenv(name) != nullptr; };
    auto zero = [](const char* name) {
        const char* e = std::getenv(name);
        return e != nullptr && e[0] == '0';
    };
    auto num = [](const char* name, long long dflt) {
        const char* e = std::getenv(name);
        return e != nullptr ? std::atoll(e) : dflt;
    };
    o->force_generic = set("COATI_HIP_FORCE_GENERIC");
    o->viterbi_bits = set("COATI_HIP_VITERBI_BITS");
    o->viterbi_ck = set("COATI_HIP_VITERBI_CK");
    o->l1_lp_off = zero("COATI_HIP_L1_LP");
    o->l1_progress = set("COATI_HIP_L1_PROGRESS");
    o->ck_per_pair = set("COATI_HIP_CK_PER_PAIR");
    o->stream_parts = static_cast<int>(num("COATI_HIP_STREAM_PARTS", -1));
    o->stream_tail_units_x10 = static_cast<int>(num("COATI_HIP_STREAM_TAIL_UNITS", 0));
    o->spec_host_rounds = set("COATI_HIP_SPEC_HOST_ROUNDS");
    if(const long long v = num("COATI_HIP_SAMPLE_BAND", 0); v >= 1 && v <= (1 << 24)) o->sample_band = static_cast<uint32_t>(v);
    o->stream_helpers = static_cast<int>(num("COATI_HIP_STREAM_HELPERS", 55));
    o->pipe_no_d2h = set("COATI_HIP_PIPE_NO_D2H");
    o->sample_sequential = set("COATI_HIP_SAMPLE_SEQUENTIAL");
    o->sample_table_off = zero("COATI_HIP_SAMPLE_TABLE");
    o->fwd_wide_build = set("COATI_HIP_FWD_WIDE_BUILD");
    o->lp_pairtab_off = zero("COATI_HIP_LP_PAIRTAB");
    o->lp3_off = zero("COATI_HIP_LP3");
    if(const char* e = std::getenv("COATI_HIP_LP_SPLICE")) o->lp_splice = std::strcmp(e, "miss") == 0 ? 2 : (e[0] == '0' ? 0 : 1);
    if(const char* e = std::getenv("COATI_HIP_CK_SPLICE")) o->ck_splice = std::strcmp(e, "miss") == 0 ? 2 : (std::strcmp(e, "nobridge") == 0 ? 3 : (e[0] == '0' ? 0 : 1));
    if(const char* e = std::getenv("COATI_HIP_FORWARD_FAST")) o->forward_fast = e[0] != '\0' && e[0] != '0';
    o->timing = set("COATI_HIP_TIMING");
    o->pipe_timing = set("COATI_HIP_PIPE_TIMING");
    if(const char* e = std::getenv("HSA_ENABLE_SDMA")) o->sdma_off = std::atoi(e) == 0;
    if(const char* e = std::getenv("COATI_HIP_PIPE")) o->pipe = std::strcmp(e, "chunks") == 0 ? 1 : (std::strcmp(e, "stream") == 0 ? 2 : 0);
    o->strip_w = static_cast<int>(num("COATI_HIP_STRIP_W", 0));
    o->fwd_w = static_cast<int>(num("COATI_HIP_FWD_W", 0));
    o->fwd_quad = static_cast<int>(num("COATI_HIP_FWD_QUAD", -1));
    o->fill_blocks_per_cu = static_cast<int>(num("COATI_HIP_FILL_BLOCKS_PER_CU", 0));
    o->lp_blocks_per_cu = static_cast<int>(num("COATI_HIP_LP_BLOCKS_PER_CU", 0));
    o->tail_pairs = num("COATI_HIP_TAIL_PAIRS", -1);
    if(const char* e = std::getenv("COATI_HIP_CK_BAND")) {
        const long x = std::atol(e);
        o->ck_band = x <= 0 ? kCkBandOff : static_cast<uint32_t>(x);
    }
    o->ck_debug = static_cast<uint32_t>(num("COATI_HIP_CK_DEBUG", 0));
    o->ck_walk_items = static_cast<int>(num("COATI_HIP_CK_WALK_ITEMS", -1));
    o->ck_fuse = static_cast<int>(num("COATI_HIP_CK_FUSE", -1));
    if(const char* e = std::getenv("COATI_HIP_CK_SPLIT")) {  // "pairs,parts[,t]": t = tapered parts (common.hpp: ck_part_cut)
        char* rest = nullptr;
        o->ck_split_set = true;
        o->ck_split_pairs = std::strtoull(e, &rest, 10);
        if(rest != nullptr && *rest == ',') o->ck_split_parts = std::strtoull(rest + 1, &rest, 10);
        o->ck_split_taper = rest != nullptr && rest[0] == ',' && rest[1] == 't';
        if(rest != nullptr && rest[0] == ',' && rest[1] == 's') o->ck_split_short_last = static_cast<uint32_t>(std::min<long>(7, std::max<long>(0, std::atol(rest + 2))));
    }
    if(const long long v = num("COATI_HIP_SPEC_CANDS", 0); v >= 1024 && v <= (1 << 22)) o->spec_cands = static_cast<uint32_t>(v);
    if(const char* e = std::getenv("COATI_HIP_SPEC_Z")) {
        const double v = std::atof(e);
        if(v >= 0.25 && v <= 10.0) o->spec_z = v;
    }
    if(const char* e = std::getenv("COATI_HIP_STREAM_UNIT")) {
        const long double v = std::strtold(e, nullptr);
        if(v >= 1.0L) o->stream_unit = v;
    }
    if(const char* e = std::getenv("COATI_HIP_MEM_BUDGET")) o->mem_budget = std::strtoull(e, nullptr, 10);
    return o;
}
}  // namespace
const EnvOptions& env_options() {
    const EnvOptions* o = g_env.load(std::memory_order_acquire);
    if(o == nullptr) {
        std::lock_guard<std::mutex> hold(g_env_lock);
        o = g_env.load(std::memory_order_acquire);
        if(o == nullptr) {
            o = read_env();
            g_env.store(o, std::memory_order_release);
        }
    }
    return *o;
}
bool forward_fast_math() { return env_options().forward_fast; }
uint32_t ck_band_setting() { return env_options().ck_band; }
}  // namespace coati_hip_detail

extern "C" void coati_hip_debug_reload_env(void) {
    std::lock_guard<std::mutex> hold(coati_hip_detail::g_env_lock);
    coati_hip_detail::g_env.store(coati_hip_detail::read_env(), std::memory_order_release);
}

namespace {
thread_local std::string g_error;
}
namespace coati_hip_abi {
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
}  // namespace coati_hip_abi

namespace {
bool device_is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

}  // namespace

namespace coati_hip_abi {
BatchDeviceView device_view(const coati_hip_batch* b) {
    const coati_hip_model* m = b->model;
    return BatchDeviceView{m->d_table,  m->k,      static_cast<uint32_t>(m->gap_len),
                           b->d_desc,   b->d_order, static_cast<uint32_t>(b->n_pairs),
                           b->d_queue,  b->d_items, b->n_items, b->d_fwd_items, b->n_fwd_items, b->d_progress, b->d_a,    b->d_b,
                           b->d_flags,  b->d_bnd,  b->bnd_floats * sizeof(float), b->d_scores,
                           b->d_ops,    b->d_ops_start, b->d_ops_len, b->d_wscratch, b->ck_slot_dwords, b->ck_split_items | (b->ck_walk_items ? kCkWalkItemsFlag : 0u),
                           b->d_mdi,    b->d_final_mdi, b->fwd_wlog2_max, b->ck_keep_all ? kCkBandOff : m->ck_band, b->long_pairs ? 1u : 0u, b->multi_strip ? 1u : 0u, b->fwd_quad ? 1u : 0u, b->fwd_fast ? 1u : 0u};
}


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
    // A fresh block is a size CLASS, not the exact need (up to an eighth above it, in steps of at least 32 MB): consecutive chunks of one job
    // differ by a few pairs, and a cached block that is a few KB short is a fresh multi-GB hipMalloc -- 0.25-0.5 s when
    // the driver has to find the pages (round 4: a 1 000 000-pair sharded job in 48 000-pair chunks took 12 s that way)
    // Small needs are taken as they are (rounded to 2 MB): a one-pair batch must not pin 32 MB of HBM, and a fresh small
    // hipMalloc is cheap.  And when the class does not fit, the EXACT need is tried before giving up -- a request close to
    // what is free must not fail because of its padding (the chunk pipeline budgets 0.8 of the free memory per slot set).
    const uint64_t exact = (need + (2ull << 20) - 1) / (2ull << 20) * (2ull << 20);
    if(need >= (256ull << 20)) {
        uint64_t step = 32ull << 20;
        while(step * 16 <= need) step *= 2;  // (a sixteenth to an eighth of the need)
        need = (need + need / 16 + step - 1) / step * step;  // (and a sixteenth of headroom: a need just above a class boundary must not start a new class)
    } else {
        need = exact;
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
        if(e == hipErrorOutOfMemory && exact < need) {
            (void)hipGetLastError();
            need = exact;
            e = hipMalloc(ptr, need);
        }
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

}  // namespace coati_hip_abi

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
    m->ck_band = ck_band_setting();
    m->n_tables = n_tables;
    m->forward_mode.store(forward_fast_math() ? COATI_HIP_FORWARD_TOLERANCE : COATI_HIP_FORWARD_EXACT);  // (the environment's default)
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
    m->helpers.reset();
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
        // (a batch that only ever ran Viterbi launches is idle when its LAST launch has ended -- its event; waiting for the
        // whole stream would wait for the next batch's kernel too, which is how the sharded job's chunks used to serialise)
        bool idle = false;
        if(m != nullptr && b->launched && !b->forward_done && b->n_launches > 0 && b->ev[(b->n_launches - 1) % coati_hip_batch::kTimingRing][1] != nullptr)
            idle = hipEventSynchronize(b->ev[(b->n_launches - 1) % coati_hip_batch::kTimingRing][1]) == hipSuccess;
        if(!idle) {
            idle = m != nullptr && hipStreamSynchronize(m->stream) == hipSuccess;
            if(idle && b->stream != nullptr && b->stream != m->stream) idle = hipStreamSynchronize(b->stream) == hipSuccess;
        }
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
    if(option == COATI_HIP_OPT_CK_BAND) {
        if(value < 0 || value > 0x7fffffff) return fail(COATI_HIP_EINVAL, "model_set_option: band of %lld steps", static_cast<long long>(value));
        model->ck_band = value == 0 ? kCkBandOff : static_cast<uint32_t>(value);
        return COATI_HIP_OK;
    }
    if(option == COATI_HIP_OPT_FORWARD_MODE) {
        if(value != COATI_HIP_FORWARD_EXACT && value != COATI_HIP_FORWARD_TOLERANCE)
            return fail(COATI_HIP_EINVAL, "model_set_option: Forward mode %lld", static_cast<long long>(value));
        model->forward_mode.store(static_cast<int>(value));
        return COATI_HIP_OK;
    }
    return fail(COATI_HIP_EINVAL, "model_set_option: unknown option %d", option);
}

int coati_hip_batch_create(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                           coati_hip_batch_t** out) {
    return coati_hip_batch_create_tables(model, n_pairs, a_cat, a_off, b_cat, b_off, nullptr, out);
}

uint64_t coati_hip_batch_pairs(const coati_hip_batch_t* b) { return b ? b->n_pairs : 0; }
uint64_t coati_hip_batch_device_bytes(const coati_hip_batch_t* b) { return b ? b->device_bytes : 0; }
uint64_t coati_hip_batch_cells(const coati_hip_batch_t* b) { return b ? b->cells : 0; }

int coati_hip_viterbi_launch(coati_hip_batch_t* b) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_launch: batch is NULL");
    coati_hip_model* m = b->model;
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t n = static_cast<uint32_t>(b->n_pairs);
    hipEvent_t* ev = b->ev[b->n_launches % coati_hip_batch::kTimingRing];
    for(int q = 0; q < 2; ++q)
        if(ev[q] == nullptr) HIP_TRY(hipEventCreate(&ev[q]));  // (created on first use: 128 events per batch cost 0.2 ms)
    HIP_TRY(hipEventRecord(ev[0], b->stream));
    if(n > 0) {
        const BatchDeviceView v = device_view(b);
        const bool force_generic = env_options().force_generic;
        if(b->ck)
            HIP_TRY(launch_viterbi_ck(v, m->n_tables == 1, b->stream));
        else if(m->gap_len == 1 && !force_generic)
            HIP_TRY(v.long_pairs != 0 ? launch_viterbi_lp(v, b->stream) : launch_viterbi_l1(v, b->stream));
        else if(b->compact)
            HIP_TRY(launch_viterbi_k(v, b->compact_narrow_only, b->stream));
        else
            HIP_TRY(launch_dp_generic(v, /*forward=*/false, b->stream));
    }
    HIP_TRY(hipEventRecord(ev[1], b->stream));  // (the traceback is fused into the fill kernel: one span per launch)
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
    HIP_TRY(hipEventSynchronize(b->ev[(b->n_launches - 1) % coati_hip_batch::kTimingRing][1]));
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
    HIP_TRY(hipEventElapsedTime(&f, ev[0], ev[1]));  // (w stays 0: the traceback is part of the fill kernel since round 1)
    if(fill_ms != nullptr) *fill_ms = f;
    if(walk_ms != nullptr) *walk_ms = w;
    return COATI_HIP_OK;
}

int coati_hip_viterbi_band_stats(coati_hip_batch_t* b, uint32_t* band_steps, uint64_t* pairs_refilled) {
    if(b == nullptr) return fail(COATI_HIP_EINVAL, "viterbi_band_stats: batch is NULL");
    if(!b->launched) return fail(COATI_HIP_ESTATE, "viterbi_band_stats: nothing was launched");
    const int rc = coati_hip_batch_sync(b);
    if(rc != COATI_HIP_OK) return rc;
    const bool banded = b->ck && !b->ck_keep_all && b->model->ck_band != kCkBandOff;
    uint32_t refilled = 0;
    if(banded && b->n_pairs > 0) HIP_TRY(hipMemcpy(&refilled, b->d_queue + 1, sizeof refilled, hipMemcpyDeviceToHost));  // (the word behind the ticket counter)
    if(band_steps != nullptr) *band_steps = banded ? b->model->ck_band : 0u;
    if(pairs_refilled != nullptr) *pairs_refilled = refilled;
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
        const bool force_generic = env_options().force_generic;
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
extern "C++" {
namespace {
// blocks handed out by coati_hip_host_alloc: asking the runtime whether a pointer is page-locked costs 5-10 us per array
// (hipPointerGetAttributes), six arrays per one-shot call; these are known without asking
std::mutex g_host_blocks_lock;
std::map<uintptr_t, uint64_t> g_host_blocks;  // start -> bytes
}  // namespace
namespace coati_hip_abi {
bool host_block_contains(const void* p) {
    const uintptr_t at = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> hold(g_host_blocks_lock);
    auto it = g_host_blocks.upper_bound(at);
    if(it == g_host_blocks.begin()) return false;
    --it;
    return at - it->first < it->second;
}
}  // namespace coati_hip_abi
}  // extern "C++"

int coati_hip_host_alloc(uint64_t bytes, void** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "host_alloc: out is NULL");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
    try {
        std::lock_guard<std::mutex> hold(g_host_blocks_lock);
        g_host_blocks[reinterpret_cast<uintptr_t>(*out)] = std::max<uint64_t>(bytes, 1);
    } catch(const std::bad_alloc&) {  // (not recorded: the runtime is asked about it instead)
    }
    return COATI_HIP_OK;
}
void coati_hip_host_free(void* p) {
    if(p == nullptr) return;
    {
        std::lock_guard<std::mutex> hold(g_host_blocks_lock);
        g_host_blocks.erase(reinterpret_cast<uintptr_t>(p));
    }
    (void)hipHostFree(p);
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
    if(b->ck && !b->ck_keep_all) {
        // the pair's checkpoints lived in a wavefront's slot and are gone, or -- own storage: multi-strip pairs, pairs cut
        // into row parts -- only a band of them was kept: run the pair again, alone, in a batch that keeps them all, and
        // decode that
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
