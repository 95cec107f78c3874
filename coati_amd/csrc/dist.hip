// dist.hip -- libcoati_hip_dist.so: the multi-GPU layer of include/coati_hip_dist.h.  One process per
// GPU; RCCL (librccl.so, linked directly) for the two real exchanges -- the model broadcast and the
// gather of results to the root; everything else goes through the public C ABI of libcoati_hip.so.
#include "coati_hip_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

static_assert(sizeof(ncclUniqueId) == COATI_HIP_DIST_ID_BYTES, "rendezvous id size");

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

#define D_HIP(expr)                                                                                            \
    do {                                                                                                       \
        const hipError_t e_ = (expr);                                                                          \
        if(e_ != hipSuccess)                                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                                \
    } while(0)
#define D_NCCL(expr)                                                                                        \
    do {                                                                                                    \
        const ncclResult_t r_ = (expr);                                                                     \
        if(r_ != ncclSuccess) return fail(COATI_HIP_EHIP, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
    } while(0)
#define D_ABI(expr)                                                               \
    do {                                                                          \
        const int rc_ = (expr);                                                   \
        if(rc_ != COATI_HIP_OK) return fail(rc_, "%s", coati_hip_last_error()); \
    } while(0)

constexpr uint64_t kTabFloats = COATI_HIP_TABLE_ROWS * COATI_HIP_TABLE_COLS;

}  // namespace

struct coati_hip_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int world = 1, rank = 0, device = 0;
    // device scratch: counts of one gather (3 words per rank: pairs, op bytes, status), and on the root the
    // landing zone of the peers' result arrays (grown on demand)
    uint64_t *d_mine = nullptr, *d_counts = nullptr;
    void* d_land = nullptr;
    uint64_t land_bytes = 0;
    double* d_reduce = nullptr;  // 64 doubles for coati_hip_dist_allreduce_f64 (no allocation inside a timed collective)
};

extern "C" {

const char* coati_hip_dist_last_error(void) { return g_error.c_str(); }

int coati_hip_dist_unique_id(void* id128) {
    if(id128 == nullptr) return fail(COATI_HIP_EINVAL, "dist_unique_id: NULL");
    ncclUniqueId id;
    D_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return COATI_HIP_OK;
}

void coati_hip_dist_destroy(coati_hip_comm_t* c) {
    if(c == nullptr) return;
    (void)hipSetDevice(c->device);
    if(c->stream != nullptr) (void)hipStreamSynchronize(c->stream);
    if(c->comm != nullptr) (void)ncclCommDestroy(c->comm);
    if(c->d_mine != nullptr) (void)hipFree(c->d_mine);
    if(c->d_counts != nullptr) (void)hipFree(c->d_counts);
    if(c->d_land != nullptr) (void)hipFree(c->d_land);
    if(c->d_reduce != nullptr) (void)hipFree(c->d_reduce);
    if(c->stream != nullptr) (void)hipStreamDestroy(c->stream);
    delete c;
}

int coati_hip_dist_init(const void* id128, int world, int rank, int device, coati_hip_comm_t** out) {
    if(out == nullptr) return fail(COATI_HIP_EINVAL, "dist_init: out is NULL");
    *out = nullptr;
    if(id128 == nullptr || world < 1 || rank < 0 || rank >= world) return fail(COATI_HIP_EINVAL, "dist_init: bad argument");
    if(coati_hip_device_count() == 0) return fail(COATI_HIP_ENODEVICE, "dist_init: no gfx950 device");
    auto* c = new(std::nothrow) coati_hip_comm;
    if(c == nullptr) return fail(COATI_HIP_ENOMEM, "dist_init: host allocation failed");
    c->world = world;
    c->rank = rank;
    c->device = device;
    struct Guard {
        coati_hip_comm* c;
        ~Guard() {
            if(c != nullptr) coati_hip_dist_destroy(c);
        }
    } guard{c};
    D_HIP(hipSetDevice(device));
    D_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    D_NCCL(ncclCommInitRank(&c->comm, world, id, rank));
    D_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_mine), 4 * sizeof(uint64_t)));  // (pairs, op bytes, status)
    D_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_counts), 4 * sizeof(uint64_t) * static_cast<size_t>(world)));
    D_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_reduce), 64 * sizeof(double)));
    guard.c = nullptr;
    *out = c;
    return COATI_HIP_OK;
}

int coati_hip_dist_rank(const coati_hip_comm_t* c) { return c != nullptr ? c->rank : -1; }
int coati_hip_dist_world(const coati_hip_comm_t* c) { return c != nullptr ? c->world : 0; }

int coati_hip_dist_allreduce_f64(coati_hip_comm_t* c, int op, double* values, uint32_t n) {
    if(c == nullptr || values == nullptr || n == 0 || n > 64 || (op != 0 && op != 1)) return fail(COATI_HIP_EINVAL, "dist_allreduce_f64: bad argument");
    D_HIP(hipSetDevice(c->device));
    double* const d = c->d_reduce;
    D_HIP(hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    D_NCCL(ncclAllReduce(d, d, n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->stream));
    D_HIP(hipMemcpyAsync(values, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    D_HIP(hipStreamSynchronize(c->stream));
    return COATI_HIP_OK;
}

int coati_hip_dist_barrier(coati_hip_comm_t* c) {
    double one = 1.0;
    return coati_hip_dist_allreduce_f64(c, 0, &one, 1);
}

int coati_hip_dist_broadcast_model(coati_hip_comm_t* c, int root, float* tables, uint32_t table_capacity_tables,
                                   uint32_t* n_tables, float consts[4], int* gap_len) {
    if(c == nullptr || tables == nullptr || n_tables == nullptr || consts == nullptr || gap_len == nullptr || root < 0 || root >= c->world)
        return fail(COATI_HIP_EINVAL, "dist_broadcast_model: bad argument");
    D_HIP(hipSetDevice(c->device));
    // header: n_tables, gap_len, the four constants as their bit patterns
    uint32_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if(c->rank == root) {
        head[0] = *n_tables;
        head[1] = static_cast<uint32_t>(*gap_len);
        std::memcpy(head + 2, consts, 4 * sizeof(float));
    }
    uint32_t* d_head = nullptr;
    D_HIP(hipMalloc(reinterpret_cast<void**>(&d_head), sizeof head));
    struct Free {
        void* p;
        ~Free() { (void)hipFree(p); }
    } free_head{d_head};
    D_HIP(hipMemcpyAsync(d_head, head, sizeof head, hipMemcpyHostToDevice, c->stream));
    D_NCCL(ncclBroadcast(d_head, d_head, 8, ncclUint32, root, c->comm, c->stream));
    D_HIP(hipMemcpyAsync(head, d_head, sizeof head, hipMemcpyDeviceToHost, c->stream));
    D_HIP(hipStreamSynchronize(c->stream));
    const uint32_t nt = head[0];
    if(nt < 1 || nt > 65535) return fail(COATI_HIP_EINVAL, "dist_broadcast_model: root announced %u tables", nt);
    if(c->rank != root && nt > table_capacity_tables)
        return fail(COATI_HIP_EINVAL, "dist_broadcast_model: %u tables do not fit the buffer of %u", nt, table_capacity_tables);
    float* d_tab = nullptr;
    D_HIP(hipMalloc(reinterpret_cast<void**>(&d_tab), nt * kTabFloats * sizeof(float)));
    Free free_tab{d_tab};
    if(c->rank == root) D_HIP(hipMemcpyAsync(d_tab, tables, nt * kTabFloats * sizeof(float), hipMemcpyHostToDevice, c->stream));
    D_NCCL(ncclBroadcast(d_tab, d_tab, nt * kTabFloats, ncclFloat32, root, c->comm, c->stream));
    if(c->rank != root) {
        D_HIP(hipMemcpyAsync(tables, d_tab, nt * kTabFloats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        *n_tables = nt;
        *gap_len = static_cast<int>(head[1]);
        std::memcpy(consts, head + 2, 4 * sizeof(float));
    }
    D_HIP(hipStreamSynchronize(c->stream));
    return COATI_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// Pure host arithmetic of the gather and of the sharded job (no HIP, no RCCL): which blocks travel, where
// they land in the root's HBM, where they end up in the caller's arrays.  The per-rank job loop below is a
// template over its ENVIRONMENT -- RCCL + HIP in production; host memory with an in-process transport
// (coati_hip_dist_simulate: every rank a thread) or with the caller's transport (coati_hip_dist_job_host: every
// rank a process, e.g. gloo in the CPU tests) -- so the multi-rank control flow that runs on the GPUs is the code
// the CPU tests run; what only hardware can exercise is the literal ncclSend / ncclRecv / hipMemcpyAsync calls.
// ---------------------------------------------------------------------------------------------------
}  // extern "C"
namespace {

constexpr int kCountWords = 3;  // per rank in the count exchange: pairs, op bytes, status (COATI_HIP_OK or the rank's error code)
// which result arrays of a block a gather moves to the root
enum : uint32_t { kScores = 1u, kOff = 2u, kLen = 4u, kOps = 8u, kAllArrays = 15u, kSummary = kScores | kLen };
// A rank's shard goes through the gather-all job in chunks of this many DP cells (48 000 pairs of 1 kb, ~17 ms of
// kernel): the kernel of chunk k + 1 hides the planning and upload of chunk k + 2 and the gather of chunk k, and a
// launch of this size loses little to its ragged end.  Two chunks are resident at a time (~2 x 14 GB).
constexpr uint64_t kDefaultChunkCells = 48000ull * 1002 * 1002;

struct Land {
    uint64_t scores, ops, off, len;  // byte offsets in the root's landing zone
};
// landing zone of the peers' blocks on the root (rank order, every array 256-byte aligned); returns its size
uint64_t landing_plan(int world, int root, const uint64_t* counts /* kCountWords per rank */, uint32_t mask, Land* land) {
    uint64_t need = 0;
    auto take = [&](uint64_t bytes, uint32_t which) {
        const uint64_t at = need;
        if(mask & which) need += (bytes + 255) / 256 * 256;
        return at;
    };
    for(int r = 0; r < world; ++r) {
        if(r == root) {
            land[r] = Land{0, 0, 0, 0};
            continue;
        }
        const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
        land[r].scores = take(n * sizeof(float), kScores);
        land[r].ops = take(ob, kOps);
        land[r].off = take(n * sizeof(uint64_t), kOff);
        land[r].len = take(n * sizeof(uint32_t), kLen);
    }
    return need;
}

// One message of the send/receive group: `count` elements of `bytes_each` from the peer's result array `which`
// (0 scores, 1 op offsets, 2 op lengths, 3 ops) to byte offset `at` of the landing zone.  Sender and receiver
// derive the same list (the sender its own entries), in the same order -- RCCL matches point-to-point calls of a
// pair of ranks in the order they are issued.
struct Transfer {
    int peer, which;
    uint64_t at, count;
    uint32_t bytes_each;
};
void transfers_of_rank(int r, const uint64_t* counts, const Land& l, uint32_t mask, std::vector<Transfer>& out) {
    const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
    if(n > 0) {
        if(mask & kScores) out.push_back(Transfer{r, 0, l.scores, n, static_cast<uint32_t>(sizeof(float))});
        if(mask & kOff) out.push_back(Transfer{r, 1, l.off, n, static_cast<uint32_t>(sizeof(uint64_t))});
        if(mask & kLen) out.push_back(Transfer{r, 2, l.len, n, static_cast<uint32_t>(sizeof(uint32_t))});
    }
    if(ob > 0 && (mask & kOps)) out.push_back(Transfer{r, 3, l.ops, ob, 1u});
}

// A rank's result block as four arrays (device pointers in the collective, host pointers in the host environments)
struct Block {
    const void *scores = nullptr, *ops = nullptr, *off = nullptr, *len = nullptr;
};
const void* block_array(const Block& b, int which) { return which == 0 ? b.scores : which == 1 ? b.off : which == 2 ? b.len : b.ops; }

// Where rank r's block goes in the root's arrays: its first entry at index pair0, its op bytes at op0.  The plain
// gather concatenates in rank order (prefix sums of the counts); the sharded job places every chunk where its
// pairs are in the INPUT order, straight from the landing zone -- no staging copy on the host.
struct Place {
    uint64_t pair0, op0;
};
void prefix_places(int world, const uint64_t* counts, Place* place) {
    uint64_t pair0 = 0, op0 = 0;
    for(int r = 0; r < world; ++r) {
        place[r] = Place{pair0, op0};
        pair0 += counts[kCountWords * r];
        op0 += counts[kCountWords * r + 1];
    }
}
// Root, after the exchange: the blocks (its own arrays, the landing zone for the peers) go to the caller's arrays.
// `copy` moves bytes (hipMemcpyAsync device->host in the collective, memcpy in the host environments).
template <typename Copy>
int unpack_blocks(int world, int root, const uint64_t* counts, uint32_t mask, const Place* place, const Block& own, const char* landing,
                  const Land* land, float* scores, uint8_t* ops, uint64_t* ops_off, uint32_t* ops_len, Copy&& copy) {
    for(int r = 0; r < world; ++r) {
        const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
        Block b = own;
        if(r != root) b = Block{landing + land[r].scores, landing + land[r].ops, landing + land[r].off, landing + land[r].len};
        const uint64_t pair0 = place[r].pair0, op0 = place[r].op0;
        if(n > 0) {
            if(scores != nullptr && (mask & kScores)) { const int rc = copy(scores + pair0, b.scores, n * sizeof(float)); if(rc != COATI_HIP_OK) return rc; }
            if(ops_off != nullptr && (mask & kOff)) { const int rc = copy(ops_off + pair0, b.off, n * sizeof(uint64_t)); if(rc != COATI_HIP_OK) return rc; }
            if(ops_len != nullptr && (mask & kLen)) { const int rc = copy(ops_len + pair0, b.len, n * sizeof(uint32_t)); if(rc != COATI_HIP_OK) return rc; }
        }
        if(ob > 0 && ops != nullptr && (mask & kOps)) { const int rc = copy(ops + op0, b.ops, ob); if(rc != COATI_HIP_OK) return rc; }
    }
    return COATI_HIP_OK;
}
// every rank's op offsets index its own ops array: rebase them to where its op bytes went
void rebase_offsets(int world, const uint64_t* counts, const Place* place, uint64_t* ops_off) {
    for(int r = 0; r < world; ++r)
        for(uint64_t p = 0; p < counts[kCountWords * r]; ++p) ops_off[place[r].pair0 + p] += place[r].op0;
}
// first failed rank of a round, or -1
int failed_rank(int world, const uint64_t* counts) {
    for(int r = 0; r < world; ++r)
        if(counts[kCountWords * r + 2] != static_cast<uint64_t>(COATI_HIP_OK)) return r;
    return -1;
}

// The sharded job's plan, identical on every rank (all hold the same lengths): shard bounds, every rank's
// chunk boundaries (a rank's shard in chunks of at most chunk_cells cells, at least one pair each), the number
// of gather rounds, and the op-byte prefix that says where a pair's ops go in the root's array.
struct JobPlan {
    std::vector<uint64_t> bounds;             // world + 1
    std::vector<std::vector<uint64_t>> cuts;  // per rank: chunk boundaries (pair indices), >= 2 entries
    std::vector<uint64_t> op_prefix;          // n_pairs + 1
    size_t rounds = 0;
};
int make_job_plan(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world, uint64_t chunk_cells, JobPlan& plan) {
    plan.bounds.assign(static_cast<size_t>(world) + 1, 0);
    const int rc = coati_hip_shard_bounds(n_pairs, a_off, b_off, world, plan.bounds.data());
    if(rc != COATI_HIP_OK) return fail(rc, "%s", coati_hip_last_error());
    plan.cuts.assign(static_cast<size_t>(world), {});
    plan.rounds = 0;
    for(int r = 0; r < world; ++r) {
        auto& cut = plan.cuts[static_cast<size_t>(r)];
        const uint64_t end = plan.bounds[static_cast<size_t>(r) + 1];
        uint64_t p = plan.bounds[static_cast<size_t>(r)], cells = 0;
        cut.push_back(p);
        for(; p < end; ++p) {
            const uint64_t w = (a_off[p + 1] - a_off[p]) * (b_off[p + 1] - b_off[p]);
            if(p > cut.back() && cells + w > chunk_cells) {
                cut.push_back(p);
                cells = 0;
            }
            cells += w;
        }
        if(cut.back() != end || cut.size() == 1) cut.push_back(end);
        plan.rounds = std::max(plan.rounds, cut.size() - 1);
    }
    plan.op_prefix.assign(n_pairs + 1, 0);
    for(uint64_t p = 0; p < n_pairs; ++p) plan.op_prefix[p + 1] = plan.op_prefix[p] + (a_off[p + 1] - a_off[p]) + (b_off[p + 1] - b_off[p]);
    return COATI_HIP_OK;
}
// what rank r contributes to round k according to the plan: pairs [p0, p0 + n), nb op bytes
void plan_block(const JobPlan& plan, int r, size_t k, uint64_t& p0, uint64_t& n, uint64_t& nb) {
    const auto& cut = plan.cuts[static_cast<size_t>(r)];
    p0 = n = nb = 0;
    if(k + 1 >= cut.size()) return;
    p0 = cut[k];
    n = cut[k + 1] - cut[k];
    nb = plan.op_prefix[cut[k + 1]] - plan.op_prefix[cut[k]];
}
// every rank checks the gathered counts of a round against the plan BEFORE anything is moved or unpacked (all
// hold the same counts and the same plan: the same verdict everywhere, and all leave together on a mismatch)
int check_round(const JobPlan& plan, int world, size_t k, const uint64_t* counts) {
    for(int r = 0; r < world; ++r) {
        uint64_t p0, n, nb;
        plan_block(plan, r, k, p0, n, nb);
        if(counts[kCountWords * r] != n || counts[kCountWords * r + 1] != nb)
            return fail(COATI_HIP_ESTATE, "dist_viterbi: rank %d sent %llu pairs / %llu op bytes in round %zu, the plan says %llu / %llu", r,
                        static_cast<unsigned long long>(counts[kCountWords * r]), static_cast<unsigned long long>(counts[kCountWords * r + 1]), k,
                        static_cast<unsigned long long>(n), static_cast<unsigned long long>(nb));
    }
    return COATI_HIP_OK;
}
// where the blocks of round k go in the root's arrays: a chunk's pairs keep their input positions
void round_places(const JobPlan& plan, int world, size_t k, Place* place) {
    for(int r = 0; r < world; ++r) {
        uint64_t p0, n, nb;
        plan_block(plan, r, k, p0, n, nb);
        place[r] = Place{p0, plan.op_prefix[p0]};
    }
}
// the largest landing zone any round of the job needs (the root reserves it BEFORE the first round, and says so
// in its status word if it cannot: no rank is ever left sending to a root that has gone)
uint64_t job_landing_need(const JobPlan& plan, int world, int root, uint32_t mask) {
    std::vector<uint64_t> counts(static_cast<size_t>(kCountWords) * static_cast<size_t>(world), 0);
    std::vector<Land> land(static_cast<size_t>(world));
    uint64_t need = 0;
    for(size_t k = 0; k < plan.rounds; ++k) {
        for(int r = 0; r < world; ++r) {
            uint64_t p0, n, nb;
            plan_block(plan, r, k, p0, n, nb);
            counts[static_cast<size_t>(kCountWords) * r] = n, counts[static_cast<size_t>(kCountWords) * r + 1] = nb;
        }
        need = std::max(need, landing_plan(world, root, counts.data(), mask, land.data()));
    }
    return need;
}

// ---------------------------------------------------------------------------------------------------
// Environments.  What a rank's side of a round needs from its surroundings:
//   exchange_counts   all ranks learn all ranks' (pairs, op bytes, status)
//   landing_reserve   root: room for the peers' blocks;  landing(): its base
//   transfer          root: receive every listed block into the landing zone; peer: send its listed arrays
//   copy_out / sync   results to the caller's arrays (device -> host, or memcpy)
// ---------------------------------------------------------------------------------------------------
struct RcclEnv {
    coati_hip_comm_t* c;
    int world() const { return c->world; }
    int rank() const { return c->rank; }
    int begin() {
        D_HIP(hipSetDevice(c->device));
        return COATI_HIP_OK;
    }
    int exchange_counts(const uint64_t* mine, uint64_t* all) {
        D_HIP(hipMemcpyAsync(c->d_mine, mine, kCountWords * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        D_NCCL(ncclAllGather(c->d_mine, c->d_counts, kCountWords, ncclUint64, c->comm, c->stream));
        D_HIP(hipMemcpyAsync(all, c->d_counts, kCountWords * sizeof(uint64_t) * static_cast<size_t>(c->world), hipMemcpyDeviceToHost, c->stream));
        D_HIP(hipStreamSynchronize(c->stream));
        return COATI_HIP_OK;
    }
    int landing_reserve(uint64_t need) {
        if(need <= c->land_bytes) return COATI_HIP_OK;
        if(c->d_land != nullptr) (void)hipFree(c->d_land);
        c->d_land = nullptr;
        c->land_bytes = 0;
        D_HIP(hipMalloc(&c->d_land, need));
        c->land_bytes = need;
        return COATI_HIP_OK;
    }
    char* landing() const { return static_cast<char*>(c->d_land); }
    int transfer(int root, const std::vector<Transfer>& xfer, const Block& own) {
        char* base = landing();
        ncclResult_t first = ncclSuccess;
        D_NCCL(ncclGroupStart());
        for(const Transfer& t : xfer) {  // (never return between GroupStart and GroupEnd: the group would stay open)
            const uint64_t bytes = t.count * t.bytes_each;
            const ncclResult_t r = c->rank == root ? ncclRecv(base + t.at, bytes, ncclUint8, t.peer, c->comm, c->stream)
                                                   : ncclSend(block_array(own, t.which), bytes, ncclUint8, root, c->comm, c->stream);
            if(r != ncclSuccess && first == ncclSuccess) first = r;
        }
        const ncclResult_t ended = ncclGroupEnd();
        if(first != ncclSuccess) return fail(COATI_HIP_EHIP, "dist_gather: send/receive failed: %s", ncclGetErrorString(first));
        if(ended != ncclSuccess) return fail(COATI_HIP_EHIP, "ncclGroupEnd failed: %s", ncclGetErrorString(ended));
        return COATI_HIP_OK;
    }
    int copy_out(void* dst, const void* src, uint64_t bytes) {
        const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
        return e == hipSuccess ? COATI_HIP_OK : fail(COATI_HIP_EHIP, "dist_gather: download failed: %s", hipGetErrorString(e));
    }
    int sync() {
        D_HIP(hipStreamSynchronize(c->stream));
        return COATI_HIP_OK;
    }
};

// Where the ROOT's thread of the last sharded job spent its time (coati_hip_dist_debug_job_times; tools/dist_sim_bench.py):
// seconds in [0] the whole job loop, [1] making / waiting for its own chunks (Chunks::start of the first, the helper thread's of the others, Chunks::finish), [2] the count exchanges, [3] the
// send / receive group (Env::transfer), [4] unpack + placement + offset rebase (Env::copy_out inside), [5] rounds.
// [0] - [1] - [2] - [3] - [4] is the loop's own logic (plans, validation, transfer lists).  One writer (the root's thread).
struct JobTimes {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ([6]: reserving the landing zone; [7]: the rank's own copy-out in the local form)
};
JobTimes g_job_times;
thread_local JobTimes* t_job_times = nullptr;  // set on the root's thread for the duration of a job
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct PhaseTimer {  // adds the time until its end to t_job_times->v[slot] (no-op on other threads)
    int slot;
    double t0;
    explicit PhaseTimer(int s) : slot(s), t0(t_job_times != nullptr ? now_s() : 0.0) {}
    ~PhaseTimer() {
        if(t_job_times != nullptr) t_job_times->v[slot] += now_s() - t0;
    }
};

// Host memory in place of HBM, an abstract transport in place of RCCL.
struct HostTransport {
    virtual ~HostTransport() = default;
    virtual int allgather(const uint64_t* mine, uint64_t* all, uint32_t words_per_rank) = 0;
    virtual int send(int peer, const void* data, uint64_t bytes) = 0;
    virtual int recv(int peer, void* data, uint64_t bytes) = 0;
};
struct HostEnv {
    HostTransport* t;
    int world_, rank_;
    std::vector<char> land;
    int world() const { return world_; }
    int rank() const { return rank_; }
    int begin() { return COATI_HIP_OK; }
    int exchange_counts(const uint64_t* mine, uint64_t* all) {
        const int rc = t->allgather(mine, all, kCountWords);
        return rc == COATI_HIP_OK ? rc : fail(rc, "host transport: the count exchange failed (%d)", rc);
    }
    int landing_reserve(uint64_t need) {
        if(need > land.size()) land.assign(need, static_cast<char>(0xDD));
        return COATI_HIP_OK;
    }
    char* landing() { return land.data(); }
    int transfer(int root, const std::vector<Transfer>& xfer, const Block& own) {
        for(const Transfer& x : xfer) {
            const uint64_t bytes = x.count * x.bytes_each;
            int rc;
            if(rank_ == root) {
                if(x.at + bytes > land.size()) return fail(COATI_HIP_ESTATE, "host transport: a block leaves the landing zone");
                rc = t->recv(x.peer, land.data() + x.at, bytes);
            } else {
                rc = t->send(root, block_array(own, x.which), bytes);
            }
            if(rc != COATI_HIP_OK) return fail(rc, "host transport: %s of array %d (%llu bytes) failed (%d)", rank_ == root ? "receive" : "send", x.which,
                                               static_cast<unsigned long long>(bytes), rc);
        }
        return COATI_HIP_OK;
    }
    int copy_out(void* dst, const void* src, uint64_t bytes) {
        std::memcpy(dst, src, bytes);
        return COATI_HIP_OK;
    }
    int sync() { return COATI_HIP_OK; }
};

// One gather round on one rank.  my_status: this rank's verdict on its own contribution (a rank whose chunk could
// not be made still takes part, with nothing to send, and tells the others); when any rank reports a failure every
// rank returns an error from THIS round -- nobody is left waiting in a later collective.  `validate` (may be empty)
// judges the counts before anything moves.  `presized`: the root reserved its landing zone before the job's first
// round (and reported a failure in its status word); otherwise -- the stand-alone gather, whose sizes are only known
// now -- the root reserves here and a second, one-word exchange tells the peers whether to send.
template <typename Env, typename Validate>
int gather_round(Env& env, int root, const Block& own, const uint64_t* mine, uint64_t* counts3, uint32_t mask, const Place* place_or_null,
                 Validate&& validate, bool presized, float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    const int world = env.world(), rank = env.rank();
    // 1. everybody learns everybody's counts and status
    int rc;
    {
        const PhaseTimer timed(2);
        rc = env.exchange_counts(mine, counts3);
    }
    if(rc != COATI_HIP_OK) return rc;
    if(const int bad = failed_rank(world, counts3); bad >= 0) {
        if(bad == rank) return static_cast<int>(counts3[kCountWords * bad + 2]);  // (its own message is already set)
        return fail(COATI_HIP_ESTATE, "dist_gather: rank %d failed with code %d", bad, static_cast<int>(counts3[kCountWords * bad + 2]));
    }
    rc = validate();
    if(rc != COATI_HIP_OK) return rc;
    std::vector<Place> place(static_cast<size_t>(world));
    if(place_or_null != nullptr)
        std::copy(place_or_null, place_or_null + world, place.begin());
    else
        prefix_places(world, counts3, place.data());
    // 2. the landing zone, then one group of sends / receives out of (into) HBM
    std::vector<Land> land(static_cast<size_t>(world));
    std::vector<Transfer> xfer;
    int root_status = COATI_HIP_OK;
    if(rank == root) {
        const uint64_t need = landing_plan(world, root, counts3, mask, land.data());
        root_status = env.landing_reserve(need);
        uint64_t total_ops = 0;
        for(int r = 0; r < world; ++r) total_ops = std::max(total_ops, place[static_cast<size_t>(r)].op0 + counts3[kCountWords * r + 1]);
        if(root_status == COATI_HIP_OK && ops != nullptr && (mask & kOps) && ops_capacity < total_ops)
            root_status = fail(COATI_HIP_EINVAL, "dist_gather: ops_capacity %llu < %llu", static_cast<unsigned long long>(ops_capacity),
                               static_cast<unsigned long long>(total_ops));
        for(int r = 0; r < world; ++r)
            if(r != root) transfers_of_rank(r, counts3, land[static_cast<size_t>(r)], mask, xfer);
    } else {
        transfers_of_rank(rank, counts3, Land{0, 0, 0, 0}, mask, xfer);
    }
    if(!presized && world > 1) {
        std::vector<uint64_t> verdicts(static_cast<size_t>(kCountWords) * static_cast<size_t>(world), 0);
        const uint64_t word[kCountWords] = {0, 0, static_cast<uint64_t>(root_status)};
        rc = env.exchange_counts(word, verdicts.data());
        if(rc != COATI_HIP_OK) return rc;
        const int root_said = static_cast<int>(verdicts[static_cast<size_t>(kCountWords) * root + 2]);
        if(root_said != COATI_HIP_OK) return rank == root ? root_status : fail(COATI_HIP_ESTATE, "dist_gather: the root failed with code %d", root_said);
    } else if(root_status != COATI_HIP_OK) {
        return root_status;  // (a presized job cannot get here: the sizes were reserved and checked before round 0)
    }
    {
        const PhaseTimer timed(3);
        rc = env.transfer(root, xfer, own);
    }
    if(rc != COATI_HIP_OK) return rc;
    if(rank != root) return env.sync();  // the block's arrays may be reused after the call
    // 3. root: download, every block to its place
    const PhaseTimer timed_unpack(4);
    rc = unpack_blocks(world, root, counts3, mask, place.data(), own, env.landing(), land.data(), scores, ops, ops_off, ops_len,
                       [&env](void* dst, const void* src, uint64_t bytes) { return env.copy_out(dst, src, bytes); });
    const int rs = env.sync();
    if(rc != COATI_HIP_OK) return rc;
    if(rs != COATI_HIP_OK) return rs;
    if(ops_off != nullptr && (mask & kOff)) rebase_offsets(world, counts3, place.data(), ops_off);
    return COATI_HIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// The sharded job on one rank: its shard in chunks, chunk k + 1 computing while chunk k is gathered.
//   gather-all (local = false): every result array goes to the root, pairs in input order.
//   local      (local = true):  a rank's ops, op offsets, op lengths and scores stay with it -- downloaded over ITS
//     PCIe link into ITS arrays (entry 0 = the first pair of its shard; offsets index its own ops array) -- and only
//     the summary (scores + op lengths, 8 bytes per pair) is gathered, if asked for.  With 2 kB of ops per 1 kb pair
//     the root's single host link otherwise carries every rank's output (the stage budget).
// `Chunks` computes: start(p0, n) -> handle; finish(handle, block, n, op bytes) waits for it; release(handle).
// ---------------------------------------------------------------------------------------------------
struct JobOut {
    float* scores = nullptr;
    uint8_t* ops = nullptr;
    uint64_t ops_capacity = 0;
    uint64_t* ops_off = nullptr;
    uint32_t* ops_len = nullptr;
    bool local = false, summary = false;
    float* all_scores = nullptr;  // local mode, root: every pair's score / op length (summary)
    uint32_t* all_len = nullptr;
};
template <typename Env, typename Chunks>
int run_shard_job(Env& env, Chunks& chunks, int root, const JobPlan& plan, uint64_t n_pairs, const JobOut& out) {
    const int world = env.world(), rank = env.rank();
    struct TimesScope {  // (the root's thread records where its time goes: JobTimes)
        bool on;
        explicit TimesScope(bool root_thread) : on(root_thread) {
            if(on) {
                g_job_times = JobTimes{};
                t_job_times = &g_job_times;
            }
        }
        ~TimesScope() { t_job_times = nullptr; }
    } times_scope(rank == root);
    const PhaseTimer timed_total(0);
    if(t_job_times != nullptr) t_job_times->v[5] = static_cast<double>(plan.rounds);
    int my_status = env.begin();
    const auto& mine = plan.cuts[static_cast<size_t>(rank)];
    const uint64_t shard0 = plan.bounds[static_cast<size_t>(rank)], shard_ops0 = plan.op_prefix[shard0];
    const uint64_t shard_ops = plan.op_prefix[plan.bounds[static_cast<size_t>(rank) + 1]] - shard_ops0;
    const uint32_t mask = out.local ? (out.summary ? static_cast<uint32_t>(kSummary) : 0u) : static_cast<uint32_t>(kAllArrays);
    // what only this rank can know goes into its status word of the first round, so that all ranks leave together
    if(my_status == COATI_HIP_OK && !out.local && rank == root && out.ops != nullptr && out.ops_capacity < plan.op_prefix[n_pairs])
        my_status = fail(COATI_HIP_EINVAL, "dist_viterbi: ops_capacity too small");
    if(my_status == COATI_HIP_OK && out.local && out.ops != nullptr && out.ops_capacity < shard_ops)
        my_status = fail(COATI_HIP_EINVAL, "dist_viterbi: ops_capacity of rank %d too small for its shard", rank);
    if(my_status == COATI_HIP_OK) my_status = chunks.check(mine.front());
    if(my_status == COATI_HIP_OK && rank == root && mask != 0u) {
        const PhaseTimer timed(6);
        my_status = env.landing_reserve(job_landing_need(plan, world, root, mask));
    }
    std::vector<uint64_t> counts(static_cast<size_t>(kCountWords) * static_cast<size_t>(world));
    std::vector<Place> place(static_cast<size_t>(world));
    auto start = [&](size_t k, void** h) -> int {
        *h = nullptr;
        if(k + 1 >= mine.size() || mine[k + 1] == mine[k]) return COATI_HIP_OK;
        return chunks.start(mine[k], mine[k + 1] - mine[k], h);
    };
    void *cur = nullptr, *next = nullptr;
    if(my_status == COATI_HIP_OK) {
        const PhaseTimer timed(1);  // (the first chunk is made on this thread)
        my_status = start(0, &cur);
    }
    int rc = COATI_HIP_OK;
    // The next chunk is planned, uploaded and launched on a HELPER thread while this thread gathers the current one:
    // planning 48 000 pairs is ~10 ms of host work, gathering them (count exchange, 100 MB of downloads) another ~5-10,
    // the kernel 17 -- one after the other on one thread the host was the bottleneck (1 000 000 pairs on one GPU: 0.65 s
    // against 0.36 s of kernel).  (Thread-local error text of the helper is carried over by hand.)
    struct Started {
        int rc = COATI_HIP_OK;
        void* handle = nullptr;
        std::string error;
    };
    for(size_t k = 0; k < plan.rounds && rc == COATI_HIP_OK; ++k) {
        std::future<Started> starting;
        if(my_status == COATI_HIP_OK && k + 1 < plan.rounds)
            starting = std::async(std::launch::async, [&start, k]() {
                Started st;
                st.rc = start(k + 1, &st.handle);
                if(st.rc != COATI_HIP_OK) st.error = g_error;
                return st;
            });
        auto join_start = [&]() {
            if(!starting.valid()) return;
            const PhaseTimer timed(1);  // (waiting for the helper thread that makes the next chunk)
            Started st = starting.get();
            next = st.handle;
            if(st.rc != COATI_HIP_OK && my_status == COATI_HIP_OK) my_status = fail(st.rc, "%s", st.error.c_str());
        };
        Block own;
        uint64_t word[kCountWords] = {0, 0, static_cast<uint64_t>(my_status)};
        if(cur != nullptr && my_status == COATI_HIP_OK) {
            const PhaseTimer timed(1);
            my_status = chunks.finish(cur, own, word[0], word[1]);
            if(my_status != COATI_HIP_OK) word[0] = word[1] = 0;
            word[2] = static_cast<uint64_t>(my_status);
        }
        uint64_t p0 = 0, n = 0, nb = 0;
        plan_block(plan, rank, k, p0, n, nb);
        const uint64_t lp = p0 - shard0, lo = plan.op_prefix[p0] - shard_ops0;  // this chunk in the rank's own arrays
        if(out.local && !Chunks::kDeliversLocal && my_status == COATI_HIP_OK && word[0] == n && word[1] == nb && n > 0) {
            // the rank's own download, on its own link, under the exchange of the summaries
            const PhaseTimer timed(7);
            int lc = COATI_HIP_OK;
            if(out.scores != nullptr) lc = env.copy_out(out.scores + lp, own.scores, n * sizeof(float));
            if(lc == COATI_HIP_OK && out.ops_off != nullptr) lc = env.copy_out(out.ops_off + lp, own.off, n * sizeof(uint64_t));
            if(lc == COATI_HIP_OK && out.ops_len != nullptr) lc = env.copy_out(out.ops_len + lp, own.len, n * sizeof(uint32_t));
            if(lc == COATI_HIP_OK && out.ops != nullptr && nb > 0) lc = env.copy_out(out.ops + lo, own.ops, nb);
            if(lc != COATI_HIP_OK) word[2] = static_cast<uint64_t>(my_status = lc);
        }
        round_places(plan, world, k, place.data());
        rc = gather_round(env, root, own, word, counts.data(), mask, place.data(), [&]() { return check_round(plan, world, k, counts.data()); },
                          /*presized=*/true, out.local ? out.all_scores : out.scores, out.local ? nullptr : out.ops, out.ops_capacity,
                          out.local ? nullptr : out.ops_off, out.local ? out.all_len : out.ops_len);
        if(out.local) {
            const int rs = env.sync();  // (the own download; the chunk's arrays go back with `cur`)
            if(rc == COATI_HIP_OK) rc = rs;
            if(rc == COATI_HIP_OK && out.ops_off != nullptr)
                for(uint64_t i = 0; i < n; ++i) out.ops_off[lp + i] += lo;
        }
        join_start();
        if(cur != nullptr) {
            const PhaseTimer timed(1);  // (giving the chunk back: part of making chunks)
            chunks.release(cur);
        }
        cur = next;
        next = nullptr;
    }
    if(cur != nullptr) chunks.release(cur);
    if(next != nullptr) chunks.release(next);
    return rc;
}

// chunks computed on the GPU: one resident batch per chunk (coati_hip_batch_create + coati_hip_viterbi_launch)
struct GpuChunks {
    static constexpr bool kDeliversLocal = false;
    coati_hip_model_t* model;
    int rank;
    const uint8_t *a_cat, *b_cat;
    uint64_t a_first, b_first;
    const uint64_t *a_off, *b_off;
    std::vector<uint64_t> loc_a, loc_b;
    int check(uint64_t first_pair) {
        if(a_off[first_pair] < a_first || b_off[first_pair] < b_first)
            return fail(COATI_HIP_EINVAL, "dist_viterbi: the sequence arrays of rank %d start behind its shard", rank);
        return COATI_HIP_OK;
    }
    int start(uint64_t p0, uint64_t n, void** h) {
        // the sequence bytes this rank was given start at offsets a_first / b_first of the concatenation: a chunk's
        // offsets are rebased to the arrays it was given
        loc_a.resize(n + 1), loc_b.resize(n + 1);
        for(uint64_t i = 0; i <= n; ++i) loc_a[i] = a_off[p0 + i] - a_first, loc_b[i] = b_off[p0 + i] - b_first;
        coati_hip_batch_t* b = nullptr;
        int rc = coati_hip_batch_create(model, n, a_cat, loc_a.data(), b_cat, loc_b.data(), &b);
        if(rc == COATI_HIP_OK) rc = coati_hip_viterbi_launch(b);
        if(rc != COATI_HIP_OK) {
            (void)fail(rc, "%s", coati_hip_last_error());
            if(b != nullptr) coati_hip_batch_destroy(b);
            b = nullptr;
        }
        *h = b;
        return rc;
    }
    int finish(void* h, Block& own, uint64_t& n, uint64_t& nb) {
        coati_hip_batch_t* b = static_cast<coati_hip_batch_t*>(h);
        void *d_scores = nullptr, *d_ops = nullptr, *d_off = nullptr, *d_len = nullptr;
        int rc = coati_hip_viterbi_wait(b);  // the results exist (other launches of the model keep running)
        if(rc == COATI_HIP_OK) rc = coati_hip_batch_result_ptrs(b, &d_scores, &d_ops, &nb, &d_off, &d_len);
        if(rc != COATI_HIP_OK) return fail(rc, "%s", coati_hip_last_error());
        n = coati_hip_batch_pairs(b);
        own = Block{d_scores, d_ops, d_off, d_len};
        return COATI_HIP_OK;
    }
    void release(void* h) { coati_hip_batch_destroy(static_cast<coati_hip_batch_t*>(h)); }
};

// The local-results job on the GPU: a rank's whole shard is ONE chunk, computed by ONE coati_hip_viterbi_batch call --
// the one-shot form that streams chunks through a persistent kernel and overlaps its own uploads and downloads
// (0.98 of the resident kernel's rate on 1 000 000 pairs) -- straight into the rank's arrays; what is
// left for the round's exchange is the summary, uploaded once (8 bytes per pair).
struct GpuLocalChunks {
    static constexpr bool kDeliversLocal = true;
    coati_hip_model_t* model;
    int rank;
    const uint8_t *a_cat, *b_cat;
    uint64_t a_first, b_first;
    const uint64_t *a_off, *b_off;
    uint64_t shard0, shard_ops0;  // the rank's first pair and its op prefix: where a chunk lies in the rank's arrays
    const uint64_t* op_prefix;
    const JobOut& out;
    hipStream_t stream;
    std::vector<uint64_t> loc_a, loc_b;
    std::vector<float> tmp_scores;
    std::vector<uint32_t> tmp_len;
    void* d_summary = nullptr;
    int check(uint64_t first_pair) {
        if(a_off[first_pair] < a_first || b_off[first_pair] < b_first)
            return fail(COATI_HIP_EINVAL, "dist_viterbi: the sequence arrays of rank %d start behind its shard", rank);
        return COATI_HIP_OK;
    }
    int start(uint64_t p0, uint64_t n, void** h) {
        loc_a.resize(n + 1), loc_b.resize(n + 1);
        for(uint64_t i = 0; i <= n; ++i) loc_a[i] = a_off[p0 + i] - a_first, loc_b[i] = b_off[p0 + i] - b_first;
        const uint64_t lp = p0 - shard0, lo = op_prefix[p0] - shard_ops0, nb = op_prefix[p0 + n] - op_prefix[p0];
        float* scores = out.scores != nullptr ? out.scores + lp : nullptr;
        uint32_t* len = out.ops_len != nullptr ? out.ops_len + lp : nullptr;
        if(out.summary && scores == nullptr) tmp_scores.resize(n), scores = tmp_scores.data();
        if(out.summary && len == nullptr) tmp_len.resize(n), len = tmp_len.data();
        int rc = coati_hip_viterbi_batch(model, n, a_cat, loc_a.data(), b_cat, loc_b.data(), scores, out.ops != nullptr ? out.ops + lo : nullptr, nb,
                                         out.ops_off != nullptr ? out.ops_off + lp : nullptr, len);
        if(rc != COATI_HIP_OK) return fail(rc, "%s", coati_hip_last_error());
        if(out.summary && n > 0) {
            D_HIP(hipMalloc(&d_summary, n * (sizeof(float) + sizeof(uint32_t))));
            // (*h is only set on success, so the caller never calls release() for a start() that failed: free here)
            const auto upload = [&]() -> int {
                D_HIP(hipMemcpyAsync(d_summary, scores, n * sizeof(float), hipMemcpyHostToDevice, stream));
                D_HIP(hipMemcpyAsync(static_cast<char*>(d_summary) + n * sizeof(float), len, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                D_HIP(hipStreamSynchronize(stream));
                return COATI_HIP_OK;
            };
            if(const int up = upload(); up != COATI_HIP_OK) {
                release(nullptr);
                return up;
            }
        }
        n_now = n, nb_now = nb;
        *h = this;
        return COATI_HIP_OK;
    }
    uint64_t n_now = 0, nb_now = 0;
    int finish(void*, Block& own, uint64_t& n, uint64_t& nb) {
        n = n_now, nb = nb_now;
        own = Block{d_summary, nullptr, nullptr, d_summary != nullptr ? static_cast<char*>(d_summary) + n_now * sizeof(float) : nullptr};
        return COATI_HIP_OK;
    }
    void release(void*) {
        if(d_summary != nullptr) (void)hipFree(d_summary);
        d_summary = nullptr;
    }
};

// chunks "computed" from given per-pair results (the host environments): a chunk as a resident batch leaves it --
// scores[n], ops slots of la + lb bytes (the ops right-aligned in the slot: the walkers write right to left),
// ops_start[n] = index of the first op in the chunk's ops array.  Only the rank's own pairs are read.
struct HostChunks {
    static constexpr bool kDeliversLocal = false;
    const JobPlan& plan;
    const float* pair_scores;
    const uint8_t* pair_ops;
    const uint32_t* pair_ops_len;
    struct Chunk {
        std::vector<float> scores;
        std::vector<uint8_t> ops;
        std::vector<uint64_t> off;
        std::vector<uint32_t> len;
        uint64_t op_bytes = 0;  // the chunk's slots together
    };
    int check(uint64_t) { return COATI_HIP_OK; }
    int start(uint64_t p0, uint64_t n, void** h) {
        auto ch = std::make_unique<Chunk>();
        const uint64_t nb = plan.op_prefix[p0 + n] - plan.op_prefix[p0];
        ch->scores.assign(pair_scores + p0, pair_scores + p0 + n);
        ch->len.assign(pair_ops_len + p0, pair_ops_len + p0 + n);
        ch->ops.assign(std::max<uint64_t>(nb, 1), 0xEE);
        ch->op_bytes = nb;
        ch->off.resize(n);
        for(uint64_t p = 0; p < n; ++p) {
            const uint64_t slot0 = plan.op_prefix[p0 + p] - plan.op_prefix[p0], slot = plan.op_prefix[p0 + p + 1] - plan.op_prefix[p0 + p];
            if(ch->len[p] > slot) return fail(COATI_HIP_EINVAL, "dist_simulate: pair %llu has more ops than its slot", static_cast<unsigned long long>(p0 + p));
            ch->off[p] = slot0 + slot - ch->len[p];
            std::memcpy(ch->ops.data() + ch->off[p], pair_ops + plan.op_prefix[p0 + p] + slot - ch->len[p], ch->len[p]);
        }
        *h = ch.release();
        return COATI_HIP_OK;
    }
    int finish(void* h, Block& own, uint64_t& n, uint64_t& nb) {
        const Chunk* ch = static_cast<const Chunk*>(h);
        n = ch->scores.size();
        nb = ch->op_bytes;
        own = Block{ch->scores.data(), ch->ops.data(), ch->off.data(), ch->len.data()};
        return COATI_HIP_OK;
    }
    void release(void* h) { delete static_cast<Chunk*>(h); }
};

// In-process transport of the simulation: every rank a thread, mailboxes per (sender, receiver), a generation
// barrier for the all-gather.  Every wait is bounded (a protocol bug fails the test, it does not hang it).
struct ThreadFabric {
    int world;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::deque<std::vector<char>>> box;  // [src * world + dst]
    std::vector<uint64_t> slots, result;              // all-gather words of the current generation / of the completed one
    int arrived = 0;
    uint64_t generation = 0;
    explicit ThreadFabric(int w) : world(w), box(static_cast<size_t>(w) * static_cast<size_t>(w)) {}
};
struct ThreadTransport final : HostTransport {
    ThreadFabric& f;
    int rank;
    ThreadTransport(ThreadFabric& fabric, int r) : f(fabric), rank(r) {}
    static constexpr auto kPatience = std::chrono::seconds(60);
    int allgather(const uint64_t* mine, uint64_t* all, uint32_t words) override {
        std::unique_lock<std::mutex> lock(f.m);
        if(f.arrived == 0) f.slots.assign(static_cast<size_t>(words) * static_cast<size_t>(f.world), 0);
        if(f.slots.size() != static_cast<size_t>(words) * static_cast<size_t>(f.world)) return COATI_HIP_ESTATE;  // ranks disagree on the collective
        std::copy(mine, mine + words, f.slots.begin() + static_cast<size_t>(words) * static_cast<size_t>(rank));
        const uint64_t gen = f.generation;
        if(++f.arrived == f.world) {
            f.arrived = 0;
            ++f.generation;
            f.result = f.slots;
            f.cv.notify_all();
        } else if(!f.cv.wait_for(lock, kPatience, [&] { return f.generation != gen; })) {
            return COATI_HIP_ESTATE;
        }
        std::copy(f.result.begin(), f.result.end(), all);
        return COATI_HIP_OK;
    }
    int send(int peer, const void* data, uint64_t bytes) override {
        std::lock_guard<std::mutex> lock(f.m);
        const char* p = static_cast<const char*>(data);
        f.box[static_cast<size_t>(rank) * static_cast<size_t>(f.world) + static_cast<size_t>(peer)].emplace_back(p, p + bytes);
        f.cv.notify_all();
        return COATI_HIP_OK;
    }
    int recv(int peer, void* data, uint64_t bytes) override {
        std::unique_lock<std::mutex> lock(f.m);
        auto& q = f.box[static_cast<size_t>(peer) * static_cast<size_t>(f.world) + static_cast<size_t>(rank)];
        if(!f.cv.wait_for(lock, kPatience, [&] { return !q.empty(); })) return COATI_HIP_ESTATE;
        if(q.front().size() != bytes) return COATI_HIP_ESTATE;  // the sender's list and the receiver's disagree
        std::memcpy(data, q.front().data(), bytes);
        q.pop_front();
        return COATI_HIP_OK;
    }
};

// the caller's transport (coati_hip_dist_job_host)
struct CallbackTransport final : HostTransport {
    const coati_hip_dist_host_transport_t& t;
    explicit CallbackTransport(const coati_hip_dist_host_transport_t& tr) : t(tr) {}
    int allgather(const uint64_t* mine, uint64_t* all, uint32_t words) override { return t.allgather(t.ctx, mine, all, words); }
    int send(int peer, const void* data, uint64_t bytes) override { return t.send(t.ctx, peer, data, bytes); }
    int recv(int peer, void* data, uint64_t bytes) override { return t.recv(t.ctx, peer, data, bytes); }
};

}  // namespace
extern "C" {

int coati_hip_dist_gather(coati_hip_comm_t* c, int root, coati_hip_batch_t* batch, uint64_t* counts, float* scores,
                          uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(c == nullptr || counts == nullptr || root < 0 || root >= c->world) return fail(COATI_HIP_EINVAL, "dist_gather: bad argument");
    try {
        RcclEnv env{c};
        std::vector<uint64_t> counts3(static_cast<size_t>(kCountWords) * static_cast<size_t>(c->world), 0);
        Block own;
        uint64_t mine[kCountWords] = {0, 0, static_cast<uint64_t>(env.begin())};
        if(batch != nullptr && mine[2] == static_cast<uint64_t>(COATI_HIP_OK)) {
            void *d_scores = nullptr, *d_ops = nullptr, *d_off = nullptr, *d_len = nullptr;
            int rc = coati_hip_viterbi_wait(batch);  // the results exist (other launches of the model keep running)
            if(rc == COATI_HIP_OK) rc = coati_hip_batch_result_ptrs(batch, &d_scores, &d_ops, &mine[1], &d_off, &d_len);
            if(rc == COATI_HIP_OK) {
                mine[0] = coati_hip_batch_pairs(batch);
                own = Block{d_scores, d_ops, d_off, d_len};
            } else {
                (void)fail(rc, "%s", coati_hip_last_error());
                mine[0] = mine[1] = 0;
                mine[2] = static_cast<uint64_t>(rc);
            }
        }
        const int rc = gather_round(env, root, own, mine, counts3.data(), kAllArrays, nullptr, [] { return COATI_HIP_OK; }, /*presized=*/false, scores,
                                    ops, ops_capacity, ops_off, ops_len);
        for(int r = 0; r < c->world; ++r) {
            counts[2 * r] = counts3[static_cast<size_t>(kCountWords) * r];
            counts[2 * r + 1] = counts3[static_cast<size_t>(kCountWords) * r + 1];
        }
        return rc;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_gather: host allocation failed");
    }
}

static int shard_job_gpu(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat, uint64_t a_first,
                         const uint64_t* a_off, const uint8_t* b_cat, uint64_t b_first, const uint64_t* b_off, const JobOut& out) {
    if(c == nullptr || model == nullptr || a_off == nullptr || b_off == nullptr || root < 0 || root >= c->world)
        return fail(COATI_HIP_EINVAL, "dist_viterbi: bad argument");
    try {
        // Chunk plan of EVERY rank (all ranks hold the same lengths, so all compute the same plan and the
        // collectives line up): a rank's shard in chunks of at most kDefaultChunkCells cells.
        JobPlan plan;
        const int rc = make_job_plan(n_pairs, a_off, b_off, c->world, out.local ? ~0ull : kDefaultChunkCells, plan);
        if(rc != COATI_HIP_OK) return rc;  // (bad offsets: the same verdict on every rank)
        RcclEnv env{c};
        if(out.local) {
            const uint64_t s0 = plan.bounds[static_cast<size_t>(c->rank)];
            GpuLocalChunks chunks{model, c->rank, a_cat, b_cat, a_first, b_first, a_off, b_off, s0, plan.op_prefix[s0], plan.op_prefix.data(), out, c->stream,
                                  {}, {}, {}, {}};
            return run_shard_job(env, chunks, root, plan, n_pairs, out);
        }
        GpuChunks chunks{model, c->rank, a_cat, b_cat, a_first, b_first, a_off, b_off, {}, {}};
        return run_shard_job(env, chunks, root, plan, n_pairs, out);
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_viterbi: host allocation failed");
    }
}

int coati_hip_dist_viterbi_shard(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                                 uint64_t a_first, const uint64_t* a_off, const uint8_t* b_cat, uint64_t b_first, const uint64_t* b_off,
                                 float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    JobOut out;
    out.scores = scores, out.ops = ops, out.ops_capacity = ops_capacity, out.ops_off = ops_off, out.ops_len = ops_len;
    return shard_job_gpu(c, root, model, n_pairs, a_cat, a_first, a_off, b_cat, b_first, b_off, out);
}

int coati_hip_dist_viterbi_shard_local(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                                       uint64_t a_first, const uint64_t* a_off, const uint8_t* b_cat, uint64_t b_first, const uint64_t* b_off,
                                       float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                                       int gather_summary, float* all_scores, uint32_t* all_len) {
    JobOut out;
    out.scores = scores, out.ops = ops, out.ops_capacity = ops_capacity, out.ops_off = ops_off, out.ops_len = ops_len;
    out.local = true, out.summary = gather_summary != 0, out.all_scores = all_scores, out.all_len = all_len;
    return shard_job_gpu(c, root, model, n_pairs, a_cat, a_first, a_off, b_cat, b_first, b_off, out);
}

int coati_hip_dist_viterbi(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off, float* scores, uint8_t* ops,
                           uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    return coati_hip_dist_viterbi_shard(c, root, model, n_pairs, a_cat, 0, a_off, b_cat, 0, b_off, scores, ops, ops_capacity, ops_off, ops_len);
}

// Debug: where the root's thread of the LAST sharded job of this process spent its time (JobTimes above).
int coati_hip_dist_debug_job_times(double* out8) {
    if(out8 == nullptr) return fail(COATI_HIP_EINVAL, "dist_debug_job_times: bad argument");
    std::copy(g_job_times.v, g_job_times.v + 8, out8);
    return COATI_HIP_OK;
}

// ---- plan exports and the host-memory job (no device, no communicator) --------------------------------------
int coati_hip_dist_chunk_plan(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world, uint64_t chunk_cells,
                              uint64_t* cut_index, uint64_t* cuts, uint64_t cuts_capacity, uint64_t* rounds) {
    if(a_off == nullptr || b_off == nullptr || world < 1 || cut_index == nullptr || rounds == nullptr) return fail(COATI_HIP_EINVAL, "dist_chunk_plan: bad argument");
    try {
        JobPlan plan;
        const int rc = make_job_plan(n_pairs, a_off, b_off, world, chunk_cells == 0 ? kDefaultChunkCells : chunk_cells, plan);
        if(rc != COATI_HIP_OK) return rc;
        uint64_t at = 0;
        for(int r = 0; r < world; ++r) {
            cut_index[r] = at;
            for(const uint64_t v : plan.cuts[static_cast<size_t>(r)]) {
                if(cuts != nullptr && at < cuts_capacity) cuts[at] = v;
                ++at;
            }
        }
        cut_index[world] = at;
        *rounds = plan.rounds;
        if(cuts != nullptr && at > cuts_capacity) return fail(COATI_HIP_EINVAL, "dist_chunk_plan: %llu boundaries do not fit", static_cast<unsigned long long>(at));
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_chunk_plan: host allocation failed");
    }
}

int coati_hip_dist_landing_plan(int world, int root, const uint64_t* counts, uint64_t* land4, uint64_t* need) {
    if(world < 1 || root < 0 || root >= world || counts == nullptr || land4 == nullptr || need == nullptr) return fail(COATI_HIP_EINVAL, "dist_landing_plan: bad argument");
    try {
        std::vector<uint64_t> c3(static_cast<size_t>(kCountWords) * static_cast<size_t>(world), 0);
        for(int r = 0; r < world; ++r) c3[static_cast<size_t>(kCountWords) * r] = counts[2 * r], c3[static_cast<size_t>(kCountWords) * r + 1] = counts[2 * r + 1];
        std::vector<Land> land(static_cast<size_t>(world));
        *need = landing_plan(world, root, c3.data(), kAllArrays, land.data());
        for(int r = 0; r < world; ++r) {
            land4[4 * r] = land[static_cast<size_t>(r)].scores, land4[4 * r + 1] = land[static_cast<size_t>(r)].ops;
            land4[4 * r + 2] = land[static_cast<size_t>(r)].off, land4[4 * r + 3] = land[static_cast<size_t>(r)].len;
        }
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_landing_plan: host allocation failed");
    }
}

int coati_hip_dist_job_host(const coati_hip_dist_host_transport_t* transport, int world, int rank, int root, uint64_t n_pairs,
                            const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells, int local, int gather_summary,
                            const float* pair_scores, const uint8_t* pair_ops, const uint32_t* pair_ops_len, float* scores, uint8_t* ops,
                            uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len, float* all_scores, uint32_t* all_len) {
    if(transport == nullptr || transport->allgather == nullptr || transport->send == nullptr || transport->recv == nullptr || world < 1 || rank < 0 ||
       rank >= world || root < 0 || root >= world || a_off == nullptr || b_off == nullptr || pair_scores == nullptr || pair_ops == nullptr || pair_ops_len == nullptr)
        return fail(COATI_HIP_EINVAL, "dist_job_host: bad argument");
    try {
        JobPlan plan;
        const int rc = make_job_plan(n_pairs, a_off, b_off, world, chunk_cells == 0 ? kDefaultChunkCells : chunk_cells, plan);
        if(rc != COATI_HIP_OK) return rc;
        CallbackTransport tr(*transport);
        HostEnv env{&tr, world, rank, {}};
        HostChunks chunks{plan, pair_scores, pair_ops, pair_ops_len};
        JobOut out;
        out.scores = scores, out.ops = ops, out.ops_capacity = ops_capacity, out.ops_off = ops_off, out.ops_len = ops_len;
        out.local = local != 0, out.summary = gather_summary != 0, out.all_scores = all_scores, out.all_len = all_len;
        return run_shard_job(env, chunks, root, plan, n_pairs, out);
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_job_host: host allocation failed");
    }
}

// every rank of the job as a thread of this process
static int simulate_impl(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells, bool local,
                         bool summary, const float* pair_scores, const uint8_t* pair_ops, const uint32_t* pair_ops_len, float* scores, uint8_t* ops,
                         uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len, float* all_scores, uint32_t* all_len) {
    if(world < 1 || root < 0 || root >= world || a_off == nullptr || b_off == nullptr || pair_scores == nullptr || pair_ops == nullptr || pair_ops_len == nullptr)
        return fail(COATI_HIP_EINVAL, "dist_simulate: bad argument");
    try {
        JobPlan plan;
        int rc = make_job_plan(n_pairs, a_off, b_off, world, chunk_cells == 0 ? kDefaultChunkCells : chunk_cells, plan);
        if(rc != COATI_HIP_OK) return rc;
        if(local && ops != nullptr && ops_capacity < plan.op_prefix[n_pairs]) return fail(COATI_HIP_EINVAL, "dist_simulate: ops_capacity too small");
        ThreadFabric fabric(world);
        std::vector<int> rcs(static_cast<size_t>(world), COATI_HIP_OK);
        std::vector<std::string> errors(static_cast<size_t>(world));
        std::vector<std::thread> threads;
        for(int r = 0; r < world; ++r) {
            threads.emplace_back([&, r]() {
                try {
                    ThreadTransport tr(fabric, r);
                    HostEnv env{&tr, world, r, {}};
                    HostChunks chunks{plan, pair_scores, pair_ops, pair_ops_len};
                    JobOut out;
                    out.local = local, out.summary = summary;
                    if(local) {
                        // every rank's own arrays are the slices of the caller's arrays that start at its shard
                        const uint64_t s0 = plan.bounds[static_cast<size_t>(r)], o0 = plan.op_prefix[s0];
                        const uint64_t o1 = plan.op_prefix[plan.bounds[static_cast<size_t>(r) + 1]];
                        out.scores = scores != nullptr ? scores + s0 : nullptr;
                        out.ops = ops != nullptr ? ops + o0 : nullptr;
                        out.ops_capacity = o1 - o0;
                        out.ops_off = ops_off != nullptr ? ops_off + s0 : nullptr;
                        out.ops_len = ops_len != nullptr ? ops_len + s0 : nullptr;
                        if(r == root) out.all_scores = all_scores, out.all_len = all_len;
                    } else if(r == root) {
                        out.scores = scores, out.ops = ops, out.ops_capacity = ops_capacity, out.ops_off = ops_off, out.ops_len = ops_len;
                    }
                    rcs[static_cast<size_t>(r)] = run_shard_job(env, chunks, root, plan, n_pairs, out);
                    if(rcs[static_cast<size_t>(r)] != COATI_HIP_OK) errors[static_cast<size_t>(r)] = g_error;  // (thread-local)
                } catch(const std::bad_alloc&) {
                    rcs[static_cast<size_t>(r)] = COATI_HIP_ENOMEM;
                    errors[static_cast<size_t>(r)] = "dist_simulate: host allocation failed";
                }
            });
        }
        for(auto& t : threads) t.join();
        // the root's verdict first (it names the cause when a rank failed), then anybody's
        if(rcs[static_cast<size_t>(root)] != COATI_HIP_OK) return fail(rcs[static_cast<size_t>(root)], "%s", errors[static_cast<size_t>(root)].c_str());
        for(int r = 0; r < world; ++r)
            if(rcs[static_cast<size_t>(r)] != COATI_HIP_OK) return fail(rcs[static_cast<size_t>(r)], "%s", errors[static_cast<size_t>(r)].c_str());
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_simulate: host allocation failed");
    } catch(const std::system_error& ex) {
        return fail(COATI_HIP_EHIP, "dist_simulate: %s", ex.what());
    }
}

int coati_hip_dist_simulate(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells,
                            const float* pair_scores, const uint8_t* pair_ops, const uint32_t* pair_ops_len, float* scores, uint8_t* ops,
                            uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    return simulate_impl(world, root, n_pairs, a_off, b_off, chunk_cells, false, false, pair_scores, pair_ops, pair_ops_len, scores, ops, ops_capacity,
                         ops_off, ops_len, nullptr, nullptr);
}

int coati_hip_dist_simulate_local(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells,
                                  int gather_summary, const float* pair_scores, const uint8_t* pair_ops, const uint32_t* pair_ops_len,
                                  float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len, float* all_scores,
                                  uint32_t* all_len) {
    return simulate_impl(world, root, n_pairs, a_off, b_off, chunk_cells, true, gather_summary != 0, pair_scores, pair_ops, pair_ops_len, scores, ops,
                         ops_capacity, ops_off, ops_len, all_scores, all_len);
}

}  // extern "C"
