// dist.hip -- libcoati_hip_dist.so: the multi-GPU layer of include/coati_hip_dist.h.  One process per
// GPU; RCCL (librccl.so, linked directly) for the two real exchanges -- the model broadcast and the
// gather of results to the root; everything else goes through the public C ABI of libcoati_hip.so.
#include "coati_hip_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
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
    guard.c = nullptr;
    *out = c;
    return COATI_HIP_OK;
}

int coati_hip_dist_rank(const coati_hip_comm_t* c) { return c != nullptr ? c->rank : -1; }
int coati_hip_dist_world(const coati_hip_comm_t* c) { return c != nullptr ? c->world : 0; }

int coati_hip_dist_allreduce_f64(coati_hip_comm_t* c, int op, double* values, uint32_t n) {
    if(c == nullptr || values == nullptr || n == 0 || n > 64 || (op != 0 && op != 1)) return fail(COATI_HIP_EINVAL, "dist_allreduce_f64: bad argument");
    D_HIP(hipSetDevice(c->device));
    double* d = nullptr;
    D_HIP(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(double)));
    struct Free {
        void* p;
        ~Free() { (void)hipFree(p); }
    } free_d{d};
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
// they land in the root's HBM, where they end up in the caller's arrays.  The collective code below and the
// host-memory simulation coati_hip_dist_simulate (tests: world 2, 3, 8 without a GPU) both run on THESE
// functions; what only hardware can exercise is the literal ncclSend / ncclRecv / hipMemcpyAsync calls.
// ---------------------------------------------------------------------------------------------------
}  // extern "C"
namespace {

constexpr int kCountWords = 3;  // per rank in the all-gather: pairs, op bytes, status (COATI_HIP_OK or the rank's error code)

struct Land {
    uint64_t scores, ops, off, len;  // byte offsets in the root's landing zone
};
// landing zone of the peers' blocks on the root (rank order, every array 256-byte aligned); returns its size
uint64_t landing_plan(int world, int root, const uint64_t* counts /* kCountWords per rank */, Land* land) {
    uint64_t need = 0;
    auto take = [&](uint64_t bytes) {
        const uint64_t at = need;
        need += (bytes + 255) / 256 * 256;
        return at;
    };
    for(int r = 0; r < world; ++r) {
        if(r == root) {
            land[r] = Land{0, 0, 0, 0};
            continue;
        }
        const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
        land[r] = Land{take(n * sizeof(float)), take(ob), take(n * sizeof(uint64_t)), take(n * sizeof(uint32_t))};
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
void transfers_of_rank(int r, const uint64_t* counts, const Land& l, std::vector<Transfer>& out) {
    const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
    if(n > 0) {
        out.push_back(Transfer{r, 0, l.scores, n, static_cast<uint32_t>(sizeof(float))});
        out.push_back(Transfer{r, 1, l.off, n, static_cast<uint32_t>(sizeof(uint64_t))});
        out.push_back(Transfer{r, 2, l.len, n, static_cast<uint32_t>(sizeof(uint32_t))});
    }
    if(ob > 0) out.push_back(Transfer{r, 3, l.ops, ob, 1u});
}

// A rank's result block as four arrays (device pointers in the collective, host pointers in the simulation)
struct Block {
    const void *scores = nullptr, *ops = nullptr, *off = nullptr, *len = nullptr;
};
const void* block_array(const Block& b, int which) { return which == 0 ? b.scores : which == 1 ? b.off : which == 2 ? b.len : b.ops; }

// Root, after the exchange: the blocks in rank order (its own arrays, the landing zone for the peers) go to the
// caller's arrays -- rank r's entries start at sum_{q<r} pairs(q), its op bytes at sum_{q<r} opbytes(q).  `copy`
// moves bytes (hipMemcpyAsync device->host in the collective, memcpy in the simulation).
template <typename Copy>
int unpack_blocks(int world, int root, const uint64_t* counts, const Block& own, const char* landing, const Land* land, float* scores,
                  uint8_t* ops, uint64_t* ops_off, uint32_t* ops_len, Copy&& copy) {
    uint64_t pair0 = 0, op0 = 0;
    for(int r = 0; r < world; ++r) {
        const uint64_t n = counts[kCountWords * r], ob = counts[kCountWords * r + 1];
        Block b = own;
        if(r != root) b = Block{landing + land[r].scores, landing + land[r].ops, landing + land[r].off, landing + land[r].len};
        if(n > 0) {
            if(scores != nullptr) { const int rc = copy(scores + pair0, b.scores, n * sizeof(float)); if(rc != COATI_HIP_OK) return rc; }
            if(ops_off != nullptr) { const int rc = copy(ops_off + pair0, b.off, n * sizeof(uint64_t)); if(rc != COATI_HIP_OK) return rc; }
            if(ops_len != nullptr) { const int rc = copy(ops_len + pair0, b.len, n * sizeof(uint32_t)); if(rc != COATI_HIP_OK) return rc; }
        }
        if(ob > 0 && ops != nullptr) { const int rc = copy(ops + op0, b.ops, ob); if(rc != COATI_HIP_OK) return rc; }
        pair0 += n;
        op0 += ob;
    }
    return COATI_HIP_OK;
}
// every rank's op offsets index its own ops array: rebase them into the concatenation
void rebase_offsets(int world, const uint64_t* counts, uint64_t* ops_off) {
    uint64_t pair0 = 0, op0 = 0;
    for(int r = 0; r < world; ++r) {
        for(uint64_t p = 0; p < counts[kCountWords * r]; ++p) ops_off[pair0 + p] += op0;
        pair0 += counts[kCountWords * r];
        op0 += counts[kCountWords * r + 1];
    }
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
// every rank checks the gathered counts of a round against the plan (so that all leave together on a mismatch)
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
// root: a round's staging arrays (the ranks' chunks concatenated in rank order, offsets rebased into st_ops) go to
// their places in the caller's arrays -- pairs keep their input order
void place_round(const JobPlan& plan, int world, size_t k, const float* st_scores, const uint8_t* st_ops, const uint64_t* st_off,
                 const uint32_t* st_len, float* scores, uint8_t* ops, uint64_t* ops_off, uint32_t* ops_len) {
    uint64_t at_p = 0, at_b = 0;
    for(int r = 0; r < world; ++r) {
        uint64_t p0, n, nb;
        plan_block(plan, r, k, p0, n, nb);
        if(n == 0) continue;
        if(scores != nullptr) std::memcpy(scores + p0, st_scores + at_p, n * sizeof(float));
        if(ops_len != nullptr) std::memcpy(ops_len + p0, st_len + at_p, n * sizeof(uint32_t));
        if(ops != nullptr && nb > 0) std::memcpy(ops + plan.op_prefix[p0], st_ops + at_b, nb);
        if(ops_off != nullptr)
            for(uint64_t p = 0; p < n; ++p) ops_off[p0 + p] = st_off[at_p + p] - at_b + plan.op_prefix[p0];
        at_p += n;
        at_b += nb;
    }
}
void round_totals(const JobPlan& plan, int world, size_t k, uint64_t& np, uint64_t& nb) {
    np = nb = 0;
    for(int r = 0; r < world; ++r) {
        uint64_t p0, n, b;
        plan_block(plan, r, k, p0, n, b);
        np += n;
        nb += b;
    }
}

// The collective gather.  my_status: this rank's verdict on its own contribution (a rank whose batch could not
// be made still takes part, with nothing to send, and tells the others); when any rank reports a failure every
// rank returns COATI_HIP_ESTATE from THIS call -- nobody is left waiting in a later collective.
int gather_impl(coati_hip_comm_t* c, int root, coati_hip_batch_t* batch, int my_status, uint64_t* counts3, float* scores, uint8_t* ops,
                uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    D_HIP(hipSetDevice(c->device));
    Block own;
    uint64_t mine[kCountWords] = {0, 0, static_cast<uint64_t>(my_status)};
    if(batch != nullptr && my_status == COATI_HIP_OK) {
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
    // 1. everybody learns everybody's counts and status
    D_HIP(hipMemcpyAsync(c->d_mine, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    D_NCCL(ncclAllGather(c->d_mine, c->d_counts, kCountWords, ncclUint64, c->comm, c->stream));
    D_HIP(hipMemcpyAsync(counts3, c->d_counts, kCountWords * sizeof(uint64_t) * static_cast<size_t>(c->world), hipMemcpyDeviceToHost, c->stream));
    D_HIP(hipStreamSynchronize(c->stream));
    if(const int bad = failed_rank(c->world, counts3); bad >= 0) {
        if(bad == c->rank) return static_cast<int>(counts3[kCountWords * bad + 2]);  // (its own message is already set)
        return fail(COATI_HIP_ESTATE, "dist_gather: rank %d failed with code %d", bad, static_cast<int>(counts3[kCountWords * bad + 2]));
    }
    // 2. one group of sends / receives out of (into) HBM
    std::vector<Land> land(static_cast<size_t>(c->world));
    std::vector<Transfer> xfer;
    if(c->rank == root) {
        const uint64_t need = landing_plan(c->world, root, counts3, land.data());
        if(need > c->land_bytes) {
            if(c->d_land != nullptr) (void)hipFree(c->d_land);
            c->d_land = nullptr;
            c->land_bytes = 0;
            // (a failed allocation here is fatal for the job: the peers are about to send.  Agreeing on it would cost
            // a second all-gather per round; the landing zone is at most the peers' result arrays, which fitted theirs.)
            D_HIP(hipMalloc(&c->d_land, need));
            c->land_bytes = need;
        }
        for(int r = 0; r < c->world; ++r)
            if(r != root) transfers_of_rank(r, counts3, land[static_cast<size_t>(r)], xfer);
    } else {
        transfers_of_rank(c->rank, counts3, Land{0, 0, 0, 0}, xfer);
    }
    char* base = static_cast<char*>(c->d_land);
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
    if(c->rank != root) {
        D_HIP(hipStreamSynchronize(c->stream));  // the batch's arrays may be reused after the call
        return COATI_HIP_OK;
    }
    // 3. root: download in rank order
    uint64_t total_ops = 0;
    for(int r = 0; r < c->world; ++r) total_ops += counts3[kCountWords * r + 1];
    if(ops != nullptr && ops_capacity < total_ops) {
        (void)hipStreamSynchronize(c->stream);
        return fail(COATI_HIP_EINVAL, "dist_gather: ops_capacity %llu < %llu", static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(total_ops));
    }
    hipStream_t stream = c->stream;
    const int rc = unpack_blocks(c->world, root, counts3, own, base, land.data(), scores, ops, ops_off, ops_len,
                                 [stream](void* dst, const void* src, uint64_t bytes) {
                                     const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
                                     return e == hipSuccess ? COATI_HIP_OK : fail(COATI_HIP_EHIP, "dist_gather: download failed: %s", hipGetErrorString(e));
                                 });
    D_HIP(hipStreamSynchronize(c->stream));
    if(rc != COATI_HIP_OK) return rc;
    if(ops_off != nullptr) rebase_offsets(c->world, counts3, ops_off);
    return COATI_HIP_OK;
}

}  // namespace
extern "C" {

int coati_hip_dist_gather(coati_hip_comm_t* c, int root, coati_hip_batch_t* batch, uint64_t* counts, float* scores,
                          uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(c == nullptr || counts == nullptr || root < 0 || root >= c->world) return fail(COATI_HIP_EINVAL, "dist_gather: bad argument");
    try {
        std::vector<uint64_t> counts3(static_cast<size_t>(kCountWords) * static_cast<size_t>(c->world), 0);
        const int rc = gather_impl(c, root, batch, COATI_HIP_OK, counts3.data(), scores, ops, ops_capacity, ops_off, ops_len);
        for(int r = 0; r < c->world; ++r) {
            counts[2 * r] = counts3[static_cast<size_t>(kCountWords) * r];
            counts[2 * r + 1] = counts3[static_cast<size_t>(kCountWords) * r + 1];
        }
        return rc;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_gather: host allocation failed");
    }
}

int coati_hip_dist_viterbi_shard(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                                 uint64_t a_first, const uint64_t* a_off, const uint8_t* b_cat, uint64_t b_first, const uint64_t* b_off,
                                 float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(c == nullptr || model == nullptr || a_off == nullptr || b_off == nullptr || root < 0 || root >= c->world)
        return fail(COATI_HIP_EINVAL, "dist_viterbi: bad argument");
    try {
        const int world = c->world;
        // Chunk plan of EVERY rank (all ranks hold the same lengths, so all compute the same plan and the
        // collectives line up): a rank's shard in chunks of at most kChunkCells cells.
        constexpr uint64_t kChunkCells = 12000ull * 1002 * 1002;
        JobPlan plan;
        int my_status = make_job_plan(n_pairs, a_off, b_off, world, kChunkCells, plan);
        if(my_status != COATI_HIP_OK) return my_status;  // (bad offsets: the same verdict on every rank)
        // what only this rank can know goes into its status word of the first round, so that all ranks leave together
        const auto& mine = plan.cuts[static_cast<size_t>(c->rank)];
        if(c->rank == root && ops != nullptr && ops_capacity < plan.op_prefix[n_pairs]) my_status = fail(COATI_HIP_EINVAL, "dist_viterbi: ops_capacity too small");
        if(my_status == COATI_HIP_OK && (a_off[mine.front()] < a_first || b_off[mine.front()] < b_first))
            my_status = fail(COATI_HIP_EINVAL, "dist_viterbi: the sequence arrays of rank %d start behind its shard", c->rank);
        // the sequence bytes this rank was given start at offsets a_first / b_first of the concatenation: a chunk's
        // offsets are rebased to the arrays it was given
        std::vector<uint64_t> loc_a, loc_b;
        auto make = [&](size_t k, coati_hip_batch_t** out) -> int {
            *out = nullptr;
            if(k + 1 >= mine.size() || mine[k + 1] == mine[k]) return COATI_HIP_OK;
            const uint64_t n = mine[k + 1] - mine[k];
            loc_a.resize(n + 1), loc_b.resize(n + 1);
            for(uint64_t i = 0; i <= n; ++i) loc_a[i] = a_off[mine[k] + i] - a_first, loc_b[i] = b_off[mine[k] + i] - b_first;
            int rc = coati_hip_batch_create(model, n, a_cat, loc_a.data(), b_cat, loc_b.data(), out);
            if(rc == COATI_HIP_OK) rc = coati_hip_viterbi_launch(*out);
            if(rc != COATI_HIP_OK) (void)fail(rc, "%s", coati_hip_last_error());
            return rc;
        };
        // per-round staging on the root: ranks' chunks arrive concatenated in rank order
        std::vector<uint64_t> counts(static_cast<size_t>(kCountWords) * static_cast<size_t>(world));
        std::vector<float> st_scores;
        std::vector<uint8_t> st_ops;
        std::vector<uint64_t> st_off;
        std::vector<uint32_t> st_len;
        coati_hip_batch_t *cur = nullptr, *next = nullptr;
        if(my_status == COATI_HIP_OK) my_status = make(0, &cur);
        int rc = COATI_HIP_OK;
        for(size_t k = 0; k < plan.rounds && rc == COATI_HIP_OK; ++k) {
            if(my_status == COATI_HIP_OK) my_status = make(k + 1, &next);  // the next chunk computes while this one is gathered
            if(c->rank == root) {
                uint64_t np = 0, nb = 0;
                round_totals(plan, world, k, np, nb);
                st_scores.resize(np), st_off.resize(np), st_len.resize(np), st_ops.resize(std::max<uint64_t>(nb, 1));
            }
            rc = gather_impl(c, root, cur, my_status, counts.data(), st_scores.data(), st_ops.data(), st_ops.size(), st_off.data(), st_len.data());
            if(cur != nullptr) coati_hip_batch_destroy(cur);
            cur = next;
            next = nullptr;
            if(rc != COATI_HIP_OK) break;
            rc = check_round(plan, world, k, counts.data());  // (every rank: same counts, same plan, same verdict)
            if(rc != COATI_HIP_OK || c->rank != root) continue;
            place_round(plan, world, k, st_scores.data(), st_ops.data(), st_off.data(), st_len.data(), scores, ops, ops_off, ops_len);
        }
        if(cur != nullptr) coati_hip_batch_destroy(cur);
        if(next != nullptr) coati_hip_batch_destroy(next);
        return rc;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_viterbi: host allocation failed");
    }
}

int coati_hip_dist_viterbi(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off, float* scores, uint8_t* ops,
                           uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    return coati_hip_dist_viterbi_shard(c, root, model, n_pairs, a_cat, 0, a_off, b_cat, 0, b_off, scores, ops, ops_capacity, ops_off, ops_len);
}

// ---- plan exports and the host-memory simulation (no device, no communicator) -------------------------------
int coati_hip_dist_chunk_plan(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world, uint64_t chunk_cells,
                              uint64_t* cut_index, uint64_t* cuts, uint64_t cuts_capacity, uint64_t* rounds) {
    if(a_off == nullptr || b_off == nullptr || world < 1 || cut_index == nullptr || rounds == nullptr) return fail(COATI_HIP_EINVAL, "dist_chunk_plan: bad argument");
    try {
        JobPlan plan;
        const int rc = make_job_plan(n_pairs, a_off, b_off, world, chunk_cells == 0 ? 12000ull * 1002 * 1002 : chunk_cells, plan);
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
        *need = landing_plan(world, root, c3.data(), land.data());
        for(int r = 0; r < world; ++r) {
            land4[4 * r] = land[static_cast<size_t>(r)].scores, land4[4 * r + 1] = land[static_cast<size_t>(r)].ops;
            land4[4 * r + 2] = land[static_cast<size_t>(r)].off, land4[4 * r + 3] = land[static_cast<size_t>(r)].len;
        }
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_landing_plan: host allocation failed");
    }
}

int coati_hip_dist_simulate(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells,
                            const float* pair_scores, const uint8_t* pair_ops, const uint32_t* pair_ops_len, float* scores, uint8_t* ops,
                            uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(world < 1 || root < 0 || root >= world || a_off == nullptr || b_off == nullptr || pair_scores == nullptr || pair_ops == nullptr || pair_ops_len == nullptr)
        return fail(COATI_HIP_EINVAL, "dist_simulate: bad argument");
    try {
        JobPlan plan;
        int rc = make_job_plan(n_pairs, a_off, b_off, world, chunk_cells == 0 ? 12000ull * 1002 * 1002 : chunk_cells, plan);
        if(rc != COATI_HIP_OK) return rc;
        if(ops != nullptr && ops_capacity < plan.op_prefix[n_pairs]) return fail(COATI_HIP_EINVAL, "dist_simulate: ops_capacity too small");
        // a rank's chunk as a resident batch leaves it: scores[n], ops slots of la+lb bytes (the ops right-aligned in
        // the slot: the walkers write right to left), ops_start[n] = index of the first op in the chunk's ops array
        struct Chunk {
            std::vector<float> scores;
            std::vector<uint8_t> ops;
            std::vector<uint64_t> off;
            std::vector<uint32_t> len;
        };
        std::vector<uint64_t> counts(static_cast<size_t>(kCountWords) * static_cast<size_t>(world));
        std::vector<Land> land(static_cast<size_t>(world));
        std::vector<Chunk> chunk(static_cast<size_t>(world));
        std::vector<char> landing;
        std::vector<float> st_scores;
        std::vector<uint8_t> st_ops;
        std::vector<uint64_t> st_off;
        std::vector<uint32_t> st_len;
        for(size_t k = 0; k < plan.rounds; ++k) {
            // every rank "computes" its chunk of this round and announces its counts
            for(int r = 0; r < world; ++r) {
                uint64_t p0, n, nb;
                plan_block(plan, r, k, p0, n, nb);
                Chunk& ch = chunk[static_cast<size_t>(r)];
                ch.scores.assign(pair_scores + p0, pair_scores + p0 + n);
                ch.len.assign(pair_ops_len + p0, pair_ops_len + p0 + n);
                ch.ops.assign(nb, 0xEE);
                ch.off.resize(n);
                for(uint64_t p = 0; p < n; ++p) {
                    const uint64_t slot0 = plan.op_prefix[p0 + p] - plan.op_prefix[p0], slot = plan.op_prefix[p0 + p + 1] - plan.op_prefix[p0 + p];
                    if(ch.len[p] > slot) return fail(COATI_HIP_EINVAL, "dist_simulate: pair %llu has more ops than its slot", static_cast<unsigned long long>(p0 + p));
                    ch.off[p] = slot0 + slot - ch.len[p];
                    std::memcpy(ch.ops.data() + ch.off[p], pair_ops + plan.op_prefix[p0 + p] + slot - ch.len[p], ch.len[p]);
                }
                counts[static_cast<size_t>(kCountWords) * r] = n, counts[static_cast<size_t>(kCountWords) * r + 1] = nb,
                counts[static_cast<size_t>(kCountWords) * r + 2] = COATI_HIP_OK;
            }
            rc = check_round(plan, world, k, counts.data());
            if(rc != COATI_HIP_OK) return rc;
            // the exchange: every peer's transfer list, executed as memcpy into the root's landing zone; the
            // receiver's list (derived from the counts alone) must name the same blocks
            const uint64_t need = landing_plan(world, root, counts.data(), land.data());
            landing.assign(need, static_cast<char>(0xDD));
            std::vector<Transfer> recv;
            for(int r = 0; r < world; ++r)
                if(r != root) transfers_of_rank(r, counts.data(), land[static_cast<size_t>(r)], recv);
            size_t at = 0;
            for(int r = 0; r < world; ++r) {
                if(r == root) continue;
                std::vector<Transfer> send;
                transfers_of_rank(r, counts.data(), Land{0, 0, 0, 0}, send);
                const Chunk& ch = chunk[static_cast<size_t>(r)];
                const Block b{ch.scores.data(), ch.ops.data(), ch.off.data(), ch.len.data()};
                for(const Transfer& t : send) {
                    if(at >= recv.size() || recv[at].peer != r || recv[at].which != t.which || recv[at].count != t.count || recv[at].bytes_each != t.bytes_each)
                        return fail(COATI_HIP_ESTATE, "dist_simulate: round %zu: send %d of rank %d has no matching receive", k, t.which, r);
                    const uint64_t bytes = t.count * t.bytes_each;
                    if(recv[at].at + bytes > need) return fail(COATI_HIP_ESTATE, "dist_simulate: a block leaves the landing zone");
                    std::memcpy(landing.data() + recv[at].at, block_array(b, t.which), bytes);
                    ++at;
                }
            }
            if(at != recv.size()) return fail(COATI_HIP_ESTATE, "dist_simulate: round %zu: %zu receives were never sent", k, recv.size() - at);
            // root: unpack, rebase, place
            uint64_t np = 0, nb = 0;
            round_totals(plan, world, k, np, nb);
            st_scores.assign(np, 0.0f), st_off.assign(np, 0), st_len.assign(np, 0), st_ops.assign(std::max<uint64_t>(nb, 1), 0);
            const Chunk& rc_own = chunk[static_cast<size_t>(root)];
            const Block own{rc_own.scores.data(), rc_own.ops.data(), rc_own.off.data(), rc_own.len.data()};
            rc = unpack_blocks(world, root, counts.data(), own, landing.data(), land.data(), st_scores.data(), st_ops.data(), st_off.data(), st_len.data(),
                               [](void* dst, const void* src, uint64_t bytes) {
                                   std::memcpy(dst, src, bytes);
                                   return COATI_HIP_OK;
                               });
            if(rc != COATI_HIP_OK) return rc;
            rebase_offsets(world, counts.data(), st_off.data());
            place_round(plan, world, k, st_scores.data(), st_ops.data(), st_off.data(), st_len.data(), scores, ops, ops_off, ops_len);
        }
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_simulate: host allocation failed");
    }
}

}  // extern "C"
