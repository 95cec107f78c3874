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
    // device scratch: counts of one gather (2 per rank, own pair first in `mine`), and on the root the
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
    D_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_mine), 2 * sizeof(uint64_t)));
    D_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_counts), 2 * sizeof(uint64_t) * static_cast<size_t>(world)));
    guard.c = nullptr;
    *out = c;
    return COATI_HIP_OK;
}

int coati_hip_dist_rank(const coati_hip_comm_t* c) { return c != nullptr ? c->rank : -1; }
int coati_hip_dist_world(const coati_hip_comm_t* c) { return c != nullptr ? c->world : 0; }

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

int coati_hip_dist_gather(coati_hip_comm_t* c, int root, coati_hip_batch_t* batch, uint64_t* counts, float* scores,
                          uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(c == nullptr || counts == nullptr || root < 0 || root >= c->world) return fail(COATI_HIP_EINVAL, "dist_gather: bad argument");
    D_HIP(hipSetDevice(c->device));
    void *d_scores = nullptr, *d_ops = nullptr, *d_off = nullptr, *d_len = nullptr;
    uint64_t mine[2] = {0, 0};
    if(batch != nullptr) {
        D_ABI(coati_hip_viterbi_wait(batch));  // the results exist (other launches of the model keep running)
        D_ABI(coati_hip_batch_result_ptrs(batch, &d_scores, &d_ops, &mine[1], &d_off, &d_len));
        mine[0] = coati_hip_batch_pairs(batch);
    }
    // 1. everybody learns everybody's counts
    D_HIP(hipMemcpyAsync(c->d_mine, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
    D_NCCL(ncclAllGather(c->d_mine, c->d_counts, 2, ncclUint64, c->comm, c->stream));
    D_HIP(hipMemcpyAsync(counts, c->d_counts, 2 * sizeof(uint64_t) * static_cast<size_t>(c->world), hipMemcpyDeviceToHost, c->stream));
    D_HIP(hipStreamSynchronize(c->stream));
    // 2. one group of sends / receives out of (into) HBM
    struct Land {
        uint64_t scores, ops, off, len;  // byte offsets in the landing zone
    };
    std::vector<Land> land(static_cast<size_t>(c->world));
    uint64_t need = 0;
    auto take = [&](uint64_t bytes) {
        const uint64_t at = need;
        need += (bytes + 255) / 256 * 256;
        return at;
    };
    if(c->rank == root) {
        for(int r = 0; r < c->world; ++r) {
            if(r == root) continue;
            const uint64_t n = counts[2 * r], ob = counts[2 * r + 1];
            land[static_cast<size_t>(r)] = Land{take(n * sizeof(float)), take(ob), take(n * sizeof(uint64_t)), take(n * sizeof(uint32_t))};
        }
        if(need > c->land_bytes) {
            if(c->d_land != nullptr) (void)hipFree(c->d_land);
            c->d_land = nullptr;
            c->land_bytes = 0;
            D_HIP(hipMalloc(&c->d_land, need));
            c->land_bytes = need;
        }
    }
    char* base = static_cast<char*>(c->d_land);
    D_NCCL(ncclGroupStart());
    if(c->rank != root) {
        if(mine[0] > 0) {
            D_NCCL(ncclSend(d_scores, mine[0], ncclFloat32, root, c->comm, c->stream));
            D_NCCL(ncclSend(d_off, mine[0], ncclUint64, root, c->comm, c->stream));
            D_NCCL(ncclSend(d_len, mine[0], ncclUint32, root, c->comm, c->stream));
        }
        if(mine[1] > 0) D_NCCL(ncclSend(d_ops, mine[1], ncclUint8, root, c->comm, c->stream));
    } else {
        for(int r = 0; r < c->world; ++r) {
            if(r == root) continue;
            const uint64_t n = counts[2 * r], ob = counts[2 * r + 1];
            const Land& l = land[static_cast<size_t>(r)];
            if(n > 0) {
                D_NCCL(ncclRecv(base + l.scores, n, ncclFloat32, r, c->comm, c->stream));
                D_NCCL(ncclRecv(base + l.off, n, ncclUint64, r, c->comm, c->stream));
                D_NCCL(ncclRecv(base + l.len, n, ncclUint32, r, c->comm, c->stream));
            }
            if(ob > 0) D_NCCL(ncclRecv(base + l.ops, ob, ncclUint8, r, c->comm, c->stream));
        }
    }
    D_NCCL(ncclGroupEnd());
    if(c->rank != root) {
        D_HIP(hipStreamSynchronize(c->stream));  // the batch's arrays may be reused after the call
        return COATI_HIP_OK;
    }
    // 3. root: download in rank order
    uint64_t pair0 = 0, op0 = 0, total_ops = 0;
    for(int r = 0; r < c->world; ++r) total_ops += counts[2 * r + 1];
    if(ops != nullptr && ops_capacity < total_ops)
        return fail(COATI_HIP_EINVAL, "dist_gather: ops_capacity %llu < %llu", static_cast<unsigned long long>(ops_capacity),
                    static_cast<unsigned long long>(total_ops));
    for(int r = 0; r < c->world; ++r) {
        const uint64_t n = counts[2 * r], ob = counts[2 * r + 1];
        const void *s_src, *o_src, *f_src, *l_src;
        if(r == root) {
            s_src = d_scores, o_src = d_ops, f_src = d_off, l_src = d_len;
        } else {
            const Land& l = land[static_cast<size_t>(r)];
            s_src = base + l.scores, o_src = base + l.ops, f_src = base + l.off, l_src = base + l.len;
        }
        if(n > 0) {
            if(scores != nullptr) D_HIP(hipMemcpyAsync(scores + pair0, s_src, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            if(ops_off != nullptr) D_HIP(hipMemcpyAsync(ops_off + pair0, f_src, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
            if(ops_len != nullptr) D_HIP(hipMemcpyAsync(ops_len + pair0, l_src, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
        if(ob > 0 && ops != nullptr) D_HIP(hipMemcpyAsync(ops + op0, o_src, ob, hipMemcpyDeviceToHost, c->stream));
        pair0 += n;
        op0 += ob;
    }
    D_HIP(hipStreamSynchronize(c->stream));
    if(ops_off != nullptr) {  // rebase every rank's offsets into the concatenation
        pair0 = op0 = 0;
        for(int r = 0; r < c->world; ++r) {
            for(uint64_t p = 0; p < counts[2 * r]; ++p) ops_off[pair0 + p] += op0;
            pair0 += counts[2 * r];
            op0 += counts[2 * r + 1];
        }
    }
    return COATI_HIP_OK;
}

int coati_hip_dist_viterbi(coati_hip_comm_t* c, int root, coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off, float* scores, uint8_t* ops,
                           uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len) {
    if(c == nullptr || model == nullptr || a_off == nullptr || b_off == nullptr || root < 0 || root >= c->world)
        return fail(COATI_HIP_EINVAL, "dist_viterbi: bad argument");
    try {
        const int world = c->world;
        std::vector<uint64_t> bounds(static_cast<size_t>(world) + 1);
        D_ABI(coati_hip_shard_bounds(n_pairs, a_off, b_off, world, bounds.data()));
        // Chunk plan of EVERY rank (all ranks hold the same input, so all compute the same plan and the
        // collectives line up): a rank's shard in chunks of at most kChunkCells cells.
        constexpr uint64_t kChunkCells = 12000ull * 1002 * 1002;
        std::vector<std::vector<uint64_t>> cuts(static_cast<size_t>(world));  // per rank: chunk boundaries (pair indices)
        size_t rounds = 0;
        for(int r = 0; r < world; ++r) {
            auto& cut = cuts[static_cast<size_t>(r)];
            uint64_t p = bounds[static_cast<size_t>(r)], cells = 0;
            cut.push_back(p);
            for(; p < bounds[static_cast<size_t>(r) + 1]; ++p) {
                const uint64_t w = (a_off[p + 1] - a_off[p]) * (b_off[p + 1] - b_off[p]);
                if(p > cut.back() && cells + w > kChunkCells) {
                    cut.push_back(p);
                    cells = 0;
                }
                cells += w;
            }
            if(cut.back() != bounds[static_cast<size_t>(r) + 1] || cut.size() == 1) cut.push_back(bounds[static_cast<size_t>(r) + 1]);
            rounds = std::max(rounds, cut.size() - 1);
        }
        // where a chunk's results go in the root's arrays: pairs keep their input order
        std::vector<uint64_t> op_prefix(n_pairs + 1, 0);
        for(uint64_t p = 0; p < n_pairs; ++p) op_prefix[p + 1] = op_prefix[p] + (a_off[p + 1] - a_off[p]) + (b_off[p + 1] - b_off[p]);
        if(c->rank == root && ops != nullptr && ops_capacity < op_prefix[n_pairs])
            return fail(COATI_HIP_EINVAL, "dist_viterbi: ops_capacity too small");
        const auto& mine = cuts[static_cast<size_t>(c->rank)];
        auto make = [&](size_t k, coati_hip_batch_t** out) -> int {
            *out = nullptr;
            if(k + 1 >= mine.size() || mine[k + 1] == mine[k]) return COATI_HIP_OK;
            D_ABI(coati_hip_batch_create(model, mine[k + 1] - mine[k], a_cat, a_off + mine[k], b_cat, b_off + mine[k], out));
            D_ABI(coati_hip_viterbi_launch(*out));
            return COATI_HIP_OK;
        };
        // per-round staging on the root: ranks' chunks arrive concatenated in rank order
        std::vector<uint64_t> counts(2 * static_cast<size_t>(world));
        std::vector<float> st_scores;
        std::vector<uint8_t> st_ops;
        std::vector<uint64_t> st_off;
        std::vector<uint32_t> st_len;
        coati_hip_batch_t *cur = nullptr, *next = nullptr;
        int rc = make(0, &cur);
        for(size_t k = 0; k < rounds && rc == COATI_HIP_OK; ++k) {
            rc = make(k + 1, &next);  // the next chunk computes while this one is gathered
            if(rc != COATI_HIP_OK) break;
            if(c->rank == root) {
                uint64_t np = 0, nb = 0;
                for(int r = 0; r < world; ++r) {
                    const auto& cut = cuts[static_cast<size_t>(r)];
                    if(k + 1 < cut.size()) {
                        np += cut[k + 1] - cut[k];
                        nb += op_prefix[cut[k + 1]] - op_prefix[cut[k]];
                    }
                }
                st_scores.resize(np), st_off.resize(np), st_len.resize(np), st_ops.resize(std::max<uint64_t>(nb, 1));
            }
            rc = coati_hip_dist_gather(c, root, cur, counts.data(), st_scores.data(), st_ops.data(), st_ops.size(), st_off.data(),
                                       st_len.data());
            if(cur != nullptr) coati_hip_batch_destroy(cur);
            cur = next;
            next = nullptr;
            if(rc != COATI_HIP_OK || c->rank != root) continue;
            uint64_t at_p = 0, at_b = 0;
            for(int r = 0; r < world; ++r) {
                const auto& cut = cuts[static_cast<size_t>(r)];
                if(k + 1 >= cut.size()) continue;
                const uint64_t p0 = cut[k], n = cut[k + 1] - cut[k], nb = op_prefix[cut[k + 1]] - op_prefix[cut[k]];
                if(counts[2 * r] != n || counts[2 * r + 1] != nb) {
                    rc = fail(COATI_HIP_ESTATE, "dist_viterbi: rank %d sent %llu pairs / %llu op bytes in round %zu, the plan says %llu / %llu", r,
                              static_cast<unsigned long long>(counts[2 * r]), static_cast<unsigned long long>(counts[2 * r + 1]), k,
                              static_cast<unsigned long long>(n), static_cast<unsigned long long>(nb));
                    break;
                }
                if(scores != nullptr) std::memcpy(scores + p0, st_scores.data() + at_p, n * sizeof(float));
                if(ops_len != nullptr) std::memcpy(ops_len + p0, st_len.data() + at_p, n * sizeof(uint32_t));
                if(ops != nullptr) std::memcpy(ops + op_prefix[p0], st_ops.data() + at_b, nb);
                if(ops_off != nullptr)
                    for(uint64_t p = 0; p < n; ++p) ops_off[p0 + p] = st_off[at_p + p] - at_b + op_prefix[p0];
                at_p += n;
                at_b += nb;
            }
        }
        if(cur != nullptr) coati_hip_batch_destroy(cur);
        if(next != nullptr) coati_hip_batch_destroy(next);
        return rc;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "dist_viterbi: host allocation failed");
    }
}

}  // extern "C"
