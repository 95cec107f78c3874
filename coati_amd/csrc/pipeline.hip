// pipeline.hip -- coati_hip_viterbi_batch: viterbi_mem + traceback_viterbi over any number of pairs that arrive in host
// memory and leave to host memory (the batched counterpart of the reference's one pair per process,
// src/lib/align_marginal.cc:69-80, utils.cc:809-812).  Two forms: ONE persistent kernel fed chunk by chunk
// (viterbi_batch_stream) and one launch per chunk over three slots.
#include "abi_internal.hpp"

#include <future>
#include <string>
#include <utility>

using namespace coati_hip_abi;

namespace {
// Is `p` page-locked host memory HIP knows about (hipHostMalloc / hipHostRegister / coati_hip_host_alloc)?
bool is_pinned_host(const void* p) {
    if(p == nullptr) return false;
    if(host_block_contains(p)) return true;  // (coati_hip_host_alloc's own blocks: no question to the runtime)
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
// Everything a streamed call allocates -- streams, control blocks, events, the wavefronts' checkpoint slots (sized for
// single-strip pairs of up to `longest_single` ancestor positions) and the stream slots a call of `total_cells` DP
// cells can use -- so that it can be done AHEAD of the call (coati_hip_model_prepare: a fresh process pays ~100 ms for
// these GBs, which an embedder can spend while it is still reading its input).  COATI_HIP_ESTATE: an allocation failed
// (not an error of a call: the chunk pipeline serves it).
// Page-locked staging a stream slot of this call should have: room for the largest chunk the cutter makes (3 units of
// cells) in both directions, and for the call's longest pair, between 4 and 48 MB.  (Page-locking is the slow part of a
// fresh process' bring-up -- ~0.25 ms per MB: nine 48 MB blocks cost coati-alignpair --batch 110 of its 360 ms on
// 10 000 pairs of 1 kb, which need a third of that.)
uint64_t stream_staging_need(uint64_t n_pairs, long double total_cells, uint64_t total_seq_bytes, uint64_t longest_pair_bytes, long double unit_cells,
                             bool in_pinned = false, bool out_pinned = false) {
    // (page-locked caller arrays are read and written by the copy engine directly: only descriptors, work items and the
    // three short result arrays go through the block -- ~160 bytes per pair; a model's first call on page-locked arrays
    // spent 15 of its 23 ms page-locking staging it never used)
    const long double mean_cells = total_cells / std::max<long double>(1, static_cast<long double>(n_pairs));
    const long double chunk_pairs = std::min<long double>(static_cast<long double>(n_pairs), 3 * unit_cells / std::max<long double>(1, mean_cells) + 1);
    const long double seq_per_pair = static_cast<long double>(total_seq_bytes) / std::max<long double>(1, static_cast<long double>(n_pairs));
    const long double per_pair = (in_pinned ? 0.0L : 1.5L) * seq_per_pair + (out_pinned ? 0.0L : 1.5L) * seq_per_pair + 160;
    const uint64_t est = static_cast<uint64_t>(1.5L * chunk_pairs * per_pair) + (in_pinned && out_pinned ? 0 : 3 * longest_pair_bytes) + (2ull << 20);
    const uint64_t step = 4ull << 20;
    return std::min<uint64_t>(kStreamSlotStaging, std::max<uint64_t>(step, (est + step - 1) / step * step));
}
// ... and the first three slots hold the call's first three chunks (1/2, 1 and 2 units of cells against 3 later on):
// a sixth, a third and two thirds of the block, in whole 4 MB (a later lap only cuts smaller chunks for them)
struct StagingNeed {
    uint64_t full = 0;   // a slot that takes 3-unit chunks
    uint64_t floor = 0;  // no slot below this: the call's longest pair must fit every slot (a chunk's first pair is taken unseen)
};
uint64_t stream_slot_staging(const StagingNeed& need, int q) {
    const uint64_t step = 4ull << 20;
    const uint64_t part = std::max(need.floor, q == 0 ? need.full / 6 : q == 1 ? need.full / 3 : q == 2 ? 2 * need.full / 3 : need.full);
    return std::min(need.full, std::max<uint64_t>(step, (part + step - 1) / step * step));
}
StagingNeed stream_staging(uint64_t n_pairs, long double total_cells, uint64_t total_seq_bytes, uint64_t longest_pair_bytes, bool in_pinned, bool out_pinned) {
    StagingNeed need;
    need.full = stream_staging_need(n_pairs, total_cells, total_seq_bytes, longest_pair_bytes, 1000.0L * 1002 * 1002, in_pinned, out_pinned);
    need.floor = std::min<uint64_t>(need.full, 4 * longest_pair_bytes + (2ull << 20));
    return need;
}
// The streamed call's last chunk (viterbi_batch_stream: "ONE last chunk"): at most kBigTailUnits units of cells, its pairs cut
// into row parts that keep their own checkpoints (1.09 MB per 1 kb pair) -- one workspace of ~3 GB, kept on the model.
// (COATI_HIP_STREAM_TAIL_UNITS: tuning / A-B switch, tenths of a unit)
// (COATI_HIP_STREAM_TAIL_UNITS, tenths, an experiment switch: clamped to 0.5 ... 8 units)
inline long double big_tail_units() {
    const int x10 = env_options().stream_tail_units_x10;
    return x10 > 0 ? std::min(80, std::max(5, x10)) / 10.0L : 2.6L;
}
int stream_big_tail_parts() {  // 0 = off
    const int v = env_options().stream_parts;
    return v < 0 ? 3 : (v >= 22 && v <= 28) ? v - 20 : 0;
}
uint64_t stream_big_tail_bytes(uint64_t wave_slot_bytes) {
    // (+ 1/8: the cutter's own safety margin on a chunk's workspace, + the chunk's descriptors and sequences: without them a
    // call whose remainder was just under kBigTailUnits left a last chunk of a few WHOLE pairs behind the row parts)
    return std::min<uint64_t>(8ull << 30, static_cast<uint64_t>(big_tail_units() * 1040) * ((wave_slot_bytes + 4096) / 8 * 9 + (32u << 10)) + (64ull << 20));
}
void stream_reserve_big_tail(coati_hip_model_t* model, uint64_t wave_slot_bytes) {  // (a failed allocation: the call runs without row parts)
    const uint64_t big = stream_big_tail_bytes(wave_slot_bytes);
    if(model->stream_tail_bytes >= big && model->stream_tail_arena[0] != nullptr) return;
    for(void*& t : model->stream_tail_arena) {
        if(t != nullptr) (void)hipFree(t);
        t = nullptr;
    }
    model->stream_tail_bytes = 0;
    if(hipMalloc(&model->stream_tail_arena[0], big) == hipSuccess)
        model->stream_tail_bytes = big;
    else
        (void)hipGetLastError();
}
int stream_reserve(coati_hip_model_t* model, uint64_t longest_single, long double total_cells, const StagingNeed& staging_need, uint64_t* wave_slot_bytes_out,
                   int* n_slots_out, long double* unit_cells_out) {
    constexpr int kSlots = kCkStreamSlots;
    for(int q = 1; q <= 2; ++q)
        if(model->slots[q].stream == nullptr && hipStreamCreateWithFlags(&model->slots[q].stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            return COATI_HIP_ESTATE;
        }
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
    const bool alloc_timing = env_options().pipe_timing;
    auto t_alloc = std::chrono::steady_clock::now();
    auto alloc_stage = [&](const char* what, uint64_t bytes) {
        if(!alloc_timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "viterbi_batch[stream]: %s (%.1f MB) %.2f ms\n", what, bytes / 1048576.0, std::chrono::duration<double, std::milli>(t - t_alloc).count());
        t_alloc = t;
    };
    if(model->stream_waves_bytes < waves_bytes) {
        if(model->d_stream_waves != nullptr) (void)hipFree(model->d_stream_waves);
        model->d_stream_waves = nullptr;
        model->stream_waves_bytes = 0;
        if(!soft(hipMalloc(&model->d_stream_waves, waves_bytes))) return COATI_HIP_ESTATE;
        model->stream_waves_bytes = waves_bytes;
        alloc_stage("wavefront slots + scratch allocated", waves_bytes);
    }
    // slots: fixed sizes (nothing can grow while the kernel runs; the chunks are cut to fit).  Workspace: room for
    // the largest pair this form accepts (kStreamPairCells) with its own checkpoints; staging block
    // [what goes up | short result arrays, and the ops when the caller's array is pageable]
    constexpr uint64_t kSlotArena = kStreamSlotArena;
    const StagingNeed kSlotStaging{std::min<uint64_t>(kStreamSlotStaging, std::max<uint64_t>(staging_need.full, 4ull << 20)), staging_need.floor};
    // How many slots can this call use?  The chunk targets below in cells: 1/2, 1, 2, then 3 units, and 1 unit each
    // once four units are left.  A one-shot process (coati-alignpair --batch: ~0.1 ms per MB of fresh hipMalloc /
    // hipHostMalloc, 12 slots are 2.9 GB) allocates what its input needs; a second call on the model takes the rest.
    long double kUnitCells = 1000.0L * 1002 * 1002;
    if(env_options().stream_unit >= 1.0L) kUnitCells = env_options().stream_unit;  // (COATI_HIP_STREAM_UNIT)
    int n_slots = kSlots;
    if(model->stream_calls == 0) {
        int est = 0;
        for(long double done = 0; done < total_cells && est < kSlots; ++est)
            done += est == 0 ? kUnitCells / 2 : (est == 1 || total_cells - done <= 4 * kUnitCells) ? kUnitCells : est == 2 ? 2 * kUnitCells : 3 * kUnitCells;
        n_slots = std::min(kSlots, std::max(3, est + 1));  // (+1: a memory-bound cut may add a chunk; fewer slots than chunks only means reuse)
        for(int q = 0; q < kSlots; ++q)
            if(model->sslots[q].arena_bytes >= kSlotArena && model->sslots[q].pinned_bytes >= stream_slot_staging(kSlotStaging, q)) n_slots = std::max(n_slots, q + 1);
    }
    // (a fresh process page-locks at ~4 GB/s and maps fresh HBM at ~10 GB/s: the slots are made side by side on the
    // model's helper threads -- the first call of a model is mostly this)
    if(!model->helpers) model->helpers = std::make_unique<HelperPool>(3);
    auto make_slot = [model, kSlotStaging](int q) -> bool {
        auto& ss = model->sslots[q];
        if(hipSetDevice(model->device) != hipSuccess) return false;
        if(ss.arena_bytes < kSlotArena) {
            if(ss.arena != nullptr) (void)hipFree(ss.arena);
            ss.arena = nullptr;
            ss.arena_bytes = 0;
            if(hipMalloc(&ss.arena, kSlotArena) != hipSuccess) return false;
            ss.arena_bytes = kSlotArena;
        }
        if(const uint64_t want = stream_slot_staging(kSlotStaging, q); ss.pinned_bytes < want) {
            if(ss.pinned != nullptr) (void)hipHostFree(ss.pinned);
            ss.pinned = nullptr;
            ss.pinned_bytes = 0;
            if(hipHostMalloc(&ss.pinned, want, hipHostMallocDefault) != hipSuccess) return false;
            ss.pinned_bytes = want;
        }
        return true;
    };
    {
        std::vector<std::future<bool>> made;
        bool all = true;
        auto ready = [&](int q) { return model->sslots[q].arena_bytes >= kSlotArena && model->sslots[q].pinned_bytes >= stream_slot_staging(kSlotStaging, q); };
        for(int q = 0; q < n_slots; ++q)
            if(q % 4 != 3 && !ready(q) && (env_options().stream_helpers & 4) != 0) made.push_back(model->helpers->submit([make_slot, q] { return make_slot(q); }));
        for(int q = 0; q < n_slots; ++q)  // (this thread takes its share)
            if((q % 4 == 3 || (env_options().stream_helpers & 4) == 0) && !ready(q)) all = make_slot(q) && all;
        for(auto& m : made) all = m.get() && all;
        if(!all) {
            (void)hipGetLastError();
            return COATI_HIP_ESTATE;
        }
    }
    {
        uint64_t made_bytes = 0;
        for(int q = 0; q < n_slots; ++q) made_bytes += model->sslots[q].arena_bytes + model->sslots[q].pinned_bytes;
        alloc_stage("stream slots ready", made_bytes);
    }
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
    *wave_slot_bytes_out = wave_slot_bytes;
    *n_slots_out = n_slots;
    *unit_cells_out = kUnitCells;
    return COATI_HIP_OK;
}

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
                         bool out_pinned, long double total_cells, uint64_t longest_single, const StagingNeed& staging_need, std::chrono::steady_clock::time_point t_call) {
    constexpr int kSlots = kCkStreamSlots;
    uint64_t wave_slot_bytes = 0;
    int n_slots = 0;
    long double kUnitCells = 0;
    if(const int rc_res = stream_reserve(model, longest_single, total_cells, staging_need, &wave_slot_bytes, &n_slots, &kUnitCells); rc_res != COATI_HIP_OK)
        return rc_res;
    hipStream_t kernel_stream = model->stream, up_stream = model->slots[1].stream, down_stream = model->slots[2].stream;
    const uint64_t host_bytes = ck_stream_host_bytes();
    auto soft = [](hipError_t e) {  // an allocation that fails here is not an error of the call
        if(e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    constexpr uint64_t kSlotArena = kStreamSlotArena;
    auto out_bytes_of = [&](uint64_t n, uint64_t ops_bytes) { return stream_out_bytes(n, ops_bytes, out_pinned); };
    auto staging_of = [&](const ChunkNeed& nd, uint64_t n) { return stream_staging_bytes(nd, n, in_pinned, out_pinned); };
    // the call's last chunks -- everything behind the first round of 4 096 wavefronts, up to ~7 500 pairs of 1 kb -- are
    // cut into row parts (finer items for the ragged end of the kernel, as a resident batch's later pairs are, abi.hip
    // "the ragged end"): their pairs keep their checkpoints, ~1.1 MB per 1 kb pair -- six larger workspaces of ~1 250 pairs
    const uint64_t tail_bytes = std::min<uint64_t>(3ull << 30, std::max<uint64_t>(kSlotArena, 1250 * (wave_slot_bytes + 4096) + (64ull << 20)));
    // OFF by default since the banded checkpoints (round 3): a tail chunk has ~1 000 pairs, so part p of a pair is drawn only
    // ~1 000 tickets after part p-1 -- by one of 4 096 wavefronts that then waits for it -- and the three parts of a pair
    // run one after the other anyway; with the leaner fill that wait costs more than the finer items save (same process,
    // page-locked arrays: 10 000 pairs 5.15 -> 4.92 ms, 40 000 pairs 16.55 -> 15.93 ms without them).  A resident batch
    // keeps its row parts: there a part's predecessor is >= 4 096 tickets back.  COATI_HIP_STREAM_PARTS=1 turns them on
    // (tests, A/B).
    const bool tail_parts_on = env_options().stream_parts == 1;
    // Round 4: ONE last chunk instead -- everything behind the first ~7 400 pairs' worth of a call, at most kBigTailUnits
    // units (2 600 pairs of 1 kb), cut into three row parts in part-major order: a part's predecessor is then ~2 500 tickets
    // back, which the 4 096 wavefronts take ~1 ms to draw while a part takes ~0.6 ms.  Page-locked arrays, same process:
    // 10 000 pairs 4.93 -> 4.65 ms, 12 000 6.57 -> 5.95, 20 000 9.60 -> 9.42, 40 000 15.77 -> 15.72; a tail of 3 600 or
    // 4 600 pairs loses again (tools/stream_tail_ab.py, profiles/r04/stream_tail_ab.txt).  Default; COATI_HIP_STREAM_PARTS=0
    // = no row parts, 22 .. 28 = 2 .. 8 parts.
    const int big_tail_parts = stream_big_tail_parts();
    const bool big_tail_on = big_tail_parts != 0;
    // (a model's FIRST call runs without either: ~3 GB of fresh hipMalloc cost a one-shot process 15-30 ms, a hundred times what
    // the row parts save it)
    const bool want_tails = tail_parts_on && (model->stream_calls > 0 || total_cells >= 30 * kUnitCells);
    if(big_tail_on && total_cells >= 6 * kUnitCells && model->stream_calls > 0) stream_reserve_big_tail(model, wave_slot_bytes);
    if(want_tails && (model->stream_tail_bytes < tail_bytes || model->stream_tail_arena[coati_hip_model::kStreamTails - 1] == nullptr)) {
        for(void*& t : model->stream_tail_arena) {
            if(t != nullptr) (void)hipFree(t);
            t = nullptr;
        }
        model->stream_tail_bytes = 0;
        bool ok = true;
        for(void*& t : model->stream_tail_arena) ok = ok && soft(hipMalloc(&t, tail_bytes));
        if(ok) model->stream_tail_bytes = tail_bytes;  // (else: no row parts in this call)
    }
    if((env_options().stream_helpers & 8) != 0) {
        std::fprintf(stderr, "viterbi_batch[stream]: begin (%llu pairs, %d slots) host words %p ctl %p waves %p + %zx table %p; inputs %p %p outputs %p %p %p %p\n",
                     static_cast<unsigned long long>(n_pairs), n_slots, model->h_stream, model->d_stream_ctl, model->d_stream_waves, model->stream_waves_bytes,
                     static_cast<void*>(model->d_table), static_cast<const void*>(a_cat), static_cast<const void*>(b_cat), static_cast<void*>(scores),
                     static_cast<void*>(ops), static_cast<void*>(ops_off), static_cast<void*>(ops_len));
        for(int q = 0; q < n_slots; ++q)
            std::fprintf(stderr, "viterbi_batch[stream]:   slot %d arena %p + %zx staging %p + %zx\n", q, model->sslots[q].arena, model->sslots[q].arena_bytes,
                         model->sslots[q].pinned, model->sslots[q].pinned_bytes);
    }
    // Page-locked result arrays: the walks store a pair's ops straight into the caller's array (write-through at system scope, as
    // they store them into the workspace otherwise; the chunk's completion word follows its pairs' acknowledged stores) -- the
    // chunks of a call's end complete within 0.3 ms of each other, and 17 MB of downloads behind the kernel's last item were
    // 0.2-0.25 ms of a 10 000-pair call.  COATI_HIP_STREAM_HELPERS bit 16 (A/B).
    // Pageable result arrays (bit 32): the same into the slot's page-locked staging block, behind the three short arrays' room --
    // what is left of the download is those three arrays, the helpers copy the ops out of the block as before.
    // The three short arrays go the same way (a pair's score, ops offset and ops length: three stores by one lane), so that a chunk's
    // completion word is all the host waits for: no download at all.
    uint8_t* ops_dev = nullptr;
    float* scores_dev = nullptr;
    uint64_t* off_dev = nullptr;
    uint32_t* len_dev = nullptr;
    bool full_direct = false;  // every array the caller passed is written by the kernel
    if(out_pinned && ops != nullptr && (env_options().stream_helpers & 16) != 0) {
        auto dev_of = [&](void* p) -> void* {
            void* dp = nullptr;
            return p != nullptr && soft(hipHostGetDevicePointer(&dp, p, 0)) ? dp : nullptr;
        };
        ops_dev = static_cast<uint8_t*>(dev_of(ops));
        scores_dev = static_cast<float*>(dev_of(scores)), off_dev = static_cast<uint64_t*>(dev_of(ops_off)), len_dev = static_cast<uint32_t*>(dev_of(ops_len));
        full_direct = ops_dev != nullptr && (scores == nullptr || scores_dev != nullptr) && (ops_off == nullptr || off_dev != nullptr) &&
                      (ops_len == nullptr || len_dev != nullptr);
    }
    const bool ops_into_staging = !out_pinned && ops != nullptr && (env_options().stream_helpers & 32) != 0;
    uint8_t* stage_dev[kSlots] = {};  // (device-visible addresses of the slots' staging blocks, asked for once per call)
    void* hs = model->h_stream;
    std::memset(hs, 0, host_bytes);
    ck_stream_host_set_slots(hs, static_cast<uint32_t>(n_slots));
    void* hs_dev = nullptr;
    if(!soft(hipHostGetDevicePointer(&hs_dev, hs, 0))) return COATI_HIP_ESTATE;
    hipEvent_t* uploaded = model->stream_events + kSlots;  // (recorded by whichever thread planned the chunk, right behind its copies)
    hipEvent_t* copied = model->stream_events;
    // the control block starts zeroed (before the launch a fill kernel may run)
    uint32_t* wave_ck = static_cast<uint32_t*>(model->d_stream_waves);
    uint32_t* wave_scratch = reinterpret_cast<uint32_t*>(static_cast<char*>(model->d_stream_waves) + static_cast<uint64_t>(ck_scratch_waves()) * wave_slot_bytes);
    hipError_t e0 = hipMemsetAsync(model->d_stream_ctl, 0, ck_stream_ctl_bytes(), kernel_stream);
    if(e0 == hipSuccess)
        e0 = launch_viterbi_ck_stream(model->d_table, model->k, model->n_tables == 1, model->d_stream_ctl, hs_dev, wave_ck, wave_slot_bytes / 4,
                                      wave_scratch, model->ck_band, kernel_stream);
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
        uint64_t kernel_ops_at = 0;  // the kernel stored the chunk's ops into the staging block, this far behind out_stage (0: it did not)
        bool no_download = false;    // ... and the short arrays too (into the staging block, or all of it into the caller's page-locked arrays)
        std::future<void> unstage;  // the chunk's results are being copied out of the staging block by a helper thread
        double t_d2h = 0, t_copied = 0;  // (COATI_HIP_PIPE_TIMING)
    };
    InFlight fl[kSlots];
    struct JoinUnstage {  // (no helper may still write the caller's arrays, or read this frame, when the call returns)
        InFlight (&fl)[kSlots];
        ~JoinUnstage() {
            for(InFlight& f : fl)
                if(f.unstage.valid()) f.unstage.wait();
        }
    } join_unstage{fl};
    if(!model->helpers) model->helpers = std::make_unique<HelperPool>(3);
    int rc = COATI_HIP_OK;
    const bool pipe_timing = env_options().pipe_timing;
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
        const bool ops_direct = ops_dev == nullptr && out_pinned && ops != nullptr && c.ops_bytes > kMinDmaBytes;
        f.ops_staged = ops_dev == nullptr && ops != nullptr && c.ops_bytes > 0 && !ops_direct;
        const uint64_t bytes = std::max<uint64_t>(group + (f.ops_staged && f.kernel_ops_at == 0 ? c.ops_bytes : 0), kMinDmaBytes + 256);
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
                if(f.no_download) {  // the kernel's stores were acknowledged before it set the word
                    f.d2h_submitted = true;
                    f.out_stage = static_cast<char*>(model->sslots[q].pinned) + f.out_off;
                    f.ops_staged = !full_direct && ops != nullptr && f.chunk.ops_bytes > 0;
                } else {
                    const hipError_t e = submit_d2h(f, q);
                    if(e != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
                }
                if(pipe_timing) f.t_d2h = t_ms();
            }
            if(!f.unstage.valid() && !f.no_download) {
                const hipError_t qd = hipEventQuery(copied[q]);
                if(qd == hipErrorNotReady) continue;
                if(qd != hipSuccess) return fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(qd));
                if(pipe_timing) f.t_copied = t_ms();
            }
            const PipeChunk& c = f.chunk;
            const uint64_t n = c.p1 - c.p0;
            // out of the staging block into the caller's arrays: megabytes when the ops array is pageable (a chunk of 2 000
            // pairs: 4 MB, 0.15-0.2 ms on one thread, and the chunks of a call's end complete together) -- on the helpers
            auto unstage = [scores, ops, ops_off, ops_len, c, n, at0 = f.out_stage, staged = f.ops_staged, kernel_ops_at = f.kernel_ops_at]() {
                char* at = at0;
                if(scores != nullptr) std::memcpy(scores + c.p0, at, n * sizeof(float));
                at += (n * sizeof(float) + 255) / 256 * 256;
                if(ops_off != nullptr) {
                    const uint64_t* src = reinterpret_cast<const uint64_t*>(at);
                    for(uint64_t p = 0; p < n; ++p) ops_off[c.p0 + p] = src[p] + c.ops_base;
                }
                at += (n * sizeof(uint64_t) + 255) / 256 * 256;
                if(ops_len != nullptr) std::memcpy(ops_len + c.p0, at, n * sizeof(uint32_t));
                at += (n * sizeof(uint32_t) + 255) / 256 * 256;
                if(staged) std::memcpy(ops + c.ops_base, kernel_ops_at != 0 ? at0 + kernel_ops_at : at, c.ops_bytes);
            };
            if(f.no_download && full_direct) {  // (nothing to copy: the results are where the caller wants them)
            } else if(f.unstage.valid()) {
                if(f.unstage.wait_for(std::chrono::seconds(0)) != std::future_status::ready) continue;
                f.unstage.get();
            } else if(f.ops_staged && c.ops_bytes > (256u << 10) && (env_options().stream_helpers & 2) != 0) {
                f.unstage = model->helpers->submit(unstage);
                continue;
            } else {
                unstage();
            }
            if(pipe_timing && f.no_download)
                std::fprintf(stderr, "viterbi_batch[stream]: chunk %u (%llu pairs) complete %.2f ms after the kernel started, seen at %.2f, on the host at %.2f ms (stored by the kernel%s)\n",
                             f.chunk_no, static_cast<unsigned long long>(n), ck_stream_host_done_ms(hs, q), f.t_d2h, t_ms(), full_direct ? "" : " into the staging block");
            else if(pipe_timing)
                std::fprintf(stderr, "viterbi_batch[stream]: chunk %u (%llu pairs) complete %.2f ms after the kernel started, on the host at %.2f ms (download %.2f .. %.2f)\n",
                             f.chunk_no, static_cast<unsigned long long>(n), ck_stream_host_done_ms(hs, q), t_ms(), f.t_d2h, f.t_copied);
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
    // A chunk, cut and ready to be planned.  The cutter runs AHEAD of the loop below: while this thread plans and uploads
    // chunk ci, the model's three helper threads plan chunks ci + 1 .. ci + 3 into their (free: first lap over the slots) slots --
    // one thread plans ~10 pairs per microsecond, so that the chip, which wants 4 096 pairs before every wavefront has one,
    // was full only ~0.6-1.0 ms after a 10 000-pair call had started (round 4; profiles/r04/stream_timeline_10000.txt).
    struct Ahead {
        bool valid = false, fits = true, tail = false;
        int slot = 0;
        PipeChunk c;
        ChunkNeed nd;
        uint64_t out_off = 0;
        BatchOpts bo;
        coati_hip_batch_t* batch = nullptr;
        double t_plan0 = 0, t_plan1 = 0;  // (COATI_HIP_PIPE_TIMING)
        std::future<std::pair<int, std::string>> task;
    };
    Ahead ahead[kSlots];
    auto cut_chunk = [&](size_t ci, int q, Ahead& a) {
        coati_hip_model::StreamSlot& sl = model->sslots[q];
        const bool big_tail = big_tail_on && ci >= 2 && cells_done >= 4.1L * kUnit && total_cells - cells_done <= big_tail_units() * kUnit && tails_used == 0 &&
                              model->stream_tail_bytes >= stream_big_tail_bytes(wave_slot_bytes) && model->stream_tail_arena[0] != nullptr;
        const long double target = big_tail ? total_cells : ci == 0 ? kUnit / 2 : (ci == 1 || total_cells - cells_done <= 4 * kUnit) ? kUnit : ci == 2 ? 2 * kUnit : 3 * kUnit;
        // row parts, in the large workspaces: the chunks behind the first 4 100 pairs' worth of cells, while at most
        // 8 300 pairs' worth are left
        const bool tail = big_tail || (ci >= 2 && cells_done >= 4.1L * kUnit && total_cells - cells_done <= 8.3L * kUnit &&
                                       tails_used < coati_hip_model::kStreamTails && model->stream_tail_bytes != 0 && tail_parts_on);
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
        a.valid = true;
        a.slot = q;
        a.tail = tail;
        a.batch = nullptr;
        a.c = PipeChunk{p0, p1, ops_base, nd.ops};
        a.nd = nd;
        p0 = p1;
        ops_base += nd.ops;
        cells_done += static_cast<long double>(nd.cells);
        const uint64_t n = a.c.p1 - a.c.p0;
        // (the cutter takes the first pair of a chunk unseen: one that does not fit a slot ends the streamed form --
        // the kernel is closed below and the chunk pipeline, whose workspaces grow, does the call; never compute the
        // staging split from an unchecked subtraction)
        a.fits = !(staging_of(nd, n) > sl.pinned_bytes || out_bytes_of(n, a.c.ops_bytes) > sl.pinned_bytes ||
                   nd.arena_streamed(tail) + nd.arena_streamed(tail) / 8 + (1u << 20) > arena_bytes);
        if(!a.fits) return;
        a.out_off = (sl.pinned_bytes - out_bytes_of(n, a.c.ops_bytes)) / 256 * 256;
        a.bo = BatchOpts{};
        a.bo.stream = up_stream;
        a.bo.arena = arena;
        a.bo.arena_bytes = arena_bytes;
        a.bo.staging = static_cast<char*>(sl.pinned);
        a.bo.staging_bytes = a.out_off;
        a.bo.seqs_pinned = in_pinned;
        a.bo.force_ck = true;
        a.bo.force_w_main = kW;  // (a small chunk is not a small batch: no narrowed strips)
        a.bo.device_validates = true;
        if(tail) {
            a.bo.tail_parts = big_tail_on ? static_cast<uint32_t>(big_tail_parts) : 3;
            ++tails_used;
        }
        a.bo.wave_slot_dwords = wave_slot_bytes / 4;
    };
    auto plan_chunk = [&](Ahead& a) -> std::pair<int, std::string> {  // (this thread or a helper: the error text is thread-local)
        if(pipe_timing) a.t_plan0 = t_ms();
        int r = batch_create_impl(model, a.c.p1 - a.c.p0, a_cat, a_off + a.c.p0, b_cat, b_off + a.c.p0, nullptr, &a.bo, &a.batch);
        if(r == COATI_HIP_OK) {  // the chunk's data (with its zeroed progress words) is on its way; once it is in HBM the kernel may know
            if(const hipError_t e = hipEventRecord(uploaded[a.slot], up_stream); e != hipSuccess) r = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(e));
        }
        if(pipe_timing) a.t_plan1 = t_ms();
        return {r, r == COATI_HIP_OK ? std::string() : std::string(coati_hip_last_error())};
    };
    struct JoinAhead {  // whatever ends the loop: no helper thread is left running on this frame, no planned chunk is leaked
        Ahead (&ahead)[kSlots];
        ~JoinAhead() {
            for(Ahead& a : ahead) {
                if(a.task.valid()) {
                    try {
                        (void)a.task.get();
                    } catch(...) {  // (a helper's std::bad_alloc: the call is already on its way out)
                    }
                }
                if(a.valid && a.batch != nullptr) coati_hip_batch_destroy(a.batch);
            }
        }
    } join_ahead{ahead};
    for(size_t ci = 0; rc == COATI_HIP_OK; ++ci) {
        const int q = static_cast<int>(ci % static_cast<size_t>(n_slots));
        InFlight& f = fl[q];
        const double t_begin = t_ms();
        rc = wait_free(q);
        if(rc != COATI_HIP_OK) break;
        Ahead& a = ahead[q];
        if(!a.valid) {
            if(p0 >= n_pairs) break;
            cut_chunk(ci, q, a);
        }
        if(!a.fits) {
            a.valid = false;
            rc = kRedo;
            break;
        }
        const PipeChunk c = a.c;
        const ChunkNeed nd = a.nd;
        const uint64_t n = c.p1 - c.p0, out_off = a.out_off;
        const double t_cut = t_ms();
        // this chunk first (its copies enter the upload stream ahead of the helpers': the kernel is waiting for THIS one),
        // then the look-ahead, then -- if a helper had this chunk -- its result
        std::pair<int, std::string> planned{COATI_HIP_OK, std::string()};
        const bool was_ahead = a.task.valid();
        if(!was_ahead) planned = plan_chunk(a);
        if(!tail_parts_on && planned.first == COATI_HIP_OK && (env_options().stream_helpers & 1) != 0) {
            for(size_t k = 1; k <= 3 && ci + k < static_cast<size_t>(n_slots) && p0 < n_pairs; ++k) {
                const int q2 = static_cast<int>((ci + k) % static_cast<size_t>(n_slots));
                Ahead& a2 = ahead[q2];
                if(a2.valid || fl[q2].batch != nullptr) continue;
                cut_chunk(ci + k, q2, a2);
                if(a2.fits) a2.task = model->helpers->submit([&plan_chunk, &a2]() { return plan_chunk(a2); });
            }
        }
        if(was_ahead) planned = a.task.get();
        {
            rc = planned.first;
            f.batch = a.batch;
            a.batch = nullptr;
            a.valid = false;
            if(rc != COATI_HIP_OK) (void)fail(rc, "%s", planned.second.c_str());
        }
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
        hipError_t e = hipSuccess;
        CkStreamDirect direct;
        f.kernel_ops_at = 0;
        f.no_download = false;
        if(full_direct) {
            direct.ops = ops_dev + c.ops_base;
            direct.scores = scores_dev != nullptr ? scores_dev + c.p0 : nullptr;
            direct.ops_start = off_dev != nullptr ? off_dev + c.p0 : nullptr;
            direct.ops_len = len_dev != nullptr ? len_dev + c.p0 : nullptr;
            direct.start_add = c.ops_base;
            f.no_download = true;
        } else if(ops_dev != nullptr) {
            direct.ops = ops_dev + c.ops_base;
        } else if(ops_into_staging && c.ops_bytes > 0) {
            coati_hip_model::StreamSlot& sl = model->sslots[q];
            if(stage_dev[q] == nullptr) {
                void* dp = nullptr;
                if(soft(hipHostGetDevicePointer(&dp, sl.pinned, 0))) stage_dev[q] = static_cast<uint8_t*>(dp);
            }
            // the block's result part as the download would have filled it -- [scores | ops offsets | ops lengths], each padded to 256
            // bytes as in the workspace -- and the ops behind the room a padded download of those would have taken
            const uint64_t group = static_cast<uint64_t>(reinterpret_cast<char*>(b->d_ops) - reinterpret_cast<char*>(b->d_scores));
            const uint64_t at = (std::max<uint64_t>(group, kMinDmaBytes + 256) + 255) / 256 * 256;
            if(stage_dev[q] != nullptr && out_off + at + c.ops_bytes <= sl.pinned_bytes) {
                uint8_t* const res = stage_dev[q] + out_off;
                f.kernel_ops_at = at;
                direct.ops = res + at;
                direct.scores = reinterpret_cast<float*>(res);
                direct.ops_start = reinterpret_cast<uint64_t*>(res + (reinterpret_cast<char*>(b->d_ops_start) - reinterpret_cast<char*>(b->d_scores)));
                direct.ops_len = reinterpret_cast<uint32_t*>(res + (reinterpret_cast<char*>(b->d_ops_len) - reinterpret_cast<char*>(b->d_scores)));
                f.no_download = true;
            }
        }
        ck_stream_fill_chunk(hs, hs_dev, q, b->arena, device_view(b), static_cast<uint32_t>(n), published, static_cast<uint32_t>(ci), direct);
        // Every copy under the persistent kernel must be done by the copy ENGINE: a copy the runtime does with a blit
        // kernel (HSA_ENABLE_SDMA=0, or its own choice) cannot start while viterbi_ck_stream holds every wavefront
        // slot.  So the wait is bounded -- 100 ms for the call's first chunk (a copy engine delivers it in well under
        // a millisecond), 5 s later on -- and a miss closes the kernel, hands the call to the chunk pipeline and is
        // remembered on the model (no later call tries the streamed form again).
        if(e == hipSuccess) {
            const auto t_up = std::chrono::steady_clock::now();
            const double bound_ms = ci == 0 ? 100.0 : 5000.0;
            for(uint64_t spins = 0;; ++spins) {
                e = hipEventQuery(uploaded[q]);
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
            std::fprintf(stderr, "viterbi_batch[stream]: chunk %zu (%llu pairs, %u items, slot %d) planned + uploaded %.2f .. %.2f ms (cut by %.2f, planned %.2f .. %.2f)\n", ci,
                         static_cast<unsigned long long>(n), b->n_items, q, t_begin, t_ms(), t_cut, a.t_plan0, a.t_plan1);
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
    if(pipe_timing || (env_options().stream_helpers & 8) != 0) std::fprintf(stderr, "viterbi_batch[stream]: done at %.2f ms (rc %d, %llu pairs)\n", t_ms(), rc, static_cast<unsigned long long>(n_pairs));
    if(rc == kRedo) {
        // a pair or an upload the streamed form cannot serve: the chunk pipeline does the call -- unless the stream itself
        // failed on top of it (a GPU fault is the likeliest reason for an upload to miss its bound): that is the error
        if(es == hipSuccess) return COATI_HIP_ESTATE;
        rc = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(es));
    }
    if(rc == COATI_HIP_OK && es != hipSuccess) rc = fail(COATI_HIP_EHIP, "viterbi_batch: %s", hipGetErrorString(es));
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

/* Ahead of a coati_hip_viterbi_batch call of about n_pairs pairs of about len_a x len_b: allocate what that call
 * would allocate first (include/coati_hip.h).  Nothing happens for inputs the persistent-kernel form does not serve. */
int coati_hip_model_prepare(coati_hip_model_t* model, uint64_t n_pairs, uint64_t len_a, uint64_t len_b) {
    if(model == nullptr) return fail(COATI_HIP_EINVAL, "model_prepare: model is NULL");
    try {
        std::lock_guard<std::mutex> one_call(model->pipeline_lock);
        HIP_TRY(hipSetDevice(model->device));
        const EnvOptions& env = env_options();
        bool streamed = model->gap_len == 1 && !env.viterbi_bits && !env.force_generic && env.pipe != 1 && len_b <= 8ull * kStrip &&
                        len_a * len_b <= kStreamPairCells && !model->stream_forbidden && !model->stream_unusable && !env.sdma_off;
        if(streamed && env.pipe != 2) streamed = n_pairs >= 4096 && len_a * len_b >= 250ull * 250ull;
        if(!streamed) return COATI_HIP_OK;
        uint64_t wave_slot_bytes = 0;
        int n_slots = 0;
        long double unit = 0;
        const long double cells = static_cast<long double>(n_pairs) * static_cast<long double>(len_a) * static_cast<long double>(len_b);
        const StagingNeed staging = stream_staging(n_pairs, cells, n_pairs * (len_a + len_b), len_a + len_b, false, false);
        (void)stream_reserve(model, len_b <= static_cast<uint64_t>(kStrip) ? len_a : 0, cells, staging, &wave_slot_bytes, &n_slots, &unit);  // (a failed allocation is the call's problem)
        if(!model->helpers) model->helpers = std::make_unique<HelperPool>(3);
        // (not the last chunk's ~3 GB workspace: ~15 ms of fresh hipMalloc against 0.25 ms per call -- a model's second call makes it)
        return COATI_HIP_OK;
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "model_prepare: host allocation failed");
    }
}

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
    double cells_sum = 0;  // (x87 long double adds made this loop 3x slower; 53 bits are plenty for the thresholds the sum meets)
    uint64_t widest = 0, max_pair_cells = 0, longest_single = 0, longest_a = 0, longest_pair_bytes = 0;
    for(uint64_t p = 0; p < n_pairs; ++p) {
        if(a_off[p + 1] < a_off[p] || b_off[p + 1] < b_off[p])
            return fail(COATI_HIP_EINVAL, "viterbi_batch: offsets of pair %llu decrease", static_cast<unsigned long long>(p));
        const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
        if(la > 0xffffffffull || lb > 0xffffffffull)
            return fail(COATI_HIP_EINVAL, "viterbi_batch: pair %llu is longer than 2^32", static_cast<unsigned long long>(p));
        const uint64_t cells = la * lb;
        cells_sum += static_cast<double>(cells);
        widest = std::max(widest, lb);
        max_pair_cells = std::max(max_pair_cells, cells);
        longest_a = std::max(longest_a, la);
        longest_pair_bytes = std::max(longest_pair_bytes, la + lb);
        if(lb > 0 && lb <= static_cast<uint64_t>(kStrip)) longest_single = std::max(longest_single, la);
    }
    const long double total_cells = cells_sum;
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
        const EnvOptions& env = env_options();
        bool streamed = gap_len == 1 && !env.viterbi_bits && !env.force_generic && env.pipe != 1 && widest <= 8 * kStrip && max_pair_cells <= kStreamPairCells;
        if(streamed && env.pipe != 2) streamed = n_pairs >= 4096 && total_cells / n_pairs >= 250.0L * 250.0L;
        // the persistent kernel owns the GPU for the length of the call: not when the embedder said no
        // (coati_hip_model_set_option), not where it failed before, and not where copies are done by kernels
        if(streamed && (model->stream_forbidden || model->stream_unusable)) streamed = false;
        if(streamed && env.sdma_off) streamed = false;
        // every pair must fit a stream slot on its own (the chunk cutter takes the first pair of a chunk unseen).
        // Ordinary pairs pass by two comparisons; the few long or wide ones are priced exactly.
        if(streamed && (longest_a > 32768 || widest > static_cast<uint64_t>(kStrip))) {
            for(uint64_t p = 0; p < n_pairs && streamed; ++p) {
                const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
                if((la > 32768 || lb > static_cast<uint64_t>(kStrip)) && !stream_pair_fits(la, lb, gap_len, in_pinned, out_pinned)) streamed = false;
            }
        }
        if(streamed) {
            const StagingNeed staging_need = stream_staging(n_pairs, total_cells, (a_off[n_pairs] - a_off[0]) + (b_off[n_pairs] - b_off[0]), longest_pair_bytes,
                                                            in_pinned, out_pinned);
            const int rc_stream = viterbi_batch_stream(model, n_pairs, a_cat, a_off, b_cat, b_off, scores, ops, ops_off, ops_len, in_pinned,
                                                       out_pinned, total_cells, longest_single, staging_need, t_entry);
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
    if(env_options().mem_budget > 0) budget = std::min<uint64_t>(budget, env_options().mem_budget);  // (COATI_HIP_MEM_BUDGET, tests: force chunking)
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
    const bool pipe_timing = env_options().pipe_timing;  // timeline of the call on stderr
    const bool no_d2h = env_options().pipe_no_d2h;   // (timing experiment: results stay on the device)
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

