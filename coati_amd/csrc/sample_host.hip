// sample_host.hip -- coati_hip_sampleback: the host loop of the stochastic traceback (kernels and rationale:
// sampleback.hip; reference: sampleback / marg_sample, src/lib/align_pair.cc:336-458, align_marginal.cc:590-593).
#include "abi_internal.hpp"

using namespace coati_hip_abi;

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
                                  float* d_lw, uint64_t* states_out, bool prepare_only = false) {
    coati_hip_model* m = b->model;
    const uint64_t n = b->n_pairs;
    constexpr uint32_t kChunkMax = 512;
    const uint32_t kMaxCands = env_options().spec_cands;  // (COATI_HIP_SPEC_CANDS; default 3 * 2^16: measured best, tools/sample_bench.py)
    // half-width of a candidate window in standard deviations of the offset; too narrow only ends
    // a chunk early (COATI_HIP_SPEC_Z overrides, for tuning)
    const double kZ = env_options().spec_z;  // (default 2; measured, 16 x 1 000 samples of 1 kb pairs: z = 5: 38.8 ms, 3: 30.9, 2: 25.9, 1.5: 26.0, 1: 35.9)
    size_t free_b = 0, total_b = 0;
    hipError_t e = hipMemGetInfo(&free_b, &total_b);
    if(e != hipSuccess) return e;
    {   // (cached blocks of this model count as free)
        std::lock_guard<std::mutex> hold(m->arena_lock);
        for(const auto& a : m->free_arenas) free_b += a.bytes;
    }
    // Round 4: the step table (sampleback.hip) -- thresholds and log-weight increments per (body cell, state), built once
    // per call -- when it fits 8 GB and a third of the free HBM (16 pairs of 1 kb: 1.15 GB); gap_len 1, either Forward mode
    // (the sampler's own arithmetic on the stored M/D/I is the libm restatement in both).  With it the
    // candidates only count draws (no temporary ops) and ONE final launch writes every sample from its resolved offset.
    // (a band of diagonals around each pair's straight line: common.hpp step_band; cells outside it are computed by the walkers)
    const uint32_t band_half = env_options().sample_band;
    uint64_t table_entries = 0, thr_entries = 0;
    std::vector<uint64_t> tab_off(n, 0), thr_off(n, 0);
    uint64_t max_cells = 0;
    for(uint64_t p = 0; p < n; ++p) {
        const StepBand band = step_band(b->desc[p].la, b->desc[p].lb, band_half);
        tab_off[p] = table_entries;
        thr_off[p] = thr_entries;
        const uint64_t cells = static_cast<uint64_t>(b->desc[p].la) * band.width;
        table_entries += 3 * cells;
        thr_entries += static_cast<uint64_t>(band.width) * band.diag_len;  // the M-state thresholds once more, diagonal-major (sampleback.hip)
        max_cells = std::max(max_cells, cells);
    }
    const bool table_off = env_options().sample_table_off;
    const uint64_t table_bytes = table_entries * step_entry_bytes() + thr_entries * 12 + 256;
    const bool use_table = m->gap_len == 1 && !table_off && table_bytes <= (8ull << 30) && table_bytes <= free_b / 3 &&
                           b->desc[0].f_compact == 0;
    // work arena for the candidates' ops: 2 GB, or a power of two below a quarter of the free HBM
    // (a stable size, so that repeated calls find their block in the cache)
    uint64_t tmp_budget = 2ull << 30;
    while(tmp_budget > (1ull << 20) && tmp_budget > free_b / 4) tmp_budget >>= 1;
    if(use_table) tmp_budget = 0;  // (no temporary ops)

    uint64_t dbg_rounds = 0, dbg_cands = 0;  // reported with COATI_HIP_TIMING=1
    struct PairState {
        u128 st0;
        uint64_t origin = 0;  // draws consumed by the samples resolved so far
        uint32_t done = 0, cnt = 0;
        double mean = 0.0, m2 = 0.0;
    };
    std::vector<PairState> ps(n);
    for(uint64_t p = 0; p < n && !prepare_only; ++p) ps[p].st0 = (static_cast<u128>(rng_state[2 * p + 1]) << 64) | rng_state[2 * p];

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
    uint64_t *d_tab_off = nullptr, *d_state0 = nullptr, *d_sample_off = nullptr, *d_base = nullptr;
    char* d_steps = nullptr;
    SpecPairState* d_states = nullptr;  // (table path: the speculation loop runs on the device, sampleback.hip "device rounds")
    SpecWindow* d_windows = nullptr;
    uint32_t* d_rank_pair = nullptr;
    SpecRound* d_round = nullptr;
    float* d_draw_table = nullptr;  // the round's stream draws, a slice per pair (spec_draws_kernel)
    uint64_t* d_thr_off = nullptr;
    char* d_thr = nullptr;
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
        d_tmp = cv.take<uint8_t>(std::max<uint64_t>(tmp_budget, 16));
        if(use_table) {
            d_tab_off = cv.take<uint64_t>(n);
            d_state0 = cv.take<uint64_t>(2 * n);
            d_sample_off = cv.take<uint64_t>(n * n_samples);
            d_base = cv.take<uint64_t>(n);
            d_states = cv.take<SpecPairState>(n);
            d_windows = cv.take<SpecWindow>(n * kSpecChunkMax);
            d_rank_pair = cv.take<uint32_t>(n);
            d_round = cv.take<SpecRound>(1);
            d_draw_table = cv.take<float>(kSpecDrawFloats + 64);
            d_thr_off = cv.take<uint64_t>(n);
            d_thr = cv.take<char>(thr_entries * 12 + 256);
            d_steps = cv.take<char>(table_entries * step_entry_bytes());
        }
    };
    if(prepare_only) {
        // coati_hip_sampleback_prepare: the call's temporaries (with the step table: ~170 MB for 16 pairs of 1 kb) and its page-locked
        // round records exist -- in the model's caches -- before the call needs them
        Carver sizing;
        carve(sizing);
        void* blk = nullptr;
        uint64_t blk_bytes = 0;
        if((e = model_take_arena(m, sizing.used, &blk, &blk_bytes)) != hipSuccess) return e;
        model_give_arena(m, blk, blk_bytes);
        void* host_block = nullptr;
        return model_pinned(m, 24 * sizeof(SpecRound) + n * sizeof(SpecPairState) + 64, &host_block);
    }
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
    const BatchDeviceView view = device_view(b);
    std::vector<uint64_t> sample_off;  // table path: draws between a pair's original state and the start of each of its samples
    if(use_table) {
        sample_off.assign(n * n_samples, 0);
        S_TRY(hipMemcpyAsync(d_tab_off, tab_off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(hipMemcpyAsync(d_state0, rng_state, 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(hipMemcpyAsync(d_base, base.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(hipMemcpyAsync(d_thr_off, thr_off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(launch_step_table(view, d_tab_off, max_cells, band_half, d_steps, d_thr_off, d_thr, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));  // (tab_off / base are stack-lifetime vectors of the caller: uploaded before they can go)
    }

    uint32_t widest = 0;
    for(uint64_t p = 0; p < n; ++p) widest = std::max(widest, b->desc[p].la + b->desc[p].lb);
    // (device rounds read the streams' draws from a 64 MB table, a slice per pair: one whole walk must fit a slice)
    const bool device_rounds = use_table && !env_options().spec_host_rounds && static_cast<uint64_t>(widest) + 128 <= kSpecDrawFloats;  // (COATI_HIP_SPEC_HOST_ROUNDS=1: the host loop below, the A/B partner)
    if(device_rounds) {
        // ---- device rounds: plan + walks + chain per round, enqueued several rounds at a time; the host reads the number
        // of unfinished pairs each round started with (a round that starts with none is three empty launches)
        // (How many: a call needs about two learning rounds plus one per ~85 samples of a pair -- the windows' sizes grow with the
        // square root of the sample's number and a pair's share of the candidates holds a chunk of about that many; 16 x 1 000
        // samples: 12-13 -- so that many are enqueued at once and two at a time after them.  Round 6: from the call's own size,
        // not from what the model's LAST call needed -- a hint that was shared, unlocked, by every call on the model and keyed on
        // the sample count alone; the first call of a process, which is all `coati sample` makes, now starts like the later ones.)
        constexpr uint32_t kBatch = 24;
        uint32_t batch_now = static_cast<uint32_t>(std::min<uint64_t>(kBatch, 2u + (static_cast<uint64_t>(n_samples) + 79u) / 80u));
        const uint32_t batch_next = 2u;
        const uint32_t max_width = widest;
        void* host_block = nullptr;
        S_TRY(model_pinned(m, kBatch * sizeof(SpecRound) + n * sizeof(SpecPairState) + 64, &host_block));
        SpecRound* h_round = static_cast<SpecRound*>(host_block);
        SpecPairState* h_states = reinterpret_cast<SpecPairState*>(h_round + kBatch);
        for(uint64_t p = 0; p < n; ++p) h_states[p] = SpecPairState{0, 0, 0, 0.0, 0.0, 0, 0, 0, 0};
        S_TRY(hipMemcpyAsync(d_states, h_states, n * sizeof(SpecPairState), hipMemcpyHostToDevice, m->stream));
        const bool timing = env_options().timing;
        for(bool finished = n_samples == 0 || n == 0; !finished;) {
            for(uint32_t r = 0; r < batch_now; ++r) {
                S_TRY(launch_spec_round(view, d_tab_off, band_half, d_steps, d_state0, d_pow, n_samples, kMaxCands, max_width, kZ, d_states, d_windows, d_rank_pair, d_round,
                                        d_draw_table, d_thr_off, d_thr, d_cdraws, d_sample_off, m->stream));
                S_TRY(hipMemcpyAsync(h_round + r, d_round, sizeof(SpecRound), hipMemcpyDeviceToHost, m->stream));
            }
            S_TRY(hipStreamSynchronize(m->stream));
            for(uint32_t r = 0; r < batch_now; ++r) {
                if(h_round[r].active == 0) {
                    finished = true;
                    break;
                }
                ++dbg_rounds;
            }
            batch_now = batch_next;
        }
        // every sample's start in its pair's stream is known (on the device): one walker per (pair, sample), results in place
        S_TRY(launch_final_walk(view, d_tab_off, band_half, d_steps, d_state0, d_pow, d_sample_off, d_base, n_samples, d_ops, d_start, d_len, d_lw, m->stream));
        S_TRY(hipMemcpyAsync(h_states, d_states, n * sizeof(SpecPairState), hipMemcpyDeviceToHost, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));
        for(uint64_t p = 0; p < n; ++p) ps[p].origin = h_states[p].origin;
        if(timing) std::fprintf(stderr, "sampleback_speculative: %llu device rounds for %llu samples\n", static_cast<unsigned long long>(dbg_rounds),
                                static_cast<unsigned long long>(n * n_samples));
    } else {
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
    const bool timing = env_options().timing;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_build = 0, t_gpu = 0, t_chain = 0;
    try {
    for(;;) {
        const double t_round0 = timing ? now_ms() : 0;
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
               (!use_table && tmp_used + static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb > tmp_budget))
                continue;  // more unfinished pairs than one round holds: this pair waits for the next round
            const uint64_t width = static_cast<uint64_t>(b->desc[p].la) + b->desc[p].lb;
            const uint32_t remaining = n_samples - s.done;
            // Before anything is known about a pair a sample is expected to take 1 + max(la, lb) draws plus a few gap
            // columns, give or take 2 % of la + lb: the first round walks 8 samples under that prior (a round costs its ~1 000
            // dependent steps whatever it holds, so the rounds of 1 and 4 samples that used to learn the mean were two
            // rounds' time for five samples); from 8 observations on the running estimate decides and the chunk is what the
            // candidate budget allows.  A window that is too narrow only ends the chunk early.
            const uint32_t want = s.cnt == 0 ? 8u : (s.cnt < 8 ? 16u : kChunkMax);
            const uint32_t chunk = std::min(remaining, want);
            if(s.cnt == 0) s.mean = 1.0 + static_cast<double>(std::max(b->desc[p].la, b->desc[p].lb)) + 0.005 * static_cast<double>(width);
            const double sigma = s.cnt >= 2 ? std::sqrt(s.m2 / (s.cnt - 1)) * (1.0 + 4.0 / s.cnt) + 1.0
                                            : 0.02 * static_cast<double>(width) + 2.0;
            for(uint32_t j = 0; j < chunk; ++j) {
                const int64_t center = std::llround(j * s.mean);
                const int64_t half = j == 0 ? 0 : static_cast<int64_t>(std::ceil(kZ * sigma * std::sqrt(static_cast<double>(j)))) + 2;
                const int64_t lo = std::max<int64_t>(center - half, j), hi = std::max<int64_t>(center + half, lo);
                const uint64_t count = static_cast<uint64_t>(hi - lo + 1);
                if(j > 0 && (cands.size() - cand_begin + count > cand_share ||
                            (!use_table && tmp_used - tmp_begin + count * std::max<uint64_t>(width, 1) > tmp_share)))
                    break;
                windows[p].push_back(Window{static_cast<uint32_t>(cands.size()), static_cast<uint32_t>(lo), static_cast<uint32_t>(hi)});
                for(int64_t off = lo; off <= hi; ++off) {
                    cands.push_back(SpecCandidate{static_cast<uint32_t>(p), static_cast<uint32_t>(off), tmp_used});
                    if(!use_table) tmp_used += width;
                }
            }
            const u128 st = s.st0 * lehmer_pow(s.origin);
            origin_states[2 * p] = static_cast<uint64_t>(st);
            origin_states[2 * p + 1] = static_cast<uint64_t>(st >> 64);
        }
        if(!any) break;
        if(cands.size() > kMaxCands || (!use_table && tmp_used > tmp_budget)) {  // a single sample does not fit the work arena
            release();
            return hipErrorOutOfMemory;
        }
        const uint32_t nc = static_cast<uint32_t>(cands.size());
        ++dbg_rounds;
        dbg_cands += nc;
        const double t_round1 = timing ? now_ms() : 0;
        S_TRY(hipMemcpyAsync(d_origin, origin_states.data(), 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(hipMemcpyAsync(d_cands, cands.data(), nc * sizeof(SpecCandidate), hipMemcpyHostToDevice, m->stream));
        if(use_table)
            S_TRY(launch_spec_len(view, d_tab_off, band_half, d_steps, d_origin, d_pow, d_cands, nc, d_cdraws, m->stream));
        else
            S_TRY(launch_spec_walk(view, d_origin, d_pow, d_cands, nc, d_tmp, d_cstart, d_clen, d_clw, d_cdraws, m->stream));
        draws.resize(nc);
        S_TRY(hipMemcpyAsync(draws.data(), d_cdraws, nc * sizeof(uint32_t), hipMemcpyDeviceToHost, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));
        const double t_round2 = timing ? now_ms() : 0;
        if(timing) std::fprintf(stderr, "sampleback_speculative: round %llu: %u candidates, lists %.3f ms, upload + walks + download %.3f ms\n",
                                static_cast<unsigned long long>(dbg_rounds), nc, t_round1 - t_round0, t_round2 - t_round1);
        t_build += t_round1 - t_round0;
        t_gpu += t_round2 - t_round1;
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
                if(use_table)
                    sample_off[out_index] = s.origin + off;
                else
                    commits.push_back(SpecCommit{cand, 0u, base[p] + (static_cast<uint64_t>(s.done) + 1) * width, out_index});
                const double x = static_cast<double>(draws[cand]);
                if(s.cnt == 0) s.mean = 0.0;  // (the prior has served: the estimate starts from the observations)
                s.cnt += 1;  // Welford
                const double d1 = x - s.mean;
                s.mean += d1 / s.cnt;
                s.m2 += d1 * (x - s.mean);
                off += draws[cand];
                s.done += 1;
            }
            s.origin += off;
        }
        if(timing) t_chain += now_ms() - t_round2;
        if(use_table) continue;  // (nothing to copy: the final launch below writes every sample)
        const uint32_t ncm = static_cast<uint32_t>(commits.size());
        S_TRY(hipMemcpyAsync(d_commits, commits.data(), ncm * sizeof(SpecCommit), hipMemcpyHostToDevice, m->stream));
        S_TRY(launch_spec_commit(d_commits, ncm, d_tmp, d_cstart, d_clen, d_clw, d_ops, d_start, d_len, d_lw, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));  // `commits`/`cands` are reused by the next round
    }
    if(use_table) {
        // every sample's start in its pair's stream is known: one walker per (pair, sample), results in place
        S_TRY(hipMemcpyAsync(d_sample_off, sample_off.data(), sample_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
        S_TRY(launch_final_walk(view, d_tab_off, band_half, d_steps, d_state0, d_pow, d_sample_off, d_base, n_samples, d_ops, d_start, d_len, d_lw, m->stream));
        S_TRY(hipStreamSynchronize(m->stream));
    }
    } catch(...) {  // host-side allocation failure: free the device work areas, report at the ABI
        release();
        throw;
    }
    if(timing)
        std::fprintf(stderr, "sampleback_speculative: %llu rounds, %llu candidate walks for %llu samples; host lists %.2f ms, device rounds %.2f ms, chains %.2f ms\n",
                     static_cast<unsigned long long>(dbg_rounds), static_cast<unsigned long long>(dbg_cands),
                     static_cast<unsigned long long>(n * n_samples), t_build, t_gpu, t_chain);
    }  // (!device_rounds)
#undef S_TRY
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
                    uint64_t* rng_state_out, bool prepare_only = false);
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

int coati_hip_sampleback_prepare(coati_hip_batch_t* b, uint32_t n_samples, int independent_streams) {
    try {
        return sampleback_impl(b, n_samples, nullptr, independent_streams, nullptr, nullptr, 0, nullptr, nullptr, nullptr, true);
    } catch(const std::bad_alloc&) {
        return fail(COATI_HIP_ENOMEM, "sampleback_prepare: host allocation failed");
    } catch(const std::exception& ex) {
        return fail(COATI_HIP_EHIP, "sampleback_prepare: %s", ex.what());
    }
}

namespace {
int sampleback_impl(coati_hip_batch_t* b, uint32_t n_samples, const uint64_t* rng_state, int independent_streams,
                    float* log_weights, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                    uint64_t* rng_state_out, bool prepare_only) {
    if(b == nullptr || (rng_state == nullptr && !prepare_only)) return fail(COATI_HIP_EINVAL, "sampleback: NULL argument");
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
    if(prepare_only) {
    } else if(independent_streams) {
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
    uint64_t *d_states = nullptr, *d_base = nullptr, *d_start = nullptr, *d_packed_off = nullptr;
    uint8_t *d_ops = nullptr, *d_packed = nullptr;
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
        d_packed_off = cv.take<uint64_t>(n_out + 1);  // (+ the total)
        d_packed = cv.take<uint8_t>(std::max<uint64_t>(total, 16));
    };
    auto release = [&]() {
        if(block == nullptr) return;
        if(hipStreamSynchronize(m->stream) == hipSuccess)
            model_give_arena(m, block, block_bytes);
        else
            (void)hipFree(block);
        block = nullptr;
    };
    bool states_final_on_host = false;
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
        const bool sequential = env_options().sample_sequential;
        if(prepare_only) {  // (the result block is taken -- and goes back to the model's cache below; now the speculation's temporaries)
            if(!independent_streams && n_samples >= 4 && !sequential && n <= 1024)
                return sampleback_speculative(b, n_samples, nullptr, base, d_ops, d_start, d_len, d_lw, nullptr, true);
            return hipSuccess;
        }
        // (speculation buys parallelism for FEW pairs with many samples; thousands of pairs are parallel as they are -- one walker
        // per pair, sample after sample -- and a round's per-pair work would only add to it: 3 000 short pairs x 12 samples
        // 6.4 ms sequentially, 9.4-14.5 ms speculated; tools/sample_many_pairs_check.py)
        if(!independent_streams && n_samples >= 4 && !sequential && n <= 1024) {
            if((e = sampleback_speculative(b, n_samples, rng_state, base, d_ops, d_start, d_len, d_lw, states.data())) != hipSuccess) return e;
            states_final_on_host = true;  // (the streams' final states were computed here: no trip through the device)
        } else {
            if((e = hipMemcpyAsync(d_states, states.data(), states.size() * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream)) != hipSuccess) return e;
            if((e = hipMemcpyAsync(d_base, base.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream)) != hipSuccess) return e;
            if((e = launch_sampleback(device_view(b), n_samples, independent_streams != 0, d_states, d_base, d_ops, d_start, d_len,
                                      d_lw, m->stream)) != hipSuccess) return e;
            if((e = hipStreamSynchronize(m->stream)) != hipSuccess) return e;
        }
        if(log_weights != nullptr && (e = hipMemcpy(log_weights, d_lw, n_out * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        // the ops go home packed back to back (a sample fills about half of its slot of la + lb bytes: 16 x 1 000 samples of
        // 1 kb pairs are 16 MB of ops in 32 MB of slots), ops_off[] says where each sample's start
        // (ops_off[] ALWAYS means "packed from ops[0]", whether or not the caller takes the ops themselves)
        if((ops != nullptr || ops_off != nullptr) && total > 0 && n_out > 0) {
            if((e = launch_ops_pack(d_ops, d_start, d_len, n_out, d_packed_off, d_packed_off + n_out, d_packed, m->stream)) != hipSuccess) return e;
            uint64_t packed_bytes = 0;
            if((e = hipMemcpyAsync(&packed_bytes, d_packed_off + n_out, sizeof packed_bytes, hipMemcpyDeviceToHost, m->stream)) != hipSuccess) return e;
            if((e = hipStreamSynchronize(m->stream)) != hipSuccess) return e;
            if(packed_bytes > total) return hipErrorInvalidValue;  // (cannot happen: a sample is at most la + lb columns)
            if(ops != nullptr && packed_bytes > 0 && (e = hipMemcpy(ops, d_packed, packed_bytes, hipMemcpyDeviceToHost)) != hipSuccess) return e;
            if(ops_off != nullptr && (e = hipMemcpy(ops_off, d_packed_off, n_out * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        } else if(ops_off != nullptr) {
            std::memset(ops_off, 0, n_out * sizeof(uint64_t));  // (no op bytes at all: every sample is empty and starts at 0)
        }
        if(ops_len != nullptr && (e = hipMemcpy(ops_len, d_len, n_out * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        if(rng_state_out != nullptr && !independent_streams) {
            if(states_final_on_host)
                std::memcpy(rng_state_out, states.data(), 2 * n * sizeof(uint64_t));
            else if((e = hipMemcpy(rng_state_out, d_states, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess)
                return e;
        }
        return hipSuccess;
    };
    const hipError_t e = attempt();
    release();
    if(e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? COATI_HIP_ENOMEM : COATI_HIP_EHIP, "sampleback: %s", hipGetErrorString(e));
    return COATI_HIP_OK;
}
}  // namespace

