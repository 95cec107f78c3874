// plan.hip -- batch_create_impl: what coati_hip_batch_create does (and what every chunk of coati_hip_viterbi_batch
// does with BatchOpts): validation of the input (process_marginal, src/lib/utils.cc:822-835), strip plans and kernel
// choice, the LPT order and the row parts of the ragged end, the layout of the ONE workspace allocation, the upload.
#include "abi_internal.hpp"

using namespace coati_hip_abi;

extern "C" {
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

}  // extern "C"

namespace coati_hip_abi {
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

}  // namespace

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
    const EnvOptions& env = env_options();  // (the COATI_HIP_* switches: read once per process, common.hpp)
    const bool timing = env.timing;
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if(!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "batch_create: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    b->desc.resize(n_pairs);
    const uint64_t L = static_cast<uint64_t>(model->gap_len);
    const bool force_generic = env.force_generic;
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
    b->fwd_fast = model->forward_mode.load() == COATI_HIP_FORWARD_TOLERANCE;  // (the mode the batch is planned for: COATI_HIP_OPT_FORWARD_MODE)
    if(L == 1 && !force_generic) {
        if(!b->fwd_fast) fwd_wlog2 = 3;
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
        if(const int w = env.fwd_w; w == 1 || w == 2 || w == 4 || w == 8 || w == 16) fwd_wlog2 = w == 1 ? 0u : w == 2 ? 1u : w == 4 ? 2u : (w == 8 ? 3u : 4u);
        // Fewer still (`coati sample`: ONE pair; BASELINE configs[3]: 16): a wavefront that is alone on its SIMD issues one
        // instruction per ~6 cycles whatever its lanes hold, so the 417 instructions of a cell are spread over FOUR lanes -- one
        // each for the M, the D and the I sum, 16 columns per wavefront (forward_l1.hip: forward_quad_strip): a step is two
        // `plus` instead of five, a 1 kb pair 63 wavefronts instead of 16.  While the quad strips fit three per SIMD and the
        // model has one table (tools/experiments/quad_sweep.py, 1 kb pairs, 1-column strips -> quad strips: 1 pair 3.07 -> 1.86 ms,
        // 16: 3.11 -> 1.91, 32: 3.18 -> 2.17, 48: 3.19 -> 2.66, 64: 3.37 -> 3.27).  COATI_HIP_FWD_QUAD=0 / 1 forces (1: wherever
        // the plan is 1 column per lane).
        if(fwd_wlog2 == 0 && model->n_tables == 1) {
            uint64_t quads = 0;
            const uint64_t quads_max = 3ull * 4ull * device_cu_count();  // three per SIMD (3 072 on the 256 CUs the sweep ran on)
            for(uint64_t p = 0; p < n_pairs && quads <= quads_max; ++p) {
                const uint64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
                quads += (la > 0 && lb > 0 && lb <= 0x7fffff00ull) ? fwd_quad_strips(static_cast<uint32_t>(lb)) : 1;
            }
            b->fwd_quad = env.fwd_quad >= 0 ? env.fwd_quad != 0 : quads <= quads_max;
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
    const bool plan_l1 = L == 1 && !force_generic;  // viterbi_ck / viterbi_l1 will run
    // COATI_HIP_VITERBI_BITS=1: the round-1 kernel (five decision bits per cell written by the fill), kept
    // as the A/B partner and second implementation of viterbi_ck
    b->ck = plan_l1 && !env.viterbi_bits;
    const bool ck_shared = model->n_tables == 1;
    // longest-processing-time-first order for the dynamic queue
    std::vector<uint32_t> order(n_pairs);
    // (the classes also for large resident batches: an exact sort of 48 000 pairs is 5.5 ms of a chunk's 12 ms of planning in
    // the sharded job; up to 20 000 pairs -- the headline's 10 000 -- the order stays exact)
    lpt_order(b->desc, order, (opts != nullptr && opts->device_validates) || n_pairs > 20000);
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
        // Narrow the strips until the batch has about TWO items per SIMD (round 4; before: one).  A wavefront that is alone
        // on its SIMD issues at under half the SIMD's rate, so a plan that stops at one item per SIMD leaves
        // half the chip's issue slots empty however the items are shaped; the sweep of {16 .. 1 024} pairs x {2 .. 32} kb
        // under every kernel and strip width (tools/planner_sweep.py -> profiles/r04/planner_sweep.txt) puts the best
        // forced choice at 1.9-2.0 items per SIMD everywhere between "a few long pairs" and "thousands of pairs":
        // 64 x 8 kb 5.7 -> 4.6 ms, 256 x 2 kb 1.47 -> 1.10, 64 x 16 kb 15.3 -> 13.7, 256 x 4 kb 3.9 -> 3.5.
        // Exception: a batch of SINGLE-strip pairs that already has an item per SIMD stays as it is (1 024 x 1 kb: whole
        // pairs through viterbi_ck 1.02 ms, cut in two 8-column strips 1.24 -- a second strip adds its pipeline lag to a
        // pair of only 1 000 rows, and multi-strip pairs lose the banded checkpoints); pairs of two or more strips gain.
        uint64_t live = 0;
        for(uint64_t p = 0; p < n_pairs; ++p) live += (b->desc[p].la > 0 && b->desc[p].lb > 0) ? 1 : 0;
        auto narrow_more = [&](uint32_t w) {
            const uint64_t items = count_items(w);
            return items < kSimds || (items < kSimds * 7 / 4 && 2 * items >= 3 * live);
        };
        while(w_main > 4 && narrow_more(w_main)) w_main /= 2;
        // While every wavefront is alone on its SIMD it issues one instruction per ~4.5 cycles whatever it is, so the time is
        // (rows + strips x hand-off lag) x instructions per step -- and viterbi_lp's 2-column step is 52 instructions
        // against 90 for 4 columns while the lag per column only doubles: 2-column strips as long as there are no more
        // of them than SIMDs (80 kb pair, 626 strips: 22 -> 18 ms).  With more (160 kb pair: 1 252) some SIMDs hold two
        // wavefronts, which then run at half speed each (the packed adds, maxima and bit deposits are limited by the
        // SIMD, not by the wavefront: tools/ubench, "cell19 lp" 4.06 cycles per instruction alone, 4.03 per SIMD with two)
        // and hold the whole pipeline back: 66 ms against 44.  viterbi_lp only (COATI_HIP_L1_LP=0 keeps viterbi_l1).
        const bool lp_allowed = !env.l1_lp_off && !env.l1_progress;
        const bool ck_forced = env.viterbi_ck || (opts != nullptr && (opts->force_w_main != 0 || opts->force_ck));
        // Round 6: 3 columns per lane where 2 do not fit the SIMDs but 3 do (the 160 kb pair: 834 strips; 4 columns leave 398 of
        // the 1 024 SIMDs without a strip): a 3-column step is 64 instructions against 81, the chain grows by lb/3 - lb/4 steps.
        if(lp_allowed && !ck_forced && w_main == 4 && count_items(2) <= kSimds) w_main = 2;
        else if(lp_allowed && !ck_forced && !env.lp3_off && w_main == 4 && count_items(3) <= kSimds) w_main = 3;
        if(const int w = env.strip_w; w == 4 || w == 8 || w == 16 || ((w == 2 || w == 3) && lp_allowed && !ck_forced)) w_main = static_cast<uint32_t>(w);
        if(opts != nullptr && opts->force_w_main != 0) w_main = opts->force_w_main;
        // Which gap_len-1 kernel.  viterbi_ck (lean fill + checkpoint traceback) wins where the fill
        // dominates; viterbi_l1 (decision bits written by the fill) keeps two regimes, both measured
        // (profiles/r02/kernel_choice.txt): batches of SHORT pairs, where a traceback round recomputes a
        // large share of the little matrix (150 nt pairs: 588 vs 440 GCUPS; from 300 nt on the two are level,
        // at 750 nt viterbi_ck leads by 16 %), and a few LONG pairs cut into narrow strips, where every
        // wavefront is alone on its SIMD and the 4x larger checkpoint stream of 4-column strips costs more
        // than the shorter cell saves (160 kb pair: 86 vs 104 ms).  COATI_HIP_VITERBI_CK=1 / _BITS=1 force one.
        if(b->ck && !env.viterbi_ck && !(opts != nullptr && (opts->force_w_main != 0 || opts->force_ck))) {
            long double cells = 0;
            uint64_t live = 0;
            for(uint64_t p = 0; p < n_pairs; ++p)
                if(b->desc[p].la > 0 && b->desc[p].lb > 0) {
                    cells += static_cast<long double>(b->desc[p].la) * b->desc[p].lb;
                    ++live;
                }
            constexpr long double kShortPair = 250.0L * 250.0L;
            // (round 4: 8-column strips stay with viterbi_ck -- its multi-strip pairs hand their boundaries over as
            // self-validating values and keep banded checkpoints now: 256 x 4 kb 3.00 ms against viterbi_l1's 3.66,
            // 64 x 16 kb 11.9 against 14.0; 4-column plans -- every wavefront nearly alone on its SIMD -- are viterbi_lp's)
            // (round 6: viterbi_ck's multi-strip pairs have the spliced traceback, one branch-free boundary store per step and 16
            // steps of straight-line code per sub-block: where a 4-column plan has MORE strips than the GPU has SIMDs -- two
            // wavefronts per SIMD, which viterbi_lp's blocks are not made for -- they stay with viterbi_ck: 64 x 8 kb 4.55 ->
            // 3.33 ms, 128 x 4 kb 2.21 -> 1.76, 32 x 16 kb 9.5 -> 6.4, 16 x 32 kb 18.9 -> 14.6; with at most a strip per SIMD
            // viterbi_lp keeps its lead: 32 x 8 kb 2.09 against 3.21, 16 x 16 kb 4.10 against 6.21 -- tools/experiments/r6_plan_sweep.py)
            const bool ck_narrow = w_main == 4 && count_items(4) > kSimds;
            if((w_main < 8 && !ck_narrow) || (live > 0 && cells / live < kShortPair)) b->ck = false;
        }
        // a decision-bit plan of 4-column strips throughout is the "few long pairs" regime: viterbi_lp fills it (the same
        // layout, half the instructions per step; COATI_HIP_L1_LP=0 keeps viterbi_l1, the A/B partner)
        b->long_pairs = !b->ck && w_main <= 4 && lp_allowed;
        const uint64_t kFillSlots = (b->ck && ck_shared ? 4 : 3) * kSimds;  // resident wavefronts of the persistent kernel
        uint64_t tail_pairs = 0;
        if(env.tail_pairs >= 0) {
            tail_pairs = std::min<uint64_t>(n_pairs, static_cast<uint64_t>(env.tail_pairs));
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
    if(L == 1 && !force_generic && fwd_wlog2 == 4 && n_pairs > 3 * 1024 * 3 / 2 && env.fwd_w == 0) {
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
    const bool plan_k = (L == 2 || L == 3) && !force_generic;
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
        if(ns > 1) b->multi_strip = true;
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
        const uint64_t nf = viterbi_only ? 1 : (b->fwd_quad && d.f_compact == 0) ? fwd_quad_strips(d.lb) : fwd_strips_w(d.lb, 1u << d.f_wlog2);
        // (viterbi_lp: the strips' record areas of the spliced traceback behind the boundary arrays, common.hpp)
        // (viterbi_ck: the same for its multi-strip pairs, ck_rec_first_float)
        const uint64_t need = std::max<uint64_t>({b->long_pairs && ns > 1 ? lp_splice_first_float(d.la, ns) + lp_splice_floats(ns)
                                                  : b->ck && plan_l1 && ns > 1 ? ck_rec_first_float(d.la, ns) + ck_rec_floats(ns)
                                                                               : (ns - 1) * 2 * (la + 1),
                                                  nf > 1 ? (nf - 1) * 3 * (la + 1) : 0,
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
                   ck_parts_fit(d.la + (std::min<uint32_t>(kStrip, d.lb) + kW - 1) / kW - 1, opts->tail_parts, false)) {
                    // (the last part shorter by what its wavefront spends on the pair's traceback: as in the resident plan below)
                    const uint32_t chunks = (d.la + (std::min<uint32_t>(kStrip, d.lb) + kW - 1) / kW - 1 + 63u) / 64u;
                    const uint32_t want = std::min<uint32_t>(7u, (3u * d.lb + 500u) / 1000u);
                    const uint32_t sl = ck_fit_short_last(chunks, opts->tail_parts, want);  // (every part keeps >= 2 chunks of the REAL cut)
                    d.v_parts = static_cast<uint8_t>(opts->tail_parts | (sl << 4));
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
    } else if(b->ck && !env.ck_per_pair) {
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
        // The cut pairs' TRACEBACKS as work items of their own, behind the last row parts in the queue (round 5): the items at
        // the end of the queue are then 0.2 ms walks instead of 0.5 ms of rows + walk, and every wavefront stays busy 0.3 ms
        // longer (idle wavefront time before the end of a 10 000-pair launch 8.5 -> 4.6 %).  It pays since a hand-over needs no
        // fences and a cut pair's checkpoints are read past the L2 by whoever walks it (viterbi_ck.hip, kThrough): 10 000 pairs
        // +1.2 % (same box, alternating, six passes), 40 000 +-0, 16 000 -0.7 %: by default for launches of up to three rounds
        // of wavefronts.  With fences (earlier in round 5) the same change LOST 1.7 %.  COATI_HIP_CK_WALK_ITEMS=0 / 1 forces.
        const bool walk_items = env.ck_walk_items >= 0 ? env.ck_walk_items != 0 : n_pairs <= 3ull * ck_scratch_waves();
        if(use_slots && n_pairs > ck_scratch_waves()) {
            split_pairs = std::min<uint64_t>(2 * ck_scratch_waves(), n_pairs - ck_scratch_waves());
            // (later in round 5: whole pairs keep TILE-major checkpoints and are ~3 % cheaper than cut ones, whose stores go through
            // the L2 one by one.  Where the tracebacks are items of their own, ONE round of wavefronts' worth of pairs in 4 parts --
            // tools/split_ab.py, one box, four runs each, 10 000 pairs: 5 904 x 3 / 4 608 x 3 / 4 096 x 4 / 4 096 x 3 / 3 584 x 3 =
            // 2 590 / 2 641 / 2 632 / 2 595 / 2 586 GCUPS; the two builds alternating: 3.926 -> 3.827 ms; 6 000 and 8 000 pairs +-0.
            // Without walk items the rule above stands: 16 000 pairs 8 192 x 3 / 8 192 x 4 / 4 096 x 4 / 4 096 x 3 = 5.94 / 5.91 /
            // 6.03 / 6.06 ms, 40 000 pairs 13.84 / 13.88 / 13.99 / 14.00.  profiles/r05/cut_sweep.txt)
            if(walk_items) split_pairs = std::min<uint64_t>(ck_scratch_waves(), n_pairs - ck_scratch_waves()), parts = 4;
            if(split_pairs < 256) split_pairs = 0;
        }
        bool taper = false;
        uint32_t short_last = 0;
        if(env.ck_split_set) {  // COATI_HIP_CK_SPLIT="pairs,parts[,t]": t = tapered parts (common.hpp: ck_part_cut)
            split_pairs = env.ck_split_pairs;
            parts = env.ck_split_parts;
            taper = env.ck_split_taper;
            short_last = taper ? 0u : env.ck_split_short_last;
            if(parts < 2 || parts > 8) split_pairs = 0;
        }
        split_pairs = std::min<uint64_t>(split_pairs, n_pairs);
        std::vector<uint32_t> cut;  // in LPT order
        for(uint64_t q = n_pairs - split_pairs; q < n_pairs; ++q) {
            PairDesc& d = b->desc[order[q]];
            // (every part at least a 64-step chunk, two on average; narrow last strips and multi-strip pairs stay whole)
            const uint32_t nsteps = d.la + (std::min<uint32_t>(kStrip, d.lb) + kW - 1) / kW - 1;
            if(d.la > 0 && d.lb > 0 && d.v_strips == 1 && d.v_wlast == kW && need_of(d) <= kSlotCap &&
               ck_parts_fit(nsteps, static_cast<uint32_t>(parts), taper)) {
                // The wavefront of a pair's LAST part also walks the pair (a 1 kb pair: ~0.17 ms, two to three chunks' worth), and
                // the last parts are the items at the end of the queue: the last part gives that many chunks to the others
                // (round 5, tools/split_ab.py on one box: 10 000 pairs 4.01 -> 3.93 ms, 40 000 14.53 -> 14.41; cuts 5+6+6 -> 6+7+4
                // chunks).  Only where every part keeps at least two chunks.
                const uint32_t chunks = (nsteps + 63u) / 64u;
                // (with the traceback as an item of its own -- COATI_HIP_CK_WALK_ITEMS -- the last part is a part like the others)
                const uint32_t want = walk_items ? 0u : env.ck_split_set ? short_last : std::min<uint32_t>(7u, (3u * d.lb + 500u) / 1000u);
                const uint32_t sl = taper ? 0u : ck_fit_short_last(chunks, static_cast<uint32_t>(parts), want);  // (every part keeps >= 2 chunks of the REAL cut)
                d.v_parts = static_cast<uint8_t>(parts | (taper ? kCkPartsTaper : 0u) | (sl << 4));
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
            b->ck_walk_items = walk_items;
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
                 : b->fwd_quad                ? fwd_quad_strips(b->desc[p].lb)
                                              : fwd_strips_w(b->desc[p].lb, 1u << b->desc[p].f_wlog2);
        for(uint32_t st = 0; st < nf; ++st) fwd_items.push_back(WorkItem{p, st});
    }
    // fused two-strip pairs (common.hpp: kCkFusedFirst): where the launch has more than two rounds of wavefronts' worth of items --
    // below that the strips of a pair are better off side by side, and the spliced traceback wants them so
    if(b->ck && L == 1) {
        // (the chunks of a streamed call -- any size, they share ONE persistent launch -- always)
        const bool fuse = env.ck_fuse >= 0 ? env.ck_fuse != 0 : (items.size() > 2ull * ck_scratch_waves() || (opts != nullptr && opts->wave_slot_dwords != 0));
        for(size_t q = 0; fuse && q + 1 < items.size(); ++q) {
            const PairDesc& d = b->desc[items[q].pair];
            if(items[q].strip == 0 && d.v_strips == 2 && d.v_wmain == kW && d.v_wlast == 4 && d.v_parts < 2 && d.la > 0 && items[q + 1].pair == items[q].pair) {
                items[q].strip |= kCkFusedFirst << 16;
                items[q + 1].strip |= kCkFusedSecond << 16;
                ++q;
            }
        }
    }
    if(b->ck_split_items > 0) {
        const uint32_t parts = ck_parts_count(b->desc[order[n_pairs - 1]].v_parts);
        // (+ one item per cut pair for its traceback where that is an item of its own: "part" number `parts`)
        for(uint32_t part = 1; part < parts + (b->ck_walk_items ? 1u : 0u); ++part)
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
    // (the ticket counter sits right in front of the progress words: one fill zeroes both before a launch, common.hpp)
    const uint64_t o_desc = carve(n_pairs * sizeof(PairDesc)), o_order = carve(n_pairs * sizeof(uint32_t)),
                   o_items = carve(items.size() * sizeof(WorkItem)), o_fwd = carve(fwd_items.size() * sizeof(WorkItem)),
                   o_queue = carve(sizeof(uint32_t)),
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
        std::memset(st + o_queue, 0, o_a - o_queue);  // (ticket counter + progress words)
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

}  // namespace coati_hip_abi
