// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// C-ABI shim over the *unmodified* reference DP engine, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile, target
// `ref`).  It can only be BUILT in the build container (the GPU box has no
// /root/reference); the built oracle/_ref/libcoati_ref.so travels to the GPU box
// with the snapshot like the product's own .so files and is used there as a
// checker (tests) and as bench.py's cpu_baseline (kind "reference") only.  It is
// used (a) to pin oracle/coati_oracle.cc bit-for-bit against the reference,
// (b) by tools/make_golden*.py to generate the fixtures under tests/golden/, and
// (c) as the timed CPU baseline beside the GPU numbers.
//
// Reference entry points driven here (src/include/coati/align_pair.hpp:157-182):
//   viterbi_mem + traceback_viterbi   (align_pair.cc:195, :319)
//   forward + sampleback              (align_pair.cc:149, :401)
//   fragmites::random string seeding  (contrib/random/random.hpp:522-540)
#include <coati/align_pair.hpp>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

void fill_aln(coati::alignment_t& aln, const float* table, float gap_open,
              float gap_extend, int gap_len) {
    aln.subst_matrix = coati::Matrixf(183, 15, table, table + 183 * 15);
    aln.gap.open = gap_open;
    aln.gap.extend = gap_extend;
    aln.gap.len = static_cast<std::size_t>(gap_len);
}

void copy_matrix(const coati::Matrixf& m, float* out) {
    if(out == nullptr) return;
    for(std::size_t i = 0; i < m.rows(); ++i)
        for(std::size_t j = 0; j < m.cols(); ++j) out[i * m.cols() + j] = m(i, j);
}

coati::random_t seeded(const char* const* seeds, int nseeds) {
    std::vector<std::string> v;
    for(int k = 0; k < nseeds; ++k) v.emplace_back(seeds[k]);
    coati::random_t rand;
    auto ss = fragmites::random::string_seed_seq(v.begin(), v.end());
    rand.Seed(ss);
    return rand;
}

}  // namespace

extern "C" {

// Viterbi fill + traceback through the reference.  `M/D/I` (rows*cols floats,
// rows = la + L, cols = lb + L) may be NULL.  aln_a/aln_b need la+lb+1 bytes.
int ref_viterbi(const float* table, float gap_open, float gap_extend, int gap_len,
                const char* a_raw, const char* b_raw, const uint8_t* a_enc,
                const uint8_t* b_enc, uint64_t la, uint64_t lb, float* M, float* D,
                float* I, char* aln_a, char* aln_b, float* score) {
    try {
        coati::alignment_t aln;
        fill_aln(aln, table, gap_open, gap_extend, gap_len);
        coati::align_pair_work_mem_t work;
        coati::seq_view_t a(a_enc, la), b(b_enc, lb);
        coati::viterbi_mem(work, a, b, aln);
        copy_matrix(work.mch, M);
        copy_matrix(work.del, D);
        copy_matrix(work.ins, I);
        std::string sa(a_raw, la), sb(b_raw, lb);
        coati::traceback_viterbi(work, sa, sb, aln, aln.gap.len);
        std::strcpy(aln_a, aln.seq(0).c_str());
        std::strcpy(aln_b, aln.seq(1).c_str());
        *score = aln.data.score;
        return 0;
    } catch(...) {
        return 1;
    }
}

// Forward fill (log semiring, 11 matrices) followed by n_samples sampleback
// walks that share one RNG stream, as marg_sample does
// (src/lib/align_marginal.cc:586-593).  `mats` (11 * rows*cols floats, order
// mch del ins mch_mch mch_del mch_ins del_mch del_del ins_mch ins_del ins_ins)
// may be NULL.  aln_out receives 2*n_samples NUL-terminated strings, each in
// a slot of (la+lb+1) bytes.
int ref_forward_sample(const float* table, float gap_open, float gap_extend,
                       int gap_len, const char* a_raw, const char* b_raw,
                       const uint8_t* a_enc, const uint8_t* b_enc, uint64_t la,
                       uint64_t lb, const char* const* seeds, int nseeds,
                       int n_samples, float* mats, char* aln_out, float* scores) {
    try {
        coati::alignment_t aln;
        fill_aln(aln, table, gap_open, gap_extend, gap_len);
        coati::align_pair_work_t work;
        coati::seq_view_t a(a_enc, la), b(b_enc, lb);
        coati::forward(work, a, b, aln);
        if(mats != nullptr) {
            const std::size_t n = work.mch.rows() * work.mch.cols();
            const coati::Matrixf* all[11] = {
                &work.mch,     &work.del,     &work.ins,     &work.mch_mch,
                &work.mch_del, &work.mch_ins, &work.del_mch, &work.del_del,
                &work.ins_mch, &work.ins_del, &work.ins_ins};
            for(int k = 0; k < 11; ++k) copy_matrix(*all[k], mats + k * n);
        }
        coati::random_t rand = seeded(seeds, nseeds);
        std::string sa(a_raw, la), sb(b_raw, lb);
        const std::size_t slot = la + lb + 1;
        for(int k = 0; k < n_samples; ++k) {
            coati::sampleback(work, sa, sb, aln, aln.gap.len, rand);
            std::strcpy(aln_out + (2 * k) * slot, aln.seq(0).c_str());
            std::strcpy(aln_out + (2 * k + 1) * slot, aln.seq(1).c_str());
            scores[k] = aln.data.score;
        }
        return 0;
    } catch(...) {
        return 1;
    }
}

// First n f24() draws after string seeding.
int ref_rng_f24(const char* const* seeds, int nseeds, int n, float* out) {
    coati::random_t rand = seeded(seeds, nseeds);
    for(int k = 0; k < n; ++k) out[k] = rand.f24();
    return 0;
}

}  // extern "C"
