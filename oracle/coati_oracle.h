/* TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of COATi's marginal
 * pairwise DP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported
 * baseline.  The product (coati_amd/, include/coati_hip.h) never links or calls
 * it and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  Checked bit-for-bit (fp32 matrices, scores, aligned
 * strings, sampled alignments and log-weights, RNG stream) against the
 * unmodified reference compiled in the build container (oracle/_ref, see
 * tests/test_oracle_vs_ref.py) and against the committed golden vectors under
 * tests/golden/ that were generated from it (tools/make_golden.py).
 *
 * All matrices are row-major fp32, rows = len_a + gap_len, cols = len_b + gap_len,
 * exactly the reference's work matrices (src/include/coati/align_pair.hpp:45-147).
 */
#ifndef COATI_ORACLE_H
#define COATI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_TROPICAL = 0, ORACLE_LOG = 1 };
enum { ORACLE_OP_M = 0, ORACLE_OP_D = 1, ORACLE_OP_I = 2 };

/* consts[4] = {no_gap, gap_stop, gap_open, gap_extend} in log space, computed
 * with the host libm exactly as semiring.hpp:117-120 does. */
void oracle_gap_consts(float gap_open, float gap_extend, float consts[4]);

/* forward_impl<S,W> (src/lib/align_pair.cc:62-139).  `edges` is NULL (3-matrix
 * work struct) or 8 matrices in the order mch_mch mch_del mch_ins del_mch
 * del_del ins_mch ins_del ins_ins (align_pair.hpp:94-103).  Returns 0. */
int oracle_fill(int semiring, const float* table, const float consts[4], int gap_len,
                const uint8_t* a, uint64_t len_a, const uint8_t* b, uint64_t len_b,
                float* M, float* D, float* I, float* edges);

/* traceback<tropical> (align_pair.cc:249-303).  Writes one op per alignment
 * COLUMN (a deletion/insertion move emits gap_len equal ops) in left-to-right
 * order; returns the number of columns (<= len_a + len_b) and the score. */
int64_t oracle_traceback(const float* M, const float* D, const float* I, uint64_t rows,
                         uint64_t cols, const float consts[4], int gap_len, uint8_t* ops,
                         float* score);

/* Per-cell decision byte, the encoding the GPU kernels store (SURVEY.md §8(a)
 * row 7): bits 0-1 = state entered after a MATCH move arrives at the cell,
 * bits 2-3 = after a DELETION move, bit 4 = after an INSERTION move (0 = M,
 * 1 = I).  Derived from the cell's final M/D/I with max_mdi / max_mi
 * (align_pair.cc:210-232) and the expressions of align_pair.cc:275-296. */
void oracle_tb_flags(const float* M, const float* D, const float* I, uint64_t rows,
                     uint64_t cols, const float consts[4], uint8_t* flags);

/* Walk the decision bytes instead of the matrices (what the GPU walker does). */
int64_t oracle_traceback_flags(const uint8_t* flags, uint64_t rows, uint64_t cols, int gap_len,
                               uint8_t start_state, uint8_t* ops);

/* viterbi_mem + traceback_viterbi with internal storage.  Returns columns. */
int64_t oracle_viterbi(const float* table, const float consts[4], int gap_len, const uint8_t* a,
                       uint64_t len_a, const uint8_t* b, uint64_t len_b, uint8_t* ops,
                       float* score);

/* Same result as oracle_viterbi but O(cols) float storage + 1 B/cell decision
 * bytes (for pairs whose three fp32 matrices do not fit in RAM). */
int64_t oracle_viterbi_lowmem(const float* table, const float consts[4], int gap_len,
                              const uint8_t* a, uint64_t len_a, const uint8_t* b, uint64_t len_b,
                              uint8_t* ops, float* score);

/* Rebuild the gapped strings the reference emits (align_pair.cc:270-302).
 * out_a/out_b need n_ops+1 bytes. */
void oracle_ops_to_strings(const uint8_t* ops, int64_t n_ops, const char* a_raw,
                           const char* b_raw, char* out_a, char* out_b);

/* Lehmer64Fast + SeedSeq<8> + string_seed_seq (contrib/random/random.hpp:80-136,
 * 334-413, 522-540). */
typedef struct {
    uint64_t lo, hi;
} oracle_rng_t;
void oracle_rng_seed(oracle_rng_t* rng, const char* const* seeds, int nseeds);
uint64_t oracle_rng_bits(oracle_rng_t* rng);
float oracle_rng_f24(oracle_rng_t* rng);

/* sampleback (align_pair.cc:401-458) over the 11 reference matrices
 * (mats = M D I followed by the 8 edge matrices in oracle_fill order). */
int64_t oracle_sampleback(const float* mats, uint64_t rows, uint64_t cols, int gap_len,
                          oracle_rng_t* rng, uint8_t* ops, float* score);

/* sampleback that needs only M/D/I: the 8 edge values are recomputed on demand
 * from the neighbours with the fill's own expressions (what the GPU walker
 * does).  Bit-identical to oracle_sampleback. */
int64_t oracle_sampleback_mdi(const float* M, const float* D, const float* I, uint64_t rows,
                              uint64_t cols, const float* table, const float consts[4],
                              int gap_len, const uint8_t* a, const uint8_t* b,
                              oracle_rng_t* rng, uint8_t* ops, float* score);

/* log-weight the sampleback arithmetic assigns to a GIVEN path (ops per
 * column, left to right).  Used to check GPU-sampled paths without requiring
 * the same random choices. */
float oracle_path_logweight(const float* M, const float* D, const float* I, uint64_t rows,
                            uint64_t cols, const float* table, const float consts[4],
                            int gap_len, const uint8_t* a, const uint8_t* b,
                            const uint8_t* ops, int64_t n_ops);

/* Size-independent check for pairs too large for the full oracle: re-derive, in
 * O(n_ops), the Viterbi value the fill assigns along a GIVEN path (ops per column),
 * with the recurrences and margin formulas of align_pair.cc:82-138 in their own
 * evaluation order.  For the optimal path this equals the DP's final score bit for
 * bit.  Returns lowest() if the ops do not describe a path of this pair. */
float oracle_path_score(const float* table, const float consts[4], int gap_len, const uint8_t* a,
                        uint64_t len_a, const uint8_t* b, uint64_t len_b, const uint8_t* ops,
                        int64_t n_ops);

/* The host libm itself, element-wise (what the reference calls): op 0 expf, 1 log1pf, 2 logf. */
void oracle_libm(int op, const float* in, uint64_t n, float* out);

/* Timed CPU baseline: run oracle_viterbi (reference data layout: three fp32
 * matrices incl. their fill) over a batch on `threads` host threads, one pair
 * per thread at a time.  Returns wall seconds. */
double oracle_viterbi_batch_timed(const float* table, const float consts[4], int gap_len,
                                  uint64_t n_pairs, const uint8_t* a_cat, const uint64_t* a_off,
                                  const uint8_t* b_cat, const uint64_t* b_off, int threads,
                                  float* scores);

#ifdef __cplusplus
}
#endif
#endif
