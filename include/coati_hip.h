/* coati_hip.h -- C ABI of libcoati_hip.so, the MI355X (gfx950) implementation of
 * COATi's marginal pairwise-alignment hot path.
 *
 * The reference (CartwrightLab/coati) has no FFI layer; the boundary this
 * library replaces is the C++ API of its pairwise DP engine as called from the
 * marginal workflow drivers:
 *
 *   viterbi_mem        src/include/coati/align_pair.hpp:170  (align_pair.cc:195)
 *   traceback_viterbi  src/include/coati/align_pair.hpp:176  (align_pair.cc:319)
 *   forward            src/include/coati/align_pair.hpp:161  (align_pair.cc:149)
 *   sampleback         src/include/coati/align_pair.hpp:180  (align_pair.cc:401)
 *   call sites         src/lib/align_marginal.cc:71,80,586,590; src/lib/align_msa.cc:307-308
 *
 * The reference aligns one pair per process (src/lib/utils.cc:810-812); this
 * ABI is batched: a *model* (183x15 marginal substitution table + gap
 * constants, all host-computed exactly as the reference computes them) and a
 * *batch* of encoded sequence pairs that stays resident in HBM.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every entry
 * point returns 0 on success and a non-zero code on failure, in which case
 * coati_hip_last_error() describes the failure (thread-local).  No exceptions
 * cross the boundary.  Handles are thread-compatible: distinct handles may be
 * used from distinct threads; one handle must not be used concurrently.
 * There is NO CPU fallback: without a usable gfx950 device every compute call
 * fails with COATI_HIP_ENODEVICE.
 *
 * Sequence encoding (marginal_seq_encoding, src/lib/utils.cc:496-528):
 *   a[]: ancestor, one byte per nucleotide, value codon61*3+phase in [0,183)
 *   b[]: descendant, one byte per nucleotide, nt16 code in [0,15)
 * Pair p occupies a_cat[a_off[p] .. a_off[p+1]) and b_cat[b_off[p] .. b_off[p+1]).
 *
 * Alignment ops: one byte per alignment COLUMN, left to right:
 *   0 = match/mismatch, 1 = deletion (gap in descendant), 2 = insertion (gap in
 *   ancestor).  From them the host rebuilds the gapped strings exactly as
 *   traceback<> does (align_pair.cc:268-302).
 */
#ifndef COATI_HIP_H
#define COATI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COATI_HIP_TABLE_ROWS 183
#define COATI_HIP_TABLE_COLS 15

enum {
    COATI_HIP_OK = 0,
    COATI_HIP_EINVAL = 1,    /* bad argument (NULL, code out of range, gap_len < 1 ...) */
    COATI_HIP_ENODEVICE = 2, /* no gfx950 device / HIP runtime unusable */
    COATI_HIP_ENOMEM = 3,    /* device or host allocation failed (the reference's
                                "sequences to align exceed available memory",
                                src/lib/align_marginal.cc:72-75) */
    COATI_HIP_EHIP = 4,      /* a HIP call or kernel failed */
    COATI_HIP_ESTATE = 5     /* call out of order (fetch before launch ...) */
};

enum { COATI_HIP_OP_MATCH = 0, COATI_HIP_OP_DEL = 1, COATI_HIP_OP_INS = 2 };

typedef struct coati_hip_model coati_hip_model_t;
typedef struct coati_hip_batch coati_hip_batch_t;

/* Library/ABI version (major << 16 | minor). */
uint32_t coati_hip_version(void);

/* Number of usable gfx950 devices (0 if none; never fails). */
int coati_hip_device_count(void);

/* Thread-local description of the last failure of any call on this thread. */
const char* coati_hip_last_error(void);

/* ---- model -------------------------------------------------------------- *
 * Replaces the read side of `const alignment_t&` in viterbi_mem/forward
 * (aln.subst_matrix, aln.gap; src/include/coati/structs.hpp:37-50,82).
 * table: 183*15 fp32, row-major, = marginal_p() output
 *        (src/lib/mutation_coati.cc:164-199).
 * no_gap/gap_stop/gap_open/gap_extend: log1pf(-g), log1pf(-e), logf(g), logf(e)
 *        computed on the HOST (align_pair.cc:66-69) so that device results are
 *        a pure function of the inputs.
 * gap_len: aln.gap.len >= 1.   device: HIP device ordinal. */
int coati_hip_model_create(const float* table, float no_gap, float gap_stop, float gap_open,
                           float gap_extend, int gap_len, int device,
                           coati_hip_model_t** out);
/* The same with n_tables substitution tables (tables: n_tables * 183*15 fp32).  What it is
 * for: `coati msa` aligns every leaf to the reference with the leaf's own branch length, i.e.
 * its own marginal table (src/lib/align_msa.cc:285-318: set_subst per leaf, then viterbi_mem);
 * one batch can then carry pairs of different tables (coati_hip_batch_create_tables). */
int coati_hip_model_create_tables(const float* tables, uint32_t n_tables, float no_gap, float gap_stop,
                                  float gap_open, float gap_extend, int gap_len, int device,
                                  coati_hip_model_t** out);
/* Batches created from a model keep it alive: destroying the model first is allowed (its
 * batches stay usable, the model's memory is released with the last of them); creating new
 * batches from a destroyed handle is not. */
void coati_hip_model_destroy(coati_hip_model_t* model);
/* A model keeps up to four HBM blocks (of at most 16 GB each) that its destroyed batches and
 * finished sampling calls no longer need -- batch workspaces, Forward matrices, the sampler's
 * temporaries -- and hands them to the next coati_hip_batch_create / forward_launch / sampleback
 * whose needs they fit.  (The reference has no counterpart: its work matrices are std::vectors that
 * die with align_pair_work_mem_t, align_pair.hpp:45-62; here a multi-GB hipMalloc costs between
 * 0.4 and 500 ms, which a loop over batches should not pay per batch.)  This call frees what is
 * cached; coati_hip_model_destroy does it too. */
int coati_hip_model_trim(coati_hip_model_t* model);
/* Per-model switches.  (No reference counterpart: they concern how this library shares the GPU with its
 * embedder, which the CPU reference never has to.)
 *   COATI_HIP_OPT_PERSISTENT_CALL (default 1): 0 forbids the device-owning form of coati_hip_viterbi_batch --
 *     one persistent kernel per call that holds every wavefront slot of the GPU until the call returns, so that
 *     any other kernel of the process (and any copy the runtime does with a kernel) waits for it.  With 0 every
 *     call uses one launch per chunk (identical results, ~0.85 instead of ~0.95 of the resident kernel's rate)
 *     and other streams of the embedder interleave normally.
 *   COATI_HIP_OPT_CK_BAND (default 64, or the environment's COATI_HIP_CK_BAND): half width, in wavefront steps, of the
 *     band around a pair's straight line (0,0) -> (len_a, len_b) inside which the gap_len-1 kernel keeps the
 *     checkpoints its traceback recomputes decisions from; 0 keeps all of them.  A narrower band writes fewer bytes
 *     (the kernel is power-bound: bytes are clock); an alignment whose path leaves the band is detected by the walk and
 *     the pair is filled a second time with everything kept -- same bits, twice the cost for that pair
 *     (coati_hip_viterbi_band_stats counts them).  Related sequences stay inside 96 steps; for inputs with indels of
 *     hundreds of bases in many pairs set a wider band or 0.  Takes effect for batches launched afterwards.
 *   COATI_HIP_OPT_FORWARD_MODE (default COATI_HIP_FORWARD_EXACT, or COATI_HIP_FORWARD_TOLERANCE where the environment has
 *     COATI_HIP_FORWARD_FAST=1): how the Forward kernels evaluate log_sum_exp (semiring.hpp:86-121, utils.hpp:134-156).
 *     EXACT: device restatements of glibc 2.35's expf / log1pf -- every M/D/I value, hence every sample's log-weight, has the
 *     bits the reference computes on an x86-64 glibc host (~100 GCUPS of Forward fill).  TOLERANCE: the hardware's exp2 / log2
 *     instructions -- log-weights within 1e-5 relative of the reference's (the contract's bound), sampled paths may differ from
 *     the reference's where two branch probabilities are that close (~380 GCUPS).  The sampler's own arithmetic on the stored
 *     M/D/I is the same in both.  Applies to batches CREATED afterwards (the strip plan depends on the mode).
 * Returns COATI_HIP_EINVAL for an unknown option or value. */
enum { COATI_HIP_OPT_PERSISTENT_CALL = 1, COATI_HIP_OPT_CK_BAND = 2, COATI_HIP_OPT_FORWARD_MODE = 3 };
enum { COATI_HIP_FORWARD_EXACT = 0, COATI_HIP_FORWARD_TOLERANCE = 1 };
int coati_hip_model_set_option(coati_hip_model_t* model, int option, int64_t value);

/* Warm-up for a coming coati_hip_viterbi_batch call of about n_pairs pairs of about len_a x len_b positions: the
 * device workspaces and page-locked staging blocks that call would allocate first (several GB: ~50-100 ms in a fresh
 * process) are allocated now -- e.g. on a helper thread while the caller is still reading its input
 * (coati-alignpair --batch does).  The hints only size things: a call that needs more allocates more.  No reference
 * counterpart (the reference allocates its three matrices per pair, align_pair.hpp:45-62). */
int coati_hip_model_prepare(coati_hip_model_t* model, uint64_t n_pairs, uint64_t len_a, uint64_t len_b);

/* ---- batch -------------------------------------------------------------- *
 * Validates and uploads n_pairs encoded pairs (host pointers) and reserves the
 * HBM workspace for the Viterbi path.  a_off/b_off have n_pairs+1 entries.
 * Every len_a must be a multiple of 3 and of gap_len, every len_b a multiple
 * of gap_len (process_marginal, src/lib/utils.cc:822-835). */
int coati_hip_batch_create(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                           const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                           coati_hip_batch_t** out);
/* The same, pair p using table table_index[p] (< n_tables of the model); NULL = table 0 for all. */
int coati_hip_batch_create_tables(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                                  const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                                  const uint32_t* table_index, coati_hip_batch_t** out);
void coati_hip_batch_destroy(coati_hip_batch_t* batch);

/* Number of pairs of a batch. */
uint64_t coati_hip_batch_pairs(const coati_hip_batch_t* batch);
/* Bytes of HBM a batch holds (inputs + workspace + results). */
uint64_t coati_hip_batch_device_bytes(const coati_hip_batch_t* batch);
/* Sum over pairs of len_a*len_b ("cell updates"). */
uint64_t coati_hip_batch_cells(const coati_hip_batch_t* batch);

/* viterbi_mem + traceback_viterbi for every pair of the batch: enqueues the
 * fill kernel and the traceback walker on the model's stream and returns
 * without waiting. */
int coati_hip_viterbi_launch(coati_hip_batch_t* batch);
/* Wait for everything enqueued on the batch's model (its stream). */
int coati_hip_batch_sync(coati_hip_batch_t* batch);
/* Wait for THIS batch's most recent Viterbi launch only: launches of other batches of the same
 * model that were enqueued after it keep running (e.g. fetch or gather the results of batch A
 * while batch B computes). */
int coati_hip_viterbi_wait(coati_hip_batch_t* batch);

/* Copy results to host (synchronises first).  Any output may be NULL.
 *   scores[n_pairs]   aln.data.score of traceback (align_pair.cc:265), fp32
 *   ops[ops_capacity] op bytes of all pairs; pair p's ops are
 *                     ops[ops_off[p] .. ops_off[p] + ops_len[p])
 *   ops_off[n_pairs], ops_len[n_pairs]
 * ops_capacity must be >= sum(len_a + len_b); pair p's ops lie inside the
 * slot [sum_{q<p}(len_a+len_b), +len_a+len_b). */
int coati_hip_viterbi_fetch(coati_hip_batch_t* batch, float* scores, uint8_t* ops,
                            uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len);

/* Device times (ms) of the last viterbi launch, from HIP events recorded on the
 * model's stream around each kernel (synchronises first). */
int coati_hip_viterbi_last_timing(coati_hip_batch_t* batch, float* fill_ms, float* walk_ms);
/* How the banded checkpoints (COATI_HIP_OPT_CK_BAND) fared in the batch's last launch (waits for it): *band_steps = the
 * band's half width the launch ran with (0: everything was kept, or another kernel than the banded one ran),
 * *pairs_refilled = pairs whose traceback left the band and that were therefore filled twice.  No reference
 * counterpart (the reference keeps three fp32 matrices, align_pair.hpp:45-62). */
int coati_hip_viterbi_band_stats(coati_hip_batch_t* batch, uint32_t* band_steps, uint64_t* pairs_refilled);
/* Same for the launch issued `launches_back` launches before the last one (0 =
 * last; the most recent 64 launches are kept), so that a caller can enqueue many
 * launches back to back and read their kernel times afterwards. */
int coati_hip_viterbi_timing(coati_hip_batch_t* batch, uint32_t launches_back, float* fill_ms,
                             float* walk_ms);

/* Device pointers of the result arrays of a batch (valid until the batch is
 * destroyed; contents valid after a synchronised viterbi launch), so that a
 * caller can hand them to a collective (RCCL gather) without a host round trip:
 *   scores    float[n_pairs]
 *   ops       uint8_t[sum(len_a+len_b)]
 *   ops_off   uint64_t[n_pairs]  (absolute offset of pair p's first op in `ops`)
 *   ops_len   uint32_t[n_pairs]
 * Any output pointer may be NULL. */
int coati_hip_batch_result_ptrs(coati_hip_batch_t* batch, void** scores, void** ops, uint64_t* ops_bytes,
                                void** ops_off, void** ops_len);

/* ---- Forward (coati sample) ------------------------------------------------ *
 * forward(): forward_impl<log, align_pair_work_t> (align_pair.cc:62-139,149) for
 * every pair of the batch.  Keeps the fp32 M/D/I of all body cells resident in HBM
 * (12 bytes per cell, reserved on the first call) for coati_hip_sampleback; the
 * eight edge matrices of align_pair_work_t are recomputed on demand.  Enqueues and
 * returns.
 * Numerics: log_sum_exp (semiring.hpp:86-121, utils.hpp:134-156) is evaluated with device
 * restatements of glibc 2.35's expf / log1pf, so every M/D/I value has the bits the reference
 * computes on an x86-64 glibc host.  A model in COATI_HIP_FORWARD_TOLERANCE mode (COATI_HIP_OPT_FORWARD_MODE above) uses the
 * hardware exp2/log2 instructions instead (3.8x the throughput, log-weights within 1e-5 relative). */
int coati_hip_forward_launch(coati_hip_batch_t* batch);
/* Terminal-adjusted M, D, I of the last cell (align_pair.cc:130-138), 3 floats per
 * pair (synchronises). */
int coati_hip_forward_final(coati_hip_batch_t* batch, float* final_mdi);
/* Parity/debug export: the Forward M, D, I of the len_a x len_b body cells of one
 * pair as three row-major matrices (the last cell NOT terminal-adjusted). */
int coati_hip_debug_forward_matrices(coati_hip_batch_t* batch, uint64_t pair, float* M, float* D, float* I,
                                     uint64_t capacity);

/* sampleback() (align_pair.cc:401-458) n_samples times for every pair of a batch
 * whose Forward matrices are resident (coati_hip_forward_launch).  Synchronous.
 *   rng_state[2*p], rng_state[2*p+1]   low / high 64 bits of pair p's Lehmer64Fast state
 *                      (contrib/random/random.hpp:80-136), i.e. what rand.Seed(seed_seq)
 *                      leaves (random.hpp:408-413) -- seeding stays on the host.
 *   independent_streams = 0: pair p's samples are drawn one after the other from that one
 *                      stream, exactly as marg_sample does (align_marginal.cc:590-593);
 *                      rng_state_out (may be NULL) receives the advanced states.
 *   independent_streams = 1: sample s of pair p starts s * 2^32 draws into the stream
 *                      (all samples in parallel; sample 0 equals the reference's first).
 * Outputs (any may be NULL), sample s of pair p at index p * n_samples + s:
 *   log_weights[]  aln.data.score of the sample
 *   ops / ops_off[] / ops_len[]  as for Viterbi; ops_capacity >= n_samples * sum(len_a+len_b) (the bound; the samples'
 *                                ops are written back to back from ops[0] on, in output order: ops_off[] is ascending) */
int coati_hip_sampleback(coati_hip_batch_t* batch, uint32_t n_samples, const uint64_t* rng_state,
                         int independent_streams, float* log_weights, uint8_t* ops, uint64_t ops_capacity,
                         uint64_t* ops_off, uint32_t* ops_len, uint64_t* rng_state_out);
/* Warm-up for a coming coati_hip_sampleback(batch, n_samples, ..., independent_streams, ...): the device blocks (result slots,
 * the speculation's candidates and step table: ~0.25 GB for 16 pairs of 1 kb) and the page-locked round records that call would
 * allocate first are allocated now and left in the model's caches.  Needs coati_hip_forward_launch to have been CALLED, not to
 * have finished: made right behind it, the allocations run while the Forward kernel does (`coati-sample` does so) -- a process
 * makes one sampleback call, and its first-call cost is then what a later call's is.  Optional; identical results.  (The
 * reference has no counterpart: marg_sample's work matrices are allocated by forward(), align_marginal.cc:584-588.) */
int coati_hip_sampleback_prepare(coati_hip_batch_t* batch, uint32_t n_samples, int independent_streams);
/* Parity/debug: the device's bit-exact restatements of the libm functions the log-semiring path
 * calls (glibc 2.35; coati_amd/csrc/glibc_math.hpp), applied element-wise.
 *   op 0: expf(x), x <= 0      (log1p_exp, utils.hpp:134-146; sample_mdi, align_pair.cc:336-358)
 *   op 1: log1pf(x), 0 <= x <= 1
 *   op 2: logf(x), x > 0
 *   op 3: log1pf(x), 2^-29 <= x <= 1: the straight-line form the Forward fill uses (same bits as op 1) */
int coati_hip_debug_libm(coati_hip_model_t* model, int op, const float* in, uint64_t n, float* out);
/* Parity/debug: the first n f24() draws (random.hpp:213-216) of a stream, computed on the device. */
int coati_hip_debug_rng_f24(coati_hip_model_t* model, const uint64_t rng_state[2], uint32_t n, float* out);

/* One-shot viterbi_mem + traceback_viterbi over any number of pairs -- the batched counterpart of the
 * reference's per-pair loop (src/lib/align_marginal.cc:69-80 called once per process, utils.cc:809-812).
 * Pairs arrive in host memory and alignments leave to host memory; uploads, compute and downloads overlap.
 * Large gap_len 1 inputs (from 4 096 pairs of >= 250 x 250 cells, no pair wider than 8 192 columns or larger
 * than 64 M cells) run as ONE persistent kernel that the host feeds chunk by chunk over 12 slots (HBM
 * workspace + page-locked staging, kept by the model between calls); while such a call runs it owns the
 * device, other work on the GPU waits until it returns.  Everything else is cut into chunks that fit the
 * device's memory and pipelined with one launch per chunk over three slots.  Either way results are
 * identical to a resident batch's.  Sequence arrays from coati_hip_host_alloc are copied from directly (DMA),
 * pageable ones pass through a slot's staging block.  Results of the persistent-kernel form are stored BY THE
 * KERNEL into host memory as each pair's traceback ends -- into page-locked result arrays themselves (nothing is
 * downloaded; all four, or none: a NULL array is fine), or into the slot's staging block, from where helper threads
 * copy them into pageable arrays; the chunk pipeline downloads them per chunk.  The result arrays hold unspecified
 * bytes outside [ops_off[p], ops_off[p] + ops_len[p]).  One call at a time per model (serialised inside). */
int coati_hip_viterbi_batch(coati_hip_model_t* model, uint64_t n_pairs, const uint8_t* a_cat,
                            const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                            float* scores, uint8_t* ops, uint64_t ops_capacity,
                            uint64_t* ops_off, uint32_t* ops_len);

/* Contiguous shards of (nearly) equal DP-cell count for `world` devices: pairs
 * [bounds[r], bounds[r+1]) go to rank r; bounds has world+1 entries, bounds[0] = 0, bounds[world] =
 * n_pairs.  The cut before rank r is the first pair index at which the running cell count reaches
 * r/world of the total, so no shard exceeds the ideal share by more than one pair.  (The reference has
 * no counterpart: one pair per process, src/lib/utils.cc:809-812.)  Pure host arithmetic: usable
 * without a device, by every rank on the same input. */
int coati_hip_shard_bounds(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world,
                           uint64_t* bounds);

/* Page-locked host memory (hipHostMalloc behind a C signature) for the arrays handed to
 * coati_hip_viterbi_batch; free with coati_hip_host_free.  Optional: any host memory works. */
int coati_hip_host_alloc(uint64_t bytes, void** out);
void coati_hip_host_free(void* p);

/* The library reads its COATI_HIP_* environment switches (A/B, test and experiment knobs; csrc/common.hpp: EnvOptions)
 * ONCE per process, on first use.  This reads them again: for test harnesses that switch kernels within one process. */
void coati_hip_debug_reload_env(void);

/* Parity/debug export: the per-cell traceback decision byte of pair `pair`
 * (bits 0-1 state after a match move arrives at the cell, bits 2-3 after a
 * deletion move, bit 4 after an insertion move) for the len_a x len_b BODY
 * cells (matrix rows/cols gap_len.., row-major), decoded from the packed HBM
 * layout.  Valid after a synchronised viterbi launch. */
int coati_hip_debug_viterbi_flags(coati_hip_batch_t* batch, uint64_t pair, uint8_t* out,
                                  uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* COATI_HIP_H */
