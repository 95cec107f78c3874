/* coati_hip_dist.h -- C ABI of libcoati_hip_dist.so: the multi-GPU layer over libcoati_hip.so.
 *
 * What it replaces: nothing in the reference is multi-device -- `coati alignpair` aligns ONE pair per
 * process (src/lib/utils.cc:809-812, src/lib/align_marginal.cc:44-88), and a user with a million pairs
 * runs a million processes.  Pairs are independent, so the batched engine shards them: ONE PROCESS PER
 * GPU, contiguous shards of equal DP-cell count (coati_hip_shard_bounds, in libcoati_hip.so), no
 * collective on the data path.  The exchanges that are real:
 *   * the model (marginal substitution table(s) + gap constants, 11 KB per table) is computed on rank 0
 *     and broadcast, so that every rank scores with bit-identical tables: ncclBroadcast;
 *   * the results return to rank 0, which writes the output: an ncclAllGather of every rank's counts,
 *     then one group of ncclSend / ncclRecv straight out of the ranks' HBM result arrays over xGMI.
 * RCCL is linked directly (librccl.so); torch is not involved.
 *
 * Conventions as in coati_hip.h: 0 on success, otherwise an error code with the text in
 * coati_hip_dist_last_error() (thread-local).  Every function taking a communicator is COLLECTIVE:
 * all ranks of the communicator must call it, in the same order.
 */
#ifndef COATI_HIP_DIST_H
#define COATI_HIP_DIST_H

#include "coati_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define COATI_HIP_DIST_ID_BYTES 128

typedef struct coati_hip_comm coati_hip_comm_t;

const char* coati_hip_dist_last_error(void);

/* Rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its 128 bytes to the other ranks out of
 * band (a file, a pipe, an environment variable, a torch store ...). */
int coati_hip_dist_unique_id(void* id128);
/* Joins the communicator: `world` processes, this one is `rank` and drives HIP device `device`
 * (ncclCommInitRank; creates the stream the collectives run on). */
int coati_hip_dist_init(const void* id128, int world, int rank, int device, coati_hip_comm_t** out);
void coati_hip_dist_destroy(coati_hip_comm_t* comm);
int coati_hip_dist_rank(const coati_hip_comm_t* comm);
int coati_hip_dist_world(const coati_hip_comm_t* comm);

/* Small collectives for drivers that time or sequence a multi-rank job (bench.py, the --devices launcher):
 * an element-wise all-reduce of up to 64 doubles (op 0 = sum, 1 = max; ncclAllReduce) and a barrier (an
 * all-reduce of one element, completed on the communicator's stream before the call returns). */
int coati_hip_dist_allreduce_f64(coati_hip_comm_t* comm, int op, double* values, uint32_t n);
int coati_hip_dist_barrier(coati_hip_comm_t* comm);

/* Broadcast of the model from `root`.  On root: tables (n_tables * 183*15 floats), *n_tables, consts
 * (no_gap, gap_stop, gap_open, gap_extend), *gap_len are inputs; on the other ranks they are outputs
 * (`tables` must hold table_capacity_tables tables; fails if the root's model has more). */
int coati_hip_dist_broadcast_model(coati_hip_comm_t* comm, int root, float* tables, uint32_t table_capacity_tables,
                                   uint32_t* n_tables, float consts[4], int* gap_len);

/* Gather of ONE launched resident batch per rank (NULL = this rank contributes nothing) to `root`.
 * Waits for the batch's Viterbi launch, then: ncclAllGather of (pairs, op bytes) of every rank into
 * counts[2 * world] (valid on every rank), grouped ncclSend / ncclRecv of scores, ops, op offsets and op
 * lengths from the batches' HBM result arrays into root's HBM, and on root the download into
 *   scores[], ops_off[], ops_len[]  -- rank r's entries start at index sum_{q<r} counts[2q]
 *   ops[]                           -- rank r's op bytes start at sum_{q<r} counts[2q+1]; ops_off[] is
 *                                      rebased to index into this concatenation.
 * Outputs are ignored on the other ranks (may be NULL). */
int coati_hip_dist_gather(coati_hip_comm_t* comm, int root, coati_hip_batch_t* batch, uint64_t* counts, float* scores,
                          uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len);

/* The whole sharded job: viterbi_mem + traceback_viterbi of n_pairs pairs over all ranks.  EVERY rank
 * passes the same input (each process reads the same file; nothing but results crosses the links);
 * rank r computes the pairs [bounds[r], bounds[r+1]) of coati_hip_shard_bounds in chunks that fit its
 * HBM, the kernel of chunk k+1 running while chunk k is gathered.  Outputs as coati_hip_viterbi_batch,
 * valid on `root` only. */
int coati_hip_dist_viterbi(coati_hip_comm_t* comm, int root, coati_hip_model_t* model, uint64_t n_pairs,
                           const uint8_t* a_cat, const uint64_t* a_off, const uint8_t* b_cat, const uint64_t* b_off,
                           float* scores, uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len);

/* The same for ranks that hold only THEIR part of the sequences (each process reads / generates its shard; the
 * LENGTHS of all pairs -- a_off / b_off, 16 bytes per pair -- are still known everywhere, they are the plan):
 * a_cat[0] is byte a_first of the concatenation that a_off indexes, b_cat[0] byte b_first; a rank's arrays must
 * cover the pairs [bounds[rank], bounds[rank+1]) of coati_hip_shard_bounds.  a_first = b_first = 0 with whole
 * arrays is coati_hip_dist_viterbi.  A rank that fails (allocation, bad codes, arrays that do not cover its
 * shard) reports it in the round's count exchange: every rank returns an error from the same round, none is
 * left waiting in a collective. */
int coati_hip_dist_viterbi_shard(coati_hip_comm_t* comm, int root, coati_hip_model_t* model, uint64_t n_pairs,
                                 const uint8_t* a_cat, uint64_t a_first, const uint64_t* a_off, const uint8_t* b_cat,
                                 uint64_t b_first, const uint64_t* b_off, float* scores, uint8_t* ops,
                                 uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len);

/* The same job with the bulk of the results LEFT ON THE RANK THAT COMPUTED THEM: rank r's scores, ops, op offsets
 * and op lengths of the pairs [bounds[r], bounds[r+1]) are downloaded over ITS OWN host link into ITS arrays (entry 0 =
 * the first pair of its shard; ops_off indexes its own ops array; ops_capacity >= the op bytes of its shard), and
 * only the summary -- score and op length of every pair, 8 bytes per pair -- is gathered over RCCL into the root's
 * all_scores[n_pairs] / all_len[n_pairs] when gather_summary != 0 (the same value on every rank; the arrays are
 * ignored elsewhere and may be NULL).  Why: a 1 kb pair leaves ~2 kB of ops, so the gather-all form sends every
 * rank's output through the root's single PCIe link (the stage budget for BASELINE configs[4]); here
 * each rank formats / writes its own slice and N links work in parallel.  Failure handling as above. */
int coati_hip_dist_viterbi_shard_local(coati_hip_comm_t* comm, int root, coati_hip_model_t* model, uint64_t n_pairs,
                                       const uint8_t* a_cat, uint64_t a_first, const uint64_t* a_off,
                                       const uint8_t* b_cat, uint64_t b_first, const uint64_t* b_off, float* scores,
                                       uint8_t* ops, uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len,
                                       int gather_summary, float* all_scores, uint32_t* all_len);

/* ---- the plan, as pure host arithmetic (no device, no communicator: usable anywhere, and what the CPU tests
 * of the multi-rank paths run on; the collectives above execute exactly these plans) ------------------------- *
 * Chunk plan of the sharded job: rank r works on pairs [cuts[cut_index[r]], ...) in chunks whose boundaries are
 * cuts[cut_index[r] .. cut_index[r+1]) (at least two entries per rank; chunks of at most chunk_cells DP cells,
 * 0 = the library's default, and at least one pair); *rounds = the number of gather rounds (the largest chunk
 * count of any rank).  cuts may be NULL to ask for the size (cut_index[world]). */
int coati_hip_dist_chunk_plan(uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, int world,
                              uint64_t chunk_cells, uint64_t* cut_index, uint64_t* cuts, uint64_t cuts_capacity,
                              uint64_t* rounds);
/* Landing zone of one gather on the root: counts[2 * world] as coati_hip_dist_gather returns them ->
 * land4[4 * world] = byte offsets of rank r's scores, ops, op offsets, op lengths (256-byte aligned, rank
 * order, nothing for the root itself), *need = bytes of HBM the zone takes. */
int coati_hip_dist_landing_plan(int world, int root, const uint64_t* counts, uint64_t* land4, uint64_t* need);

/* Debug / measurement: where the ROOT's thread of the last sharded job of this process (any of the entry points above or
 * below) spent its time, in seconds: out8[0] the whole job loop, [1] making / waiting for its own chunks, [2] count
 * exchanges, [3] the send / receive group, [4] unpack + placement + offset rebase, [5] the number of rounds, [6] reserving
 * the landing zone, [7] the rank's own copy-out (local form).  What is left of [0] is the loop's own logic (plans,
 * validation, transfer lists).  tools/dist_sim_bench.py.  ONE record per process, written by the root's thread of whichever
 * job ran last: valid only after a job has returned and while no other job of the process is in flight (two communicators
 * running jobs at once, or a read during a job, race on it -- a measurement aid, not an API to build on). */
int coati_hip_dist_debug_job_times(double* out8);

/* The per-rank job loop of coati_hip_dist_viterbi_shard[_local] is one piece of code over an ENVIRONMENT: RCCL + HIP
 * in the entry points above, host memory + a host transport in the three below (test infrastructure, exported so
 * that the tests reach it through the ABI).  Every rank "computes" its chunks from given per-pair results: pair p's
 * score, its ops (its slot of len_a+len_b bytes at the op prefix of p, ops right-aligned in the slot as the walkers
 * leave them) and their number; only the entries of the rank's own shard are read.
 *
 * coati_hip_dist_simulate[_local]: all ranks as threads of the calling process, an in-process transport (bounded
 * waits).  _simulate: outputs as coati_hip_dist_viterbi on the root.  _simulate_local: scores / ops / ops_off /
 * ops_len are global-size arrays in which rank r's own arrays are the slices that start at its shard (index
 * bounds[r], op byte = op prefix of bounds[r]); all_scores / all_len: the root's summary. */
int coati_hip_dist_simulate(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off,
                            uint64_t chunk_cells, const float* pair_scores, const uint8_t* pair_ops,
                            const uint32_t* pair_ops_len, float* scores, uint8_t* ops, uint64_t ops_capacity,
                            uint64_t* ops_off, uint32_t* ops_len);
int coati_hip_dist_simulate_local(int world, int root, uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off,
                                  uint64_t chunk_cells, int gather_summary, const float* pair_scores,
                                  const uint8_t* pair_ops, const uint32_t* pair_ops_len, float* scores, uint8_t* ops,
                                  uint64_t ops_capacity, uint64_t* ops_off, uint32_t* ops_len, float* all_scores,
                                  uint32_t* all_len);
/* ONE rank of the job in the calling process, the exchanges done by the caller's functions (each returns 0 or an
 * error code): allgather -- every rank contributes words_per_rank 64-bit words, all[] receives them in rank order;
 * send / recv -- `bytes` bytes to / from `peer`, matched in the order they are issued between two ranks.  The CPU
 * tests run two such processes over torch.distributed's gloo backend.  local / gather_summary as in
 * coati_hip_dist_viterbi_shard_local (local = 0: outputs on the root as coati_hip_dist_viterbi_shard). */
typedef struct coati_hip_dist_host_transport {
    void* ctx;
    int (*allgather)(void* ctx, const uint64_t* mine, uint64_t* all, uint32_t words_per_rank);
    int (*send)(void* ctx, int peer, const void* data, uint64_t bytes);
    int (*recv)(void* ctx, int peer, void* data, uint64_t bytes);
} coati_hip_dist_host_transport_t;
int coati_hip_dist_job_host(const coati_hip_dist_host_transport_t* transport, int world, int rank, int root,
                            uint64_t n_pairs, const uint64_t* a_off, const uint64_t* b_off, uint64_t chunk_cells,
                            int local, int gather_summary, const float* pair_scores, const uint8_t* pair_ops,
                            const uint32_t* pair_ops_len, float* scores, uint8_t* ops, uint64_t ops_capacity,
                            uint64_t* ops_off, uint32_t* ops_len, float* all_scores, uint32_t* all_len);

#ifdef __cplusplus
}
#endif
#endif /* COATI_HIP_DIST_H */
