// hip_stub.cc -- libcoati_hip.so stand-in for the sanitizer build of the host layer (make asan): every
// entry point of include/coati_hip.h answers COATI_HIP_ENODEVICE, so the AddressSanitizer / UBSan runs of
// the parsers, writers and drivers stay CPU-only (GPU ASan is not available on the pool) and need no ROCm
// runtime at all.  Never shipped, never linked by the product targets.
#include "coati_hip.h"

namespace {
constexpr int kNo = COATI_HIP_ENODEVICE;
}
extern "C" {
uint32_t coati_hip_version(void) { return 1u; }
int coati_hip_device_count(void) { return 0; }
const char* coati_hip_last_error(void) { return "stub libcoati_hip: no device in the sanitizer build"; }
int coati_hip_model_create(const float*, float, float, float, float, int, int, coati_hip_model_t** out) { if(out) *out = nullptr; return kNo; }
int coati_hip_model_create_tables(const float*, uint32_t, float, float, float, float, int, int, coati_hip_model_t** out) { if(out) *out = nullptr; return kNo; }
void coati_hip_model_destroy(coati_hip_model_t*) {}
int coati_hip_model_trim(coati_hip_model_t*) { return kNo; }
int coati_hip_model_set_option(coati_hip_model_t*, int, int64_t) { return kNo; }
int coati_hip_batch_create(coati_hip_model_t*, uint64_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, coati_hip_batch_t** out) { if(out) *out = nullptr; return kNo; }
int coati_hip_batch_create_tables(coati_hip_model_t*, uint64_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint32_t*, coati_hip_batch_t** out) { if(out) *out = nullptr; return kNo; }
void coati_hip_batch_destroy(coati_hip_batch_t*) {}
uint64_t coati_hip_batch_pairs(const coati_hip_batch_t*) { return 0; }
uint64_t coati_hip_batch_device_bytes(const coati_hip_batch_t*) { return 0; }
uint64_t coati_hip_batch_cells(const coati_hip_batch_t*) { return 0; }
int coati_hip_viterbi_launch(coati_hip_batch_t*) { return kNo; }
int coati_hip_batch_sync(coati_hip_batch_t*) { return kNo; }
int coati_hip_viterbi_wait(coati_hip_batch_t*) { return kNo; }
int coati_hip_viterbi_fetch(coati_hip_batch_t*, float*, uint8_t*, uint64_t, uint64_t*, uint32_t*) { return kNo; }
int coati_hip_viterbi_last_timing(coati_hip_batch_t*, float*, float*) { return kNo; }
int coati_hip_viterbi_band_stats(coati_hip_batch_t*, uint32_t*, uint64_t*) { return kNo; }
int coati_hip_model_prepare(coati_hip_model_t*, uint64_t, uint64_t, uint64_t) { return kNo; }
void coati_hip_debug_reload_env(void) {}
int coati_hip_viterbi_timing(coati_hip_batch_t*, uint32_t, float*, float*) { return kNo; }
int coati_hip_batch_result_ptrs(coati_hip_batch_t*, void**, void**, uint64_t*, void**, void**) { return kNo; }
int coati_hip_forward_launch(coati_hip_batch_t*) { return kNo; }
int coati_hip_forward_final(coati_hip_batch_t*, float*) { return kNo; }
int coati_hip_debug_forward_matrices(coati_hip_batch_t*, uint64_t, float*, float*, float*, uint64_t) { return kNo; }
int coati_hip_sampleback(coati_hip_batch_t*, uint32_t, const uint64_t*, int, float*, uint8_t*, uint64_t, uint64_t*, uint32_t*, uint64_t*) { return kNo; }
int coati_hip_sampleback_prepare(coati_hip_batch_t*, uint32_t, int) { return kNo; }
int coati_hip_debug_libm(coati_hip_model_t*, int, const float*, uint64_t, float*) { return kNo; }
int coati_hip_debug_rng_f24(coati_hip_model_t*, const uint64_t*, uint32_t, float*) { return kNo; }
int coati_hip_viterbi_batch(coati_hip_model_t*, uint64_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, float*, uint8_t*, uint64_t, uint64_t*, uint32_t*) { return kNo; }
int coati_hip_shard_bounds(uint64_t, const uint64_t*, const uint64_t*, int, uint64_t*) { return kNo; }
int coati_hip_host_alloc(uint64_t, void** out) { if(out) *out = nullptr; return kNo; }
void coati_hip_host_free(void*) {}
int coati_hip_debug_viterbi_flags(coati_hip_batch_t*, uint64_t, uint8_t*, uint64_t) { return kNo; }
}
