// nodevice_stubs.cpp -- DIAGNOSTIC BUILD ONLY (make -C yaha_amd/csrc asan): the device entry points of include/yaha_hip.h answered with YGPU_ENODEV, so that
// the HOST stages (reader, .nib2 / index formats, argument handling, OQC/FBS, SAM text, pipeline) can be linked without HIP and run under
// AddressSanitizer / UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool).  Never part of the
// product: libyaha_hip.so is built from device/*.hip and has no such stubs.
#include "../../yaha_amd/csrc/host/yaha_host.h"
extern "C" {
int  ygpu_init(int, const ygpu_index_view *, const ygpu_params *, ygpu_ctx **out) { if (out) *out = nullptr; return YGPU_ENODEV; }
int  ygpu_device_count(void) { return 0; }
int  ygpu_init_multi(const int *, int n, int cpd, const ygpu_index_view *, const ygpu_params *, ygpu_ctx **out, int *rc_each) { for (int k = 0; k < n; k++) { if (out) for (int j = 0; j < cpd; j++) out[k * cpd + j] = nullptr; if (rc_each) rc_each[k] = YGPU_ENODEV; } return YGPU_ENODEV; }
int  ygpu_clone(const ygpu_ctx *, ygpu_ctx **out) { if (out) *out = nullptr; return YGPU_ENODEV; }
void ygpu_destroy(ygpu_ctx *) {}
int  ygpu_memory(ygpu_ctx *, uint64_t *, uint64_t *, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_park(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_get_arena_profile(ygpu_ctx *, ygpu_arena_profile *) { return YGPU_ENODEV; }
int  ygpu_presize(ygpu_ctx *, const ygpu_arena_profile *) { return YGPU_ENODEV; }
const char *ygpu_last_error(const ygpu_ctx *) { return "sanitizer build of the host stages: no device code linked"; }
int  ygpu_upload(ygpu_ctx *, const ygpu_read_batch *) { return YGPU_ENODEV; }
int  ygpu_upload_nowait(ygpu_ctx *, const ygpu_read_batch *) { return YGPU_ENODEV; }
int  ygpu_run(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_collect(ygpu_ctx *, ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_result_size(ygpu_ctx *, uint64_t *, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_collect_into(ygpu_ctx *, uint32_t *, ygpu_clump *, uint32_t *, ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_set_postfilter(ygpu_ctx *, const ygpu_postfilter_params *) { return YGPU_ENODEV; }
int  ygpu_postfilter(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_postfilter_snapshot(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_selftest_primitives(ygpu_ctx *, uint32_t, uint32_t, int) { return YGPU_ENODEV; }
int  ygpu_trace_volume(ygpu_ctx *, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_inject_results(ygpu_ctx *, const ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_postfilter_drop(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_filtered_size(ygpu_ctx *, uint64_t *, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_collect_filtered(ygpu_ctx *, uint32_t *, ygpu_out_clump *, uint32_t *, ygpu_filtered_batch *) { return YGPU_ENODEV; }
void *ygpu_host_alloc(size_t) { return nullptr; }
void ygpu_host_free(void *) {}
int  ygpu_submit(ygpu_ctx *, const ygpu_read_batch *, ygpu_ticket *) { return YGPU_ENODEV; }
int  ygpu_poll(ygpu_ctx *, ygpu_ticket) { return YGPU_ENODEV; }
int  ygpu_wait(ygpu_ctx *, ygpu_ticket, ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_last_timing(ygpu_ctx *, float *, int *, const char *const **, const float **) { return YGPU_ENODEV; }
int  ygpu_seed_join(ygpu_ctx *, const ygpu_fragment **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_chain(ygpu_ctx *, const ygpu_fragment **, const uint32_t **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_dp_batch(ygpu_ctx *, const ygpu_dp_problem *, uint32_t, const ygpu_dp_result **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_dp_batch_ex(ygpu_ctx *, const ygpu_dp_problem *, uint32_t, int, const ygpu_dp_result **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
}
namespace yaha {
int  visibleDevices() { return 0; }
bool buildIndexDevice(int, const Genome &, int, int, int, IndexImage &, FILE *, std::string &err) { err = "sanitizer build: no device code"; return false; }
}
