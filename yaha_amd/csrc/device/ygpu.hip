// placeholder -- replaced by the real device implementation
#include "../../../include/yaha_hip.h"
extern "C" {
int  ygpu_init(int, const ygpu_index_view *, const ygpu_params *, ygpu_ctx **out) { *out = nullptr; return YGPU_ENODEV; }
void ygpu_destroy(ygpu_ctx *) {}
const char *ygpu_last_error(const ygpu_ctx *) { return "not implemented"; }
int  ygpu_upload(ygpu_ctx *, const ygpu_read_batch *) { return YGPU_ENODEV; }
int  ygpu_run(ygpu_ctx *) { return YGPU_ENODEV; }
int  ygpu_collect(ygpu_ctx *, ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_last_timing(ygpu_ctx *, float *, int *, const char *const **, const float **) { return YGPU_ENODEV; }
int  ygpu_seed_join(ygpu_ctx *, const ygpu_fragment **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_chain(ygpu_ctx *, const ygpu_fragment **, const uint32_t **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_dp_batch(ygpu_ctx *, const ygpu_dp_problem *, uint32_t, const ygpu_dp_result **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
}
