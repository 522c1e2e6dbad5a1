// placeholder, replaced below
#include "common.h"
using namespace sg;
extern "C" {
size_t sg_spconv_hash_workspace_bytes(int) { return 0; }
int sg_spconv_subm_rulebook(const int32_t *, int, const int32_t *, int32_t *, void *, size_t, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
int sg_spconv_down_build(const int32_t *, int, const int32_t *, int32_t *, int32_t *, void *, size_t, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
int sg_spconv_down_fill(const int32_t *, int, const int32_t *, int, int32_t *, int32_t *, void *, size_t, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
int sg_spconv_inverse_rulebook(const int32_t *, const int32_t *, int, int32_t *, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
size_t sg_spconv_plan_workspace_bytes(int) { return 0; }
int sg_spconv_plan(const int32_t *, int, int, int32_t *, uint32_t *, void *, size_t, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
int sg_spconv_weight_to_kio(const float *, int, int, int, float *, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
int sg_spconv_gather_conv_f32(const float *, int, const int32_t *, int, int, int, int, const float *, const float *, const float *, const float *, const int32_t *, const uint32_t *, float *, sg_stream_t) { return SG_ERR_UNSUPPORTED; }
}
