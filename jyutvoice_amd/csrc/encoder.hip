// placeholder replaced below
#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
namespace jv {
int enc_ws_create(Context&) { return JV_OK; }
void enc_ws_destroy(Context&) {}
}
extern "C" {
int jv_encoder_fwd(jv_context*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, const float*, int, int, float*, float*, float*, float*, void*) { return jv::fail(JV_ERR_STATE, "encoder not built yet"); }
int jv_length_regulate(jv_context*, const float*, const int64_t*, const float*, int, int, float, float*, int64_t*, int, float*, float*, void*) { return jv::fail(JV_ERR_STATE, "encoder not built yet"); }
}
