// placeholder replaced below
#include "../../include/jyutvoice_hip.h"
#include "jv_model.h"
namespace jv {
int hift_ws_create(Context&) { return JV_OK; }
void hift_ws_destroy(Context&) {}
}
extern "C" {
int jv_hift_f0(jv_context*, const float*, const int32_t*, int, int, float*, void*) { return jv::fail(JV_ERR_STATE, "hift not built yet"); }
int jv_hift_source(jv_context*, const float*, const float*, const float*, int, int, float*, void*) { return jv::fail(JV_ERR_STATE, "hift not built yet"); }
int jv_hift_decode(jv_context*, const float*, const float*, const int32_t*, int, int, float*, void*) { return jv::fail(JV_ERR_STATE, "hift not built yet"); }
}
