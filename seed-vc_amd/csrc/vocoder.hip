// placeholder until the vocoder engines land (same C ABI)
#include "model_util.h"
using namespace svc;
extern "C" {
int svc_bigvgan_create(const svc_bigvgan_config_t*, const svc_tensor_desc_t*, int, void*, svc_bigvgan_t**) { set_error("bigvgan: not built yet"); return 1; }
void svc_bigvgan_destroy(svc_bigvgan_t*) {}
int svc_bigvgan_forward(svc_bigvgan_t*, const float*, int, int, float*, void*) { set_error("bigvgan: not built yet"); return 1; }
int svc_hift_create(const svc_hift_config_t*, const svc_tensor_desc_t*, int, void*, svc_hift_t**) { set_error("hift: not built yet"); return 1; }
void svc_hift_destroy(svc_hift_t*) {}
int svc_hift_forward(svc_hift_t*, const float*, const float*, const float*, const float*, int, int, float*, float*, void*) { set_error("hift: not built yet"); return 1; }
int svc_op_conv1d(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, int, void*) { set_error("conv1d: not built yet"); return 1; }
int svc_op_conv_transpose1d(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, void*) { set_error("convT: not built yet"); return 1; }
}
