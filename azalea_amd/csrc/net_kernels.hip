// placeholder -- replaced by the MFMA implementation
#include "net.h"
#include <string>
static std::string g_net_err;
struct AzxNet { int dummy; };
int azx_net_create(AzxNet **, int, int, int, int, hipStream_t) { g_net_err = "resnet evaluator not built yet"; return AZX_ESTATE; }
void azx_net_destroy(AzxNet *) {}
const char *azx_net_error() { return g_net_err.c_str(); }
int azx_net_set_weights(AzxNet *, int, const char *const *, const void *const *, const int64_t *, int) { return AZX_ESTATE; }
bool azx_net_ready(const AzxNet *) { return false; }
void azx_net_eval(AzxNet *, const DevEngine &, hipStream_t) {}
int azx_net_forward_host(AzxNet *, int, int, const int32_t *, const int32_t *, float *, float *, hipStream_t) { return AZX_ESTATE; }
