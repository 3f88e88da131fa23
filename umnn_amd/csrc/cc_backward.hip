// placeholder until the backward kernels land (next commit)
#include "cc_host.h"

extern "C" long long umnn_cc_backward_workspace_bytes(const umnn_mlp* net, long long B, int d, int E) {
    (void)net; (void)B; (void)d; (void)E;
    return 0;
}

extern "C" int umnn_cc_backward(const umnn_mlp* net, const float* x0, const float* x, const float* h, const float* g, const float* g_fx,
                                const float* cc_w, const float* cc_s, int nb_steps, long long B, int d, int E,
                                float* dx0, float* dx, float* dh, float* dtheta,
                                void* workspace, long long workspace_bytes, void* stream) {
    return umnn_fail(UMNN_EUNSUPPORTED, "backward kernel not built yet");
}
