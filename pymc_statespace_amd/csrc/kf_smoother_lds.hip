#include "kf_internal.h"
int kf_launch_smoother_lds(kf_handle* h, const KfSmoothArgs& a, hipStream_t stream) {
  (void)a; (void)stream;
  KF_SET_ERR(h, "kf_smoother: not built yet");
  return KF_ERR_UNSUPPORTED;
}
