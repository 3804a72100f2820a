#include "kf_internal.h"
int kf_launch_backward_lds(kf_handle* h, const KfBwdArgs& a, hipStream_t stream) {
  (void)a; (void)stream;
  KF_SET_ERR(h, "kf_filter_backward: not built yet");
  return KF_ERR_UNSUPPORTED;
}
