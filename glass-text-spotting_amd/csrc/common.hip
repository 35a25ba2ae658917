#include "common.h"

static thread_local char g_err[512] = "";

void glass_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* glass_last_error(void) { return g_err; }
extern "C" int glass_abi_version(void) { return 8; }      // 8: + glass_pointwise_split_dual_supported, glass_conv1x1_pointwise_split_dual_nhwc; 7: + glass_winograd43_splitk_*, glass_conv3x3_winograd43_splitk_nhwc; 6: the persistent recurrent entries take `call_status`, + glass_recurrence_test_hook; 5: + glass_pointwise_split_*, glass_conv1x1_pointwise_split_nhwc; 4: + glass_bilstm_recurrence_persistent, glass_recurrence_status, glass_linear_splitk*, glass_conv3x3_winograd_body_nhwc; 2: + glass_backbone_stem_*, glass_conv3x3_winograd43_body_nhwc, glass_roi_align_rotated_up2; 3: + glass_text_argmax, glass_postprocess_words takes its outputs instead of the probabilities
extern "C" int glass_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
