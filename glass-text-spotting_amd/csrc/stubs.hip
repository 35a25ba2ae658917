// Temporary: entry points not implemented yet return an error (never a silent fallback).
#include "common.h"
#define STUB(name) glass_set_error(name ": not implemented yet"); return GLASS_EINVAL
extern "C" int64_t glass_rpn_topk_workspace(int, int) { return 0; }
extern "C" int glass_rpn_level_topk_decode(const float*, int, const float*, int, int, int, int, int, int, float, const float*, const float*, int, int, int, int, float*, float*, int*, void*, int64_t, glass_stream_t) { STUB("glass_rpn_level_topk_decode"); }
extern "C" int64_t glass_nms_workspace(int, int) { return 0; }
extern "C" int glass_rotated_nms_select(const float*, const float*, const int*, const int*, int, int, const int*, float, float, int, int, int, float*, float*, int*, int*, void*, int64_t, glass_stream_t) { STUB("glass_rotated_nms_select"); }
extern "C" int glass_box_decode(const float*, const float*, const float*, const float*, int, const float*, float*, float*, float*, glass_stream_t) { STUB("glass_box_decode"); }
extern "C" int glass_gc_attention_inplace(float*, int, int, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, glass_stream_t) { STUB("glass_gc_attention_inplace"); }
extern "C" int glass_bilstm_recurrence(const float*, const float*, float*, int, int, int, glass_stream_t) { STUB("glass_bilstm_recurrence"); }
extern "C" int glass_attention_decode(const float*, const float*, const glass_decoder_weights*, const int*, int, int, int, int, int, int, float*, glass_stream_t) { STUB("glass_attention_decode"); }
