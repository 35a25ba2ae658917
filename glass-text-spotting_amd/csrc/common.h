// Shared helpers for libglass_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "glass_hip.h"

void glass_set_error(const char* fmt, ...);

#define GLASS_CHECK_ARG(cond, ...)                \
  do {                                            \
    if (!(cond)) {                                \
      glass_set_error(__VA_ARGS__);               \
      return GLASS_EINVAL;                        \
    }                                             \
  } while (0)

#define GLASS_CHECK_LAUNCH(what)                                                   \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      glass_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return GLASS_EHIP;                                                           \
    }                                                                              \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
