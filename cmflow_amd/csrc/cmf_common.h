// Shared helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMF_WAVE 64

#define CMF_CHECK_ARG(cond) \
    do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)

static inline int cmf_launch_status() { return (int)hipGetLastError(); }

static inline int cmf_divup(long long a, long long b) { return (int)((a + b - 1) / b); }
