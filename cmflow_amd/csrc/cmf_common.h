// Shared helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMF_WAVE 64

#define CMF_CHECK_ARG(cond) \
    do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)

static inline int cmf_launch_status() { return (int)hipGetLastError(); }

static inline int cmf_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// Library-owned scratch for the few entry points that need more working memory than their reference signature
// carries (cmf_ball_query: spilled hit lists at nsample > 32; cmf_group_points_grad: the inverse index).  One buffer per
// (device, stream, slot), grown on demand and kept for the life of the process: work on one stream is ordered, so the
// next call on that stream may reuse it, and calls on different streams get different buffers.  [The first version
// used hipMallocAsync / hipFreeAsync per call: normally a pool hit, but measured at 4.6 ms for one 64 MB request after
// the pool had been trimmed -- 25x the kernel it served.]  Returns nullptr on allocation failure.  group_points.hip.
void *cmf_stream_scratch(hipStream_t stream, int slot, size_t bytes);
