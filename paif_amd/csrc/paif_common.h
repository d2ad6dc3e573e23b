// Shared helpers for the gfx950 kernels of libpaif_hip.so (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "paif_hip.h"

namespace paif {

void set_error(const char* fmt, ...);

#define PAIF_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      paif::set_error(__VA_ARGS__);     \
      return (code);                    \
    }                                   \
  } while (0)

// Launch check: hipGetLastError after a launch (does not synchronise).
#define PAIF_LAUNCH_CHECK(name)                                            \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      paif::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                     \
    }                                                                      \
  } while (0)

static inline hipStream_t as_stream(paif_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (round-robin
// dispatch, MI355X_MICROARCH "Workgroup dispatch"), so give each XCD a CONTIGUOUS chunk of tiles:
// spatially adjacent tiles (which share halo pixels) then hit the same 4 MiB L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ float prelu_f(float v, float a) { return v >= 0.f ? v : a * v; }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

}  // namespace paif
