// Input pipeline, device side: the decoded uint8 images are uploaded as bytes (4x less
// PCIe traffic than float32) and converted here to the reference loader's tensors (TaskFusion_dataset2.py:57-70,85-98):
// vis uint8 HWC -> float32 CHW / 255, ir uint8 HW -> float32 [1,H,W] / 255, label uint8 -> int64.  The division is the same
// IEEE fp32 division numpy performs, so the result is bit-identical to the host path.
#include "paif_common.h"

namespace {

// src [B][HW][C] bytes -> dst [B][C][HW] floats = src / 255
__global__ void u8_to_planes_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int B, size_t HW, int C) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t b = t / C;
    dst[i] = __fdiv_rn((float)src[(b * HW + px) * C + c], 255.0f);
  }
}

__global__ void u8_to_i64_kernel(const unsigned char* __restrict__ src, long long* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (long long)src[i];
}

inline int grid_for(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int paif_u8_to_planes_fwd(const unsigned char* src, float* dst, int B, int HW, int C, paif_stream_t stream) {
  PAIF_REQUIRE(src && dst && B > 0 && HW > 0 && C > 0, PAIF_EINVAL, "u8_to_planes: bad arguments");
  hipLaunchKernelGGL(u8_to_planes_kernel, dim3(grid_for((size_t)B * HW * C)), dim3(256), 0, paif::as_stream(stream), src, dst, B,
                     (size_t)HW, C);
  PAIF_LAUNCH_CHECK("u8_to_planes");
  return 0;
}

extern "C" int paif_u8_to_i64_fwd(const unsigned char* src, long long* dst, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(src && dst && n > 0, PAIF_EINVAL, "u8_to_i64: bad arguments");
  hipLaunchKernelGGL(u8_to_i64_kernel, dim3(grid_for(n)), dim3(256), 0, paif::as_stream(stream), src, dst, n);
  PAIF_LAUNCH_CHECK("u8_to_i64");
  return 0;
}
