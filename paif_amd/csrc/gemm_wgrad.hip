// Linear-layer weight gradient (training step, DESIGN.md "Plan for the training step" item 3):
//   dW[n][k] = sum_m dY[m][n] * X[m][k]          db[n] = sum_m dY[m][n]
// for y = x W^T + b of the MiT blocks / head MLPs (core/mix_transformer.py:22-25,66-69; core/segformer_head.py:19).
// dY is the gradient at the GEMM output (the activation derivative, where the forward fused one, is the caller's).
//
// Same operand trick as conv_wgrad.hip: with K = 2 consecutive tokens, v_mfma_f32_32x32x2_f32 takes
//   A[i = n][kk] = dY[token kk][n0 + i],   B[kk][j = k] = X[token kk][k0 + j]
// straight from the row-major activations (a wave-load is 2 tokens x 128 B).  A workgroup owns a 32 (n) x 128 (k) tile of
// dW (4 waves x 32 columns of k, all reading the same dY strip) and a slice of the tokens (blockIdx.z); the slices'
// partial tiles are summed in slice order by a second pass (deterministic).  Exact fp32.
#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         float* __restrict__ part_w, float* __restrict__ part_b, int M, int N, int K,
                                                         int mper) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 128 + wave * 32;
  const int mbeg = blockIdx.z * mper, mend = min(M, mbeg + mper);
  const int n = min(n0 + c, N - 1), kc = min(k0 + c, K - 1);      // clamped columns: their results are never stored
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int m = mbeg; m < mend; m += 2) {
    const int mm = min(m + kk, M - 1);
    const bool ok = m + kk < mend;
    float a = dy[(size_t)mm * lddy + n];
    float b = x[(size_t)mm * ldx + kc];
    a = ok ? a : 0.f;
    bsum += a;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  float* pw = part_w + (size_t)blockIdx.z * N * K;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * kk;     // C/D layout: row = n, column = lane & 31 = k
    if (nn < N && k0 + c < K) pw[(size_t)nn * K + k0 + c] = acc[r];
  }
  if (part_b && blockIdx.y == 0 && wave == 0) {
    bsum += __shfl_xor(bsum, 32);                            // the two token halves of the pair
    if (kk == 0 && n0 + c < N) part_b[(size_t)blockIdx.z * N + n0 + c] = bsum;
  }
}

__global__ __launch_bounds__(256) void gemm_wgrad_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                                float* __restrict__ dw, float* __restrict__ db, int splits, int N,
                                                                int K, int accumulate) {
  const size_t total = (size_t)N * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total + (db ? N : 0); i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    if (i < total) {
      for (int s = 0; s < splits; ++s) v += part_w[(size_t)s * total + i];
      dw[i] = accumulate ? dw[i] + v : v;
    } else {
      const size_t n = i - total;
      for (int s = 0; s < splits; ++s) v += part_b[(size_t)s * N + n];
      db[n] = accumulate ? db[n] + v : v;
    }
  }
}

}  // namespace

// token slices: enough workgroups to fill the chip, at least 256 tokens per slice, at most 64 slices
extern "C" int paif_gemm_wgrad_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const long tiles = (long)((N + 31) / 32) * ((K + 127) / 128);
  long s = (2048 + tiles - 1) / tiles;
  const long maxs = (M + 255) / 256;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

extern "C" int paif_gemm_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                               float* workspace, int accumulate, paif_stream_t stream) {
  PAIF_REQUIRE(dy && x && dw && workspace, PAIF_EINVAL, "gemm_wgrad: null pointer");
  PAIF_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K, PAIF_EINVAL, "gemm_wgrad: shape %dx%dx%d ld %d/%d", M, N, K, lddy, ldx);
  PAIF_REQUIRE(splits >= 1 && splits <= 64, PAIF_EINVAL, "gemm_wgrad: splits=%d", splits);
  int mper = (M + splits - 1) / splits;
  mper += mper & 1;                                           // token pairs never straddle a slice
  float* part_w = workspace;
  float* part_b = db ? workspace + (size_t)splits * N * K : nullptr;
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(gemm_wgrad_kernel, dim3((N + 31) / 32, (K + 127) / 128, splits), dim3(256), 0, st, dy, lddy, x, ldx, part_w, part_b, M,
                     N, K, mper);
  PAIF_LAUNCH_CHECK("gemm_wgrad");
  const size_t total = (size_t)N * K + (db ? N : 0);
  const int rb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gemm_wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, part_w, part_b, dw, db, splits, N, K, accumulate);
  PAIF_LAUNCH_CHECK("gemm_wgrad_reduce");
  return 0;
}
