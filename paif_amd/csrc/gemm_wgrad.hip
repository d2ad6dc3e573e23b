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

// Operands go through a wave-private LDS region in their natural [token][channel] layout: the MFMA takes ONE dword per lane per
// operand (lane (kk, c) = channel c of token 2j + kk), and fetching those dwords straight from global memory made every MFMA wait
// on a 256-B wave load (the kernel ran at the CU's L1 rate, ~23 TFLOP/s).  Staged with coalesced float4 loads (8 tokens x 128 B
// per wave instruction), the same dword is a conflict-free ds_read_b32.
constexpr int GW_CH = 32;     // tokens per staged chunk

__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         float* __restrict__ part_w, float* __restrict__ part_b, int M, int N, int K,
                                                         int mper) {
  __shared__ __align__(16) float lds[4][2][GW_CH * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 128 + wave * 32;
  const int mbeg = blockIdx.z * mper, mend = min(M, mbeg + mper);
  float* sD = lds[wave][0];
  float* sX = lds[wave][1];
  const int tl = lane >> 3, q = lane & 7;                      // staging role: token lane, channel quad
  // clamped columns (ragged N / K): their products land in accumulator rows / columns that are never stored
  const int nq = min(n0 + 4 * q, max(N - 4, 0)), kq = min(k0 + 4 * q, max(K - 4, 0));
  const bool nvec = (N % 4 == 0 && lddy % 4 == 0 && n0 + 4 * q + 4 <= N), kvec = (K % 4 == 0 && ldx % 4 == 0 && k0 + 4 * q + 4 <= K);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int m0 = mbeg; m0 < mend; m0 += GW_CH) {
#pragma unroll
    for (int it = 0; it < GW_CH / 8; ++it) {
      const int m = m0 + it * 8 + tl;
      const int mm = min(m, M - 1);
      const bool in = m < mend;
      float4 d, v;
      if (nvec) d = *reinterpret_cast<const float4*>(dy + (size_t)mm * lddy + n0 + 4 * q);
      else {
        const float* r_ = dy + (size_t)mm * lddy;
        d = make_float4(r_[min(n0 + 4 * q, N - 1)], r_[min(n0 + 4 * q + 1, N - 1)], r_[min(n0 + 4 * q + 2, N - 1)], r_[min(n0 + 4 * q + 3, N - 1)]);
      }
      if (kvec) v = *reinterpret_cast<const float4*>(x + (size_t)mm * ldx + k0 + 4 * q);
      else {
        const float* r_ = x + (size_t)mm * ldx;
        v = make_float4(r_[min(k0 + 4 * q, K - 1)], r_[min(k0 + 4 * q + 1, K - 1)], r_[min(k0 + 4 * q + 2, K - 1)], r_[min(k0 + 4 * q + 3, K - 1)]);
      }
      if (!in) d = make_float4(0.f, 0.f, 0.f, 0.f);             // tokens past the slice contribute nothing
      *reinterpret_cast<float4*>(sD + (it * 8 + tl) * 32 + 4 * q) = d;
      *reinterpret_cast<float4*>(sX + (it * 8 + tl) * 32 + 4 * q) = v;
    }
    (void)nq; (void)kq;
    __builtin_amdgcn_wave_barrier();          // wave-private region, LDS operations of one wave complete in order:
    asm volatile("" ::: "memory");            // only the compiler must not reorder across the phases
#pragma unroll 4
    for (int j = 0; j < GW_CH / 2; ++j) {
      const float a = sD[(2 * j + kk) * 32 + c];
      const float b = sX[(2 * j + kk) * 32 + c];
      bsum += a;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
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
