// LayerNorm affine gradients (training step, DESIGN.md plan item 3):  dgamma[c] = sum_rows dy[r][c] * xhat[r][c],
// dbeta[c] = sum_rows dy[r][c],  xhat = (x - mean_r) * rstd_r   (nn.LayerNorm of the MiT blocks, core/mix_transformer.py).
// A workgroup takes a chunk of rows: its 4 waves first form mean / rstd of those rows (two-pass, like layernorm_kernel) into
// LDS, then thread c walks the chunk column-wise (coalesced across threads).  Per-workgroup partials [blk][2][C] are summed
// in block order by a second pass (deterministic).
#include "paif_common.h"

namespace {

constexpr int RB = 64;   // rows per workgroup

__global__ __launch_bounds__(256) void layernorm_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int M, int C, float eps) {
  __shared__ float s_mean[RB], s_rstd[RB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * RB;
  const int nrows = min(RB, M - r0);
  for (int r = wave; r < nrows; r += 4) {
    const float* row = x + (size_t)(r0 + r) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = row[c] - mean; ss = fmaf(d, d, ss); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
    if (lane == 0) { s_mean[r] = mean; s_rstd[r] = 1.0f / sqrtf(ss / (float)C + eps); }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float dg = 0.f, db = 0.f;
    for (int r = 0; r < nrows; ++r) {
      const size_t o = (size_t)(r0 + r) * C + c;
      const float d = dy[o];
      dg = fmaf(d, (x[o] - s_mean[r]) * s_rstd[r], dg);
      db += d;
    }
    partial[((size_t)blockIdx.x * 2) * C + c] = dg;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = db;
  }
}

// one wave per (which, c): its 64 lanes stride over the block partials, then a fixed xor-shuffle tree (deterministic)
__global__ __launch_bounds__(256) void layernorm_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int nblk, int C, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= 2 * C) return;      // wave-uniform
  const int which = i / C, c = i - which * C;
  float v = 0.f;
  for (int b = lane; b < nblk; b += 64) v += partial[((size_t)b * 2 + which) * C + c];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if (lane == 0) {
    float* dst = which ? dbeta : dgamma;
    dst[c] = accumulate ? dst[c] + v : v;
  }
}

}  // namespace

extern "C" int paif_layernorm_wgrad_blocks(int M) { return (M + RB - 1) / RB; }

extern "C" int paif_layernorm_wgrad(const float* x, const float* dy, float* dgamma, float* dbeta, float* workspace, int M, int C,
                                    float eps, int accumulate, paif_stream_t stream) {
  PAIF_REQUIRE(x && dy && dgamma && dbeta && workspace && M > 0 && C > 0, PAIF_EINVAL, "layernorm_wgrad: bad arguments");
  const int nblk = (M + RB - 1) / RB;
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(layernorm_wgrad_kernel, dim3(nblk), dim3(256), 0, st, x, dy, workspace, M, C, eps);
  PAIF_LAUNCH_CHECK("layernorm_wgrad");
  hipLaunchKernelGGL(layernorm_wgrad_reduce_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, st, workspace, dgamma, dbeta, nblk, C, accumulate);
  PAIF_LAUNCH_CHECK("layernorm_wgrad_reduce");
  return 0;
}
