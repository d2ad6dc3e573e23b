// forward_object's extra step on the fused plane (core/model_fusion_auto.py:743-751, same at :1074-1082):
//   f = where(f > 1, 1, f); f = where(f < 0, 0, f); f = (f - min(f)) / (max(f) - min(f))      (batch-global min / max)
// and its backward with torch's semantics: the two torch.where pass gradient only where 0 <= f <= 1, torch.min / torch.max
// (full reductions) spread their gradient EVENLY over all elements equal to the extremum, and an element that owes that
// value to the clamp passes nothing on.  Two-pass reductions in a fixed order (no float atomics): deterministic.
#include "paif_common.h"

namespace {

constexpr int OG_PIX = 2048;  // elements per block of a reduction pass

__device__ __forceinline__ float clamp01(float x) { return x > 1.0f ? 1.0f : (x < 0.0f ? 0.0f : x); }   // NaN passes, like torch.where

__global__ __launch_bounds__(256) void plane_minmax_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int nblk,
                                                                   size_t n) {
  const size_t start = (size_t)blockIdx.x * OG_PIX;
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = start + threadIdx.x; i < start + OG_PIX && i < n; i += 256) {
    const float v = clamp01(x[i]);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, m));
    mx = fmaxf(mx, __shfl_xor(mx, m));
  }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    partial[nblk + blockIdx.x] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

__device__ __forceinline__ void reduce_minmax(const float* __restrict__ partial, int npartial, float& mn, float& mx) {
  mn = INFINITY; mx = -INFINITY;
  for (int i = threadIdx.x; i < npartial; i += 256) {
    mn = fminf(mn, partial[i]);
    mx = fmaxf(mx, partial[npartial + i]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, m));
    mx = fmaxf(mx, __shfl_xor(mx, m));
  }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
}

__global__ __launch_bounds__(256) void plane_normalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, int npartial,
                                                              float* __restrict__ out, float* __restrict__ minmax_out, size_t n) {
  float mn, mx;
  reduce_minmax(partial, npartial, mn, mx);    // every block, same order: min / max are order-independent anyway
  if (minmax_out && blockIdx.x == 0 && threadIdx.x == 0) { minmax_out[0] = mn; minmax_out[1] = mx; }
  const float range = mx - mn;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = __fdiv_rn(__fsub_rn(clamp01(x[i]), mn), range);
}

// partial[blk] = (sum d*(c - mx), sum d*(c - mn), #(c == mn), #(c == mx))   with c = clamp01(x)
__global__ __launch_bounds__(256) void plane_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                               const float* __restrict__ minmax, float* __restrict__ partial, size_t n) {
  const float mn = minmax[0], mx = minmax[1];
  const size_t start = (size_t)blockIdx.x * OG_PIX;
  float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;
  for (size_t i = start + threadIdx.x; i < start + OG_PIX && i < n; i += 256) {
    const float c = clamp01(x[i]), d = dout[i];
    a0 = fmaf(d, c - mx, a0);
    a1 = fmaf(d, c - mn, a1);
    c0 += (c == mn) ? 1.f : 0.f;
    c1 += (c == mx) ? 1.f : 0.f;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); c0 += __shfl_xor(c0, m); c1 += __shfl_xor(c1, m);
  }
  __shared__ float4 sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = make_float4(a0, a1, c0, c1);
  __syncthreads();
  if (threadIdx.x == 0)
    *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 4) =
        make_float4((sm[0].x + sm[1].x) + (sm[2].x + sm[3].x), (sm[0].y + sm[1].y) + (sm[2].y + sm[3].y),
                    (sm[0].z + sm[1].z) + (sm[2].z + sm[3].z), (sm[0].w + sm[1].w) + (sm[2].w + sm[3].w));
}

__global__ __launch_bounds__(256) void plane_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                              const float* __restrict__ minmax, const float* __restrict__ partial,
                                                              int npartial, float* __restrict__ dx, size_t n) {
  __shared__ float4 tot;
  if (threadIdx.x < 64) {
    float sx = 0.f, sy = 0.f;
    double cz = 0.0, cw = 0.0;     // tie counts: per-block values are exact small integers; their total is exact in double (fp32 stops at 2^24 elements)
    for (int i = threadIdx.x; i < npartial; i += 64) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)i * 4);
      sx += v.x; sy += v.y; cz += (double)v.z; cw += (double)v.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      sx += __shfl_xor(sx, m); sy += __shfl_xor(sy, m); cz += __shfl_xor(cz, m); cw += __shfl_xor(cw, m);
    }
    if (threadIdx.x == 0) tot = make_float4(sx, sy, (float)cz, (float)cw);
  }
  __syncthreads();
  const float mn = minmax[0], mx = minmax[1];
  const float inv = 1.0f / (mx - mn);
  const float d_mn_share = (tot.x * inv * inv) / tot.z;     // d loss / d min, per element equal to the minimum
  const float d_mx_share = (-tot.y * inv * inv) / tot.w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i], c = clamp01(v);
    float d = dout[i] * inv;
    if (c == mn) d += d_mn_share;
    if (c == mx) d += d_mx_share;
    dx[i] = (v >= 0.f && v <= 1.f) ? d : 0.f;               // the two torch.where clamps
  }
}

// Stand-alone eca_layer.forward (operations_m.py:353-367): per-image channel sums of an NHWC map as `chunks` partial sums per image
// ([B][chunks][32], the layout paif_eca_finish_fwd reduces: the fused ECABasicBlock gets them from its conv's epilogue).
// One block per (image, chunk); lane (tid & 7) owns a channel quad, fixed-order tree -> deterministic.
__global__ __launch_bounds__(256) void channel_sum_chunks_kernel(const float* __restrict__ x, float* __restrict__ partial, int chunks,
                                                                 size_t HW) {
  const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
  const size_t per = (HW + chunks - 1) / chunks;
  const size_t p0 = (size_t)ch * per, p1 = p0 + per < HW ? p0 + per : HW;
  const int q = threadIdx.x & 7;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t px = p0 + (threadIdx.x >> 3); px < p1; px += 32) {
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + px) * 32 + q * 4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  __shared__ float4 sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st >= 8; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const float4 o = sm[threadIdx.x + st];
      float4& m = sm[threadIdx.x];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
    }
    __syncthreads();
  }
  if (threadIdx.x < 8) *reinterpret_cast<float4*>(partial + ((size_t)b * chunks + ch) * 32 + threadIdx.x * 4) = sm[threadIdx.x];
}

inline int grid1d(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int paif_plane_minmax_blocks(size_t n) { return (int)((n + OG_PIX - 1) / OG_PIX); }

int paif_plane_clamp_minmax_fwd(const float* x, float* out, float* partial, float* minmax_out, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(x && out && partial && minmax_out && n > 0, PAIF_EINVAL, "plane_clamp_minmax_fwd: bad arguments");
  hipStream_t st = paif::as_stream(stream);
  const int nblk = paif_plane_minmax_blocks(n);
  hipLaunchKernelGGL(plane_minmax_partial_kernel, dim3(nblk), dim3(256), 0, st, x, partial, nblk, n);
  PAIF_LAUNCH_CHECK("plane_clamp_minmax_fwd(partial)");
  hipLaunchKernelGGL(plane_normalize_kernel, dim3(grid1d(n)), dim3(256), 0, st, x, partial, nblk, out, minmax_out, n);
  PAIF_LAUNCH_CHECK("plane_clamp_minmax_fwd(normalize)");
  return 0;
}

int paif_plane_clamp_minmax_bwd_input(const float* dout, const float* x, const float* minmax, float* partial, float* dx, size_t n,
                                      paif_stream_t stream) {
  PAIF_REQUIRE(dout && x && minmax && partial && dx && n > 0, PAIF_EINVAL, "plane_clamp_minmax_bwd: bad arguments");
  hipStream_t st = paif::as_stream(stream);
  const int nblk = paif_plane_minmax_blocks(n);
  hipLaunchKernelGGL(plane_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, dout, x, minmax, partial, n);
  PAIF_LAUNCH_CHECK("plane_clamp_minmax_bwd(reduce)");
  hipLaunchKernelGGL(plane_bwd_apply_kernel, dim3(grid1d(n)), dim3(256), 0, st, dout, x, minmax, partial, nblk, dx, n);
  PAIF_LAUNCH_CHECK("plane_clamp_minmax_bwd(apply)");
  return 0;
}

int paif_channel_sum_chunks_fwd(const float* x, float* partial, int chunks, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && partial && chunks > 0 && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_sum_chunks: bad arguments");
  hipLaunchKernelGGL(channel_sum_chunks_kernel, dim3(B * chunks), dim3(256), 0, paif::as_stream(stream), x, partial, chunks, (size_t)H * W);
  PAIF_LAUNCH_CHECK("channel_sum_chunks");
  return 0;
}

}  // extern "C"
