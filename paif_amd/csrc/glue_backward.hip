// Backward of the fusion->segmentation glue (core/model_fusion_auto.py:712-727) for the PGD loop:
//   ycc = RGB2YCrCb(vis); rgb = clamp(YCrCb2RGB(cat(fused, Cr, Cb)), 0, 1); n = (rgb - mn)/(mx - mn)  (batch-global);
//   seg_in = (255 n - mean_c)/std_c
// torch semantics reproduced: the clamp (two torch.where) passes gradient only where 0 <= r <= 1; torch.min/max
// (full reductions) spread their gradient EVENLY over all elements equal to the extremum, and elements that owe
// that value to the clamp pass nothing on.
#include "paif_common.h"

namespace {

constexpr int GB_PIX = 2048;  // pixels per block of the reduction pass

__device__ __forceinline__ void recompose(float Y, float cr, float cb, float& R, float& G, float& Bl) {
  const float c1 = __fadd_rn(cr, -0.5f), c2 = __fadd_rn(cb, -0.5f);
  const float y0 = __fadd_rn(Y, 0.0f);
  R = fmaf(c2, 0.0f, fmaf(c1, 1.403f, y0 * 1.0f));
  G = fmaf(c2, -0.344f, fmaf(c1, -0.714f, y0 * 1.0f));
  Bl = fmaf(c2, 1.773f, fmaf(c1, 0.0f, y0 * 1.0f));
}

__device__ __forceinline__ float inv_std(int c) { return c == 0 ? 1.0f / 58.395f : (c == 1 ? 1.0f / 57.12f : 1.0f / 57.375f); }

// partial[blk] = (sum d_n*(x - mx), sum d_n*(x - mn), #(x == mn), #(x == mx))
__global__ __launch_bounds__(256) void glue_bwd_reduce_kernel(const float* __restrict__ dseg, const float* __restrict__ fused,
                                                              const float* __restrict__ ycc, const float* __restrict__ minmax,
                                                              float* __restrict__ partial, int B, size_t HW) {
  const float mn = minmax[0], mx = minmax[1];
  const size_t total = (size_t)B * HW;
  const size_t start = (size_t)blockIdx.x * GB_PIX;
  float a0 = 0.f, a1 = 0.f, c0 = 0.f, c1 = 0.f;
  for (size_t i = start + threadIdx.x; i < start + GB_PIX && i < total; i += 256) {
    const size_t b = i / HW, px = i - b * HW;
    float r[3];
    recompose(fused[i], ycc[(b * 3 + 1) * HW + px], ycc[(b * 3 + 2) * HW + px], r[0], r[1], r[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = fminf(fmaxf(r[c], 0.f), 1.f);
      const float dn = dseg[(b * 3 + c) * HW + px] * 255.0f * inv_std(c);
      a0 = fmaf(dn, x - mx, a0);
      a1 = fmaf(dn, x - mn, a1);
      c0 += (x == mn) ? 1.f : 0.f;
      c1 += (x == mx) ? 1.f : 0.f;
    }
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

__global__ __launch_bounds__(256) void glue_bwd_apply_kernel(const float* __restrict__ dseg, const float* __restrict__ fused,
                                                             const float* __restrict__ ycc, const float* __restrict__ minmax,
                                                             const float* __restrict__ partial, int npartial,
                                                             const float* __restrict__ dfused_direct, float* __restrict__ dfused,
                                                             float* __restrict__ dcrcb, int B, size_t HW) {
  // every block reduces the partials in the same fixed order (one wave, lane-strided + shuffle tree)
  __shared__ float4 tot;
  if (threadIdx.x < 64) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < npartial; i += 64) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      s.x += __shfl_xor(s.x, m); s.y += __shfl_xor(s.y, m); s.z += __shfl_xor(s.z, m); s.w += __shfl_xor(s.w, m);
    }
    if (threadIdx.x == 0) tot = s;
  }
  __syncthreads();
  const float mn = minmax[0], mx = minmax[1];
  const float range = mx - mn, inv = 1.0f / range;
  const float d_mn_share = (tot.x * inv * inv) / tot.z;     // d loss / d mn, per element equal to mn
  const float d_mx_share = (-tot.y * inv * inv) / tot.w;
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, px = i - b * HW;
    float r[3], dr[3];
    recompose(fused[i], ycc[(b * 3 + 1) * HW + px], ycc[(b * 3 + 2) * HW + px], r[0], r[1], r[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = fminf(fmaxf(r[c], 0.f), 1.f);
      float dx = dseg[(b * 3 + c) * HW + px] * 255.0f * inv_std(c) * inv;
      if (x == mn) dx += d_mn_share;
      if (x == mx) dx += d_mx_share;
      dr[c] = (r[c] >= 0.f && r[c] <= 1.f) ? dx : 0.f;   // the two torch.where clamps
    }
    float dy = (dr[0] + dr[1]) + dr[2];
    if (dfused_direct) dy += dfused_direct[i];
    dfused[i] = dy;
    dcrcb[(b * 2 + 0) * HW + px] = 1.403f * dr[0] - 0.714f * dr[1];
    dcrcb[(b * 2 + 1) * HW + px] = -0.344f * dr[1] + 1.773f * dr[2];
  }
}

// d vis from (dY, dCr, dCb):  Y = .299R+.587G+.114B, Cr = (R-Y).713+.5, Cb = (B-Y).564+.5
__global__ void rgb2ycrcb_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ dcrcb, float* __restrict__ dvis, int B,
                                     size_t HW) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, px = i - b * HW;
    const float dcr = dcrcb[(b * 2 + 0) * HW + px], dcb = dcrcb[(b * 2 + 1) * HW + px];
    const float dyt = dY[i] - 0.713f * dcr - 0.564f * dcb;
    dvis[(b * 3 + 0) * HW + px] = 0.299f * dyt + 0.713f * dcr;
    dvis[(b * 3 + 1) * HW + px] = 0.587f * dyt;
    dvis[(b * 3 + 2) * HW + px] = 0.114f * dyt + 0.564f * dcb;
  }
}

// PGD update (attack/attack.py:504-512): delta <- clamp(clamp(delta + alpha*sign(g), -eps, eps), 0 - X, 1 - X)
__global__ void pgd_step_kernel(float* __restrict__ delta, const float* __restrict__ g, const float* __restrict__ X, float alpha,
                                float eps, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float sg = gi > 0.f ? 1.f : (gi < 0.f ? -1.f : 0.f);
    float d = delta[i] + alpha * sg;
    d = fminf(fmaxf(d, -eps), eps);
    const float x = X[i];
    d = fminf(fmaxf(d, 0.f - x), 1.f - x);
    delta[i] = d;
  }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaf(a, x[i], y[i]);
}

inline int grid1d(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int paif_glue_bwd_blocks(int B, int H, int W) { return (int)(((size_t)B * H * W + GB_PIX - 1) / GB_PIX); }

int paif_glue_bwd_input(const float* dseg, const float* fused, const float* ycc, const float* minmax, const float* dfused_direct,
                        float* partial, float* dfused, float* dcrcb, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(dseg && fused && ycc && minmax && partial && dfused && dcrcb && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "glue_bwd: bad arguments");
  hipStream_t st = paif::as_stream(stream);
  const size_t HW = (size_t)H * W;
  const int nblk = paif_glue_bwd_blocks(B, H, W);
  hipLaunchKernelGGL(glue_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, dseg, fused, ycc, minmax, partial, B, HW);
  PAIF_LAUNCH_CHECK("glue_bwd(reduce)");
  hipLaunchKernelGGL(glue_bwd_apply_kernel, dim3(grid1d((size_t)B * HW)), dim3(256), 0, st, dseg, fused, ycc, minmax, partial, nblk,
                     dfused_direct, dfused, dcrcb, B, HW);
  PAIF_LAUNCH_CHECK("glue_bwd(apply)");
  return 0;
}

int paif_rgb2ycrcb_bwd_input(const float* dY, const float* dcrcb, float* dvis, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(dY && dcrcb && dvis && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "rgb2ycrcb_bwd: bad arguments");
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(rgb2ycrcb_bwd_kernel, dim3(grid1d((size_t)B * HW)), dim3(256), 0, paif::as_stream(stream), dY, dcrcb, dvis, B, HW);
  PAIF_LAUNCH_CHECK("rgb2ycrcb_bwd");
  return 0;
}

int paif_pgd_step(float* delta, const float* grad_sum, const float* X, float alpha, float eps, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(delta && grad_sum && X, PAIF_EINVAL, "pgd_step: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pgd_step_kernel, dim3(grid1d(n)), dim3(256), 0, paif::as_stream(stream), delta, grad_sum, X, alpha, eps, n);
  PAIF_LAUNCH_CHECK("pgd_step");
  return 0;
}

int paif_axpy(float* y, const float* x, float a, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(y && x, PAIF_EINVAL, "axpy: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid1d(n)), dim3(256), 0, paif::as_stream(stream), y, x, a, n);
  PAIF_LAUNCH_CHECK("axpy");
  return 0;
}

}  // extern "C"
