// Training-step kernels (BASELINE configs[4]): train-mode BatchNorm (batch statistics, forward and
// backward), the small parameter-gradient reductions (PReLU slope, depthwise / stem / tail / spatial-attention conv
// weights, ECA conv1d, conv biases), stochastic-depth / Dropout2d masks and scaling, and the multi-tensor AdamW update.
// The big parameter gradients live in conv_wgrad.hip (dense convs), gemm_wgrad.hip (Linear) and norm_wgrad.hip (LayerNorm).
//
// Every reduction is two-pass and deterministic: per-workgroup partials in a fixed lane order, then a second kernel adds
// the partials in block order (in double) and ACCUMULATES into the destination (gradients accumulate like torch's
// `loss.backward()`; the caller zeroes the gradient arena once per step).  No float atomics.
//
// Layout: NHWC rows [M, C] (M = B*H*W), one float4 (4 channels) per lane.  Workgroup geometry for per-channel
// reductions: QB (a power of two <= 32 dividing C/4) channel quads x PB = 256/QB rows in parallel; blockIdx.y walks quad groups.
#include "paif_common.h"

namespace {

constexpr int MAXGRID = 256 * 8;

inline int grid_for(size_t work_items, int per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)MAXGRID) g = MAXGRID;
  if (g < 1) g = 1;
  return (int)g;
}

struct RowGeom {
  int QB, PB, cgroups, rows_per_block, nblk;
};

// C % 4 == 0.  QB = the largest power of two <= 32 dividing the quad count (so that lane % QB is the quad lane)
inline bool row_geom(int M, int C, RowGeom& g) {
  if (C <= 0 || (C & 3) || M <= 0) return false;
  const int Q = C / 4;
  int qb = 32;
  while (Q % qb) qb >>= 1;
  g.QB = qb;
  g.cgroups = Q / qb;
  g.PB = 256 / g.QB;
  long rpb = ((long)M + 1023) / 1024;            // <= 1024 row blocks
  if (rpb < 256) rpb = 256;
  rpb = (rpb + g.PB - 1) / g.PB * g.PB;
  g.rows_per_block = (int)rpb;
  g.nblk = (int)(((long)M + rpb - 1) / rpb);
  return true;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4shfl_xor(float4 a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4fma(float4& acc, float4 a, float4 b) {
  acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y); acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
}
__device__ __forceinline__ void f4fma_s(float4& acc, float s, float4 b) {
  acc.x = fmaf(s, b.x, acc.x); acc.y = fmaf(s, b.y, acc.y); acc.z = fmaf(s, b.z, acc.z); acc.w = fmaf(s, b.w, acc.w);
}

// Reduce NACC float4 accumulators over the PB row lanes of a workgroup (thread = (row lane, quad ql = tid % QB)) and store
// the workgroup's partial dst[a][C] for the quads [qbase, qbase + QB).  Fixed order: xor-shuffles inside a wave, then the
// four waves in wave order through LDS (8 accumulators at a time).
template <int NACC>
__device__ __forceinline__ void reduce_rows_store(float4 (&acc)[NACC], float* __restrict__ dst, int C, int QB, int qbase) {
  __shared__ float4 red[4][8][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
    for (int m = QB; m < 64; m <<= 1) acc[a] = f4add(acc[a], f4shfl_xor(acc[a], m));
#pragma unroll
  for (int a0 = 0; a0 < NACC; a0 += 8) {
    if (lane < QB) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
        if (a0 + a < NACC) red[wave][a][lane] = acc[a0 + a];
    }
    __syncthreads();
    if ((int)threadIdx.x < 8 * QB) {
      const int a = threadIdx.x / QB, ql = threadIdx.x - a * QB;
      if (a0 + a < NACC) {
        float4 s = f4add(f4add(red[0][a][ql], red[1][a][ql]), f4add(red[2][a][ql], red[3][a][ql]));
        *reinterpret_cast<float4*>(dst + (size_t)(a0 + a) * C + (size_t)(qbase + ql) * 4) = s;
      }
    }
    __syncthreads();
  }
}

// out[i] (+)= sum over blocks of ws[blk][i]   (i < n), double accumulation in block order
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ ws, int nblk, int n, float* __restrict__ out,
                                                           int accumulate) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)ws[(size_t)b * n + i];
    out[i] = (accumulate ? out[i] : 0.f) + (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// layout helper: NHWC [B,HW,ldx] channels [0,C) -> NCHW [B,C,HW]
// ---------------------------------------------------------------------------------------------
__global__ void nhwc_slice_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, size_t HW, int C, int ldx) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t b = t / C;
    y[i] = x[(b * HW + px) * ldx + c];
  }
}

// y[M,Cd] = x[M,Cs] zero padded
__global__ void pad_channels_kernel(const float* __restrict__ x, float* __restrict__ y, size_t M, int Cs, int Cd) {
  const size_t total = M * Cd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cd);
    const size_t m = i / Cd;
    y[i] = c < Cs ? x[m * Cs + c] : 0.f;
  }
}

// Cell_Decom.decomposition's return value (core/model_fusion_auto.py:522-535): LF = cat(LF_eps0, LF_eps1) and
// HF = cat(x - LF_eps0, x - LF_eps1) on the channel axis.  x [M,32], lf [2][M,32] -> lfcat, hfcat [M,64]
__global__ void decomp_cat_kernel(const float* __restrict__ x, const float* __restrict__ lf, float* __restrict__ lfcat,
                                  float* __restrict__ hfcat, size_t M) {
  const size_t n4 = M * 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i >> 4;
    const int q = (int)(i & 15), e = q >> 3, qq = q & 7;
    const float4 l = *reinterpret_cast<const float4*>(lf + ((size_t)e * M + px) * 32 + qq * 4);
    const float4 xv = *reinterpret_cast<const float4*>(x + px * 32 + qq * 4);
    reinterpret_cast<float4*>(lfcat)[i] = l;
    reinterpret_cast<float4*>(hfcat)[i] = make_float4(xv.x - l.x, xv.y - l.y, xv.z - l.z, xv.w - l.w);
  }
}

// ---------------------------------------------------------------------------------------------
// train-mode BatchNorm2d statistics over the rows of x [M,C]  (nn.BatchNorm2d in ResidualModule / DilConv,
// operations_m.py:451-464,494-506; mmcv ConvModule's BN in the head, core/segformer_head.py:50-55)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ ws, int M, int C, int QB,
                                                       int rows_per_block) {
  __shared__ double red[4][2][32][4];
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB, PB = 256 / QB;
  const int q = blockIdx.y * QB + ql;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  for (int r = r0 + pl; r < r1; r += PB) {
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + q * 4);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = QB; m < 64; m <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i] += __shfl_xor(s[i], m); ss[i] += __shfl_xor(ss[i], m); }
  if (lane < QB)
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[wave][0][lane][i] = s[i]; red[wave][1][lane][i] = ss[i]; }
  __syncthreads();
  if ((int)threadIdx.x < 2 * QB) {
    const int which = threadIdx.x / QB, l = threadIdx.x - which * QB;
    double* dst = ws + ((size_t)blockIdx.x * 2 + which) * C + (size_t)(blockIdx.y * QB + l) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = ((red[0][which][l][i] + red[1][which][l][i]) + red[2][which][l][i]) + red[3][which][l][i];
  }
}

// mean / biased var -> scale = gamma * invstd, shift = beta - mean * scale; running statistics (momentum, unbiased var)
__global__ void bn_stats_finish_kernel(const double* __restrict__ ws, int nblk, int M, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                       float* __restrict__ scale_out, float* __restrict__ shift_out) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += ws[((size_t)b * 2) * C + c]; ss += ws[((size_t)b * 2 + 1) * C + c]; }
    const double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, varf = (float)var;
    const float invstd = 1.0f / sqrtf(varf + eps);
    mean_out[c] = meanf;
    invstd_out[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale_out[c] = sc;
    shift_out[c] = bt - meanf * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    if (running_var) {
      const float unbiased = M > 1 ? (float)(var * ((double)M / (double)(M - 1))) : varf;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
}

// eval-mode BatchNorm as the same 4 statistics (running mean / var): lets one backward kernel serve both modes
__global__ void bn_eval_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                     const float* __restrict__ rv, float eps, int C, float* __restrict__ mean_out,
                                     float* __restrict__ invstd_out, float* __restrict__ scale_out, float* __restrict__ shift_out) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    mean_out[c] = rm[c];
    invstd_out[c] = invstd;
    scale_out[c] = sc;
    shift_out[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
  }
}

// out = act(x * scale[c] + shift[c]) + res0 + res1  (z_out: the pre-activation, optional)
__global__ __launch_bounds__(256) void affine_act_res_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, const float* __restrict__ prelu,
                                                             const float* __restrict__ res0, const float* __restrict__ res1,
                                                             float* __restrict__ out, float* __restrict__ z_out, size_t n4, int Q) {
  const float slope = act == 1 ? *prelu : 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i % Q);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = scale ? reinterpret_cast<const float4*>(scale)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? reinterpret_cast<const float4*>(shift)[q] : f4zero();
    float4 z = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (z_out) reinterpret_cast<float4*>(z_out)[i] = z;
    if (act == 1) { z.x = paif::prelu_f(z.x, slope); z.y = paif::prelu_f(z.y, slope); z.z = paif::prelu_f(z.z, slope); z.w = paif::prelu_f(z.w, slope); }
    else if (act == 2) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
    if (res0) z = f4add(z, reinterpret_cast<const float4*>(res0)[i]);
    if (res1) z = f4add(z, reinterpret_cast<const float4*>(res1)[i]);
    reinterpret_cast<float4*>(out)[i] = z;
  }
}

__device__ __forceinline__ float dact(float z, int act, float slope) {
  return act == 1 ? (z >= 0.f ? 1.f : slope) : (act == 2 ? (z > 0.f ? 1.f : 0.f) : 1.f);
}

// BatchNorm(+activation) backward, pass 1: per-channel sums of dz, dz*xhat, and g*z over z<0 (PReLU slope)
//   z = x*scale + shift, xhat = (x - mean)*invstd, dz = g * act'(z)
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                int act, const float* __restrict__ prelu, float* __restrict__ ws,
                                                                int M, int C, int QB, int rows_per_block) {
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB, PB = 256 / QB;
  const int q = blockIdx.y * QB + ql;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const float slope = act == 1 ? *prelu : 0.f;
  const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
  const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
  float4 acc[3] = {f4zero(), f4zero(), f4zero()};
  for (int r = r0 + pl; r < r1; r += PB) {
    const size_t o = (size_t)r * C + q * 4;
    const float4 gv = *reinterpret_cast<const float4*>(g + o);
    const float4 xv = *reinterpret_cast<const float4*>(x + o);
    const float4 z = make_float4(fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w));
    const float4 dz = make_float4(gv.x * dact(z.x, act, slope), gv.y * dact(z.y, act, slope), gv.z * dact(z.z, act, slope),
                                  gv.w * dact(z.w, act, slope));
    const float4 xh = make_float4((xv.x - mu.x) * is.x, (xv.y - mu.y) * is.y, (xv.z - mu.z) * is.z, (xv.w - mu.w) * is.w);
    acc[0] = f4add(acc[0], dz);
    f4fma(acc[1], dz, xh);
    if (act == 1) {
      acc[2].x += z.x < 0.f ? gv.x * z.x : 0.f; acc[2].y += z.y < 0.f ? gv.y * z.y : 0.f;
      acc[2].z += z.z < 0.f ? gv.z * z.z : 0.f; acc[2].w += z.w < 0.f ? gv.w * z.w : 0.f;
    }
  }
  reduce_rows_store<3>(acc, ws + (size_t)blockIdx.x * 3 * C, C, QB, blockIdx.y * QB);
}

// sums[0][c] = sum dz, sums[1][c] = sum dz*xhat; dbeta += sums[0], dgamma += sums[1], dslope += sum_c sums[2]
__global__ void bn_act_bwd_finish_kernel(const float* __restrict__ ws, int nblk, int C, float* __restrict__ sums,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dslope) {
  __shared__ double sl[256];
  double slope_part = 0.0;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
      const float* p = ws + (size_t)b * 3 * C;
      s0 += p[c]; s1 += p[C + c]; s2 += p[2 * C + c];
    }
    sums[c] = (float)s0; sums[C + c] = (float)s1;
    if (dbeta) dbeta[c] += (float)s0;
    if (dgamma) dgamma[c] += (float)s1;
    slope_part += s2;
  }
  sl[threadIdx.x] = slope_part;
  __syncthreads();
  if (threadIdx.x == 0 && dslope) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += sl[i];
    dslope[0] += (float)t;
  }
}

// pass 2: dx = scale * (dz - sums0/M - xhat * sums1/M)      (scale = gamma * invstd)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               int act, const float* __restrict__ prelu, const float* __restrict__ sums,
                                                               float inv_m, float* __restrict__ dx, size_t n4, int Q) {
  const float slope = act == 1 ? *prelu : 0.f;
  const int C = Q * 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i % Q);
    const float4 gv = reinterpret_cast<const float4*>(g)[i], xv = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
    const float4 s0 = reinterpret_cast<const float4*>(sums)[q], s1 = reinterpret_cast<const float4*>(sums + C)[q];
    float4 o;
#define PAIF_BN_APPLY(f)                                            \
    {                                                               \
      const float z = fmaf(xv.f, sc.f, sh.f);                       \
      const float dz = gv.f * dact(z, act, slope);                  \
      const float xh = (xv.f - mu.f) * is.f;                        \
      o.f = sc.f * (dz - s0.f * inv_m - xh * (s1.f * inv_m));       \
    }
    PAIF_BN_APPLY(x) PAIF_BN_APPLY(y) PAIF_BN_APPLY(z) PAIF_BN_APPLY(w)
#undef PAIF_BN_APPLY
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// PReLU backward on an element stream: dx = t * P'(r) + add (dx optional), partial[blk] = sum t*r over r<0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ t, const float* __restrict__ r,
                                                        const float* __restrict__ add, const float* __restrict__ prelu,
                                                        float* __restrict__ dx, float* __restrict__ partial, size_t n4) {
  __shared__ float red[4];
  const float slope = *prelu;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 tv = reinterpret_cast<const float4*>(t)[i], rv = reinterpret_cast<const float4*>(r)[i];
    s += (rv.x < 0.f ? tv.x * rv.x : 0.f) + (rv.y < 0.f ? tv.y * rv.y : 0.f) + (rv.z < 0.f ? tv.z * rv.z : 0.f) +
         (rv.w < 0.f ? tv.w * rv.w : 0.f);
    if (dx) {
      float4 o = make_float4(tv.x * (rv.x >= 0.f ? 1.f : slope), tv.y * (rv.y >= 0.f ? 1.f : slope),
                             tv.z * (rv.z >= 0.f ? 1.f : slope), tv.w * (rv.w >= 0.f ? 1.f : slope));
      if (add) o = f4add(o, reinterpret_cast<const float4*>(add)[i]);
      reinterpret_cast<float4*>(dx)[i] = o;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] += factor * sum(partial)
__global__ void scalar_finish_kernel(const float* __restrict__ partial, int n, float factor, float* __restrict__ out) {
  __shared__ double sl[64];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  sl[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += sl[i];
    out[0] += (float)(t * (double)factor);
  }
}

// tail: fused = tanh(PReLU(z)) (core/model_fusion_auto.py:615-619,634): dz = dfused*(1-fused^2)*P'(z); slope partials
__global__ __launch_bounds__(256) void tail_dz_kernel(const float* __restrict__ dfused, const float* __restrict__ fused,
                                                      const float* __restrict__ z, const float* __restrict__ prelu,
                                                      float* __restrict__ dz, float* __restrict__ partial, size_t n) {
  __shared__ float red[4];
  const float slope = *prelu;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float f = fused[i], zz = z[i];
    const float t = dfused[i] * (1.f - f * f);
    dz[i] = t * (zz >= 0.f ? 1.f : slope);
    s += zz < 0.f ? t * zz : 0.f;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
// column sums of x [M, ld] (first C columns): conv bias gradients (Cell_Decom.conv1x1_lf/hf.bias)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, float* __restrict__ ws, int M, int C, int QB,
                                                     int rows_per_block) {
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB, PB = 256 / QB;
  const int q = blockIdx.y * QB + ql;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 acc[1] = {f4zero()};
  for (int r = r0 + pl; r < r1; r += PB) acc[0] = f4add(acc[0], *reinterpret_cast<const float4*>(x + (size_t)r * ld + q * 4));
  reduce_rows_store<1>(acc, ws + (size_t)blockIdx.x * C, C, QB, blockIdx.y * QB);
}

// ---------------------------------------------------------------------------------------------
// depthwise-conv weight (and bias) gradient:  dW[c][ky][kx] = sum_px dy[px][c] * in(x)[px + off][c]
//   DilConv's depthwise (operations_m.py:497-498, in = ReLU), MiT Mlp.dwconv (core/mix_transformer.py:376-387, + bias)
// partial layout ws[blk][K*K + 1][C] (the last plane = sum dy)
// ---------------------------------------------------------------------------------------------
template <int K, int DIL>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ ws, int in_relu, int B, int H, int W, int C, int QB,
                                                           int rows_per_block) {
  constexpr int P = DIL * (K - 1) / 2;
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB, PB = 256 / QB;
  const int q = blockIdx.y * QB + ql;
  const int M = B * H * W;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 acc[K * K + 1];
#pragma unroll
  for (int a = 0; a < K * K + 1; ++a) acc[a] = f4zero();
  for (int r = r0 + pl; r < r1; r += PB) {
    const int xx = r % W;
    const int t = r / W;
    const int yy = t % H;
    const size_t img = (size_t)(t - yy) * W;
    const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * C + q * 4);
    acc[K * K] = f4add(acc[K * K], d);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ys = yy + ky * DIL - P;
      const int yc = min(max(ys, 0), H - 1);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int xs = xx + kx * DIL - P;
        const int xc = min(max(xs, 0), W - 1);
        float4 v = *reinterpret_cast<const float4*>(x + (img + (size_t)yc * W + xc) * C + q * 4);   // unconditional, zero by select
        if (ys < 0 || ys >= H || xs < 0 || xs >= W) v = f4zero();
        if (in_relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        f4fma(acc[ky * K + kx], d, v);
      }
    }
  }
  reduce_rows_store<K * K + 1>(acc, ws + (size_t)blockIdx.x * (K * K + 1) * C, C, QB, blockIdx.y * QB);
}

// dW[c][tap] += sum_blk ws[blk][tap][c];  db[c] += sum_blk ws[blk][KK][c]
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ ws, int nblk, int KK, int C,
                                                                  float* __restrict__ dw, float* __restrict__ db) {
  const int total = (KK + 1) * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int a = i / C, c = i - a * C;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)ws[(size_t)b * total + i];
    if (a < KK) dw[(size_t)c * KK + a] += (float)s;
    else if (db) db[c] += (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// stem weight gradient: feat = PReLU(conv3x3 1->32 (img)) (core/model_fusion_auto.py:607-614); the pre-activation is
// recomputed from the 9 image taps.  partial ws[blk][10][32]: 9 tap planes of dW, then the slope plane (summed over c later)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, size_t img_bstride, const float* __restrict__ dfeat,
                                                         const float* __restrict__ w, const float* __restrict__ prelu,
                                                         float* __restrict__ ws, int B, int H, int W, int rows_per_block) {
  const int ql = threadIdx.x & 7, pl = threadIdx.x >> 3;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(4 * ql + c) * 9 + k];
  const float slope = *prelu;
  const int M = B * H * W;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 acc[10];
#pragma unroll
  for (int a = 0; a < 10; ++a) acc[a] = f4zero();
  for (int r = r0 + pl; r < r1; r += 32) {
    const int xx = r % W;
    const int t = r / W;
    const int yy = t % H, b = t / H;
    const float* plane = img + (size_t)b * img_bstride;
    float tap[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ys = yy + ky - 1, xs = xx + kx - 1;
        const float v = plane[(size_t)min(max(ys, 0), H - 1) * W + min(max(xs, 0), W - 1)];
        tap[ky * 3 + kx] = (ys < 0 || ys >= H || xs < 0 || xs >= W) ? 0.f : v;
      }
    const float4 d = *reinterpret_cast<const float4*>(dfeat + (size_t)r * 32 + ql * 4);
    float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) z[c] = fmaf(tap[k], wr[c][k], z[c]);
    const float4 dz = make_float4(d.x * (z[0] >= 0.f ? 1.f : slope), d.y * (z[1] >= 0.f ? 1.f : slope),
                                  d.z * (z[2] >= 0.f ? 1.f : slope), d.w * (z[3] >= 0.f ? 1.f : slope));
#pragma unroll
    for (int k = 0; k < 9; ++k) f4fma_s(acc[k], tap[k], dz);
    acc[9].x += z[0] < 0.f ? d.x * z[0] : 0.f; acc[9].y += z[1] < 0.f ? d.y * z[1] : 0.f;
    acc[9].z += z[2] < 0.f ? d.z * z[2] : 0.f; acc[9].w += z[3] < 0.f ? d.w * z[3] : 0.f;
  }
  reduce_rows_store<10>(acc, ws + (size_t)blockIdx.x * 10 * 32, 32, 8, 0);
}

// dW[c][k] += sum ws[blk][k][c]; dslope += sum_c sum ws[blk][9][c]
__global__ void stem_wgrad_finish_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ dw, float* __restrict__ dslope) {
  __shared__ double sl[32];
  const int i = threadIdx.x;   // 320 threads: (plane a = i / 32, c = i % 32)
  const int a = i / 32, c = i - a * 32;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)ws[(size_t)b * 320 + i];
  if (a < 9) dw[c * 9 + a] += (float)s;
  else sl[c] = s;
  __syncthreads();
  if (i == 0 && dslope) {
    double t = 0.0;
    for (int k = 0; k < 32; ++k) t += sl[k];
    dslope[0] += (float)t;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient of a Cm -> 1 conv (stride 1, "same" padding) from the 1-channel output gradient:
//   dW[c][ky][kx] = sum_px s[px] * m[px + (ky-P, kx-P)][c]
// stem_out.1 (Cm = 16, k = 3, core/model_fusion_auto.py:617), spatial_attn_layer_M's conv (Cm = 4, k = 5, :1361)
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void corr1_wgrad_kernel(const float* __restrict__ s, const float* __restrict__ m, float* __restrict__ ws,
                                                          int B, int H, int W, int Cm, int QB, int rows_per_block) {
  constexpr int P = (K - 1) / 2;
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB, PB = 256 / QB;
  const int M = B * H * W;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 acc[K * K];
#pragma unroll
  for (int a = 0; a < K * K; ++a) acc[a] = f4zero();
  for (int r = r0 + pl; r < r1; r += PB) {
    const int xx = r % W;
    const int t = r / W;
    const int yy = t % H;
    const size_t img = (size_t)(t - yy) * W;
    const float sv = s[r];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ys = yy + ky - P;
      const int yc = min(max(ys, 0), H - 1);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int xs = xx + kx - P;
        const int xc = min(max(xs, 0), W - 1);
        float4 v = *reinterpret_cast<const float4*>(m + (img + (size_t)yc * W + xc) * Cm + ql * 4);
        if (ys < 0 || ys >= H || xs < 0 || xs >= W) v = f4zero();
        f4fma_s(acc[ky * K + kx], sv, v);
      }
    }
  }
  reduce_rows_store<K * K>(acc, ws + (size_t)blockIdx.x * K * K * Cm, Cm, QB, 0);
}

// dW[c][tap] += sum_blk ws[blk][tap][c]
__global__ void corr1_wgrad_finish_kernel(const float* __restrict__ ws, int nblk, int KK, int Cm, float* __restrict__ dw) {
  const int total = KK * Cm;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = i / Cm, c = i - a * Cm;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)ws[(size_t)b * total + i];
    dw[(size_t)c * KK + a] += (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// ECA conv1d weight gradient (operations_m.py:353-367): gate = sigmoid(conv1d_k(mean)), dpre = dgate*gate*(1-gate)
//   dW[j] = sum_b sum_c dpre[b][c] * mean[b][c + j - pad]
// pool_partial: the forward conv's per-tile channel sums; dgate_partial: eca_bwd's per-block sums of du*o
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void eca_wgrad_kernel(const float* __restrict__ pool_partial, int tiles_per_img,
                                                         const float* __restrict__ dgate_partial, int blocks_per_img,
                                                         const float* __restrict__ gate, int k, float inv_hw, float* __restrict__ ws) {
  // one workgroup per image: thread (part = tid >> 5, c = tid & 31) sums every 32nd partial, the 32 parts are added in a
  // fixed order (deterministic); ws[b][j] = this image's contribution to dW[j]
  __shared__ float ps[32][32], pd[32][32];
  __shared__ float mean[32], dpre[32];
  const int b = blockIdx.x, c = threadIdx.x & 31, part = threadIdx.x >> 5;
  float s = 0.f, d = 0.f;
  for (int t = part; t < tiles_per_img; t += 32) s += pool_partial[((size_t)b * tiles_per_img + t) * 32 + c];
  for (int t = part; t < blocks_per_img; t += 32) d += dgate_partial[((size_t)b * blocks_per_img + t) * 32 + c];
  ps[part][c] = s; pd[part][c] = d;
  __syncthreads();
  if (threadIdx.x < 32) {
    double ts = 0.0, td = 0.0;
    for (int i = 0; i < 32; ++i) { ts += (double)ps[i][c]; td += (double)pd[i][c]; }
    const float gt = gate[b * 32 + c];
    mean[c] = (float)ts * inv_hw;
    dpre[c] = (float)td * gt * (1.f - gt);
  }
  __syncthreads();
  if ((int)threadIdx.x < k) {
    const int j = threadIdx.x, pad = (k - 1) / 2;
    double t = 0.0;
    for (int cc = 0; cc < 32; ++cc) {
      const int src = cc + j - pad;
      if (src >= 0 && src < 32) t += (double)dpre[cc] * (double)mean[src];
    }
    ws[b * 9 + j] = (float)t;
  }
}

__global__ void eca_wgrad_finish_kernel(const float* __restrict__ ws, int B, int k, float* __restrict__ dw) {
  const int j = threadIdx.x;
  if (j >= k) return;
  double t = 0.0;
  for (int b = 0; b < B; ++b) t += (double)ws[b * 9 + j];
  dw[j] += (float)t;
}

// ---------------------------------------------------------------------------------------------
// Cell_Decom 1x1 weight gradient, unfolded: the forward ran the folded 96-channel form over [x, LF1, LF2]
// (paif_pack_decomp1x1_weight); with G = dY^T [x, LF1, LF2]  ([32][96]):
//   dW[:, 0:32] += G1, dW[:, 32:64] += G2, dW[:, 64:96] += Gx - G1, dW[:, 96:128] += Gx - G2
// (W's input channels are cat[LF1, LF2, x-LF1, x-LF2], core/model_fusion_auto.py:512-513,522-535)
// ---------------------------------------------------------------------------------------------
__global__ void unfold_decomp1x1_wgrad_kernel(const float* __restrict__ G, float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 32 * 128) return;
  const int n = i >> 7, c = i & 127;
  const float* g = G + n * 96;
  float v;
  if (c < 32) v = g[32 + c];
  else if (c < 64) v = g[64 + c - 32];
  else if (c < 96) v = g[c - 64] - g[32 + c - 64];
  else v = g[c - 96] - g[64 + c - 96];
  dw[i] += v;
}

// conv weight gradient computed as a GEMM over im2col columns: dw[n][c][tap] += dWp[n][tap*Cin + c]
__global__ void unpack_conv_gemm_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin, int KK, int Kpad) {
  const size_t total = (size_t)Cout * Cin * KK;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % KK);
    const size_t t = i / KK;
    const int c = (int)(t % Cin);
    const size_t n = t / Cin;
    dw[i] += dwp[n * Kpad + (size_t)tap * Cin + c];
  }
}

// ---------------------------------------------------------------------------------------------
// stochastic depth / Dropout2d: counter-based keep masks and the per-sample(-channel) scaling
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ULL;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// out[i] = u(seed, offset + i) >= p ? 1/(1-p) : 0, u = (splitmix64(seed * 0x100000001B3 + offset + i) >> 11) * 2^-53
// (the same counter-based uniform as paif_amd/synthetic.py:hash_uniform, so a host can replay any mask)
__global__ void keep_mask_kernel(float* __restrict__ out, int n, unsigned long long seed, unsigned long long offset, float p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = splitmix64(seed * 0x100000001B3ULL + offset + (unsigned long long)i);
  const double u = (double)(key >> 11) * (1.0 / 9007199254740992.0);
  out[i] = u >= (double)p ? 1.0f / (1.0f - p) : 0.f;
}

// out[b][r][c] = x[b][r][c] * s[b] (or s[b][c]) + res
__global__ __launch_bounds__(256) void rowscale_add_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           const float* __restrict__ res, float* __restrict__ out, size_t n4,
                                                           size_t per_b4, int Q, int per_channel) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t b = i / per_b4;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    if (per_channel) {
      const float4 sv = reinterpret_cast<const float4*>(s + b * Q * 4)[i % Q];
      v = make_float4(v.x * sv.x, v.y * sv.y, v.z * sv.z, v.w * sv.w);
    } else {
      const float sv = s[b];
      v = make_float4(v.x * sv, v.y * sv, v.z * sv, v.w * sv);
    }
    if (res) v = f4add(v, reinterpret_cast<const float4*>(res)[i]);
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// multi-tensor AdamW over a flat arena (utils/optimizer.py:3-33 = torch.optim.AdamW semantics, eps 1e-8, no amsgrad):
//   p *= 1 - lr*wd;  m += (1-b1)*(g-m);  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// chunk_group[i]: parameter group of arena chunk i (1024 floats); 255 = skip (padding)
// ---------------------------------------------------------------------------------------------
struct AdamGroups {
  float decay[8];       // 1 - lr*wd
  float step_size[8];   // lr / bc1
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const unsigned char* __restrict__ chunk_group, size_t nchunks,
                                                    AdamGroups grp, float one_minus_b1, float b2, float one_minus_b2, float bc2_sqrt,
                                                    float eps) {
  for (size_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int gi = chunk_group[ch];
    if (gi >= 8) continue;
    const float decay = grp.decay[gi], step = grp.step_size[gi];
    const size_t i = ch * 256 + threadIdx.x;
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define PAIF_ADAM(f)                                              \
    {                                                             \
      pv.f = pv.f * decay;                                        \
      mv.f = mv.f + one_minus_b1 * (gv.f - mv.f);                 \
      vv.f = vv.f * b2 + one_minus_b2 * (gv.f * gv.f);            \
      const float denom = sqrtf(vv.f) / bc2_sqrt + eps;           \
      pv.f = pv.f - step * (mv.f / denom);                        \
    }
    PAIF_ADAM(x) PAIF_ADAM(y) PAIF_ADAM(z) PAIF_ADAM(w)
#undef PAIF_ADAM
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int paif_nhwc_slice_to_nchw_fwd(const float* x, float* y, int B, int HW, int C, int ldx, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && ldx >= C, PAIF_EINVAL, "nhwc_slice_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_slice_to_nchw_kernel, dim3(grid_for((size_t)B * HW * C, 256)), dim3(256), 0, paif::as_stream(stream), x, y, B,
                     (size_t)HW, C, ldx);
  PAIF_LAUNCH_CHECK("nhwc_slice_to_nchw");
  return 0;
}

int paif_pad_channels_fwd(const float* x, float* y, size_t M, int Cs, int Cd, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && M > 0 && Cs > 0 && Cd >= Cs, PAIF_EINVAL, "pad_channels: bad arguments");
  hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(M * Cd, 256)), dim3(256), 0, paif::as_stream(stream), x, y, M, Cs, Cd);
  PAIF_LAUNCH_CHECK("pad_channels");
  return 0;
}

int paif_decomp_cat_fwd(const float* x, const float* lf, float* lfcat, float* hfcat, size_t M, paif_stream_t stream) {
  PAIF_REQUIRE(x && lf && lfcat && hfcat && M > 0, PAIF_EINVAL, "decomp_cat: bad arguments");
  hipLaunchKernelGGL(decomp_cat_kernel, dim3(grid_for(M * 16, 256)), dim3(256), 0, paif::as_stream(stream), x, lf, lfcat, hfcat, M);
  PAIF_LAUNCH_CHECK("decomp_cat");
  return 0;
}

size_t paif_row_reduce_workspace_floats(int M, int C, int nacc) {
  RowGeom g;
  if (!row_geom(M, C, g)) return 0;
  return (size_t)g.nblk * nacc * C;
}

int paif_bn_stats_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      float* workspace, paif_stream_t stream) {
  PAIF_REQUIRE(x && mean && invstd && scale && shift && workspace && M > 0, PAIF_EINVAL, "bn_stats: bad arguments");
  RowGeom g;
  PAIF_REQUIRE(row_geom(M, C, g), PAIF_ENOSUP, "bn_stats: C=%d", C);
  hipStream_t st = paif::as_stream(stream);
  double* ws = reinterpret_cast<double*>(workspace);   // 2 floats per double: workspace = 2 * row_reduce_workspace_floats(M, C, 2)
  hipLaunchKernelGGL(bn_stats_kernel, dim3(g.nblk, g.cgroups), dim3(256), 0, st, x, ws, M, C, g.QB, g.rows_per_block);
  PAIF_LAUNCH_CHECK("bn_stats");
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, g.nblk, M, C, gamma, beta, eps, momentum,
                     running_mean, running_var, mean, invstd, scale, shift);
  PAIF_LAUNCH_CHECK("bn_stats_finish");
  return 0;
}

int paif_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                       float* mean, float* invstd, float* scale, float* shift, paif_stream_t stream) {
  PAIF_REQUIRE(running_mean && running_var && mean && invstd && scale && shift && C > 0, PAIF_EINVAL, "bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, paif::as_stream(stream), gamma, beta, running_mean,
                     running_var, eps, C, mean, invstd, scale, shift);
  PAIF_LAUNCH_CHECK("bn_eval_stats");
  return 0;
}

int paif_affine_act_res_fwd(const float* x, const float* scale, const float* shift, int act, const float* prelu, const float* res0,
                            const float* res1, float* out, float* z_out, size_t M, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && out && M > 0 && C > 0 && (C & 3) == 0, PAIF_EINVAL, "affine_act_res: bad arguments");
  PAIF_REQUIRE(act >= 0 && act <= 2 && (act != 1 || prelu), PAIF_EINVAL, "affine_act_res: act=%d", act);
  const size_t n4 = M * (C / 4);
  hipLaunchKernelGGL(affine_act_res_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, paif::as_stream(stream), x, scale, shift, act, prelu,
                     res0, res1, out, z_out, n4, C / 4);
  PAIF_LAUNCH_CHECK("affine_act_res");
  return 0;
}

int paif_bn_act_bwd(const float* g, const float* x, const float* scale, const float* shift, const float* mean, const float* invstd,
                    int act, const float* prelu, float* dx, float* dgamma, float* dbeta, float* dslope, float* sums, float* workspace,
                    int training, int M, int C, paif_stream_t stream) {
  PAIF_REQUIRE(g && x && scale && shift && mean && invstd && dx && sums && workspace && M > 0, PAIF_EINVAL, "bn_act_bwd: bad arguments");
  PAIF_REQUIRE(act >= 0 && act <= 2 && (act != 1 || prelu), PAIF_EINVAL, "bn_act_bwd: act=%d", act);
  RowGeom gm;
  PAIF_REQUIRE(row_geom(M, C, gm), PAIF_ENOSUP, "bn_act_bwd: C=%d", C);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(gm.nblk, gm.cgroups), dim3(256), 0, st, g, x, scale, shift, mean, invstd, act, prelu,
                     workspace, M, C, gm.QB, gm.rows_per_block);
  PAIF_LAUNCH_CHECK("bn_act_bwd(reduce)");
  hipLaunchKernelGGL(bn_act_bwd_finish_kernel, dim3(1), dim3(256), 0, st, workspace, gm.nblk, C, sums, dgamma, dbeta,
                     act == 1 ? dslope : (float*)nullptr);
  PAIF_LAUNCH_CHECK("bn_act_bwd(finish)");
  const size_t n4 = (size_t)M * (C / 4);
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, st, g, x, scale, shift, mean, invstd, act, prelu, sums,
                     training ? 1.0f / (float)M : 0.f, dx, n4, C / 4);
  PAIF_LAUNCH_CHECK("bn_act_bwd(apply)");
  return 0;
}

int paif_prelu_bwd(const float* t, const float* r, const float* add, const float* prelu, float factor, float* dx, float* dslope,
                   float* workspace, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(t && r && prelu && dslope && workspace && n > 0 && (n & 3) == 0, PAIF_EINVAL, "prelu_bwd: bad arguments");
  const int nblk = grid_for(n / 4, 256);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nblk), dim3(256), 0, st, t, r, add, prelu, dx, workspace, n / 4);
  PAIF_LAUNCH_CHECK("prelu_bwd");
  hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(64), 0, st, workspace, nblk, factor, dslope);
  PAIF_LAUNCH_CHECK("prelu_bwd(finish)");
  return 0;
}

int paif_tail_dz(const float* dfused, const float* fused, const float* z, const float* prelu, float* dz, float* dslope, float* workspace,
                 size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(dfused && fused && z && prelu && dz && dslope && workspace && n > 0, PAIF_EINVAL, "tail_dz: bad arguments");
  const int nblk = grid_for(n, 256);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(tail_dz_kernel, dim3(nblk), dim3(256), 0, st, dfused, fused, z, prelu, dz, workspace, n);
  PAIF_LAUNCH_CHECK("tail_dz");
  hipLaunchKernelGGL(scalar_finish_kernel, dim3(1), dim3(64), 0, st, workspace, nblk, 1.0f, dslope);
  PAIF_LAUNCH_CHECK("tail_dz(finish)");
  return 0;
}

int paif_colsum(const float* x, int ld, float* out, float* workspace, int M, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && out && workspace && M > 0 && ld >= C && (ld & 3) == 0, PAIF_EINVAL, "colsum: bad arguments");
  RowGeom g;
  PAIF_REQUIRE(row_geom(M, C, g), PAIF_ENOSUP, "colsum: C=%d", C);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(colsum_kernel, dim3(g.nblk, g.cgroups), dim3(256), 0, st, x, ld, workspace, M, C, g.QB, g.rows_per_block);
  PAIF_LAUNCH_CHECK("colsum");
  hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, g.nblk, C, out, 1);
  PAIF_LAUNCH_CHECK("colsum(finish)");
  return 0;
}

int paif_dwconv_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int k, int dil, int in_relu, int B, int H,
                      int W, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && dy && dw && workspace && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv_wgrad: bad arguments");
  RowGeom g;
  PAIF_REQUIRE(row_geom(B * H * W, C, g), PAIF_ENOSUP, "dwconv_wgrad: C=%d", C);
  hipStream_t st = paif::as_stream(stream);
  const dim3 grid(g.nblk, g.cgroups), blk(256);
  switch (k * 10 + dil) {
    case 31: hipLaunchKernelGGL((dwconv_wgrad_kernel<3, 1>), grid, blk, 0, st, x, dy, workspace, in_relu, B, H, W, C, g.QB, g.rows_per_block); break;
    case 32: hipLaunchKernelGGL((dwconv_wgrad_kernel<3, 2>), grid, blk, 0, st, x, dy, workspace, in_relu, B, H, W, C, g.QB, g.rows_per_block); break;
    case 51: hipLaunchKernelGGL((dwconv_wgrad_kernel<5, 1>), grid, blk, 0, st, x, dy, workspace, in_relu, B, H, W, C, g.QB, g.rows_per_block); break;
    case 52: hipLaunchKernelGGL((dwconv_wgrad_kernel<5, 2>), grid, blk, 0, st, x, dy, workspace, in_relu, B, H, W, C, g.QB, g.rows_per_block); break;
    default:
      paif::set_error("dwconv_wgrad: kernel %d dil %d not built", k, dil);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("dwconv_wgrad");
  const int total = (k * k + 1) * C;
  hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, g.nblk, k * k, C, dw, db);
  PAIF_LAUNCH_CHECK("dwconv_wgrad(finish)");
  return 0;
}

int paif_stem_wgrad(const float* img, size_t img_bstride, const float* dfeat, const float* w, const float* prelu, float* dw, float* dslope,
                    float* workspace, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(img && dfeat && w && prelu && dw && dslope && workspace && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "stem_wgrad: bad arguments");
  PAIF_REQUIRE(img_bstride >= (size_t)H * W, PAIF_EINVAL, "stem_wgrad: batch stride");
  RowGeom g;
  row_geom(B * H * W, 32, g);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(g.nblk), dim3(256), 0, st, img, img_bstride, dfeat, w, prelu, workspace, B, H, W, g.rows_per_block);
  PAIF_LAUNCH_CHECK("stem_wgrad");
  hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3(1), dim3(320), 0, st, workspace, g.nblk, dw, dslope);
  PAIF_LAUNCH_CHECK("stem_wgrad(finish)");
  return 0;
}

int paif_corr1_wgrad(const float* s, const float* m, float* dw, float* workspace, int Cm, int k, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(s && m && dw && workspace && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "corr1_wgrad: bad arguments");
  RowGeom g;
  PAIF_REQUIRE(row_geom(B * H * W, Cm, g) && g.cgroups == 1, PAIF_ENOSUP, "corr1_wgrad: Cm=%d", Cm);
  hipStream_t st = paif::as_stream(stream);
  switch (k) {
    case 3: hipLaunchKernelGGL(corr1_wgrad_kernel<3>, dim3(g.nblk), dim3(256), 0, st, s, m, workspace, B, H, W, Cm, g.QB, g.rows_per_block); break;
    case 5: hipLaunchKernelGGL(corr1_wgrad_kernel<5>, dim3(g.nblk), dim3(256), 0, st, s, m, workspace, B, H, W, Cm, g.QB, g.rows_per_block); break;
    default:
      paif::set_error("corr1_wgrad: kernel %d not built", k);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("corr1_wgrad");
  hipLaunchKernelGGL(corr1_wgrad_finish_kernel, dim3(1), dim3(256), 0, st, workspace, g.nblk, k * k, Cm, dw);
  PAIF_LAUNCH_CHECK("corr1_wgrad(finish)");
  return 0;
}

int paif_eca_wgrad(const float* pool_partial, const float* dgate_partial, int dgate_blocks_per_img, const float* gate, int k, float* dw,
                   float* workspace, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(pool_partial && dgate_partial && gate && dw && workspace && B > 0 && dgate_blocks_per_img > 0, PAIF_EINVAL,
               "eca_wgrad: bad arguments");
  PAIF_REQUIRE(k >= 1 && k <= 9 && (k & 1), PAIF_ENOSUP, "eca_wgrad: k=%d", k);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(eca_wgrad_kernel, dim3(B), dim3(1024), 0, st, pool_partial, paif_conv2d_blocks(1, H, W), dgate_partial,
                     dgate_blocks_per_img, gate, k, 1.0f / ((float)H * (float)W), workspace);
  PAIF_LAUNCH_CHECK("eca_wgrad");
  hipLaunchKernelGGL(eca_wgrad_finish_kernel, dim3(1), dim3(32), 0, st, workspace, B, k, dw);
  PAIF_LAUNCH_CHECK("eca_wgrad(finish)");
  return 0;
}

int paif_unfold_decomp1x1_wgrad(const float* G, float* dw, paif_stream_t stream) {
  PAIF_REQUIRE(G && dw, PAIF_EINVAL, "unfold_decomp1x1_wgrad: null pointer");
  hipLaunchKernelGGL(unfold_decomp1x1_wgrad_kernel, dim3(16), dim3(256), 0, paif::as_stream(stream), G, dw);
  PAIF_LAUNCH_CHECK("unfold_decomp1x1_wgrad");
  return 0;
}

int paif_unpack_conv_gemm_wgrad(const float* dwp, float* dw, int Cout, int Cin, int k, int Kpad, paif_stream_t stream) {
  PAIF_REQUIRE(dwp && dw && Cout > 0 && Cin > 0 && k > 0 && Kpad >= k * k * Cin, PAIF_EINVAL, "unpack_conv_gemm_wgrad: bad arguments");
  hipLaunchKernelGGL(unpack_conv_gemm_wgrad_kernel, dim3(grid_for((size_t)Cout * Cin * k * k, 256)), dim3(256), 0, paif::as_stream(stream),
                     dwp, dw, Cout, Cin, k * k, Kpad);
  PAIF_LAUNCH_CHECK("unpack_conv_gemm_wgrad");
  return 0;
}

int paif_keep_mask(float* out, int n, unsigned long long seed, unsigned long long offset, float p, paif_stream_t stream) {
  PAIF_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f, PAIF_EINVAL, "keep_mask: bad arguments");
  hipLaunchKernelGGL(keep_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, paif::as_stream(stream), out, n, seed, offset, p);
  PAIF_LAUNCH_CHECK("keep_mask");
  return 0;
}

int paif_rowscale_add_fwd(const float* x, const float* s, const float* res, float* out, int B, size_t rows_per_b, int C, int per_channel,
                          paif_stream_t stream) {
  PAIF_REQUIRE(x && s && out && B > 0 && rows_per_b > 0 && C > 0 && (C & 3) == 0, PAIF_EINVAL, "rowscale_add: bad arguments");
  const size_t per_b4 = rows_per_b * (C / 4), n4 = per_b4 * B;
  hipLaunchKernelGGL(rowscale_add_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, paif::as_stream(stream), x, s, res, out, n4, per_b4, C / 4,
                     per_channel);
  PAIF_LAUNCH_CHECK("rowscale_add");
  return 0;
}

int paif_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* chunk_group, size_t nchunks, int ngroups,
                    const float* group_decay, const float* group_step_size, float one_minus_beta1, float beta2, float one_minus_beta2,
                    float bc2_sqrt, float eps, paif_stream_t stream) {
  PAIF_REQUIRE(p && g && m && v && chunk_group && nchunks > 0 && group_decay && group_step_size, PAIF_EINVAL, "adamw_step: bad arguments");
  PAIF_REQUIRE(ngroups >= 1 && ngroups <= 8, PAIF_ENOSUP, "adamw_step: %d parameter groups (max 8)", ngroups);
  AdamGroups grp;
  for (int i = 0; i < 8; ++i) {   // HOST arrays (the schedule changes them every step)
    grp.decay[i] = i < ngroups ? group_decay[i] : 1.f;
    grp.step_size[i] = i < ngroups ? group_step_size[i] : 0.f;
  }
  const int grid = (int)(nchunks < (size_t)MAXGRID * 4 ? nchunks : (size_t)MAXGRID * 4);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, paif::as_stream(stream), p, g, m, v, chunk_group, nchunks, grp, one_minus_beta1,
                     beta2, one_minus_beta2, bc2_sqrt, eps);
  PAIF_LAUNCH_CHECK("adamw_step");
  return 0;
}

}  // extern "C"
