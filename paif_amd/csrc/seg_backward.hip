// Input-gradient (dgrad) kernels of the segmentation network for the PGD inner loop
// (attack/attack.py:443-501: loss.backward() through WeTr).  Weight gradients are NOT produced: the
// attack only consumes d(loss)/d(input) (the reference additionally accumulates unused .grad on every
// parameter).  GEMM dgrads reuse gemm_mfma.hip with transposed weights.
#include "paif_common.h"

namespace {

constexpr int MAXGRID = 256 * 8;
inline int grid_for(size_t work_items, int per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)MAXGRID) g = MAXGRID;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float gelu_erf(float x) { return paif::gelu_erf_fast(x); }   // paif_common.h: x * Phi(x), one polynomial + v_exp_f32
__device__ __forceinline__ float gelu_grad(float x) { return paif::gelu_grad_fast(x); }

// ---------------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. the input:  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// optional `add`: dx += add (merges the residual branch's gradient)
// ---------------------------------------------------------------------------------------------
template <int G, int QPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, const float* __restrict__ add,
                                                            float* __restrict__ dx, int M, int C, float eps) {
  const int rows_per_block = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const int nq = C / 4;
  const float invC = 1.0f / (float)C;
  for (int row = blockIdx.x * rows_per_block + gr; row < M; row += gridDim.x * rows_per_block) {
    float4 v[QPL], g[QPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      // unconditional loads on a clamped quad index, zeroed afterwards (a load under a branch is waited for at the
      // join); dy is requested here too so that both streams are in flight during the statistics
      const int q = gl + i * G;
      const int qc = min(q, nq - 1);
      v[i] = *reinterpret_cast<const float4*>(x + (size_t)row * C + qc * 4);
      g[i] = *reinterpret_cast<const float4*>(dy + (size_t)row * C + qc * 4);
      if (q >= nq) v[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s * invC;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const int q = gl + i * G;
      if (q < nq) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) ss += __shfl_xor(ss, m);
    const float rstd = 1.0f / sqrtf(ss * invC + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const int q = gl + i * G;
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + min(q, nq - 1) * 4);
      if (q < nq) {
        const float4 d4 = g[i];
        g[i] = make_float4(d4.x * g4.x, d4.y * g4.y, d4.z * g4.z, d4.w * g4.w);
        v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
        sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        sgx += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
      }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
      sg += __shfl_xor(sg, m);
      sgx += __shfl_xor(sgx, m);
    }
    const float mg = sg * invC, mgx = sgx * invC;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const int q = gl + i * G;
      float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (add) a4 = *reinterpret_cast<const float4*>(add + (size_t)row * C + min(q, nq - 1) * 4);   // launch-uniform branch
      if (q < nq) {
        float4 o;
        o.x = rstd * (g[i].x - mg - v[i].x * mgx) + a4.x;
        o.y = rstd * (g[i].y - mg - v[i].y * mgx) + a4.y;
        o.z = rstd * (g[i].z - mg - v[i].z * mgx) + a4.z;
        o.w = rstd * (g[i].w - mg - v[i].w * mgx) + a4.w;
        *reinterpret_cast<float4*>(dx + (size_t)row * C + q * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Mlp.dwconv + GELU backward, two launches of one kernel:
//   MODE 1: g = dy * gelu'(dwconv3(x) + bias)          (recomputes the pre-activation)
//   MODE 2: dx = dwconv3 with the 180-degree-rotated kernel (no bias) applied to g
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void dwconv3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ dy,
                                                          float* __restrict__ y, int B, int H, int W, int C, int pix_per_block) {
  const int nq = C / 4;
  const int qchunks = (nq + 63) / 64;
  const int qc = blockIdx.x % qchunks;
  const int pblk = blockIdx.x / qchunks;
  const int q = qc * 64 + (threadIdx.x & 63);
  const int prow = threadIdx.x >> 6;
  if (q >= nq) return;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(size_t)(4 * q + c) * 9 + (MODE == 2 ? 8 - k : k)];
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1) b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
  const size_t npix = (size_t)B * H * W;
  const size_t pbeg = (size_t)pblk * pix_per_block, pend = min(npix, pbeg + (size_t)pix_per_block);
  for (size_t pix = pbeg + prow; pix < pend; pix += 4) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const float* base = x + (rowid - y0) * W * C + q * 4;
    float4 acc = b4;
    {
      // all 9 taps in flight: unconditional loads on clamped coordinates, padding by select (a load under a branch
      // is waited for at the join, which serialises the taps' latencies)
      float4 tv[9];
#pragma unroll
      for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 3; ++dx_) {
          const int yy = min(max(y0 + dy_ - 1, 0), H - 1), xx = min(max(x0 + dx_ - 1, 0), W - 1);
          tv[dy_ * 3 + dx_] = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * C);
        }
#pragma unroll
      for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 3; ++dx_) {
          const int yy = y0 + dy_ - 1, xx = x0 + dx_ - 1;
          const int k = dy_ * 3 + dx_;
          float4 v = tv[k];
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) v = make_float4(0.f, 0.f, 0.f, 0.f);
          acc.x = fmaf(v.x, wr[0][k], acc.x); acc.y = fmaf(v.y, wr[1][k], acc.y);
          acc.z = fmaf(v.z, wr[2][k], acc.z); acc.w = fmaf(v.w, wr[3][k], acc.w);
        }
    }
    float4 o = acc;
    if (MODE == 1) {
      const float4 d4 = *reinterpret_cast<const float4*>(dy + pix * C + q * 4);
      o = make_float4(d4.x * gelu_grad(acc.x), d4.y * gelu_grad(acc.y), d4.z * gelu_grad(acc.z), d4.w * gelu_grad(acc.w));
    }
    *reinterpret_cast<float4*>(y + pix * C + q * 4) = o;
  }
}

// Row-walking form of the same two passes (C % 256 == 0; seg_kernels.hip dwconv3_bias_gelu_rows_kernel has the forward): a lane
// owns one channel quad of one column and walks BR_ROWS output rows, the 3-row x 3-tap window in registers (static slots) --
// 3 loads per output instead of 9 and no per-pixel 64-bit div / mod.
#ifndef PAIF_DW_ROWS
#define PAIF_DW_ROWS 8
#endif
#ifndef PAIF_DW_AHEAD
#define PAIF_DW_AHEAD 2
#endif
constexpr int BR_ROWS = PAIF_DW_ROWS, BR_COLS = 4, BR_AHEAD = PAIF_DW_AHEAD;
template <int MODE>
__global__ __launch_bounds__(256) void dwconv3_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ dy,
                                                               float* __restrict__ y, int B, int H, int W, int C, int qchunks, int ctiles,
                                                               int strips) {
  int t = blockIdx.x;
  const int qc = t % qchunks; t /= qchunks;
  const int ct = t % ctiles; t /= ctiles;
  const int st = t % strips;
  const int b = t / strips;
  const int q = qc * 64 + (threadIdx.x & 63);
  const int xx0 = ct * BR_COLS + (threadIdx.x >> 6);
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(size_t)(4 * q + c) * 9 + (MODE == 2 ? 8 - k : k)];
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1) b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
  const int ybeg = st * BR_ROWS, yend = min(H, ybeg + BR_ROWS);
  const bool colv = xx0 < W;
  const int xc[3] = {min(max(xx0 - 1, 0), W - 1), min(xx0, W - 1), min(xx0 + 1, W - 1)};
  const bool xok[3] = {xx0 - 1 >= 0, true, xx0 + 1 < W};
  const float* img = x + (size_t)b * H * W * C + q * 4;
  constexpr int NW = 3 + BR_AHEAD;        // slot (row + 1 - ybeg) % NW holds input row `row`
  float4 win[NW][3];
  auto load_row = [&](int yy, float4 (&dst)[3]) {
    const int yc = min(max(yy, 0), H - 1);
    const bool rok = yy >= 0 && yy < H;
#pragma unroll
    for (int dx_ = 0; dx_ < 3; ++dx_) {
      float4 v = *reinterpret_cast<const float4*>(img + ((size_t)yc * W + xc[dx_]) * C);      // unconditional, clamped
      if (!(rok && xok[dx_])) v = make_float4(0.f, 0.f, 0.f, 0.f);
      dst[dx_] = v;
    }
  };
#pragma unroll
  for (int s = 0; s < NW - 1; ++s) load_row(ybeg - 1 + s, win[s]);
  const size_t ooff = ((size_t)b * H * W + min(xx0, W - 1)) * C + q * 4;
#pragma unroll
  for (int i = 0; i < BR_ROWS; ++i) {
    const int yy = ybeg + i;
    if (yy >= yend) break;                                     // workgroup-uniform
    load_row(yy + 1 + BR_AHEAD, win[(i + NW - 1) % NW]);
    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) d4 = *reinterpret_cast<const float4*>(dy + ooff + (size_t)min(yy, H - 1) * W * C);   // unconditional, clamped
    float4 acc = b4;
#pragma unroll
    for (int dy_ = 0; dy_ < 3; ++dy_)
#pragma unroll
      for (int dx_ = 0; dx_ < 3; ++dx_) {
        const float4 v = win[(i + dy_) % NW][dx_];
        const int k = dy_ * 3 + dx_;
        acc.x = fmaf(v.x, wr[0][k], acc.x); acc.y = fmaf(v.y, wr[1][k], acc.y);
        acc.z = fmaf(v.z, wr[2][k], acc.z); acc.w = fmaf(v.w, wr[3][k], acc.w);
      }
    float4 o = acc;
    if (MODE == 1) o = make_float4(d4.x * gelu_grad(acc.x), d4.y * gelu_grad(acc.y), d4.z * gelu_grad(acc.z), d4.w * gelu_grad(acc.w));
    if (colv) *reinterpret_cast<float4*>(y + ooff + (size_t)yy * W * C) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// col2im (adjoint of paif_im2col_fwd), gather form: dx[b,iy,ix,c] = sum over the windows covering (iy,ix)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W,
                                                     int Cin, int KH, int stride, int pad, int OH, int OW, int Kpad) {
  const size_t total = (size_t)B * H * W * Cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % Cin);
    size_t t = i / Cin;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H);
    const int b = (int)(t / H);
    float s = 0.f;
    for (int ky = 0; ky < KH; ++ky) {
      const int ny = iy + pad - ky;
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= OH) continue;
      for (int kx = 0; kx < KH; ++kx) {
        const int nx = ix + pad - kx;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= OW) continue;
        s += dcol[(((size_t)b * OH + oy) * OW + ox) * Kpad + (ky * KH + kx) * Cin + c];
      }
    }
    dx[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint of paif_resize_bilinear_into_fwd (gather form): dx[b,iy,ix,c] = sum_o w(o -> i) * dout[b,o,coff+c]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(float scale, int o, int isz, int& i0, int& i1, float& l1) {
  float f = scale * ((float)o + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  i1 = i0 + (i0 < isz - 1 ? 1 : 0);
  l1 = f - (float)i0;
}

__global__ __launch_bounds__(256) void resize_adjoint_kernel(const float* __restrict__ dout, float* __restrict__ dx, int B, int IH,
                                                             int IW, int C, int OH, int OW, int ldo, int coff) {
  const int cq = C / 4;
  const size_t total = (size_t)B * IH * IW * cq;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const int ry = (OH + IH - 1) / IH + 1, rx = (OW + IW - 1) / IW + 1;  // footprint radius in output pixels
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq);
    size_t t = i / cq;
    const int ix = (int)(t % IW); t /= IW;
    const int iy = (int)(t % IH);
    const int b = (int)(t / IH);
    const int oyc = (int)(((float)iy + 0.5f) / sy), oxc = (int)(((float)ix + 0.5f) / sx);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = max(0, oyc - 2 * ry); oy <= min(OH - 1, oyc + 2 * ry); ++oy) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = max(0, oxc - 2 * rx); ox <= min(OW - 1, oxc + 2 * rx); ++ox) {
        int x0, x1; float lx;
        src_index(sx, ox, IW, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx == 0.f) continue;
        const float4 d = *reinterpret_cast<const float4*>(dout + (((size_t)b * OH + oy) * OW + ox) * ldo + coff + c4 * 4);
        const float wgt = wy * wx;
        acc.x = fmaf(wgt, d.x, acc.x); acc.y = fmaf(wgt, d.y, acc.y); acc.z = fmaf(wgt, d.z, acc.z); acc.w = fmaf(wgt, d.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Seg_loss (attack/attack.py:103-114,446-448): bilinear upsample of the logits to the label size +
// CrossEntropyLoss(ignore_index): forward partial sums and dlogits at full resolution (C padded to CP)
// ---------------------------------------------------------------------------------------------
constexpr int CE_MAXC = 32;

template <bool BWD>
__global__ __launch_bounds__(256) void upsample_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                          float* __restrict__ partial, float* __restrict__ dfull,
                                                          const float* __restrict__ scale_ptr, int B, int IH, int IW, int C, int OH,
                                                          int OW, int ignore, int CP) {
  const size_t total = (size_t)B * OH * OW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  float nll_sum = 0.f, cnt = 0.f;
  float gscale = 0.f;
  if (BWD) gscale = *scale_ptr;  // dloss / count
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    size_t t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    const long long lab = label[i];
    const bool valid = lab != (long long)ignore;
    if (!valid) {
      if (BWD) for (int c = 0; c < CP; ++c) dfull[i * CP + c] = 0.f;
      continue;
    }
    int y0, y1, x0, x1; float ly, lx;
    src_index(sy, oy, IH, y0, y1, ly);
    src_index(sx, ox, IW, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = logits + (((size_t)b * IH + y0) * IW + x0) * C;
    const float* p01 = logits + (((size_t)b * IH + y0) * IW + x1) * C;
    const float* p10 = logits + (((size_t)b * IH + y1) * IW + x0) * C;
    const float* p11 = logits + (((size_t)b * IH + y1) * IW + x1) * C;
    // the interpolated logits are recomputed per pass instead of being kept in a runtime-indexed array
    // (which would live in scratch memory)
#define PAIF_INTERP(c) (hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]))
    float mx = -INFINITY;
#pragma unroll 1
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, PAIF_INTERP(c));
    float se = 0.f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) se += expf(PAIF_INTERP(c) - mx);
    const float lse = mx + logf(se);
    if (!BWD) {
      nll_sum += lse - PAIF_INTERP((int)lab);
      cnt += 1.f;
    } else {
#pragma unroll 1
      for (int c = 0; c < CP; ++c)
        dfull[i * CP + c] = (c < C) ? gscale * (expf(PAIF_INTERP(c) - lse) - (c == (int)lab ? 1.f : 0.f)) : 0.f;
    }
#undef PAIF_INTERP
  }
  if (!BWD) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      nll_sum += __shfl_xor(nll_sum, m);
      cnt += __shfl_xor(cnt, m);
    }
    __shared__ float s1[4], s2[4];
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = nll_sum; s2[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      partial[blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
      partial[gridDim.x + blockIdx.x] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    }
  }
}

// loss = sum(nll)/sum(count) from the per-block partials, fixed order; also writes 1/count * upstream
__global__ void ce_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  // single wave: lane-strided sums in fixed order, then a fixed shuffle tree -> deterministic
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) { a += partial[i]; c += partial[nblk + i]; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); c += __shfl_xor(c, m); }
  if (threadIdx.x == 0) { out[0] = a / c; out[1] = c; }
}

// w [N][K] -> wt [K][Npad] (zero padded): the dgrad GEMM's weight operand
__global__ void transpose_pad_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K, int Npad) {
  const size_t total = (size_t)K * Npad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % Npad), k = (int)(i / Npad);
    wt[i] = n < N ? w[(size_t)n * K + k] : 0.f;
  }
}

}  // namespace

extern "C" {

int paif_layernorm_bwd_input(const float* x, const float* gamma, const float* dy, const float* add, float* dx, int M, int C,
                             float eps, paif_stream_t stream) {
  PAIF_REQUIRE(x && gamma && dy && dx && M > 0, PAIF_EINVAL, "layernorm_bwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C % 4 == 0 && C <= 2048, PAIF_ENOSUP, "layernorm_bwd: C=%d", C);
  hipStream_t st = paif::as_stream(stream);
  const int nq = C / 4;
#define LNB_LAUNCH(G, QPL) \
  hipLaunchKernelGGL((layernorm_bwd_kernel<G, QPL>), dim3(grid_for((size_t)M, 256 / G)), dim3(256), 0, st, x, gamma, dy, add, dx, M, C, eps)
  if (nq <= 16) LNB_LAUNCH(16, 1);
  else if (nq <= 32) LNB_LAUNCH(32, 1);
  else if (nq <= 64) LNB_LAUNCH(64, 1);
  else if (nq <= 128) LNB_LAUNCH(64, 2);
  else if (nq <= 256) LNB_LAUNCH(64, 4);
  else LNB_LAUNCH(64, 8);
#undef LNB_LAUNCH
  PAIF_LAUNCH_CHECK("layernorm_bwd");
  return 0;
}

int paif_dwconv3_bias_gelu_bwd_input(const float* x, const float* w, const float* bias, const float* dy, float* tmp, float* dx,
                                     int B, int H, int W, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && w && bias && dy && tmp && dx && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv3_bias_gelu_bwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C % 4 == 0, PAIF_ENOSUP, "dwconv3_bias_gelu_bwd: C=%d", C);
  const int qchunks = (C / 4 + 63) / 64;
  const size_t npix = (size_t)B * H * W;
  size_t pblocks = (npix + 63) / 64;
  const size_t cap = (size_t)(MAXGRID * 2) / qchunks + 1;
  if (pblocks > cap) pblocks = cap;
  const int pix_per_block = (int)((npix + pblocks - 1) / pblocks);
  pblocks = (npix + pix_per_block - 1) / pix_per_block;
  hipStream_t st = paif::as_stream(stream);
  if (C % 256 == 0) {
    const int ctiles = (W + BR_COLS - 1) / BR_COLS, strips = (H + BR_ROWS - 1) / BR_ROWS;
    const size_t nblk = (size_t)B * strips * ctiles * qchunks;
    PAIF_REQUIRE(nblk < ((size_t)1 << 31), PAIF_EINVAL, "dwconv3_bias_gelu_bwd: %dx%dx%dx%d is too large for one launch", B, H, W, C);
    hipLaunchKernelGGL(dwconv3_bwd_rows_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, x, w, bias, dy, tmp, B, H, W, C, qchunks, ctiles, strips);
    PAIF_LAUNCH_CHECK("dwconv3_bias_gelu_bwd(1)");
    hipLaunchKernelGGL(dwconv3_bwd_rows_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, st, tmp, w, bias, dy, dx, B, H, W, C, qchunks, ctiles, strips);
    PAIF_LAUNCH_CHECK("dwconv3_bias_gelu_bwd(2)");
    return 0;
  }
  hipLaunchKernelGGL(dwconv3_bwd_kernel<1>, dim3((unsigned)(pblocks * qchunks)), dim3(256), 0, st, x, w, bias, dy, tmp, B, H, W, C,
                     pix_per_block);
  PAIF_LAUNCH_CHECK("dwconv3_bias_gelu_bwd(1)");
  hipLaunchKernelGGL(dwconv3_bwd_kernel<2>, dim3((unsigned)(pblocks * qchunks)), dim3(256), 0, st, tmp, w, bias, dy, dx, B, H, W, C,
                     pix_per_block);
  PAIF_LAUNCH_CHECK("dwconv3_bias_gelu_bwd(2)");
  return 0;
}

int paif_col2im_fwd(const float* dcol, float* dx, int B, int H, int W, int Cin, int k, int stride, int pad, int Kpad,
                    paif_stream_t stream) {
  PAIF_REQUIRE(dcol && dx && B > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0, PAIF_EINVAL, "col2im: bad arguments");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  PAIF_REQUIRE(OH > 0 && OW > 0 && Kpad >= k * k * Cin, PAIF_EINVAL, "col2im: bad geometry");
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((size_t)B * H * W * Cin, 256)), dim3(256), 0, paif::as_stream(stream), dcol, dx, B,
                     H, W, Cin, k, stride, pad, OH, OW, Kpad);
  PAIF_LAUNCH_CHECK("col2im");
  return 0;
}

int paif_resize_bilinear_adjoint_fwd(const float* dout, float* dx, int B, int IH, int IW, int C, int OH, int OW, int ldo, int coff,
                                     paif_stream_t stream) {
  PAIF_REQUIRE(dout && dx && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PAIF_EINVAL, "resize_bilinear_adjoint: bad arguments");
  PAIF_REQUIRE(C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo, PAIF_EINVAL, "resize_bilinear_adjoint: C=%d ldo=%d coff=%d",
               C, ldo, coff);
  hipLaunchKernelGGL(resize_adjoint_kernel, dim3(grid_for((size_t)B * IH * IW * (C / 4), 256)), dim3(256), 0, paif::as_stream(stream),
                     dout, dx, B, IH, IW, C, OH, OW, ldo, coff);
  PAIF_LAUNCH_CHECK("resize_bilinear_adjoint");
  return 0;
}

int paif_upsample_ce_blocks(int B, int OH, int OW) { return grid_for((size_t)B * OH * OW, 256); }

int paif_upsample_ce_fwd(const float* logits, const long long* label, float* partial, float* loss_count, int B, int IH, int IW, int C,
                         int OH, int OW, int ignore_index, paif_stream_t stream) {
  PAIF_REQUIRE(logits && label && partial && loss_count && B > 0, PAIF_EINVAL, "upsample_ce_fwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C <= CE_MAXC, PAIF_ENOSUP, "upsample_ce_fwd: C=%d (max %d)", C, CE_MAXC);
  const int nblk = paif_upsample_ce_blocks(B, OH, OW);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(upsample_ce_kernel<false>, dim3(nblk), dim3(256), 0, st, logits, label, partial, (float*)nullptr,
                     (const float*)nullptr, B, IH, IW, C, OH, OW, ignore_index, 0);
  PAIF_LAUNCH_CHECK("upsample_ce_fwd");
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, partial, nblk, loss_count);
  PAIF_LAUNCH_CHECK("ce_finish");
  return 0;
}

int paif_upsample_ce_bwd(const float* logits, const long long* label, const float* gscale, float* dfull, int B, int IH, int IW, int C,
                         int OH, int OW, int ignore_index, int CP, paif_stream_t stream) {
  PAIF_REQUIRE(logits && label && gscale && dfull && B > 0, PAIF_EINVAL, "upsample_ce_bwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C <= CE_MAXC && CP >= C && CP % 4 == 0, PAIF_ENOSUP, "upsample_ce_bwd: C=%d CP=%d", C, CP);
  hipLaunchKernelGGL(upsample_ce_kernel<true>, dim3(paif_upsample_ce_blocks(B, OH, OW)), dim3(256), 0, paif::as_stream(stream), logits,
                     label, (float*)nullptr, dfull, gscale, B, IH, IW, C, OH, OW, ignore_index, CP);
  PAIF_LAUNCH_CHECK("upsample_ce_bwd");
  return 0;
}

int paif_transpose_pad_fwd(const float* w, float* wt, int N, int K, int Npad, paif_stream_t stream) {
  PAIF_REQUIRE(w && wt && N > 0 && K > 0 && Npad >= N, PAIF_EINVAL, "transpose_pad: bad arguments");
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(grid_for((size_t)K * Npad, 256)), dim3(256), 0, paif::as_stream(stream), w, wt, N, K,
                     Npad);
  PAIF_LAUNCH_CHECK("transpose_pad");
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// evaluation harness kernels (test_original.py:180,207-211): bilinear upsample + argmax, confusion matrix
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ logits, long long* __restrict__ pred, int B,
                                                              int IH, int IW, int C, int OH, int OW) {
  const size_t total = (size_t)B * OH * OW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    size_t t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    int y0, y1, x0, x1; float ly, lx;
    src_index(sy, oy, IH, y0, y1, ly);
    src_index(sx, ox, IW, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = logits + (((size_t)b * IH + y0) * IW + x0) * C;
    const float* p01 = logits + (((size_t)b * IH + y0) * IW + x1) * C;
    const float* p10 = logits + (((size_t)b * IH + y1) * IW + x0) * C;
    const float* p11 = logits + (((size_t)b * IH + y1) * IW + x1) * C;
    float best = -INFINITY; int bi = 0;
    for (int c = 0; c < C; ++c) {
      const float v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
      if (v > best) { best = v; bi = c; }   // first maximum, like torch.argmax
    }
    pred[i] = bi;
  }
}

// rows = true class, cols = predicted; pairs with a class outside [0, n) are dropped (sklearn labels=0..n-1)
__global__ void confusion_kernel(const long long* __restrict__ label, const long long* __restrict__ pred,
                                 unsigned long long* __restrict__ conf, size_t n, int ncls) {
  __shared__ unsigned int local[32 * 32];
  for (int i = threadIdx.x; i < ncls * ncls; i += blockDim.x) local[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long long l = label[i], p = pred[i];
    if (l >= 0 && l < ncls && p >= 0 && p < ncls) atomicAdd(&local[l * ncls + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ncls * ncls; i += blockDim.x)
    if (local[i]) atomicAdd(&conf[i], (unsigned long long)local[i]);   // integer atomics: order-independent, exact
}
}  // namespace

extern "C" int paif_upsample_argmax_fwd(const float* logits, long long* pred, int B, int IH, int IW, int C, int OH, int OW,
                                        paif_stream_t stream) {
  PAIF_REQUIRE(logits && pred && B > 0 && IH > 0 && IW > 0 && C > 0 && OH > 0 && OW > 0, PAIF_EINVAL, "upsample_argmax: bad arguments");
  hipLaunchKernelGGL(upsample_argmax_kernel, dim3(grid_for((size_t)B * OH * OW, 256)), dim3(256), 0, paif::as_stream(stream), logits, pred,
                     B, IH, IW, C, OH, OW);
  PAIF_LAUNCH_CHECK("upsample_argmax");
  return 0;
}

extern "C" int paif_confusion_matrix_accum(const long long* label, const long long* pred, unsigned long long* conf, size_t n, int ncls,
                                           paif_stream_t stream) {
  PAIF_REQUIRE(label && pred && conf && ncls > 0 && ncls <= 32, PAIF_EINVAL, "confusion_matrix: bad arguments (ncls <= 32)");
  if (n == 0) return 0;
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(n, 256)), dim3(256), 0, paif::as_stream(stream), label, pred, conf, n, ncls);
  PAIF_LAUNCH_CHECK("confusion_matrix");
  return 0;
}
