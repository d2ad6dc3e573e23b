// HBM-bound kernels of the MixTransformer encoder / SegFormer head (gfx950).  Token tensors [B,N,C] with
// N = H*W row-major ARE NHWC images, so the whole segmentation net runs channels-last with no transposes.
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

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C % 4 == 0, C <= 2048).  G lanes per row (G = 16/32/64), rows held in
// registers (two-pass mean / variance), shuffle reductions inside the lane group.
// ---------------------------------------------------------------------------------------------
template <int G, int QPL>  // lanes per row, float4 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, int M, int C,
                                                        float eps) {
  const int rows_per_block = 256 / G;
  const int gl = threadIdx.x % G, gr = threadIdx.x / G;
  const int nq = C / 4;
  for (int row = blockIdx.x * rows_per_block + gr; row < M; row += gridDim.x * rows_per_block) {
    float4 v[QPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      // unconditional load on a clamped quad index, zeroed afterwards (a load under a branch is waited for at the join)
      const int q = gl + i * G;
      v[i] = *reinterpret_cast<const float4*>(x + (size_t)row * C + min(q, nq - 1) * 4);
      if (q >= nq) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const int q = gl + i * G;
      if (q < nq) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        ss += (a * a + b * b) + (c * c + d * d);
      }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) ss += __shfl_xor(ss, m);
    const float rstd = 1.0f / sqrtf(ss / (float)C + eps);
#pragma unroll
    for (int i = 0; i < QPL; ++i) {
      const int q = gl + i * G;
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + min(q, nq - 1) * 4);
      const float4 b4 = *reinterpret_cast<const float4*>(beta + min(q, nq - 1) * 4);
      if (q < nq) {
        float4 o;
        o.x = (v[i].x - mean) * rstd * g4.x + b4.x;
        o.y = (v[i].y - mean) * rstd * g4.y + b4.y;
        o.z = (v[i].z - mean) * rstd * g4.z + b4.z;
        o.w = (v[i].w - mean) * rstd * g4.w + b4.w;
        *reinterpret_cast<float4*>(y + (size_t)row * C + q * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// im2col for strided convs on NHWC input: out[b, oy, ox, (ky*KW+kx)*Cin + c] (K padded to Kpad with zeros)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W,
                                                     int Cin, int KH, int stride, int pad, int OH, int OW, int Kpad) {
  const size_t total = (size_t)B * OH * OW * Kpad;
  const int K = KH * KH * Cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const size_t tok = i / Kpad;
    float v = 0.f;
    if (k < K) {
      const int c = k % Cin, tap = k / Cin;
      const int ky = tap / KH, kx = tap - ky * KH;
      const int ox = (int)(tok % OW);
      const size_t t2 = tok / OW;
      const int oy = (int)(t2 % OH);
      const int b = (int)(t2 / OH);
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * Cin + c];
    }
    col[i] = v;
  }
}

// float4 variant (Cin % 4 == 0, Kpad == K): each lane moves 4 channels of one tap
__global__ __launch_bounds__(256) void im2col4_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W,
                                                      int Cin, int KH, int stride, int pad, int OH, int OW) {
  const int cq = Cin / 4;
  const int kq = KH * KH * cq;
  const size_t total = (size_t)B * OH * OW * kq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k4 = (int)(i % kq);
    const size_t tok = i / kq;
    const int c4 = k4 % cq, tap = k4 / cq;
    const int ky = tap / KH, kx = tap - ky * KH;
    const int ox = (int)(tok % OW);
    const size_t t2 = tok / OW;
    const int oy = (int)(t2 % OH);
    const int b = (int)(t2 / OH);
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + iy) * W + ix) * Cin + c4 * 4);
    *reinterpret_cast<float4*>(col + i * 4) = v;
  }
}

// conv weight [Cout][Cin][KH][KW] -> GEMM weight [Cout][Kpad], k = (ky*KW+kx)*Cin + c, zero padded
__global__ void pack_conv_gemm_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH,
                                             int Kpad) {
  const int K = KH * KH * Cin;
  const size_t total = (size_t)Cout * Kpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad), n = (int)(i / Kpad);
    float v = 0.f;
    if (k < K) {
      const int c = k % Cin, tap = k / Cin;
      v = w[((size_t)n * Cin + c) * KH * KH + tap];
    }
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Mlp.dwconv (3x3 depthwise, bias) + GELU on NHWC tokens; thread keeps one channel quad (weights in
// registers) and walks pixels.  core/mix_transformer.py:376-387, :49 (act)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv3_bias_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int B,
                                                                int H, int W, int C, int pix_per_block) {
  const int nq = C / 4;
  const int qchunks = (nq + 63) / 64;
  const int qc = blockIdx.x % qchunks;
  const int pblk = blockIdx.x / qchunks;
  const int q = qc * 64 + (threadIdx.x & 63);
  const int prow = threadIdx.x >> 6;  // 4 pixels in flight per block
  if (q >= nq) return;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(size_t)(4 * q + c) * 9 + k];
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
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
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int yy = min(max(y0 + dy - 1, 0), H - 1), xx = min(max(x0 + dx - 1, 0), W - 1);
          tv[dy * 3 + dx] = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * C);
        }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int yy = y0 + dy - 1, xx = x0 + dx - 1;
          const int k = dy * 3 + dx;
          float4 v = tv[k];
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) v = make_float4(0.f, 0.f, 0.f, 0.f);
          acc.x = fmaf(v.x, wr[0][k], acc.x); acc.y = fmaf(v.y, wr[1][k], acc.y);
          acc.z = fmaf(v.z, wr[2][k], acc.z); acc.w = fmaf(v.w, wr[3][k], acc.w);
        }
    }
    *reinterpret_cast<float4*>(y + pix * C + q * 4) = make_float4(gelu_erf(acc.x), gelu_erf(acc.y), gelu_erf(acc.z), gelu_erf(acc.w));
  }
}

// Row-walking form (C % 256 == 0: every hidden width of MiT-b0..b5): a lane owns ONE channel quad of ONE column and walks DG_ROWS
// output rows with the 3-row x 3-tap window in registers (static slots: the row loop is fully unrolled) -- 3 loads per output
// instead of 9, each a 16-byte piece of a 1 KB contiguous run per wave, no per-pixel 64-bit div / mod.  The per-pixel form above
// ran at ~1.2 TB/s over its stream (100 us per call averaged over a mit_b3 forward at B=8 480x640).
#ifndef PAIF_DW_ROWS
#define PAIF_DW_ROWS 8
#endif
#ifndef PAIF_DW_AHEAD
#define PAIF_DW_AHEAD 2
#endif
constexpr int DG_ROWS = PAIF_DW_ROWS, DG_COLS = 4, DG_AHEAD = PAIF_DW_AHEAD;
__global__ __launch_bounds__(256) void dwconv3_bias_gelu_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ y, int B,
                                                                     int H, int W, int C, int qchunks, int ctiles, int strips) {
  int t = blockIdx.x;
  const int qc = t % qchunks; t /= qchunks;
  const int ct = t % ctiles; t /= ctiles;
  const int st = t % strips;
  const int b = t / strips;
  const int q = qc * 64 + (threadIdx.x & 63);
  const int xx0 = ct * DG_COLS + (threadIdx.x >> 6);
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(size_t)(4 * q + c) * 9 + k];
  const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * q);
  const int ybeg = st * DG_ROWS, yend = min(H, ybeg + DG_ROWS);
  const bool colv = xx0 < W;
  const int xc[3] = {min(max(xx0 - 1, 0), W - 1), min(xx0, W - 1), min(xx0 + 1, W - 1)};
  const bool xok[3] = {xx0 - 1 >= 0, true, xx0 + 1 < W};
  const float* img = x + (size_t)b * H * W * C + q * 4;
  constexpr int NW = 3 + DG_AHEAD;        // window slots: slot (row + 1 - ybeg) % NW holds input row `row`
  float4 win[NW][3];
  auto load_row = [&](int yy, float4 (&dst)[3]) {
    const int yc = min(max(yy, 0), H - 1);
    const bool rok = yy >= 0 && yy < H;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float4 v = *reinterpret_cast<const float4*>(img + ((size_t)yc * W + xc[dx]) * C);      // unconditional, clamped
      if (!(rok && xok[dx])) v = make_float4(0.f, 0.f, 0.f, 0.f);
      dst[dx] = v;
    }
  };
#pragma unroll
  for (int s = 0; s < NW - 1; ++s) load_row(ybeg - 1 + s, win[s]);
  float* orow = y + ((size_t)b * H * W + xx0) * C + q * 4;
#pragma unroll
  for (int i = 0; i < DG_ROWS; ++i) {
    const int yy = ybeg + i;
    if (yy >= yend) break;                                     // workgroup-uniform
    load_row(yy + 1 + DG_AHEAD, win[(i + NW - 1) % NW]);       // the slot of row yy - 2, free since the previous iteration
    float4 acc = b4;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 v = win[(i + dy) % NW][dx];
        const int k = dy * 3 + dx;
        acc.x = fmaf(v.x, wr[0][k], acc.x); acc.y = fmaf(v.y, wr[1][k], acc.y);
        acc.z = fmaf(v.z, wr[2][k], acc.z); acc.w = fmaf(v.w, wr[3][k], acc.w);
      }
    if (colv)
      *reinterpret_cast<float4*>(orow + (size_t)yy * W * C) = make_float4(gelu_erf(acc.x), gelu_erf(acc.y), gelu_erf(acc.z), gelu_erf(acc.w));
  }
}

// ---------------------------------------------------------------------------------------------
// SegFormerHead with linear_fuse folded in front of the upsampling (core/segformer_head.py:63-80): the 1x1 conv over
// cat[up(y4), up(y3), up(y2), y1] is linear, and bilinear interpolation is linear and per channel, so
//   conv(cat) = up(W4 y4) + up(W3 y3) + up(W2 y2) + W1 y1         (W_i = the i-th 256-column block of the fuse weight)
// z_i = W_i y_i are formed at each stage's own resolution (1.2 instead of 10.1 GFLOP per 480x640 pair); this kernel adds
// the four maps at 1/4 resolution and applies the folded BatchNorm + ReLU:  out = relu((z1 + up z2 + up z3 + up z4) * scale + shift)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 bilerp4(const float* __restrict__ xb, int IH, int IW, int C, float sy, float sx, int oy, int ox) {
  float fy = sy * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
  float fx = sx * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float4 v00 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * IW + x0) * C);
  const float4 v01 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * IW + x1) * C);
  const float4 v10 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * IW + x0) * C);
  const float4 v11 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * IW + x1) * C);
  float4 o;
  o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
  o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
  o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
  o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  return o;
}

struct HeadSumArgs {
  const float* z[4];     // z[0] at the output size, z[1..3] coarser
  int ih[4], iw[4];
};

__global__ __launch_bounds__(256) void head_sum_kernel(HeadSumArgs a, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ out, int B, int OH, int OW, int C) {
  const int cq = C / 4;
  const size_t total = (size_t)B * OH * OW * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq);
    const size_t tok = i / cq;
    const int ox = (int)(tok % OW);
    const size_t t2 = tok / OW;
    const int oy = (int)(t2 % OH);
    const int b = (int)(t2 / OH);
    float4 acc = *reinterpret_cast<const float4*>(a.z[0] + tok * C + c4 * 4);
    // summed in the concat order of the reference's K axis: _c4, _c3, _c2, then _c1 (already in acc)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 3; k >= 1; --k) {
      const float4 v = bilerp4(a.z[k] + (size_t)b * a.ih[k] * a.iw[k] * C + c4 * 4, a.ih[k], a.iw[k], C, (float)a.ih[k] / (float)OH,
                               (float)a.iw[k] / (float)OW, oy, ox);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4), sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
    acc.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f); acc.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f);
    acc.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f); acc.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f);
    *reinterpret_cast<float4*>(out + tok * C + c4 * 4) = acc;
  }
}

// d_pre = d_x * (x > 0) * scale[c]: backward of the folded BatchNorm + ReLU on the head's 1/4-resolution map
__global__ __launch_bounds__(256) void relu_mask_scale_kernel(const float* __restrict__ dx, const float* __restrict__ x,
                                                              const float* __restrict__ scale, float* __restrict__ out, size_t n4, int Q) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 d = reinterpret_cast<const float4*>(dx)[i], v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[i % Q];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x > 0.f ? d.x * sc.x : 0.f, v.y > 0.f ? d.y * sc.y : 0.f,
                                                    v.z > 0.f ? d.z * sc.z : 0.f, v.w > 0.f ? d.w * sc.w : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// bilinear resize (align_corners=False) of an NHWC map into a channel slice of a wider NHWC tensor:
// SegFormerHead's F.interpolate + torch.cat (core/segformer_head.py:66-77).  Identity size = plain copy.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_into_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int IH,
                                                          int IW, int C, int OH, int OW, int ldo, int coff) {
  const int cq = C / 4;
  const size_t total = (size_t)B * OH * OW * cq;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq);
    const size_t tok = i / cq;
    const int ox = (int)(tok % OW);
    const size_t t2 = tok / OW;
    const int oy = (int)(t2 % OH);
    const int b = (int)(t2 / OH);
    // ATen area_pixel_compute_source_index (align_corners=False): max(0, scale*(dst+0.5)-0.5)
    float fy = sy * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = sx * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* xb = x + (size_t)b * IH * IW * C + c4 * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * IW + x0) * C);
    const float4 v01 = *reinterpret_cast<const float4*>(xb + ((size_t)y0 * IW + x1) * C);
    const float4 v10 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * IW + x0) * C);
    const float4 v11 = *reinterpret_cast<const float4*>(xb + ((size_t)y1 * IW + x1) * C);
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<float4*>(out + tok * ldo + coff + c4 * 4) = o;
  }
}

// NHWC [B,HW,C] <-> NCHW [B,C,HW] (tiny tensors only: 3-channel input, 9-channel logits)
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, size_t HW, int C) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i % HW;
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t b = t / C;
    y[i] = x[(b * HW + px) * C + c];
  }
}
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, size_t HW, int C) {
  const size_t total = (size_t)B * HW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t t = i / C;
    const size_t px = t % HW, b = t / HW;
    y[i] = x[(b * C + c) * HW + px];
  }
}

}  // namespace

extern "C" {

int paif_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int C, float eps,
                       paif_stream_t stream) {
  PAIF_REQUIRE(x && gamma && beta && y && M > 0, PAIF_EINVAL, "layernorm: bad arguments");
  PAIF_REQUIRE(C > 0 && C % 4 == 0 && C <= 2048, PAIF_ENOSUP, "layernorm: C=%d (need C %% 4 == 0, C <= 2048)", C);
  hipStream_t st = paif::as_stream(stream);
  const int nq = C / 4;
#define LN_LAUNCH(G, QPL) \
  hipLaunchKernelGGL((layernorm_kernel<G, QPL>), dim3(grid_for((size_t)M, 256 / G)), dim3(256), 0, st, x, gamma, beta, y, M, C, eps)
  if (nq <= 16) LN_LAUNCH(16, 1);
  else if (nq <= 32) LN_LAUNCH(32, 1);
  else if (nq <= 64) LN_LAUNCH(64, 1);
  else if (nq <= 128) LN_LAUNCH(64, 2);
  else if (nq <= 256) LN_LAUNCH(64, 4);
  else LN_LAUNCH(64, 8);
#undef LN_LAUNCH
  PAIF_LAUNCH_CHECK("layernorm");
  return 0;
}

int paif_im2col_fwd(const float* x, float* col, int B, int H, int W, int Cin, int k, int stride, int pad, int Kpad,
                    paif_stream_t stream) {
  PAIF_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0, PAIF_EINVAL, "im2col: bad arguments");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  PAIF_REQUIRE(OH > 0 && OW > 0 && Kpad >= k * k * Cin, PAIF_EINVAL, "im2col: bad geometry");
  hipStream_t st = paif::as_stream(stream);
  if (Cin % 4 == 0 && Kpad == k * k * Cin) {
    const size_t total = (size_t)B * OH * OW * (k * k * Cin / 4);
    hipLaunchKernelGGL(im2col4_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, x, col, B, H, W, Cin, k, stride, pad, OH, OW);
  } else {
    const size_t total = (size_t)B * OH * OW * Kpad;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, x, col, B, H, W, Cin, k, stride, pad, OH, OW,
                       Kpad);
  }
  PAIF_LAUNCH_CHECK("im2col");
  return 0;
}

int paif_pack_conv_gemm_weight(const float* w, float* out, int Cout, int Cin, int k, int Kpad, paif_stream_t stream) {
  PAIF_REQUIRE(w && out && Cout > 0 && Cin > 0 && k > 0 && Kpad >= k * k * Cin, PAIF_EINVAL, "pack_conv_gemm_weight: bad arguments");
  hipLaunchKernelGGL(pack_conv_gemm_weight_kernel, dim3(grid_for((size_t)Cout * Kpad, 256)), dim3(256), 0, paif::as_stream(stream),
                     w, out, Cout, Cin, k, Kpad);
  PAIF_LAUNCH_CHECK("pack_conv_gemm_weight");
  return 0;
}

int paif_dwconv3_bias_gelu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                               paif_stream_t stream) {
  PAIF_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv3_bias_gelu: bad arguments");
  PAIF_REQUIRE(C > 0 && C % 4 == 0, PAIF_ENOSUP, "dwconv3_bias_gelu: C=%d", C);
  const int qchunks = (C / 4 + 63) / 64;
  const size_t npix = (size_t)B * H * W;
  if (C % 256 == 0) {
    const int ctiles = (W + DG_COLS - 1) / DG_COLS, strips = (H + DG_ROWS - 1) / DG_ROWS;
    const size_t nblk = (size_t)B * strips * ctiles * qchunks;
    PAIF_REQUIRE(nblk < ((size_t)1 << 31), PAIF_EINVAL, "dwconv3_bias_gelu: %dx%dx%dx%d is too large for one launch", B, H, W, C);
    hipLaunchKernelGGL(dwconv3_bias_gelu_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, paif::as_stream(stream), x, w, bias, y, B, H, W, C,
                       qchunks, ctiles, strips);
    PAIF_LAUNCH_CHECK("dwconv3_bias_gelu");
    return 0;
  }
  // enough blocks to fill the chip, >= 64 pixels per block so the 40 weight registers are amortised
  size_t pblocks = (npix + 63) / 64;
  const size_t cap = (size_t)(MAXGRID * 2) / qchunks + 1;
  if (pblocks > cap) pblocks = cap;
  const int pix_per_block = (int)((npix + pblocks - 1) / pblocks);
  pblocks = (npix + pix_per_block - 1) / pix_per_block;
  hipLaunchKernelGGL(dwconv3_bias_gelu_kernel, dim3((unsigned)(pblocks * qchunks)), dim3(256), 0, paif::as_stream(stream), x, w, bias,
                     y, B, H, W, C, pix_per_block);
  PAIF_LAUNCH_CHECK("dwconv3_bias_gelu");
  return 0;
}

int paif_resize_bilinear_into_fwd(const float* x, float* out, int B, int IH, int IW, int C, int OH, int OW, int ldo, int coff,
                                  paif_stream_t stream) {
  PAIF_REQUIRE(x && out && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PAIF_EINVAL, "resize_bilinear_into: bad arguments");
  PAIF_REQUIRE(C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo, PAIF_EINVAL, "resize_bilinear_into: C=%d ldo=%d coff=%d", C,
               ldo, coff);
  hipLaunchKernelGGL(resize_into_kernel, dim3(grid_for((size_t)B * OH * OW * (C / 4), 256)), dim3(256), 0, paif::as_stream(stream), x,
                     out, B, IH, IW, C, OH, OW, ldo, coff);
  PAIF_LAUNCH_CHECK("resize_bilinear_into");
  return 0;
}

int paif_head_sum_fwd(const float* z1, const float* z2, const float* z3, const float* z4, const int* hw, const float* scale,
                      const float* shift, float* out, int B, int C, paif_stream_t stream) {
  PAIF_REQUIRE(z1 && z2 && z3 && z4 && hw && scale && shift && out && B > 0 && C > 0 && (C & 3) == 0, PAIF_EINVAL, "head_sum: bad arguments");
  HeadSumArgs a;
  const float* zs[4] = {z1, z2, z3, z4};
  for (int k = 0; k < 4; ++k) { a.z[k] = zs[k]; a.ih[k] = hw[2 * k]; a.iw[k] = hw[2 * k + 1]; }
  hipLaunchKernelGGL(head_sum_kernel, dim3(grid_for((size_t)B * a.ih[0] * a.iw[0] * (C / 4), 256)), dim3(256), 0, paif::as_stream(stream), a,
                     scale, shift, out, B, a.ih[0], a.iw[0], C);
  PAIF_LAUNCH_CHECK("head_sum");
  return 0;
}

int paif_relu_mask_scale_fwd(const float* dx, const float* x, const float* scale, float* out, size_t M, int C, paif_stream_t stream) {
  PAIF_REQUIRE(dx && x && scale && out && M > 0 && C > 0 && (C & 3) == 0, PAIF_EINVAL, "relu_mask_scale: bad arguments");
  const size_t n4 = M * (C / 4);
  hipLaunchKernelGGL(relu_mask_scale_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, paif::as_stream(stream), dx, x, scale, out, n4, C / 4);
  PAIF_LAUNCH_CHECK("relu_mask_scale");
  return 0;
}

int paif_nhwc_to_nchw_fwd(const float* x, float* y, int B, int HW, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, PAIF_EINVAL, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * HW * C, 256)), dim3(256), 0, paif::as_stream(stream), x, y, B,
                     (size_t)HW, C);
  PAIF_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}

int paif_nchw_to_nhwc_fwd(const float* x, float* y, int B, int HW, int C, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, PAIF_EINVAL, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)B * HW * C, 256)), dim3(256), 0, paif::as_stream(stream), x, y, B,
                     (size_t)HW, C);
  PAIF_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}

}  // extern "C"

// NCHW [B,C,HW] -> NHWC [B,HW,CP] with zero padding of channels [C,CP) (dlogits entering the dgrad GEMM)
namespace {
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int B, size_t HW, int C, int CP) {
  const size_t total = (size_t)B * HW * CP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CP);
    const size_t t = i / CP;
    const size_t px = t % HW, b = t / HW;
    y[i] = c < C ? x[(b * C + c) * HW + px] : 0.f;
  }
}
}  // namespace

extern "C" int paif_nchw_to_nhwc_pad_fwd(const float* x, float* y, int B, int HW, int C, int CP, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && CP >= C, PAIF_EINVAL, "nchw_to_nhwc_pad: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for((size_t)B * HW * CP, 256)), dim3(256), 0, paif::as_stream(stream), x, y, B,
                     (size_t)HW, C, CP);
  PAIF_LAUNCH_CHECK("nchw_to_nhwc_pad");
  return 0;
}
