// Input-gradient kernels of the fusion network (PGD inner loop, attack/attack.py:443-512).
// Dense-conv dgrads reuse conv_mfma.hip (transposed + 180-degree-rotated weights, activation-derivative
// hooks in the staging/epilogue); this file holds the rest: weight re-layout, tail / stem transposed
// convs, depthwise dgrad, ECA and spatial-attention backward, guided-filter backward.
#include "paif_common.h"

namespace {

constexpr int MAXGRID = 256 * 8;
inline int grid_for(size_t work_items, int per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)MAXGRID) g = MAXGRID;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// weights of the dgrad conv w.r.t. source channels [coff, coff+cs) of a forward conv w [Co][Ctot][k][k]:
//   wt[n][c][ky][kx] = w[c][coff + n][k-1-ky][k-1-kx]      -> [cs][Co][k][k]
// ---------------------------------------------------------------------------------------------
__global__ void conv_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int Ctot, int k, int coff,
                                         int cs) {
  const int kk = k * k;
  const int total = cs * Co * kk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int tap = i % kk;
    const int c = (i / kk) % Co;
    const int n = i / (kk * Co);
    wt[i] = w[((size_t)c * Ctot + coff + n) * kk + (kk - 1 - tap)];
  }
}

// Cell_Decom 1x1 folded to an ordinary [32][96][1][1] weight over [x, LF1, LF2] (see conv_mfma.hip)
__global__ void fold_decomp1x1_kernel(const float* __restrict__ w, float* __restrict__ wf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 32 * 96) return;
  const int n = i / 96, c = i % 96;
  const float* wn = w + n * 128;
  float v;
  if (c < 32) v = wn[64 + c] + wn[96 + c];
  else if (c < 64) v = wn[c - 32] - wn[64 + c - 32];
  else v = wn[32 + c - 64] - wn[96 + c - 64];
  wf[i] = v;
}

// ---------------------------------------------------------------------------------------------
// tail backward: fused = tanh(PReLU(z)), z = conv3x3 16->1 (t16).  d_t16[q][c] = sum_taps dz[q - off] * w[c][tap]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ dfused, const float* __restrict__ fused,
                                                       const float* __restrict__ z, const float* __restrict__ w,
                                                       const float* __restrict__ prelu, float* __restrict__ dt16, int B, int H, int W) {
  const int q = threadIdx.x & 3;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(4 * q + c) * 9 + k];
  const float slope = *prelu;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2); pix < npix; pix += (size_t)gridDim.x * 64) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const size_t img = (rowid - y0) * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y0 - (ky - 1);
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x0 - (kx - 1);
        if (xx < 0 || xx >= W) continue;
        const size_t o = img + (size_t)yy * W + xx;
        const float f = fused[o];
        const float dz = dfused[o] * (1.f - f * f) * (z[o] >= 0.f ? 1.f : slope);
        const int k = ky * 3 + kx;
        acc.x = fmaf(dz, wr[0][k], acc.x); acc.y = fmaf(dz, wr[1][k], acc.y);
        acc.z = fmaf(dz, wr[2][k], acc.z); acc.w = fmaf(dz, wr[3][k], acc.w);
      }
    }
    *reinterpret_cast<float4*>(dt16 + pix * 16 + q * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// stem backward: feat = PReLU(conv3x3 1->32 (img)); d_img[q] = sum_c sum_taps (dfeat*P'(feat))[q - off][c] * w[c][tap]
// (sign(feat) = sign(pre-activation) for a positive PReLU slope, which the host checks)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_bwd_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat,
                                                       const float* __restrict__ w, const float* __restrict__ prelu,
                                                       float* __restrict__ dimg, int B, int H, int W) {
  const int q = threadIdx.x & 7;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(4 * q + c) * 9 + k];
  const float slope = *prelu;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const size_t img = (rowid - y0) * W;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // one tap row (3 x 2 float4) in flight: unconditional loads on clamped coordinates, padding by select
      const int yy = y0 - (ky - 1);
      const int yc = min(max(yy, 0), H - 1);
      float4 dv[3], fv[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xc = min(max(x0 - (kx - 1), 0), W - 1);
        const size_t o = (img + (size_t)yc * W + xc) * 32 + q * 4;
        dv[kx] = *reinterpret_cast<const float4*>(dfeat + o);
        fv[kx] = *reinterpret_cast<const float4*>(feat + o);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x0 - (kx - 1);
        float4 d = dv[kx];
        const float4 f = fv[kx];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) d = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = ky * 3 + kx;
        s = fmaf(d.x * (f.x >= 0.f ? 1.f : slope), wr[0][k], s);
        s = fmaf(d.y * (f.y >= 0.f ? 1.f : slope), wr[1][k], s);
        s = fmaf(d.z * (f.z >= 0.f ? 1.f : slope), wr[2][k], s);
        s = fmaf(d.w * (f.w >= 0.f ? 1.f : slope), wr[3][k], s);
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) s += __shfl_xor(s, m);
    if (q == 0) dimg[pix] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// depthwise dgrad: out = dwconv(d_t; 180-degree-rotated w) * (aux > 0) + add     (DilConv: t = dw(relu(x)))
// ---------------------------------------------------------------------------------------------
template <int K, int DIL>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const float* __restrict__ dt, const float* __restrict__ w,
                                                         const float* __restrict__ aux, const float* __restrict__ add,
                                                         float* __restrict__ out, int B, int H, int W) {
  constexpr int P = DIL * (K - 1) / 2;
  const int q = threadIdx.x & 7;
  float wr[4][K * K];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < K * K; ++k) wr[c][k] = w[(4 * q + c) * K * K + (K * K - 1 - k)];
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int xx0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int yy0 = (int)(rowid % H);
    const float* base = dt + (rowid - yy0) * W * 32 + q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // unconditional loads on clamped coordinates, padding by select (see dwconv_kernel): the whole 3x3 stencil, or one
    // tap row for k >= 5, is in flight at a time
    constexpr int ROWS_IN_FLIGHT = K == 3 ? 3 : 1;
#pragma unroll
    for (int dy0 = 0; dy0 < K; dy0 += ROWS_IN_FLIGHT) {
      float4 tv[ROWS_IN_FLIGHT][K];
#pragma unroll
      for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
        const int yy = min(max(yy0 + (dy0 + r) * DIL - P, 0), H - 1);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int xx = min(max(xx0 + dx * DIL - P, 0), W - 1);
          tv[r][dx] = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * 32);
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
        const int dy = dy0 + r;
        const int yy = yy0 + dy * DIL - P;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int xx = xx0 + dx * DIL - P;
          float4 v = tv[r][dx];
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) v = make_float4(0.f, 0.f, 0.f, 0.f);
          const int k = dy * K + dx;
          acc.x = fmaf(v.x, wr[0][k], acc.x); acc.y = fmaf(v.y, wr[1][k], acc.y);
          acc.z = fmaf(v.z, wr[2][k], acc.z); acc.w = fmaf(v.w, wr[3][k], acc.w);
        }
      }
    }
    if (aux) {
      const float4 m = *reinterpret_cast<const float4*>(aux + pix * 32 + q * 4);
      acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
      acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
    }
    if (add) {
      const float4 a4 = *reinterpret_cast<const float4*>(add + pix * 32 + q * 4);
      acc.x += a4.x; acc.y += a4.y; acc.z += a4.z; acc.w += a4.w;
    }
    *reinterpret_cast<float4*>(out + pix * 32 + q * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// ECA backward.  Forward: out = PReLU(u), u = o*gate[b,c] + r, gate = sigmoid(conv1d_k(mean_hw(o))).
//   du = dout * P'(u);  dr = du;  do = du*gate + coef[b,c],
//   coef = conv1d^T( dgate * gate*(1-gate) ) / (H*W),  dgate[b,c] = sum_px du*o
// ---------------------------------------------------------------------------------------------
constexpr int ECA_PIX_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void eca_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ u,
                                                             const float* __restrict__ o, const float* __restrict__ prelu,
                                                             float* __restrict__ partial, size_t pix_per_img, int blocks_per_img) {
  __shared__ float4 red[32][8];
  const int q = threadIdx.x & 7, pr = threadIdx.x >> 3;
  const int b = blockIdx.y, blk = blockIdx.x;
  const float slope = *prelu;
  const size_t p0 = (size_t)blk * ECA_PIX_PER_BLOCK, p1 = min(pix_per_img, p0 + (size_t)ECA_PIX_PER_BLOCK);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t px = p0 + pr; px < p1; px += 32) {
    const size_t off = ((size_t)b * pix_per_img + px) * 32 + q * 4;
    const float4 d = *reinterpret_cast<const float4*>(dout + off);
    const float4 uu = *reinterpret_cast<const float4*>(u + off);
    const float4 oo = *reinterpret_cast<const float4*>(o + off);
    s.x += d.x * (uu.x >= 0.f ? 1.f : slope) * oo.x; s.y += d.y * (uu.y >= 0.f ? 1.f : slope) * oo.y;
    s.z += d.z * (uu.z >= 0.f ? 1.f : slope) * oo.z; s.w += d.w * (uu.w >= 0.f ? 1.f : slope) * oo.w;
  }
  red[pr][q] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    float4 t = red[0][threadIdx.x];
    for (int i = 1; i < 32; ++i) { const float4 v = red[i][threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *reinterpret_cast<float4*>(partial + ((size_t)b * blocks_per_img + blk) * 32 + threadIdx.x * 4) = t;
  }
}

__global__ void eca_bwd_gate_kernel(const float* __restrict__ partial, const float* __restrict__ gate, const float* __restrict__ w1d,
                                    int k, int blocks_per_img, float inv_hw, float* __restrict__ coef) {
  __shared__ float dy[32];
  const int b = blockIdx.x, c = threadIdx.x;  // 32 threads
  float s = 0.f;
  for (int i = 0; i < blocks_per_img; ++i) s += partial[((size_t)b * blocks_per_img + i) * 32 + c];
  const float g = gate[b * 32 + c];
  dy[c] = s * g * (1.f - g);
  __syncthreads();
  const int pad = (k - 1) / 2;
  float dm = 0.f;
  for (int j = 0; j < k; ++j) {
    const int cc = c - j + pad;  // y[cc] used mean[cc + j - pad] = mean[c]
    if (cc >= 0 && cc < 32) dm = fmaf(dy[cc], w1d[j], dm);
  }
  coef[b * 32 + c] = dm * inv_hw;
}

__global__ __launch_bounds__(256) void eca_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ u,
                                                            const float* __restrict__ gate, const float* __restrict__ coef,
                                                            const float* __restrict__ prelu, float* __restrict__ d_o,
                                                            float* __restrict__ d_r, size_t pix_per_img, size_t npix) {
  const int q = threadIdx.x & 7;
  const float slope = *prelu;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const size_t b = pix / pix_per_img;
    const float4 g = *reinterpret_cast<const float4*>(gate + b * 32 + q * 4);
    const float4 cf = *reinterpret_cast<const float4*>(coef + b * 32 + q * 4);
    const float4 d = *reinterpret_cast<const float4*>(dout + pix * 32 + q * 4);
    const float4 uu = *reinterpret_cast<const float4*>(u + pix * 32 + q * 4);
    float4 du;
    du.x = d.x * (uu.x >= 0.f ? 1.f : slope); du.y = d.y * (uu.y >= 0.f ? 1.f : slope);
    du.z = d.z * (uu.z >= 0.f ? 1.f : slope); du.w = d.w * (uu.w >= 0.f ? 1.f : slope);
    *reinterpret_cast<float4*>(d_r + pix * 32 + q * 4) = du;
    *reinterpret_cast<float4*>(d_o + pix * 32 + q * 4) =
        make_float4(fmaf(du.x, g.x, cf.x), fmaf(du.y, g.y, cf.y), fmaf(du.z, g.z, cf.z), fmaf(du.w, g.w, cf.w));
  }
}

// ---------------------------------------------------------------------------------------------
// spatial-attention blend backward.  Forward: agg = s*ir + (1-s)*vis, s = sigmoid(conv5x5(comp)),
// comp = (max_c ir, mean_c ir, max_c vis, mean_c vis).
//   pass 1: dpre = (sum_c dagg*(ir-vis)) * s*(1-s)
//   pass 2: dcomp[j] = sum_taps dpre[q - off] * w[j][tap];  d_ir = dagg*s + dcomp[1]/32 + [c == argmax_c ir] dcomp[0] (+add)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spa_bwd_dpre_kernel(const float* __restrict__ dagg, const float* __restrict__ ir,
                                                           const float* __restrict__ vis, const float* __restrict__ s,
                                                           float* __restrict__ dpre, size_t npix) {
  const int q = threadIdx.x & 7;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 d = *reinterpret_cast<const float4*>(dagg + pix * 32 + q * 4);
    const float4 a = *reinterpret_cast<const float4*>(ir + pix * 32 + q * 4);
    const float4 b = *reinterpret_cast<const float4*>(vis + pix * 32 + q * 4);
    float t = (d.x * (a.x - b.x) + d.y * (a.y - b.y)) + (d.z * (a.z - b.z) + d.w * (a.w - b.w));
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) t += __shfl_xor(t, m);
    if (q == 0) {
      const float sc = s[pix];
      dpre[pix] = t * sc * (1.f - sc);
    }
  }
}

__global__ __launch_bounds__(256) void spa_bwd_apply_kernel(const float* __restrict__ dagg, const float* __restrict__ dpre,
                                                            const float* __restrict__ w, const float* __restrict__ ir,
                                                            const float* __restrict__ vis, const float* __restrict__ s,
                                                            const float* __restrict__ add_ir, const float* __restrict__ add_vis,
                                                            float* __restrict__ d_ir, float* __restrict__ d_vis, int B, int H, int W) {
  __shared__ float ws[100];
  if (threadIdx.x < 100) ws[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 7;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const size_t img = (rowid - y0) * W;
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = q; tap < 25; tap += 8) {  // 25 taps over the 8 lanes of the pixel
      const int ky = tap / 5, kx = tap - ky * 5;
      const int yy = y0 - (ky - 2), xx = x0 - (kx - 2);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float dp = dpre[img + (size_t)yy * W + xx];
        dc[0] = fmaf(dp, ws[tap], dc[0]); dc[1] = fmaf(dp, ws[25 + tap], dc[1]);
        dc[2] = fmaf(dp, ws[50 + tap], dc[2]); dc[3] = fmaf(dp, ws[75 + tap], dc[3]);
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      dc[0] += __shfl_xor(dc[0], m); dc[1] += __shfl_xor(dc[1], m);
      dc[2] += __shfl_xor(dc[2], m); dc[3] += __shfl_xor(dc[3], m);
    }
    const float4 a = *reinterpret_cast<const float4*>(ir + pix * 32 + q * 4);
    const float4 b = *reinterpret_cast<const float4*>(vis + pix * 32 + q * 4);
    // first index of the channel maximum (torch.max(dim) returns one index; ties -> lowest)
    float ma = a.x; int ia = 4 * q;
    if (a.y > ma) { ma = a.y; ia = 4 * q + 1; } if (a.z > ma) { ma = a.z; ia = 4 * q + 2; } if (a.w > ma) { ma = a.w; ia = 4 * q + 3; }
    float mb = b.x; int ib = 4 * q;
    if (b.y > mb) { mb = b.y; ib = 4 * q + 1; } if (b.z > mb) { mb = b.z; ib = 4 * q + 2; } if (b.w > mb) { mb = b.w; ib = 4 * q + 3; }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      const float oa = __shfl_xor(ma, m); const int oia = __shfl_xor(ia, m);
      if (oa > ma || (oa == ma && oia < ia)) { ma = oa; ia = oia; }
      const float ob = __shfl_xor(mb, m); const int oib = __shfl_xor(ib, m);
      if (ob > mb || (ob == mb && oib < ib)) { mb = ob; ib = oib; }
    }
    const float sc = s[pix], om = 1.f - sc;
    const float4 d = *reinterpret_cast<const float4*>(dagg + pix * 32 + q * 4);
    const float mi = dc[1] * (1.0f / 32.0f), mv = dc[3] * (1.0f / 32.0f);
    float4 gi = make_float4(d.x * sc + mi, d.y * sc + mi, d.z * sc + mi, d.w * sc + mi);
    float4 gv = make_float4(d.x * om + mv, d.y * om + mv, d.z * om + mv, d.w * om + mv);
    if ((ia >> 2) == q) { const int j = ia & 3; if (j == 0) gi.x += dc[0]; else if (j == 1) gi.y += dc[0]; else if (j == 2) gi.z += dc[0]; else gi.w += dc[0]; }
    if ((ib >> 2) == q) { const int j = ib & 3; if (j == 0) gv.x += dc[2]; else if (j == 1) gv.y += dc[2]; else if (j == 2) gv.z += dc[2]; else gv.w += dc[2]; }
    if (add_ir) { const float4 t = *reinterpret_cast<const float4*>(add_ir + pix * 32 + q * 4); gi.x += t.x; gi.y += t.y; gi.z += t.z; gi.w += t.w; }
    if (add_vis) { const float4 t = *reinterpret_cast<const float4*>(add_vis + pix * 32 + q * 4); gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
    *reinterpret_cast<float4*>(d_ir + pix * 32 + q * 4) = gi;
    *reinterpret_cast<float4*>(d_vis + pix * 32 + q * 4) = gv;
  }
}

// ---------------------------------------------------------------------------------------------
// Guided-filter backward (see guided_filter.hip for the forward and the row-streaming box filter).
// With M(v) = box(v)/N (self-adjoint up to the normalisation: M^T(u) = box(u/N)):
//   stage 1 (per eps e, accumulating):  dA' = M^T(dLF_e * g), db' = M^T(dLF_e), MA = M(A_e)
//       dA = dA' - db'*mg;  dcov = dA/(var+eps);  dvar_e = sum_c -dA*A_e/(var+eps)
//       t_my += db' - dcov*mg;  t_mgy += dcov;  dmg += sum_c(-db'*A_e - dcov*my) - 2*mg*dvar_e;  dmgg += dvar_e
//       dg_direct += sum_c dLF_e * MA
//   stage 2:  dy = M^T(t_my) + g*M^T(t_mgy);  dg = dg_direct + sum_c y*M^T(t_mgy) + M^T(dmg) + 2 g M^T(dmgg)
//             then the guide g = max_c y - min_c y routes dg to the arg-max (+) and arg-min (-) channels.
// ---------------------------------------------------------------------------------------------
constexpr int R = 4, KB = 9, NCOL = 32, OCOL = 24, ROWS_PER_SEG = 60;

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float boxn(int row, int col, int H, int W) {
  const int cy = min(row + R, H - 1) - max(row - R, 0) + 1;
  const int cx = min(col + R, W - 1) - max(col - R, 0) + 1;
  return (float)(cy * cx);
}

// gstat [B,H,W,2] = (mean_g, var_g): small forward by-product recomputed here (2 box filters on 1 channel)
__global__ __launch_bounds__(256) void gf_gstat_kernel(const float* __restrict__ guide, float* __restrict__ gstat, int B, int H, int W) {
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * 256) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const float* base = guide + (rowid - y0) * W;
    float s = 0.f, ss = 0.f;
    for (int yy = max(y0 - R, 0); yy <= min(y0 + R, H - 1); ++yy)
      for (int xx = max(x0 - R, 0); xx <= min(x0 + R, W - 1); ++xx) {
        const float g = base[(size_t)yy * W + xx];
        s += g;
        ss = fmaf(g, g, ss);
      }
    const float n = boxn(y0, x0, H, W);
    const float mg = s / n;
    gstat[pix * 2] = mg;
    gstat[pix * 2 + 1] = ss / n - mg * mg;
  }
}

// stage 1, one eps per launch (e = 0 writes, e = 1 accumulates)
__global__ __launch_bounds__(256) void gf_bwd1_kernel(const float* __restrict__ guide, const float* __restrict__ gstat,
                                                      const float* __restrict__ Ae, const float* __restrict__ be,
                                                      const float* __restrict__ dLF, float eps, int accumulate,
                                                      float* __restrict__ t_my, float* __restrict__ t_mgy, float* __restrict__ t_g,
                                                      int B, int H, int W, int nstrip, int nseg) {
  __shared__ float4 s1[2][NCOL][8];
  __shared__ float4 s2[2][NCOL][8];
  __shared__ float4 s3[2][NCOL][8];
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  int t = blockIdx.x;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int col = strip * OCOL - R + xi;
  const bool colin = col >= 0 && col < W;
  const int ybeg = seg * ROWS_PER_SEG, yend = min(H, ybeg + ROWS_PER_SEG);
  const size_t img = (size_t)b * H * W;
  float4 r1[KB], r2[KB], r3[KB];
  const int r0 = ybeg - R, rend = yend + R;
  for (int rr = r0; rr < rend; rr += KB) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int row = rr + k;
      if (row >= rend) break;
      float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f), u2 = u1, a4 = u1;
      if (colin && row >= 0 && row < H) {
        const size_t px = img + (size_t)row * W + col;
        const float4 d = *reinterpret_cast<const float4*>(dLF + px * 32 + q * 4);
        a4 = *reinterpret_cast<const float4*>(Ae + px * 32 + q * 4);
        const float invn = 1.0f / boxn(row, col, H, W);
        u2 = f4scale(d, invn);            // dLF / N   (source-side normalisation of M^T)
        u1 = f4scale(u2, guide[px]);      // dLF * g / N
      }
      r1[k] = u1; r2[k] = u2; r3[k] = a4;
      const int orow = row - R;
      if (orow < ybeg) continue;
      float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1, v3 = v1;
#pragma unroll
      for (int j = 0; j < KB; ++j) { v1 = f4add(v1, r1[j]); v2 = f4add(v2, r2[j]); v3 = f4add(v3, r3[j]); }
      const int buf = orow & 1;
      s1[buf][xi][q] = v1; s2[buf][xi][q] = v2; s3[buf][xi][q] = v3;
      __syncthreads();
      if (xi >= R && xi < NCOL - R && col < W) {
        float4 dAp = make_float4(0.f, 0.f, 0.f, 0.f), dbp = dAp, MA = dAp;
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          dAp = f4add(dAp, s1[buf][xi + j][q]);
          dbp = f4add(dbp, s2[buf][xi + j][q]);
          MA = f4add(MA, s3[buf][xi + j][q]);
        }
        const size_t px = img + (size_t)orow * W + col;
        const float n = boxn(orow, col, H, W);
        MA = f4scale(MA, 1.0f / n);
        const float mg = gstat[px * 2], var = gstat[px * 2 + 1];
        const float den = var + eps;
        const float4 A = *reinterpret_cast<const float4*>(Ae + px * 32 + q * 4);
        const float4 bb = *reinterpret_cast<const float4*>(be + px * 32 + q * 4);
        const float4 dl = *reinterpret_cast<const float4*>(dLF + px * 32 + q * 4);
        float4 dA = make_float4(dAp.x - dbp.x * mg, dAp.y - dbp.y * mg, dAp.z - dbp.z * mg, dAp.w - dbp.w * mg);
        float4 dcov = make_float4(dA.x / den, dA.y / den, dA.z / den, dA.w / den);
        float dvar = -((dcov.x * A.x + dcov.y * A.y) + (dcov.z * A.z + dcov.w * A.w));
        // my = b + A*mg
        float dmg = -((dbp.x * A.x + dcov.x * (bb.x + A.x * mg)) + (dbp.y * A.y + dcov.y * (bb.y + A.y * mg)) +
                      (dbp.z * A.z + dcov.z * (bb.z + A.z * mg)) + (dbp.w * A.w + dcov.w * (bb.w + A.w * mg)));
        float dgd = (dl.x * MA.x + dl.y * MA.y) + (dl.z * MA.z + dl.w * MA.w);
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
          dvar += __shfl_xor(dvar, m);
          dmg += __shfl_xor(dmg, m);
          dgd += __shfl_xor(dgd, m);
        }
        dmg -= 2.f * mg * dvar;
        float4 o_my = make_float4(dbp.x - dcov.x * mg, dbp.y - dcov.y * mg, dbp.z - dcov.z * mg, dbp.w - dcov.w * mg);
        float4 o_mgy = dcov;
        float4 og = make_float4(dmg, dvar, dgd, 0.f);
        if (accumulate) {
          o_my = f4add(o_my, *reinterpret_cast<const float4*>(t_my + px * 32 + q * 4));
          o_mgy = f4add(o_mgy, *reinterpret_cast<const float4*>(t_mgy + px * 32 + q * 4));
          if (q == 0) og = f4add(og, *reinterpret_cast<const float4*>(t_g + px * 4));
        }
        *reinterpret_cast<float4*>(t_my + px * 32 + q * 4) = o_my;
        *reinterpret_cast<float4*>(t_mgy + px * 32 + q * 4) = o_mgy;
        if (q == 0) *reinterpret_cast<float4*>(t_g + px * 4) = og;
      }
    }
  }
}

// stage 2: dy (+ guide routing + add), the guide gradient itself is folded into dy
__global__ __launch_bounds__(256) void gf_bwd2_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                      const float* __restrict__ t_my, const float* __restrict__ t_mgy,
                                                      const float* __restrict__ t_g, const float* __restrict__ add,
                                                      float* __restrict__ dy, int B, int H, int W, int nstrip, int nseg) {
  __shared__ float4 s1[2][NCOL][8];
  __shared__ float4 s2[2][NCOL][8];
  __shared__ float2 s3[2][NCOL];
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  int t = blockIdx.x;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int col = strip * OCOL - R + xi;
  const bool colin = col >= 0 && col < W;
  const int ybeg = seg * ROWS_PER_SEG, yend = min(H, ybeg + ROWS_PER_SEG);
  const size_t img = (size_t)b * H * W;
  float4 r1[KB], r2[KB];
  float2 r3[KB];
  const int r0 = ybeg - R, rend = yend + R;
  for (int rr = r0; rr < rend; rr += KB) {
#pragma unroll
    for (int k = 0; k < KB; ++k) {
      const int row = rr + k;
      if (row >= rend) break;
      float4 u1 = make_float4(0.f, 0.f, 0.f, 0.f), u2 = u1;
      float2 u3 = make_float2(0.f, 0.f);
      if (colin && row >= 0 && row < H) {
        const size_t px = img + (size_t)row * W + col;
        const float invn = 1.0f / boxn(row, col, H, W);
        u1 = f4scale(*reinterpret_cast<const float4*>(t_my + px * 32 + q * 4), invn);
        u2 = f4scale(*reinterpret_cast<const float4*>(t_mgy + px * 32 + q * 4), invn);
        const float4 tg = *reinterpret_cast<const float4*>(t_g + px * 4);
        u3 = make_float2(tg.x * invn, tg.y * invn);
      }
      r1[k] = u1; r2[k] = u2; r3[k] = u3;
      const int orow = row - R;
      if (orow < ybeg) continue;
      float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v1;
      float2 v3 = make_float2(0.f, 0.f);
#pragma unroll
      for (int j = 0; j < KB; ++j) { v1 = f4add(v1, r1[j]); v2 = f4add(v2, r2[j]); v3.x += r3[j].x; v3.y += r3[j].y; }
      const int buf = orow & 1;
      s1[buf][xi][q] = v1; s2[buf][xi][q] = v2;
      if (q == 0) s3[buf][xi] = v3;
      __syncthreads();
      if (xi >= R && xi < NCOL - R && col < W) {
        float4 b1 = make_float4(0.f, 0.f, 0.f, 0.f), b2 = b1;
        float2 b3 = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          b1 = f4add(b1, s1[buf][xi + j][q]);
          b2 = f4add(b2, s2[buf][xi + j][q]);
          const float2 g3 = s3[buf][xi + j];
          b3.x += g3.x; b3.y += g3.y;
        }
        const size_t px = img + (size_t)orow * W + col;
        const float g = guide[px];
        const float4 yy = *reinterpret_cast<const float4*>(y + px * 32 + q * 4);
        float4 o = make_float4(fmaf(g, b2.x, b1.x), fmaf(g, b2.y, b1.y), fmaf(g, b2.z, b1.z), fmaf(g, b2.w, b1.w));
        float dgy = (yy.x * b2.x + yy.y * b2.y) + (yy.z * b2.z + yy.w * b2.w);
        // arg-max / arg-min channel of y at this pixel (first index on ties)
        float mx = yy.x; int imx = 4 * q;
        if (yy.y > mx) { mx = yy.y; imx = 4 * q + 1; } if (yy.z > mx) { mx = yy.z; imx = 4 * q + 2; } if (yy.w > mx) { mx = yy.w; imx = 4 * q + 3; }
        float mn = yy.x; int imn = 4 * q;
        if (yy.y < mn) { mn = yy.y; imn = 4 * q + 1; } if (yy.z < mn) { mn = yy.z; imn = 4 * q + 2; } if (yy.w < mn) { mn = yy.w; imn = 4 * q + 3; }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
          dgy += __shfl_xor(dgy, m);
          const float ox = __shfl_xor(mx, m); const int oix = __shfl_xor(imx, m);
          if (ox > mx || (ox == mx && oix < imx)) { mx = ox; imx = oix; }
          const float on = __shfl_xor(mn, m); const int oin = __shfl_xor(imn, m);
          if (on < mn || (on == mn && oin < imn)) { mn = on; imn = oin; }
        }
        const float dg = t_g[px * 4 + 2] + dgy + b3.x + 2.f * g * b3.y;
        if ((imx >> 2) == q) { const int j = imx & 3; if (j == 0) o.x += dg; else if (j == 1) o.y += dg; else if (j == 2) o.z += dg; else o.w += dg; }
        if ((imn >> 2) == q) { const int j = imn & 3; if (j == 0) o.x -= dg; else if (j == 1) o.y -= dg; else if (j == 2) o.z -= dg; else o.w -= dg; }
        if (add) o = f4add(o, *reinterpret_cast<const float4*>(add + px * 32 + q * 4));
        *reinterpret_cast<float4*>(dy + px * 32 + q * 4) = o;
      }
    }
  }
}

}  // namespace

extern "C" {

int paif_conv_weight_dgrad(const float* w, float* wt, int Co, int Ctot, int k, int coff, int cs, paif_stream_t stream) {
  PAIF_REQUIRE(w && wt && Co > 0 && Ctot > 0 && k > 0 && coff >= 0 && cs > 0 && coff + cs <= Ctot, PAIF_EINVAL,
               "conv_weight_dgrad: bad arguments");
  const int total = cs * Co * k * k;
  hipLaunchKernelGGL(conv_weight_dgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), w, wt, Co, Ctot, k,
                     coff, cs);
  PAIF_LAUNCH_CHECK("conv_weight_dgrad");
  return 0;
}

int paif_fold_decomp1x1_weight(const float* w, float* wf, paif_stream_t stream) {
  PAIF_REQUIRE(w && wf, PAIF_EINVAL, "fold_decomp1x1_weight: null pointer");
  hipLaunchKernelGGL(fold_decomp1x1_kernel, dim3(12), dim3(256), 0, paif::as_stream(stream), w, wf);
  PAIF_LAUNCH_CHECK("fold_decomp1x1_weight");
  return 0;
}

int paif_tail_bwd_input(const float* dfused, const float* fused, const float* z, const float* w, const float* prelu, float* dt16,
                        int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(dfused && fused && z && w && prelu && dt16 && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "tail_bwd: bad arguments");
  hipLaunchKernelGGL(tail_bwd_kernel, dim3(grid_for((size_t)B * H * W, 64)), dim3(256), 0, paif::as_stream(stream), dfused, fused, z, w,
                     prelu, dt16, B, H, W);
  PAIF_LAUNCH_CHECK("tail_bwd");
  return 0;
}

int paif_stem_bwd_input(const float* dfeat, const float* feat, const float* w, const float* prelu, float* dimg, int B, int H, int W,
                        paif_stream_t stream) {
  PAIF_REQUIRE(dfeat && feat && w && prelu && dimg && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "stem_bwd: bad arguments");
  hipLaunchKernelGGL(stem_bwd_kernel, dim3(grid_for((size_t)B * H * W, 32)), dim3(256), 0, paif::as_stream(stream), dfeat, feat, w,
                     prelu, dimg, B, H, W);
  PAIF_LAUNCH_CHECK("stem_bwd");
  return 0;
}

int paif_dwconv_bwd_input(const float* dt, const float* w, const float* aux, const float* add, float* out, int k, int dil, int B, int H,
                          int W, paif_stream_t stream) {
  PAIF_REQUIRE(dt && w && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv_bwd: bad arguments");
  const dim3 g(grid_for((size_t)B * H * W, 32)), blk(256);
  hipStream_t st = paif::as_stream(stream);
  switch (k * 10 + dil) {
    case 31: hipLaunchKernelGGL((dwconv_bwd_kernel<3, 1>), g, blk, 0, st, dt, w, aux, add, out, B, H, W); break;
    case 32: hipLaunchKernelGGL((dwconv_bwd_kernel<3, 2>), g, blk, 0, st, dt, w, aux, add, out, B, H, W); break;
    case 51: hipLaunchKernelGGL((dwconv_bwd_kernel<5, 1>), g, blk, 0, st, dt, w, aux, add, out, B, H, W); break;
    case 52: hipLaunchKernelGGL((dwconv_bwd_kernel<5, 2>), g, blk, 0, st, dt, w, aux, add, out, B, H, W); break;
    default:
      paif::set_error("dwconv_bwd: kernel %d dil %d not built", k, dil);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("dwconv_bwd");
  return 0;
}

int paif_eca_bwd_blocks(int H, int W) { return (int)(((size_t)H * W + ECA_PIX_PER_BLOCK - 1) / ECA_PIX_PER_BLOCK); }

int paif_eca_bwd_input(const float* dout, const float* u, const float* o, const float* gate, const float* w1d, int k,
                       const float* prelu, float* partial, float* coef, float* d_o, float* d_r, int B, int H, int W,
                       paif_stream_t stream) {
  PAIF_REQUIRE(dout && u && o && gate && w1d && prelu && partial && coef && d_o && d_r && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "eca_bwd: bad arguments");
  hipStream_t st = paif::as_stream(stream);
  const int bpi = paif_eca_bwd_blocks(H, W);
  const size_t ppi = (size_t)H * W;
  hipLaunchKernelGGL(eca_bwd_reduce_kernel, dim3(bpi, B), dim3(256), 0, st, dout, u, o, prelu, partial, ppi, bpi);
  PAIF_LAUNCH_CHECK("eca_bwd(reduce)");
  hipLaunchKernelGGL(eca_bwd_gate_kernel, dim3(B), dim3(32), 0, st, partial, gate, w1d, k, bpi, 1.0f / ((float)H * (float)W), coef);
  PAIF_LAUNCH_CHECK("eca_bwd(gate)");
  hipLaunchKernelGGL(eca_bwd_apply_kernel, dim3(grid_for(ppi * B, 32)), dim3(256), 0, st, dout, u, gate, coef, prelu, d_o, d_r, ppi,
                     ppi * B);
  PAIF_LAUNCH_CHECK("eca_bwd(apply)");
  return 0;
}

int paif_spa_blend_bwd_input(const float* dagg, const float* w, const float* ir, const float* vis, const float* s, const float* add_ir,
                             const float* add_vis, float* dpre, float* d_ir, float* d_vis, int B, int H, int W,
                             paif_stream_t stream) {
  PAIF_REQUIRE(dagg && w && ir && vis && s && dpre && d_ir && d_vis && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "spa_blend_bwd: bad arguments");
  hipStream_t st = paif::as_stream(stream);
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(spa_bwd_dpre_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, dagg, ir, vis, s, dpre, npix);
  PAIF_LAUNCH_CHECK("spa_blend_bwd(dpre)");
  hipLaunchKernelGGL(spa_bwd_apply_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, dagg, dpre, w, ir, vis, s, add_ir, add_vis, d_ir,
                     d_vis, B, H, W);
  PAIF_LAUNCH_CHECK("spa_blend_bwd(apply)");
  return 0;
}

// the round-1 form (32-column strips x 60-row segments, one eps per stage-1 launch); since round 6 the cross-check and the
// any-size fallback of gf_backward.hip's paif_guided_filter_bwd_input (PAIF_GF_BWD=v1).  gstat: [B,H,W,2] of the caller's workspace.
__attribute__((visibility("hidden"))) int paifi_gf_bwd_input_v1(const float* guide, const float* y, const float* ab, const float* dlf,
                                                                float eps0, float eps1, const float* add, float* gstat, float* t_my,
                                                                float* t_mgy, float* t_g, float* dy, int B, int H, int W,
                                                                paif_stream_t stream) {
  PAIF_REQUIRE(guide && y && ab && dlf && gstat && t_my && t_mgy && t_g && dy && B > 0, PAIF_EINVAL, "guided_filter_bwd: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter_bwd: H,W must exceed 9");
  hipStream_t st = paif::as_stream(stream);
  const size_t plane = (size_t)B * H * W * 32;
  const int nstrip = (W + OCOL - 1) / OCOL, nseg = (H + ROWS_PER_SEG - 1) / ROWS_PER_SEG;
  hipLaunchKernelGGL(gf_gstat_kernel, dim3(grid_for((size_t)B * H * W, 256)), dim3(256), 0, st, guide, gstat, B, H, W);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(gstat)");
  hipLaunchKernelGGL(gf_bwd1_kernel, dim3(B * nstrip * nseg), dim3(256), 0, st, guide, gstat, ab, ab + plane, dlf, eps0, 0, t_my, t_mgy,
                     t_g, B, H, W, nstrip, nseg);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(1a)");
  hipLaunchKernelGGL(gf_bwd1_kernel, dim3(B * nstrip * nseg), dim3(256), 0, st, guide, gstat, ab + 2 * plane, ab + 3 * plane,
                     dlf + plane, eps1, 1, t_my, t_mgy, t_g, B, H, W, nstrip, nseg);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(1b)");
  hipLaunchKernelGGL(gf_bwd2_kernel, dim3(B * nstrip * nseg), dim3(256), 0, st, guide, y, t_my, t_mgy, t_g, add, dy, B, H, W, nstrip,
                     nseg);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(2)");
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// SPAattention (operations_m.py:148-204): comp = (max_c o, mean_c o); s = sigmoid(conv kxk 2->1 (comp));
// out = PReLU(o*s + r).  Forward in two passes (pool, then conv+apply); backward mirrors spa_blend's.
// ---------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void channel_pool1_kernel(const float* __restrict__ o, float* __restrict__ comp, size_t npix) {
  const int q = threadIdx.x & 7;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 a = *reinterpret_cast<const float4*>(o + pix * 32 + q * 4);
    float mx = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), sm = (a.x + a.y) + (a.z + a.w);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, m));
      sm += __shfl_xor(sm, m);
    }
    if (q == 0) *reinterpret_cast<float2*>(comp + pix * 2) = make_float2(mx, sm * (1.0f / 32.0f));
  }
}

__global__ __launch_bounds__(256) void spa1_apply_kernel(const float* __restrict__ comp, const float* __restrict__ w, int k,
                                                         const float* __restrict__ o, const float* __restrict__ r,
                                                         const float* __restrict__ prelu, float* __restrict__ s_out,
                                                         float* __restrict__ u_out, float* __restrict__ out, int B, int H, int W) {
  __shared__ float ws[2 * 49];
  const int kk = k * k, pad = k / 2;
  if ((int)threadIdx.x < 2 * kk) ws[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 7;
  const float slope = *prelu;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const float* cbase = comp + (rowid - y0) * W * 2;
    float acc = 0.f;
    for (int tap = q; tap < kk; tap += 8) {
      const int dy = tap / k, dx = tap - dy * k;
      const int yy = y0 + dy - pad, xx = x0 + dx - pad;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float2 c = *reinterpret_cast<const float2*>(cbase + ((size_t)yy * W + xx) * 2);
        acc = fmaf(c.x, ws[tap], acc);
        acc = fmaf(c.y, ws[kk + tap], acc);
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) acc += __shfl_xor(acc, m);
    const float sc = 1.0f / (1.0f + expf(-acc));
    const float4 ov = *reinterpret_cast<const float4*>(o + pix * 32 + q * 4);
    const float4 rv = *reinterpret_cast<const float4*>(r + pix * 32 + q * 4);
    float4 u;
    u.x = __fadd_rn(__fmul_rn(ov.x, sc), rv.x); u.y = __fadd_rn(__fmul_rn(ov.y, sc), rv.y);
    u.z = __fadd_rn(__fmul_rn(ov.z, sc), rv.z); u.w = __fadd_rn(__fmul_rn(ov.w, sc), rv.w);
    if (u_out) *reinterpret_cast<float4*>(u_out + pix * 32 + q * 4) = u;
    if (s_out && q == 0) s_out[pix] = sc;
    *reinterpret_cast<float4*>(out + pix * 32 + q * 4) =
        make_float4(paif::prelu_f(u.x, slope), paif::prelu_f(u.y, slope), paif::prelu_f(u.z, slope), paif::prelu_f(u.w, slope));
  }
}

// pass 1 of the backward: dpre = (sum_c du*o) * s(1-s), du = dout*P'(u)
__global__ __launch_bounds__(256) void spa1_bwd_dpre_kernel(const float* __restrict__ dout, const float* __restrict__ u,
                                                            const float* __restrict__ o, const float* __restrict__ s,
                                                            const float* __restrict__ prelu, float* __restrict__ dpre, size_t npix) {
  const int q = threadIdx.x & 7;
  const float slope = *prelu;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 d = *reinterpret_cast<const float4*>(dout + pix * 32 + q * 4);
    const float4 uu = *reinterpret_cast<const float4*>(u + pix * 32 + q * 4);
    const float4 oo = *reinterpret_cast<const float4*>(o + pix * 32 + q * 4);
    float t = (d.x * (uu.x >= 0.f ? 1.f : slope) * oo.x + d.y * (uu.y >= 0.f ? 1.f : slope) * oo.y) +
              (d.z * (uu.z >= 0.f ? 1.f : slope) * oo.z + d.w * (uu.w >= 0.f ? 1.f : slope) * oo.w);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) t += __shfl_xor(t, m);
    if (q == 0) {
      const float sc = s[pix];
      dpre[pix] = t * sc * (1.f - sc);
    }
  }
}

__global__ __launch_bounds__(256) void spa1_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ u,
                                                             const float* __restrict__ o, const float* __restrict__ s,
                                                             const float* __restrict__ dpre, const float* __restrict__ w, int k,
                                                             const float* __restrict__ prelu, float* __restrict__ d_o,
                                                             float* __restrict__ d_r, int B, int H, int W) {
  __shared__ float ws[2 * 49];
  const int kk = k * k, pad = k / 2;
  if ((int)threadIdx.x < 2 * kk) ws[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int q = threadIdx.x & 7;
  const float slope = *prelu;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const size_t img = (rowid - y0) * W;
    float dc0 = 0.f, dc1 = 0.f;
    for (int tap = q; tap < kk; tap += 8) {
      const int ky = tap / k, kx = tap - ky * k;
      const int yy = y0 - (ky - pad), xx = x0 - (kx - pad);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float dp = dpre[img + (size_t)yy * W + xx];
        dc0 = fmaf(dp, ws[tap], dc0);
        dc1 = fmaf(dp, ws[kk + tap], dc1);
      }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { dc0 += __shfl_xor(dc0, m); dc1 += __shfl_xor(dc1, m); }
    const float4 oo = *reinterpret_cast<const float4*>(o + pix * 32 + q * 4);
    float mx = oo.x; int im = 4 * q;
    if (oo.y > mx) { mx = oo.y; im = 4 * q + 1; } if (oo.z > mx) { mx = oo.z; im = 4 * q + 2; } if (oo.w > mx) { mx = oo.w; im = 4 * q + 3; }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      const float om = __shfl_xor(mx, m); const int oi = __shfl_xor(im, m);
      if (om > mx || (om == mx && oi < im)) { mx = om; im = oi; }
    }
    const float4 d = *reinterpret_cast<const float4*>(dout + pix * 32 + q * 4);
    const float4 uu = *reinterpret_cast<const float4*>(u + pix * 32 + q * 4);
    float4 du;
    du.x = d.x * (uu.x >= 0.f ? 1.f : slope); du.y = d.y * (uu.y >= 0.f ? 1.f : slope);
    du.z = d.z * (uu.z >= 0.f ? 1.f : slope); du.w = d.w * (uu.w >= 0.f ? 1.f : slope);
    *reinterpret_cast<float4*>(d_r + pix * 32 + q * 4) = du;
    const float sc = s[pix], mean_g = dc1 * (1.0f / 32.0f);
    float4 go = make_float4(du.x * sc + mean_g, du.y * sc + mean_g, du.z * sc + mean_g, du.w * sc + mean_g);
    if ((im >> 2) == q) { const int j = im & 3; if (j == 0) go.x += dc0; else if (j == 1) go.y += dc0; else if (j == 2) go.z += dc0; else go.w += dc0; }
    *reinterpret_cast<float4*>(d_o + pix * 32 + q * 4) = go;
  }
}

}  // namespace

extern "C" int paif_spa1_fwd(const float* o, const float* r, const float* w, int k, const float* prelu, float* comp, float* s_out,
                             float* u_out, float* out, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(o && r && w && prelu && comp && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "spa1: bad arguments");
  PAIF_REQUIRE(k == 3 || k == 5 || k == 7, PAIF_ENOSUP, "spa1: k=%d", k);
  hipStream_t st = paif::as_stream(stream);
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(channel_pool1_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, o, comp, npix);
  PAIF_LAUNCH_CHECK("spa1(pool)");
  hipLaunchKernelGGL(spa1_apply_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, comp, w, k, o, r, prelu, s_out, u_out, out, B, H, W);
  PAIF_LAUNCH_CHECK("spa1(apply)");
  return 0;
}

extern "C" int paif_spa1_bwd_input(const float* dout, const float* u, const float* o, const float* s, const float* w, int k,
                                   const float* prelu, float* dpre, float* d_o, float* d_r, int B, int H, int W,
                                   paif_stream_t stream) {
  PAIF_REQUIRE(dout && u && o && s && w && prelu && dpre && d_o && d_r && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "spa1_bwd: bad arguments");
  PAIF_REQUIRE(k == 3 || k == 5 || k == 7, PAIF_ENOSUP, "spa1_bwd: k=%d", k);
  hipStream_t st = paif::as_stream(stream);
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(spa1_bwd_dpre_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, dout, u, o, s, prelu, dpre, npix);
  PAIF_LAUNCH_CHECK("spa1_bwd(dpre)");
  hipLaunchKernelGGL(spa1_bwd_apply_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, st, dout, u, o, s, dpre, w, k, prelu, d_o, d_r, B, H,
                     W);
  PAIF_LAUNCH_CHECK("spa1_bwd(apply)");
  return 0;
}
