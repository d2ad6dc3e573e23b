// HBM-bound kernels of the fusion network and of the fusion->segmentation glue (gfx950).
// Layout: NHWC fp32, one float4 (4 channels) per lane, 8 lanes per pixel -> a wave64 load/store
// touches 8 pixels x 128 B = 1 KiB contiguous.  Channel reductions (max/min/mean over the 32
// channels of a pixel) are 8-lane shuffle reductions; grid-wide reductions are two-pass and
// deterministic (per-block partials reduced in a fixed order, no float atomics).
#include <stdarg.h>

#include "paif_common.h"

namespace paif {
static thread_local char g_err[256] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace paif

namespace {

constexpr int MAXGRID = 256 * 8;  // memory-bound kernels: cap the grid and grid-stride (guide, G11)

inline int grid_for(size_t work_items, int per_block) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)MAXGRID) g = MAXGRID;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// RGB -> YCrCb, NCHW planes (core/model_fusion_auto.py:69-92)
// ---------------------------------------------------------------------------------------------
__global__ void rgb2ycrcb_kernel(const float* __restrict__ rgb, float* __restrict__ ycc, int B, size_t HW) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, px = i - b * HW;
    const float* s = rgb + b * 3 * HW + px;
    const float R = s[0], G = s[HW], Bl = s[2 * HW];
    // written exactly as the reference's expression tree (left-to-right adds, no re-association)
    const float Y = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, R), __fmul_rn(0.587f, G)), __fmul_rn(0.114f, Bl));
    const float Cr = __fadd_rn(__fmul_rn(__fsub_rn(R, Y), 0.713f), 0.5f);
    const float Cb = __fadd_rn(__fmul_rn(__fsub_rn(Bl, Y), 0.564f), 0.5f);
    float* d = ycc + b * 3 * HW + px;
    d[0] = Y; d[HW] = Cr; d[2 * HW] = Cb;
  }
}

// YCrCb2RGB stand-alone (core/model_fusion_auto.py:94-111): (im_flat + bias).mm(mat), no clamp
__global__ void ycrcb2rgb_kernel(const float* __restrict__ ycc, float* __restrict__ rgb, int B, size_t HW) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / HW, px = i - b * HW;
    const float* s = ycc + b * 3 * HW + px;
    const float y0 = __fadd_rn(s[0], 0.0f);
    const float cr = __fadd_rn(s[HW], -0.5f);
    const float cb = __fadd_rn(s[2 * HW], -0.5f);
    // mat rows: [1,1,1], [1.403,-0.714,0], [0,-0.344,1.773]; k-ordered fma chain (same as recompose_clamp_kernel)
    float* d = rgb + b * 3 * HW + px;
    d[0] = fmaf(cb, 0.0f, fmaf(cr, 1.403f, y0 * 1.0f));
    d[HW] = fmaf(cb, -0.344f, fmaf(cr, -0.714f, y0 * 1.0f));
    d[2 * HW] = fmaf(cb, 1.773f, fmaf(cr, 0.0f, y0 * 1.0f));
  }
}

// ---------------------------------------------------------------------------------------------
// stem: 3x3 conv 1->32 (no bias) + PReLU, fused with the guide max_c - min_c
// thread = (pixel, channel quad); 8 lanes of a pixel read the same 9 taps (broadcast from L1)
// ---------------------------------------------------------------------------------------------
// One workgroup = one 128-pixel piece of an image row (4 steps of 32 pixels x 8 channel quads): the row / image
// coordinates are block-uniform scalars, so the per-pixel index arithmetic is a column clamp -- the flat-pixel form
// spent most of its instructions on 64-bit div/mod (154 us for a 315 MB write-only stream).
constexpr int STEM_CHUNK = 128;
#ifndef PAIF_STEM_GRID
#define PAIF_STEM_GRID 4096
#endif
constexpr int STEM_GRID = PAIF_STEM_GRID;   // workgroups of a launch: 16 per CU, each walking ~5 items at B=8 480x640
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                   const float* __restrict__ prelu, float* __restrict__ feat,
                                                   float* __restrict__ guide, int B, int H, int W,
                                                   size_t img_bstride, int chunks, unsigned short* __restrict__ feat16, int twin_f16) {
  const int q = threadIdx.x & 7;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(4 * q + c) * 9 + k];
  const float slope = *prelu;
  // a workgroup walks (row, chunk) items with a grid stride: the 36 weights of a thread are loaded once per workgroup, not once per
  // 128 pixels (19,200 workgroups at B=8 480x640 before)
  // Round 4: the 3 x (128 + 2) image patch of an item goes through LDS once (zero padding written there: no per-lane selects) and every
  // lane reads its nine taps from it -- before, each lane issued 9 global loads per pixel (the eight lanes of a pixel the same nine):
  // 36 vector-memory instructions per thread and item against one 16-byte store per pixel, i.e. the kernel was bound by the
  // address unit, not by its 325 MB of writes.  Two patch buffers (item parity): one barrier per item.
  __shared__ float patch[2][3][STEM_CHUNK + 2];
  const int nitems = B * H * chunks;
  int par = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x, par ^= 1) {
  const int chunk = item % chunks;
  const int row = item / chunks;       // b * H + y
  const int y = row % H, b = row / H;
  const float* base = img + (size_t)b * img_bstride;
  for (int i = threadIdx.x; i < 3 * (STEM_CHUNK + 2); i += 256) {
    const int r = i / (STEM_CHUNK + 2), c = i - r * (STEM_CHUNK + 2);
    const int yy = y + r - 1, xx = chunk * STEM_CHUNK + c - 1;
    const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float v = base[(size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];   // unconditional, clamped
    patch[par][r][c] = ok ? v : 0.f;
  }
  __syncthreads();   // the only barrier of an item: a wave that writes buffer p again (two items later) has passed the NEXT item's barrier,
                     // which every wave reaches only after its reads of p
  float* frow = feat ? feat + (size_t)row * W * 32 : nullptr;   // (NULL with a 16-bit twin, round 6: the fp16 forward keeps the twin only)
  unsigned short* frow16 = feat16 ? feat16 + (size_t)row * W * 32 : nullptr;   // optional 16-bit twin of the map (bf16 / fp16 storage modes)
  float* grow = guide ? guide + (size_t)row * W : nullptr;
#pragma unroll
  for (int it = 0; it < STEM_CHUNK / 32; ++it) {
    const int xi = it * 32 + (threadIdx.x >> 3);
    const int x = chunk * STEM_CHUNK + xi;
    float v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[r * 3 + c] = patch[par][r][xi + c];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) s = fmaf(v[k], wr[c][k], s);
      o[c] = paif::prelu_f(s, slope);
    }
    if (frow && x < W) paif::store_nt(frow + (size_t)x * 32 + q * 4, make_float4(o[0], o[1], o[2], o[3]));
    if (frow16 && x < W)      // the 16-bit twin: bf16, or IEEE fp16 (launch-uniform)
      *reinterpret_cast<uint2*>(frow16 + (size_t)x * 32 + q * 4) = twin_f16 ? paif::f32_to_f16x4(make_float4(o[0], o[1], o[2], o[3]))
                                                                             : paif::f32_to_bf16x4(make_float4(o[0], o[1], o[2], o[3]));
    if (grow) {
      float mx = fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3]));
      float mn = fminf(fminf(o[0], o[1]), fminf(o[2], o[3]));
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, m));
        mn = fminf(mn, __shfl_xor(mn, m));
      }
      if (q == 0 && x < W) grow[x] = mx - mn;
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------
// depthwise k x k (groups = 32), optional ReLU on the input; taps come from L1/L2 (each float4 is
// re-read by the k*k neighbouring pixels' threads of the same or an adjacent wave).
// ---------------------------------------------------------------------------------------------
template <int K, int DIL, int BF = 0>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ out, int in_relu, int B, int H, int W) {
  constexpr int P = DIL * (K - 1) / 2;
  const int q = threadIdx.x & 7;
  float wr[4][K * K];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < K * K; ++k) wr[c][k] = w[(4 * q + c) * K * K + k];
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int xx0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int yy0 = (int)(rowid % H);
    const size_t base = (rowid - yy0) * W * 32 + q * 4;      // element offset (fp32 or bf16 storage: paif_common.h)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // Loads are unconditional on clamped coordinates and padding is a select afterwards: a load under a branch is
    // waited for at the join, which serialised the k*k taps' latencies (1.85 TB/s); one tap row (k loads) is in
    // flight at a time for k >= 5, the whole 3x3 stencil for k == 3.
    constexpr int ROWS_IN_FLIGHT = K == 3 ? 3 : 1;
#pragma unroll
    for (int dy0 = 0; dy0 < K; dy0 += ROWS_IN_FLIGHT) {
      float4 v[ROWS_IN_FLIGHT][K];
#pragma unroll
      for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
        const int yy = min(max(yy0 + (dy0 + r) * DIL - P, 0), H - 1);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int xx = min(max(xx0 + dx * DIL - P, 0), W - 1);
          v[r][dx] = paif::ldq<BF>(x, base + ((size_t)yy * W + xx) * 32);
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS_IN_FLIGHT; ++r) {
        const int dy = dy0 + r;
        const int yy = yy0 + dy * DIL - P;
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const int xx = xx0 + dx * DIL - P;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          float4 t = v[r][dx];
          if (in_relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
          if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
          const int k = dy * K + dx;
          acc.x = fmaf(t.x, wr[0][k], acc.x); acc.y = fmaf(t.y, wr[1][k], acc.y);
          acc.z = fmaf(t.z, wr[2][k], acc.z); acc.w = fmaf(t.w, wr[3][k], acc.w);
        }
      }
    }
    paif::stq_nt<BF>(out, pix * 32 + q * 4, acc);
  }
}

// bf16-stored maps, 3x3 depthwise (dilation DIL): a lane owns 8 channels (16 bytes) of one COLUMN and walks DW_ROWS output rows,
// keeping the 2 DIL + 1 input rows of its three taps in registers -- 3 loads per output instead of 9, and each is one 16-byte piece
// of a 4 KB contiguous row segment per workgroup.  (The per-pixel 9-load form re-read its vertical taps from HBM: PMC traffic
// 1.6 x the map, 153-165 us at B=8 480x640.)
constexpr int DW_ROWS = 24, DW_COLS = 64;
template <int DIL, int F = 1>
__global__ __launch_bounds__(256) void dwconv3_bf16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                           unsigned short* __restrict__ out, int in_relu, int B, int H, int W,
                                                           int ctiles, int strips) {
  constexpr int NR = 2 * DIL + 1;
  const int q = threadIdx.x & 3;
  float wr[8][9];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(8 * q + c) * 9 + k];
  int t = blockIdx.x;
  const int ct = t % ctiles; t /= ctiles;
  const int st = t % strips;
  const int b = t / strips;
  const int xx0 = ct * DW_COLS + (threadIdx.x >> 2);
  const int ybeg = st * DW_ROWS, yend = min(H, ybeg + DW_ROWS);
  const bool colv = xx0 < W;
  const int xc[3] = {min(max(xx0 - DIL, 0), W - 1), min(xx0, W - 1), min(max(xx0 + DIL, 0), W - 1)};
  const bool xok[3] = {xx0 - DIL >= 0, true, xx0 + DIL < W};
  const unsigned short* img = x + (size_t)b * H * W * 32 + q * 8;
  constexpr int AHEAD = 2;   // rows loaded ahead of their first use: the row loop is otherwise one memory latency per row
  constexpr int NW = NR + AHEAD;
  uint4 win[NW][3];      // input rows y - DIL .. y + DIL + AHEAD (window slot s = row y - DIL + s), the three taps each; zero = padding
  auto load_row = [&](int yy, uint4 (&dst)[3]) {
    const int yc = min(max(yy, 0), H - 1);
    const bool rok = yy >= 0 && yy < H;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      uint4 v = *reinterpret_cast<const uint4*>(img + ((size_t)yc * W + xc[dx]) * 32);      // unconditional, clamped
      if (!(rok && xok[dx])) v = make_uint4(0u, 0u, 0u, 0u);
      dst[dx] = v;
    }
  };
#pragma unroll
  for (int s = 0; s < NW - 1; ++s) load_row(ybeg - DIL + s, win[s + 1]);    // slots 1 .. NW-1 hold rows ybeg-DIL .. ; shifted below
  for (int y = ybeg; y < yend; ++y) {
#pragma unroll
    for (int s = 0; s < NW - 1; ++s)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) win[s][dx] = win[s + 1][dx];
    load_row(y + DIL + AHEAD, win[NW - 1]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const uint4 v = win[dy * DIL][dx];
        const int k = dy * 3 + dx;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo, hi;
          if constexpr (F == 2) {
            const paif::f16x2_t h = __builtin_bit_cast(paif::f16x2_t, u[e]);
            lo = (float)h[0]; hi = (float)h[1];
          } else {
            lo = __uint_as_float(u[e] << 16); hi = __uint_as_float(u[e] & 0xffff0000u);
          }
          if (in_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
          acc[2 * e] = fmaf(lo, wr[2 * e][k], acc[2 * e]);
          acc[2 * e + 1] = fmaf(hi, wr[2 * e + 1][k], acc[2 * e + 1]);
        }
      }
    if (colv) {
      const uint2 o0 = paif::f32_to_h4<F>(make_float4(acc[0], acc[1], acc[2], acc[3]));
      const uint2 o1 = paif::f32_to_h4<F>(make_float4(acc[4], acc[5], acc[6], acc[7]));
      typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
      const u32x4_nt ov = {o0.x, o0.y, o1.x, o1.y};
      __builtin_nontemporal_store(ov, reinterpret_cast<u32x4_nt*>(out + (((size_t)b * H + y) * W + xx0) * 32 + q * 8));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ChannelPool(ir, vis): (max_c ir, mean_c ir, max_c vis, mean_c vis) -> float4 per pixel
// ---------------------------------------------------------------------------------------------
template <int BF = 0>
__global__ __launch_bounds__(256) void channel_pool2_kernel(const float* __restrict__ ir, const float* __restrict__ vis,
                                                            float* __restrict__ comp, size_t npix) {
  const int q = threadIdx.x & 7;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 a = paif::ldq_nt<BF>(ir, pix * 32 + q * 4);
    const float4 b = paif::ldq_nt<BF>(vis, pix * 32 + q * 4);
    float mxa = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), sa = (a.x + a.y) + (a.z + a.w);
    float mxb = fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)), sb = (b.x + b.y) + (b.z + b.w);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      mxa = fmaxf(mxa, __shfl_xor(mxa, m)); sa += __shfl_xor(sa, m);
      mxb = fmaxf(mxb, __shfl_xor(mxb, m)); sb += __shfl_xor(sb, m);
    }
    if (q == 0) *reinterpret_cast<float4*>(comp + pix * 4) = make_float4(mxa, sa * (1.0f / 32.0f), mxb, sb * (1.0f / 32.0f));
  }
}

// ChannelPool of ONE map into its half of the interleaved [B,H,W,4] plane (comp pre-offset by 0 or 2): the stand-alone pass behind a
// producer whose epilogue does not pool (paif_conv_desc.cpool); same lane mapping and summation order as channel_pool2_kernel
template <int BF = 0>
__global__ __launch_bounds__(256) void channel_pool1_kernel(const float* __restrict__ x, float* __restrict__ comp, size_t npix) {
  const int q = threadIdx.x & 7;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 a = paif::ldq_nt<BF>(x, pix * 32 + q * 4);
    float mxa = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), sa = (a.x + a.y) + (a.z + a.w);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      mxa = fmaxf(mxa, __shfl_xor(mxa, m)); sa += __shfl_xor(sa, m);
    }
    if (q == 0) *reinterpret_cast<float2*>(comp + pix * 4) = make_float2(mxa, sa * (1.0f / 32.0f));
  }
}

// Cell_Decom.get_residue on an existing feature map: max_c - min_c
__global__ __launch_bounds__(256) void channel_residue_kernel(const float* __restrict__ x, float* __restrict__ g, size_t npix) {
  const int q = threadIdx.x & 7;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const float4 a = *reinterpret_cast<const float4*>(x + pix * 32 + q * 4);
    float mx = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), mn = fminf(fminf(a.x, a.y), fminf(a.z, a.w));
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, m));
      mn = fminf(mn, __shfl_xor(mn, m));
    }
    if (q == 0) g[pix] = mx - mn;
  }
}

// scale = sigmoid(conv5x5 4->1 (comp)); agg = scale*ir + (1-scale)*vis
template <int BF = 0>
__global__ __launch_bounds__(256) void spa_blend_kernel(const float* __restrict__ comp, const float* __restrict__ w,
                                                        const float* __restrict__ ir, const float* __restrict__ vis,
                                                        float* __restrict__ agg, float* __restrict__ scale_out, int B,
                                                        int H, int W) {
  __shared__ float ws[100];
  if (threadIdx.x < 100) ws[threadIdx.x] = w[threadIdx.x];  // [1][4][5][5]
  __syncthreads();
  const int q = threadIdx.x & 7;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const float* cbase = comp + (rowid - y0) * W * 4;
    // 25 taps split over the 8 lanes of the pixel: lane q takes taps q, q+8, q+16, (q+24)
    // Loads are unconditional on clamped coordinates (all four in flight; a load under a branch is waited for at the join) and
    // padding / the missing 4th tap of lanes 1..7 are selects: a masked tap adds fma(0, w, s) = s, the sum is unchanged bit for bit
    float s = 0.f;
    float4 cv[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tap = min(q + 8 * i, 24);
      const int dy = tap / 5, dx = tap - dy * 5;
      const int yy = y0 + dy - 2, xx = x0 + dx - 2;
      ok[i] = q + 8 * i < 25 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      cv[i] = *reinterpret_cast<const float4*>(cbase + ((size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tap = min(q + 8 * i, 24);
      const float4 c = ok[i] ? cv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      s = fmaf(c.x, ws[tap], s); s = fmaf(c.y, ws[25 + tap], s);
      s = fmaf(c.z, ws[50 + tap], s); s = fmaf(c.w, ws[75 + tap], s);
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) s += __shfl_xor(s, m);
    const float sc = 1.0f / (1.0f + expf(-s));
    const float4 a = paif::ldq_nt<BF>(ir, pix * 32 + q * 4);
    const float4 b = paif::ldq_nt<BF>(vis, pix * 32 + q * 4);
    const float om = 1.0f - sc;
    float4 o;
    o.x = __fadd_rn(__fmul_rn(sc, a.x), __fmul_rn(om, b.x));
    o.y = __fadd_rn(__fmul_rn(sc, a.y), __fmul_rn(om, b.y));
    o.z = __fadd_rn(__fmul_rn(sc, a.z), __fmul_rn(om, b.z));
    o.w = __fadd_rn(__fmul_rn(sc, a.w), __fmul_rn(om, b.w));
    paif::stq_nt<BF>(agg, pix * 32 + q * 4, o);
    if (scale_out && q == 0) scale_out[pix] = sc;
  }
}

// Round 6, 16-bit maps: the same arithmetic (bit for bit: the 8 tap-subset partial sums of spa_blend_kernel's lanes and its shuffle
// tree, formed by one thread) on a 4 x 64 pixel tile per workgroup iteration.  spa_blend_kernel has the 8 lanes of a pixel gather the
// 25 taps of the pooled plane from global memory (4 clamped 16-byte gathers per lane) and move the maps 8 bytes per lane: 7 vector-memory
// instructions per wave for 8 pixels -- bound by the address unit (139 us at B=8 480x640 for 511 MB: 3.7 TB/s).  Here the (4+4) x (64+4)
// patch of the pooled plane goes through LDS once (zero padding written there), a thread convolves ONE pixel from it, and the maps
// move 16 bytes per lane (4 lanes per pixel), all eight loads of a thread issued before the convolution.
constexpr int SB_TW = 64, SB_TH = 4;
template <int F>
__global__ __launch_bounds__(256) void spa_blend_tile_kernel(const float* __restrict__ comp, const float* __restrict__ w,
                                                             const float* __restrict__ ir, const float* __restrict__ vis,
                                                             float* __restrict__ agg, int B, int H, int W, int tilesX, int tilesY, int ntiles) {
  typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
  __shared__ float ws[100];
  __shared__ float4 ct[SB_TH + 4][SB_TW + 4];
  __shared__ float sc[SB_TH * SB_TW];
  const int t = threadIdx.x;
  if (t < 100) ws[t] = w[t];
  const unsigned short* ir16 = reinterpret_cast<const unsigned short*>(ir);
  const unsigned short* vis16 = reinterpret_cast<const unsigned short*>(vis);
  unsigned short* agg16 = reinterpret_cast<unsigned short*>(agg);
  const int pc = t >> 2, part = t & 3;                      // blend phase: pixel column of the tile, 8-channel part
  const int cr = t >> 6, cc = t & 63;                       // convolution phase: this thread's pixel of the tile
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tilesX;
    const int r_ = tile / tilesX;
    const int ty = r_ % tilesY, b = r_ / tilesY;
    const int y0 = ty * SB_TH, x0 = tx * SB_TW;
    // the maps: one 16-byte piece per row of the tile and map, all in flight across the convolution (clamped addresses, stores masked)
    u32x4_nt av[SB_TH], bv[SB_TH];
    size_t eo[SB_TH];
    bool ok[SB_TH];
#pragma unroll
    for (int p = 0; p < SB_TH; ++p) {
      const int yy = y0 + p, xx = x0 + pc;
      ok[p] = yy < H && xx < W;
      eo[p] = (((size_t)b * H + min(yy, H - 1)) * W + min(xx, W - 1)) * 32 + part * 8;
      av[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(ir16 + eo[p]));
      bv[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(vis16 + eo[p]));
    }
    const float* cbase = comp + (size_t)b * H * W * 4;
    for (int i = t; i < (SB_TH + 4) * (SB_TW + 4); i += 256) {
      const int r = i / (SB_TW + 4), c = i - r * (SB_TW + 4);
      const int gy = y0 - 2 + r, gx = x0 - 2 + c;
      const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float4 v = *reinterpret_cast<const float4*>(cbase + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 4);
      ct[r][c] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();   // the patch is in (and every wave has left the previous tile's blend phase: `sc` may be rewritten)
    {
      float p8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) p8[q] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int tap = q + 8 * i;
          if (tap < 25) {                                    // lane q of spa_blend_kernel: taps q, q + 8, q + 16, (q + 24), in that order
            const int dy = tap / 5, dx = tap - dy * 5;
            const float4 c4 = ct[cr + dy][cc + dx];
            float s_ = p8[q];
            s_ = fmaf(c4.x, ws[tap], s_); s_ = fmaf(c4.y, ws[25 + tap], s_);
            s_ = fmaf(c4.z, ws[50 + tap], s_); s_ = fmaf(c4.w, ws[75 + tap], s_);
            p8[q] = s_;
          }
        }
      const float s_ = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));   // its xor-shuffle tree
      sc[t] = 1.0f / (1.0f + expf(-s_));
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < SB_TH; ++p) {
      const float s_ = sc[p * SB_TW + pc], om = 1.0f - s_;
      const float4 a0 = paif::h4_to_f32<F>(make_uint2(av[p].x, av[p].y)), a1 = paif::h4_to_f32<F>(make_uint2(av[p].z, av[p].w));
      const float4 b0 = paif::h4_to_f32<F>(make_uint2(bv[p].x, bv[p].y)), b1 = paif::h4_to_f32<F>(make_uint2(bv[p].z, bv[p].w));
      float4 o0, o1;
      o0.x = __fadd_rn(__fmul_rn(s_, a0.x), __fmul_rn(om, b0.x)); o0.y = __fadd_rn(__fmul_rn(s_, a0.y), __fmul_rn(om, b0.y));
      o0.z = __fadd_rn(__fmul_rn(s_, a0.z), __fmul_rn(om, b0.z)); o0.w = __fadd_rn(__fmul_rn(s_, a0.w), __fmul_rn(om, b0.w));
      o1.x = __fadd_rn(__fmul_rn(s_, a1.x), __fmul_rn(om, b1.x)); o1.y = __fadd_rn(__fmul_rn(s_, a1.y), __fmul_rn(om, b1.y));
      o1.z = __fadd_rn(__fmul_rn(s_, a1.z), __fmul_rn(om, b1.z)); o1.w = __fadd_rn(__fmul_rn(s_, a1.w), __fmul_rn(om, b1.w));
      const uint2 u0 = paif::f32_to_h4<F>(o0), u1 = paif::f32_to_h4<F>(o1);
      const u32x4_nt ov = {u0.x, u0.y, u1.x, u1.y};
      if (ok[p]) __builtin_nontemporal_store(ov, reinterpret_cast<u32x4_nt*>(agg16 + eo[p]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ECA tail: mean over HxW from the conv's per-tile partials -> conv1d(k) over channels -> sigmoid
//           -> out = PReLU(o * s[c] + r)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void eca_scale_kernel(const float* __restrict__ partial, const float* __restrict__ w1d, int k,
                                                         int tiles_per_img, float inv_hw, float* __restrict__ s_out) {
  // one block per image: thread (part = tid>>5, c = tid&31) sums tiles part, part+32, ... then the 32 parts
  // are added in a fixed order -> deterministic (no float atomics)
  __shared__ float part_sum[32][32];
  __shared__ float mean[32];
  const int b = blockIdx.x, c = threadIdx.x & 31, part = threadIdx.x >> 5;
  float s = 0.f;
  const float* p = partial + (size_t)b * tiles_per_img * 32 + c;
  for (int t = part; t < tiles_per_img; t += 32) s += p[(size_t)t * 32];
  part_sum[part][c] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += part_sum[i][c];
    mean[c] = tot * inv_hw;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float y = 0.f;
    const int pad = (k - 1) / 2;
    for (int j = 0; j < k; ++j) {
      const int cc = c + j - pad;
      if (cc >= 0 && cc < 32) y = fmaf(mean[cc], w1d[j], y);
    }
    s_out[b * 32 + c] = 1.0f / (1.0f + expf(-y));
  }
}

template <int BF = 0>
__global__ __launch_bounds__(256) void eca_apply_kernel(const float* __restrict__ o, const float* __restrict__ r,
                                                        const float* __restrict__ s, const float* __restrict__ prelu,
                                                        float* __restrict__ out, float* __restrict__ u_out, size_t pix_per_img,
                                                        size_t npix) {
  const int q = threadIdx.x & 7;
  const float slope = *prelu;
  for (size_t pix = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix; pix += (size_t)gridDim.x * 32) {
    const size_t b = pix / pix_per_img;
    const float4 sv = *reinterpret_cast<const float4*>(s + b * 32 + q * 4);
    const float4 ov = paif::ldq_nt<BF>(o, pix * 32 + q * 4);
    const float4 rv = paif::ldq_nt<BF>(r, pix * 32 + q * 4);
    float4 u4;
    u4.x = __fadd_rn(__fmul_rn(ov.x, sv.x), rv.x); u4.y = __fadd_rn(__fmul_rn(ov.y, sv.y), rv.y);
    u4.z = __fadd_rn(__fmul_rn(ov.z, sv.z), rv.z); u4.w = __fadd_rn(__fmul_rn(ov.w, sv.w), rv.w);
    if (u_out) paif::stq_nt<BF>(u_out, pix * 32 + q * 4, u4);
    paif::stq_nt<BF>(out, pix * 32 + q * 4,
                   make_float4(paif::prelu_f(u4.x, slope), paif::prelu_f(u4.y, slope), paif::prelu_f(u4.z, slope), paif::prelu_f(u4.w, slope)));
  }
}

// ---------------------------------------------------------------------------------------------
// tail: 3x3 conv 16->1 + PReLU + tanh.  x NHWC16; 4 lanes per pixel (one float4 each), taps from L1.
// ---------------------------------------------------------------------------------------------
template <int BF = 0>
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ prelu, float* __restrict__ fused,
                                                   float* __restrict__ z_out, int B, int H, int W) {
  const int q = threadIdx.x & 3;
  float wr[4][9];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[c][k] = w[(4 * q + c) * 9 + k];  // w[0][cin][3][3]
  const float slope = *prelu;
  const size_t npix = (size_t)B * H * W;
  for (size_t pix = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2); pix < npix; pix += (size_t)gridDim.x * 64) {
    const int x0 = (int)(pix % W);
    const size_t rowid = pix / W;
    const int y0 = (int)(rowid % H);
    const size_t base = (rowid - y0) * W * 16 + q * 4;       // element offset
    float s = 0.f;
    // all 9 taps in flight: unconditional loads on clamped coordinates, padding by select (see dwconv_kernel)
    float4 v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = min(max(y0 + dy - 1, 0), H - 1), xx = min(max(x0 + dx - 1, 0), W - 1);
        v[dy * 3 + dx] = paif::ldq<BF>(x, base + ((size_t)yy * W + xx) * 16);
      }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int yy = y0 + dy - 1, xx = x0 + dx - 1;
        const int k = dy * 3 + dx;
        float4 t = v[k];
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) t = make_float4(0.f, 0.f, 0.f, 0.f);
        s = fmaf(t.x, wr[0][k], s); s = fmaf(t.y, wr[1][k], s); s = fmaf(t.z, wr[2][k], s); s = fmaf(t.w, wr[3][k], s);
      }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q == 0) {
      fused[pix] = tanhf(paif::prelu_f(s, slope));
      if (z_out) z_out[pix] = s;
    }
  }
}

__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n4,
                           const float* at, const float* bt, float* ot, size_t tail) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < tail) ot[threadIdx.x] = at[threadIdx.x] + bt[threadIdx.x];
}

// elementwise add / casts on bf16-stored maps (4 elements per thread)
template <int F>
__global__ void add_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = paif::ldq<F>(a, i * 4), y = paif::ldq<F>(b, i * 4);
    paif::stq<F>(o, i * 4, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
  }
}
template <int FI, int FO>     // storage formats in / out (0 fp32, 1 bf16, 2 fp16)
__global__ void cast_storage_kernel(const float* __restrict__ a, float* __restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    paif::stq_nt<FO>(o, i * 4, paif::ldq_nt<FI>(a, i * 4));
}

// ---------------------------------------------------------------------------------------------
// glue: cat(fused,Cr,Cb) -> YCrCb2RGB -> clamp -> partial min/max ; then normalise
// ---------------------------------------------------------------------------------------------
constexpr int MM_BLOCK_PIX = 2048;  // pixels per block (fixed: the partial count is part of the ABI)

__global__ __launch_bounds__(256) void recompose_clamp_kernel(const float* __restrict__ fused, const float* __restrict__ ycc,
                                                              float* __restrict__ rgb, float* __restrict__ partial, int nblk,
                                                              int B, size_t HW) {
  const size_t total = (size_t)B * HW;
  const size_t start = (size_t)blockIdx.x * MM_BLOCK_PIX;
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = start + threadIdx.x; i < start + MM_BLOCK_PIX && i < total; i += 256) {
    const size_t b = i / HW, px = i - b * HW;
    const float Y = fused[i];
    const float cr = __fadd_rn(ycc[(b * 3 + 1) * HW + px], -0.5f);
    const float cb = __fadd_rn(ycc[(b * 3 + 2) * HW + px], -0.5f);
    const float y0 = __fadd_rn(Y, 0.0f);
    // (im_flat + bias).mm(mat), mat rows: [1,1,1], [1.403,-0.714,0], [0,-0.344,1.773]; k-ordered fma chain
    float R = fmaf(cb, 0.0f, fmaf(cr, 1.403f, y0 * 1.0f));
    float G = fmaf(cb, -0.344f, fmaf(cr, -0.714f, y0 * 1.0f));
    float Bl = fmaf(cb, 1.773f, fmaf(cr, 0.0f, y0 * 1.0f));
    R = fminf(fmaxf(R, 0.f), 1.f); G = fminf(fmaxf(G, 0.f), 1.f); Bl = fminf(fmaxf(Bl, 0.f), 1.f);
    rgb[(b * 3 + 0) * HW + px] = R; rgb[(b * 3 + 1) * HW + px] = G; rgb[(b * 3 + 2) * HW + px] = Bl;
    mn = fminf(mn, fminf(R, fminf(G, Bl)));
    mx = fmaxf(mx, fmaxf(R, fmaxf(G, Bl)));
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

__global__ __launch_bounds__(256) void minmax_normalize_kernel(const float* __restrict__ rgb, const float* __restrict__ partial,
                                                               int npartial, float* __restrict__ out,
                                                               float* __restrict__ minmax_out, int B, size_t HW) {
  // every block reduces the (small) partial array itself: min/max are order-independent -> exact
  float mn = INFINITY, mx = -INFINITY;
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
  if (minmax_out && blockIdx.x == 0 && threadIdx.x == 0) { minmax_out[0] = mn; minmax_out[1] = mx; }
  const float range = mx - mn;
  const size_t total = (size_t)B * 3 * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i / HW) % 3);
    const float mean = c == 0 ? 123.675f : (c == 1 ? 116.28f : 103.53f);
    const float sd = c == 0 ? 58.395f : (c == 1 ? 57.12f : 57.375f);
    float v = __fdiv_rn(__fsub_rn(rgb[i], mn), range);
    v = __fmul_rn(v, 255.0f);
    out[i] = __fdiv_rn(__fsub_rn(v, mean), sd);
  }
}

// Fused-image writer post-processing (test_original.py:181-197) after the recomposition + clamp: q = uint8(255*rgb)
// (truncation, fp32 product), batch-global min/max of q (= trunc of the float min/max: monotone), (q - mn)/(mx - mn) in
// float64, uint8(255*x) again.  rgb NCHW [B,3,H,W] -> out NHWC uint8 [B,H,W,3].
__global__ __launch_bounds__(256) void fused_uint8_kernel(const float* __restrict__ rgb, const float* __restrict__ partial, int npartial,
                                                          unsigned char* __restrict__ out, int B, size_t HW) {
  float mn = INFINITY, mx = -INFINITY;
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
  const int qmn = (int)(unsigned char)__fmul_rn(255.0f, mn), qmx = (int)(unsigned char)__fmul_rn(255.0f, mx);
  const double range = (double)(qmx - qmn);
  const size_t total = (size_t)B * HW * 3;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const size_t t = i / 3;
    const size_t px = t % HW, b = t / HW;
    const int q = (int)(unsigned char)__fmul_rn(255.0f, rgb[(b * 3 + c) * HW + px]);
    out[i] = (unsigned char)(255.0 * ((double)(q - qmn) / range));
  }
}

}  // namespace

extern "C" {

int paif_fused_uint8_fwd(const float* rgb, const float* minmax_partial, int npartial, unsigned char* out, int B, int H, int W,
                         paif_stream_t stream) {
  PAIF_REQUIRE(rgb && minmax_partial && out && npartial > 0 && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "fused_uint8: bad arguments");
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(fused_uint8_kernel, dim3(grid_for((size_t)B * 3 * HW, 256)), dim3(256), 0, paif::as_stream(stream), rgb,
                     minmax_partial, npartial, out, B, HW);
  PAIF_LAUNCH_CHECK("fused_uint8");
  return 0;
}

int paif_version(void) { return PAIF_ABI_VERSION; }
const char* paif_last_error(void) { return paif::g_err; }

int paif_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    paif::set_error("device_cus: no HIP device");
    return PAIF_EINVAL;
  }
  return prop.multiProcessorCount;
}

int paif_rgb2ycrcb_fwd(const float* rgb, float* ycc, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(rgb && ycc && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "rgb2ycrcb: bad arguments");
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(rgb2ycrcb_kernel, dim3(grid_for(B * HW, 256)), dim3(256), 0, paif::as_stream(stream), rgb, ycc, B, HW);
  PAIF_LAUNCH_CHECK("rgb2ycrcb");
  return 0;
}

int paif_ycrcb2rgb_fwd(const float* ycc, float* rgb, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(ycc && rgb && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "ycrcb2rgb: bad arguments");
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(ycrcb2rgb_kernel, dim3(grid_for(B * HW, 256)), dim3(256), 0, paif::as_stream(stream), ycc, rgb, B, HW);
  PAIF_LAUNCH_CHECK("ycrcb2rgb");
  return 0;
}

static int stem_launch(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* guide,
                       unsigned short* feat16, int twin_f16, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(img && w && prelu && (feat || feat16) && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "stem: bad arguments");
  PAIF_REQUIRE(img_bstride >= (size_t)H * W, PAIF_EINVAL, "stem: batch stride %zu < H*W", img_bstride);
  const int chunks = (W + STEM_CHUNK - 1) / STEM_CHUNK;
  PAIF_REQUIRE((size_t)B * H * chunks < ((size_t)1 << 31), PAIF_EINVAL, "stem: %dx%dx%d is too large for one launch", B, H, W);
  hipLaunchKernelGGL(stem_kernel, dim3((unsigned)min(B * H * chunks, STEM_GRID)), dim3(256), 0, paif::as_stream(stream), img, w,
                     prelu, feat, guide, B, H, W, img_bstride, chunks, feat16, twin_f16);
  PAIF_LAUNCH_CHECK("stem");
  return 0;
}

int paif_stem_fwd(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* guide,
                  int B, int H, int W, paif_stream_t stream) {
  return stem_launch(img, img_bstride, w, prelu, feat, guide, nullptr, 0, B, H, W, stream);
}

int paif_stem_fwd_twin(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* feat_bf16,
                       float* guide, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(feat_bf16, PAIF_EINVAL, "stem_twin: null bf16 map");
  return stem_launch(img, img_bstride, w, prelu, feat, guide, reinterpret_cast<unsigned short*>(feat_bf16), 0, B, H, W, stream);
}

int paif_stem_fwd_twin_f16(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* feat_f16,
                           float* guide, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(feat_f16, PAIF_EINVAL, "stem_twin: null fp16 map");
  return stem_launch(img, img_bstride, w, prelu, feat, guide, reinterpret_cast<unsigned short*>(feat_f16), 1, B, H, W, stream);
}

int paif_dwconv_fwd(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W,
                    paif_stream_t stream) {
  PAIF_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv: bad arguments");
  const dim3 g(grid_for((size_t)B * H * W, 32)), blk(256);
  hipStream_t st = paif::as_stream(stream);
  const int key = k * 10 + dil;
  switch (key) {
    case 31: hipLaunchKernelGGL((dwconv_kernel<3, 1>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    case 32: hipLaunchKernelGGL((dwconv_kernel<3, 2>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    case 51: hipLaunchKernelGGL((dwconv_kernel<5, 1>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    case 52: hipLaunchKernelGGL((dwconv_kernel<5, 2>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    case 71: hipLaunchKernelGGL((dwconv_kernel<7, 1>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    case 72: hipLaunchKernelGGL((dwconv_kernel<7, 2>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    default:
      paif::set_error("dwconv: kernel %d dil %d not built", k, dil);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("dwconv");
  return 0;
}

int paif_channel_pool2_fwd(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(ir && vis && comp && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_pool2: bad arguments");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(channel_pool2_kernel<0>, dim3(grid_for(npix, 32)), dim3(256), 0, paif::as_stream(stream), ir, vis, comp,
                     npix);
  PAIF_LAUNCH_CHECK("channel_pool2");
  return 0;
}

int paif_channel_residue_fwd(const float* x, float* guide, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && guide && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_residue: bad arguments");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(channel_residue_kernel, dim3(grid_for(npix, 32)), dim3(256), 0, paif::as_stream(stream), x, guide, npix);
  PAIF_LAUNCH_CHECK("channel_residue");
  return 0;
}

int paif_spa_blend_fwd(const float* comp, const float* w, const float* ir, const float* vis, float* agg, float* scale_out,
                       int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(comp && w && ir && vis && agg && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "spa_blend: bad arguments");
  hipLaunchKernelGGL(spa_blend_kernel<0>, dim3(grid_for((size_t)B * H * W, 32)), dim3(256), 0, paif::as_stream(stream), comp,
                     w, ir, vis, agg, scale_out, B, H, W);
  PAIF_LAUNCH_CHECK("spa_blend");
  return 0;
}

int paif_eca_finish_fwd(const float* o, const float* r, const float* pool_partial, const float* w1d, int k,
                        const float* prelu, float* gate, float* out, float* u_out, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(o && r && pool_partial && w1d && prelu && gate && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "eca_finish: bad arguments");
  PAIF_REQUIRE(k >= 1 && k <= 9 && (k & 1), PAIF_ENOSUP, "eca_finish: k=%d", k);
  hipStream_t st = paif::as_stream(stream);
  const int tiles_per_img = paif_conv2d_blocks(1, H, W);
  float* s = gate;
  hipLaunchKernelGGL(eca_scale_kernel, dim3(B), dim3(1024), 0, st, pool_partial, w1d, k, tiles_per_img,
                     1.0f / ((float)H * (float)W), s);
  PAIF_LAUNCH_CHECK("eca_scale");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(eca_apply_kernel<0>, dim3(grid_for(npix, 32)), dim3(256), 0, st, o, r, s, prelu, out, u_out, (size_t)H * W, npix);
  PAIF_LAUNCH_CHECK("eca_apply");
  return 0;
}

int paif_tail_fwd(const float* x, const float* w, const float* prelu, float* fused, float* z_out, int B, int H, int W,
                  paif_stream_t stream) {
  PAIF_REQUIRE(x && w && prelu && fused && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "tail: bad arguments");
  hipLaunchKernelGGL(tail_kernel<0>, dim3(grid_for((size_t)B * H * W, 64)), dim3(256), 0, paif::as_stream(stream), x, w, prelu,
                     fused, z_out, B, H, W);
  PAIF_LAUNCH_CHECK("tail");
  return 0;
}

int paif_add_fwd(const float* a, const float* b, float* out, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(a && b && out, PAIF_EINVAL, "add: null pointer");
  if (n == 0) return 0;
  const size_t n4 = n / 4, tail = n - n4 * 4;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n4 ? n4 : 1, 256)), dim3(256), 0, paif::as_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n4,
                     a + n4 * 4, b + n4 * 4, out + n4 * 4, tail);
  PAIF_LAUNCH_CHECK("add");
  return 0;
}

// ---- 16-bit-stored activation maps (bf16: BASELINE configs[1], PAIF_ST_BF16; fp16: PAIF_ST_F16, round 5): the same kernels with
// 8-byte loads / round-to-nearest-even stores; fp32 arithmetic.  x / out / ir / vis / agg / o / r are 16-bit data behind the float*.
// One implementation per kernel, templated on the format F (1 bf16, 2 fp16); the extern "C" twins follow. ----
}  // extern "C"

namespace {

template <int F>
int dwconv16(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && w && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "dwconv(16-bit): bad arguments");
  const dim3 g(grid_for((size_t)B * H * W, 32)), blk(256);
  hipStream_t st = paif::as_stream(stream);
  const int ctiles = (W + DW_COLS - 1) / DW_COLS, strips = (H + DW_ROWS - 1) / DW_ROWS;
  const dim3 g8((unsigned)(B * ctiles * strips));
  const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
  unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
  switch (k * 10 + dil) {
    case 31: hipLaunchKernelGGL((dwconv3_bf16_kernel<1, F>), g8, blk, 0, st, x16, w, o16, in_relu, B, H, W, ctiles, strips); break;
    case 32: hipLaunchKernelGGL((dwconv3_bf16_kernel<2, F>), g8, blk, 0, st, x16, w, o16, in_relu, B, H, W, ctiles, strips); break;
    case 51: hipLaunchKernelGGL((dwconv_kernel<5, 1, F>), g, blk, 0, st, x, w, out, in_relu, B, H, W); break;
    default:
      paif::set_error("dwconv(16-bit): kernel %d dil %d not built", k, dil);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("dwconv(16-bit)");
  return 0;
}

template <int F>
int channel_pool2_16(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(ir && vis && comp && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_pool2(16-bit): bad arguments");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(channel_pool2_kernel<F>, dim3(grid_for(npix, 32)), dim3(256), 0, paif::as_stream(stream), ir, vis, comp, npix);
  PAIF_LAUNCH_CHECK("channel_pool2(16-bit)");
  return 0;
}

template <int F>
int spa_blend16(const float* comp, const float* w, const float* ir, const float* vis, float* agg, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(comp && w && ir && vis && agg && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "spa_blend(16-bit): bad arguments");
  const char* e_ = getenv("PAIF_SPA_TILED");       // PAIF_SPA_TILED=0: the pixel-per-8-lanes kernel (A/B runs and the bit-equality test; read per call)
  const bool tiled = !(e_ && e_[0] == '0');
  const int tilesX = (W + SB_TW - 1) / SB_TW, tilesY = (H + SB_TH - 1) / SB_TH;
  const long long ntiles = (long long)B * tilesX * tilesY;
  if (tiled && ntiles < (1ll << 31)) {
    hipLaunchKernelGGL(spa_blend_tile_kernel<F>, dim3((unsigned)(ntiles < 4096 ? ntiles : 4096)), dim3(256), 0, paif::as_stream(stream), comp, w, ir,
                       vis, agg, B, H, W, tilesX, tilesY, (int)ntiles);
    PAIF_LAUNCH_CHECK("spa_blend(16-bit, tiled)");
    return 0;
  }
  hipLaunchKernelGGL(spa_blend_kernel<F>, dim3(grid_for((size_t)B * H * W, 32)), dim3(256), 0, paif::as_stream(stream), comp, w, ir, vis, agg,
                     (float*)nullptr, B, H, W);
  PAIF_LAUNCH_CHECK("spa_blend(16-bit)");
  return 0;
}

template <int F>
int eca_finish16(const float* o, const float* r, const float* pool_partial, const float* w1d, int k, const float* prelu, float* gate,
                 float* out, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(o && r && pool_partial && w1d && prelu && gate && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "eca_finish(16-bit): bad arguments");
  PAIF_REQUIRE(k >= 1 && k <= 9 && (k & 1), PAIF_ENOSUP, "eca_finish: k=%d", k);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(eca_scale_kernel, dim3(B), dim3(1024), 0, st, pool_partial, w1d, k, paif_conv2d_blocks(1, H, W), 1.0f / ((float)H * (float)W), gate);
  PAIF_LAUNCH_CHECK("eca_scale");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(eca_apply_kernel<F>, dim3(grid_for(npix, 32)), dim3(256), 0, st, o, r, gate, prelu, out, (float*)nullptr, (size_t)H * W, npix);
  PAIF_LAUNCH_CHECK("eca_apply(16-bit)");
  return 0;
}

template <int F>
int tail16(const float* x, const float* w, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && w && prelu && fused && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "tail(16-bit): bad arguments");
  hipLaunchKernelGGL(tail_kernel<F>, dim3(grid_for((size_t)B * H * W, 64)), dim3(256), 0, paif::as_stream(stream), x, w, prelu, fused,
                     (float*)nullptr, B, H, W);
  PAIF_LAUNCH_CHECK("tail(16-bit)");
  return 0;
}

template <int F>
int add16(const float* a, const float* b, float* out, size_t n, paif_stream_t stream) {
  PAIF_REQUIRE(a && b && out && n % 4 == 0, PAIF_EINVAL, "add(16-bit): null pointer or n %% 4 != 0");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_bf16_kernel<F>, dim3(grid_for(n / 4, 256)), dim3(256), 0, paif::as_stream(stream), a, b, out, n / 4);
  PAIF_LAUNCH_CHECK("add(16-bit)");
  return 0;
}

}  // namespace

extern "C" {

static int channel_pool1_launch(const float* x, float* comp_off, int B, int H, int W, int fmt, paif_stream_t stream) {
  PAIF_REQUIRE(x && comp_off && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_pool1: bad arguments");
  const size_t npix = (size_t)B * H * W;
  const dim3 g(grid_for(npix, 32)), blk(256);
  hipStream_t st = paif::as_stream(stream);
  if (fmt == 0) hipLaunchKernelGGL(channel_pool1_kernel<0>, g, blk, 0, st, x, comp_off, npix);
  else if (fmt == 1) hipLaunchKernelGGL(channel_pool1_kernel<1>, g, blk, 0, st, x, comp_off, npix);
  else hipLaunchKernelGGL(channel_pool1_kernel<2>, g, blk, 0, st, x, comp_off, npix);
  PAIF_LAUNCH_CHECK("channel_pool1");
  return 0;
}
int paif_channel_pool1_fwd(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream) { return channel_pool1_launch(x, comp_off, B, H, W, 0, stream); }
int paif_channel_pool1_fwd_bf16(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream) { return channel_pool1_launch(x, comp_off, B, H, W, 1, stream); }
int paif_channel_pool1_fwd_f16(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream) { return channel_pool1_launch(x, comp_off, B, H, W, 2, stream); }

int paif_dwconv_fwd_bf16(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W, paif_stream_t stream) {
  return dwconv16<1>(x, w, out, k, dil, in_relu, B, H, W, stream);
}
int paif_dwconv_fwd_f16(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W, paif_stream_t stream) {
  return dwconv16<2>(x, w, out, k, dil, in_relu, B, H, W, stream);
}
int paif_channel_pool2_fwd_bf16(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream) {
  return channel_pool2_16<1>(ir, vis, comp, B, H, W, stream);
}
int paif_channel_pool2_fwd_f16(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream) {
  return channel_pool2_16<2>(ir, vis, comp, B, H, W, stream);
}
int paif_spa_blend_fwd_bf16(const float* comp, const float* w, const float* ir, const float* vis, float* agg, int B, int H, int W,
                            paif_stream_t stream) {
  return spa_blend16<1>(comp, w, ir, vis, agg, B, H, W, stream);
}
int paif_spa_blend_fwd_f16(const float* comp, const float* w, const float* ir, const float* vis, float* agg, int B, int H, int W,
                           paif_stream_t stream) {
  return spa_blend16<2>(comp, w, ir, vis, agg, B, H, W, stream);
}
int paif_eca_finish_fwd_bf16(const float* o, const float* r, const float* pool_partial, const float* w1d, int k, const float* prelu,
                             float* gate, float* out, int B, int H, int W, paif_stream_t stream) {
  return eca_finish16<1>(o, r, pool_partial, w1d, k, prelu, gate, out, B, H, W, stream);
}
int paif_eca_finish_fwd_f16(const float* o, const float* r, const float* pool_partial, const float* w1d, int k, const float* prelu,
                            float* gate, float* out, int B, int H, int W, paif_stream_t stream) {
  return eca_finish16<2>(o, r, pool_partial, w1d, k, prelu, gate, out, B, H, W, stream);
}
int paif_tail_fwd_bf16(const float* x, const float* w, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream) {
  return tail16<1>(x, w, prelu, fused, B, H, W, stream);
}
int paif_tail_fwd_f16(const float* x, const float* w, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream) {
  return tail16<2>(x, w, prelu, fused, B, H, W, stream);
}
int paif_add_fwd_bf16(const float* a, const float* b, float* out, size_t n, paif_stream_t stream) { return add16<1>(a, b, out, n, stream); }
int paif_add_fwd_f16(const float* a, const float* b, float* out, size_t n, paif_stream_t stream) { return add16<2>(a, b, out, n, stream); }

/* mode: 0 bf16 -> fp32, 1 fp32 -> bf16, 2 fp32 -> fp16, 3 fp16 -> fp32 (round to nearest even on the way down) */
int paif_cast_storage_fwd(const float* src, float* dst, size_t n, int mode, paif_stream_t stream) {
  PAIF_REQUIRE(src && dst && n % 4 == 0, PAIF_EINVAL, "cast_storage: null pointer or n %% 4 != 0");
  PAIF_REQUIRE(mode >= 0 && mode <= 3, PAIF_EINVAL, "cast_storage: mode=%d", mode);
  if (n == 0) return 0;
  const dim3 g(grid_for(n / 4, 256)), blk(256);
  hipStream_t st = paif::as_stream(stream);
  switch (mode) {
    case 0: hipLaunchKernelGGL((cast_storage_kernel<1, 0>), g, blk, 0, st, src, dst, n / 4); break;
    case 1: hipLaunchKernelGGL((cast_storage_kernel<0, 1>), g, blk, 0, st, src, dst, n / 4); break;
    case 2: hipLaunchKernelGGL((cast_storage_kernel<0, 2>), g, blk, 0, st, src, dst, n / 4); break;
    default: hipLaunchKernelGGL((cast_storage_kernel<2, 0>), g, blk, 0, st, src, dst, n / 4); break;
  }
  PAIF_LAUNCH_CHECK("cast_storage");
  return 0;
}

int paif_minmax_blocks(int B, int H, int W) {
  const size_t total = (size_t)B * H * W;
  return (int)((total + MM_BLOCK_PIX - 1) / MM_BLOCK_PIX);
}

int paif_recompose_clamp_fwd(const float* fused, const float* ycc, float* rgb_out, float* minmax_partial, int B, int H,
                             int W, paif_stream_t stream) {
  PAIF_REQUIRE(fused && ycc && rgb_out && minmax_partial && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "recompose_clamp: bad arguments");
  const int nblk = paif_minmax_blocks(B, H, W);
  hipLaunchKernelGGL(recompose_clamp_kernel, dim3(nblk), dim3(256), 0, paif::as_stream(stream), fused, ycc, rgb_out,
                     minmax_partial, nblk, B, (size_t)H * W);
  PAIF_LAUNCH_CHECK("recompose_clamp");
  return 0;
}

int paif_minmax_normalize_fwd(const float* rgb, const float* minmax_partial, int npartial, float* out, float* minmax_out,
                              int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(rgb && minmax_partial && out && npartial > 0 && B > 0 && H > 0 && W > 0, PAIF_EINVAL,
               "minmax_normalize: bad arguments");
  const size_t HW = (size_t)H * W;
  hipLaunchKernelGGL(minmax_normalize_kernel, dim3(grid_for((size_t)B * 3 * HW, 256)), dim3(256), 0, paif::as_stream(stream),
                     rgb, minmax_partial, npartial, out, minmax_out, B, HW);
  PAIF_LAUNCH_CHECK("minmax_normalize");
  return 0;
}

}  // extern "C"
