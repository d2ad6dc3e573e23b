// Guided filter (radius 4, 1-channel guide, 32-channel NHWC target), both eps of
// Cell_Decom.decomposition (core/model_fusion_auto.py:522-535; third-party
// guided_filter_pytorch.GuidedFilter -- algorithm per oracle/shims/guided_filter_pytorch).
//
// HBM-bound.  Row-streaming separable 9x9 box sums:
//   workgroup = 32 loaded columns (24 output columns + 4 halo each side) x 8 channel quads;
//   it walks down a segment of rows keeping the last 9 rows in REGISTERS (static ring, loop
//   unrolled by 9) -> vertical 9-term sums are direct (no running add/subtract drift);
//   the vertical sums of one row go to LDS (double-buffered by row parity, one barrier per row)
//   and each thread adds its 9 horizontal neighbours.
// Box sums are exact border-clipped window sums; N = (#valid rows) x (#valid cols).
// fp32 throughout (SURVEY.md section 7 hard part 1: A = cov/(var+eps) with eps = 1e-4).
#include <stdlib.h>
#include <string.h>

#include "paif_common.h"

namespace {

constexpr int R = 4;
constexpr int K = 2 * R + 1;   // 9
constexpr int NCOL = 32;       // loaded columns per workgroup
constexpr int OCOL = NCOL - 2 * R;  // 24 output columns
constexpr int ROWS_PER_SEG = 60;

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// ---- stage 1: A_e = cov/(var+eps_e), b_e = mean_y - A_e*mean_g ---------------------------------
__global__ __launch_bounds__(256) void gf_ab_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                    float* __restrict__ ab, float eps0, float eps1, int B, int H, int W,
                                                    int nstrip, int nseg) {
  __shared__ float4 sy[2][NCOL][8];
  __shared__ float4 sgy[2][NCOL][8];
  __shared__ float2 sg[2][NCOL];
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  int t = blockIdx.x;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int col = strip * OCOL - R + xi;          // image column of this thread
  const bool colin = col >= 0 && col < W;
  const int ybeg = seg * ROWS_PER_SEG, yend = min(H, ybeg + ROWS_PER_SEG);
  const size_t img = (size_t)b * H * W;
  const size_t plane = (size_t)B * H * W * 32;

  float4 ry[K];
  float rg[K];
  const int r0 = ybeg - R, r1 = yend + R;  // rows streamed: [r0, r1)
  for (int rr = r0; rr < r1; rr += K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int row = rr + k;
      if (row >= r1) break;  // block-uniform
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      float g = 0.f;
      if (colin && row >= 0 && row < H) {
        const size_t px = img + (size_t)row * W + col;
        v = *reinterpret_cast<const float4*>(y + px * 32 + q * 4);
        g = guide[px];
      }
      ry[k] = v;
      rg[k] = g;
      const int orow = row - R;  // output row whose 9-row window is now complete
      if (orow < ybeg) continue;  // block-uniform (ring not full yet)
      float4 vy = make_float4(0.f, 0.f, 0.f, 0.f), vgy = vy;
      float vg = 0.f, vgg = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        vy = f4add(vy, ry[j]);
        vgy = f4fma(rg[j], ry[j], vgy);
        vg += rg[j];
        vgg = fmaf(rg[j], rg[j], vgg);
      }
      const int buf = orow & 1;
      sy[buf][xi][q] = vy;
      sgy[buf][xi][q] = vgy;
      if (q == 0) sg[buf][xi] = make_float2(vg, vgg);
      __syncthreads();
      if (xi >= R && xi < NCOL - R && col < W) {
        float4 by = make_float4(0.f, 0.f, 0.f, 0.f), bgy = by;
        float bg = 0.f, bgg = 0.f;
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          by = f4add(by, sy[buf][xi + j][q]);
          bgy = f4add(bgy, sgy[buf][xi + j][q]);
          const float2 gg = sg[buf][xi + j];
          bg += gg.x;
          bgg += gg.y;
        }
        const int cy = min(orow + R, H - 1) - max(orow - R, 0) + 1;
        const int cx = min(col + R, W - 1) - max(col - R, 0) + 1;
        const float n = (float)(cy * cx);
        const float mg = bg / n;
        const float var = bgg / n - mg * mg;
        const float4 my = make_float4(by.x / n, by.y / n, by.z / n, by.w / n);
        const float4 cov = make_float4(bgy.x / n - mg * my.x, bgy.y / n - mg * my.y, bgy.z / n - mg * my.z,
                                       bgy.w / n - mg * my.w);
        const size_t o = (img + (size_t)orow * W + col) * 32 + q * 4;
        const float d0 = var + eps0, d1 = var + eps1;
        const float4 A0 = make_float4(cov.x / d0, cov.y / d0, cov.z / d0, cov.w / d0);
        const float4 A1 = make_float4(cov.x / d1, cov.y / d1, cov.z / d1, cov.w / d1);
        *reinterpret_cast<float4*>(ab + o) = A0;
        *reinterpret_cast<float4*>(ab + plane + o) =
            make_float4(my.x - A0.x * mg, my.y - A0.y * mg, my.z - A0.z * mg, my.w - A0.w * mg);
        *reinterpret_cast<float4*>(ab + 2 * plane + o) = A1;
        *reinterpret_cast<float4*>(ab + 3 * plane + o) =
            make_float4(my.x - A1.x * mg, my.y - A1.y * mg, my.z - A1.z * mg, my.w - A1.w * mg);
      }
    }
  }
}

// ---- stage 2: LF_e = box(A_e)/N * g + box(b_e)/N ; blockIdx.y = e -------------------------------
__global__ __launch_bounds__(256) void gf_lf_kernel(const float* __restrict__ guide, const float* __restrict__ ab,
                                                    float* __restrict__ lf, int B, int H, int W, int nstrip, int nseg) {
  __shared__ float4 sa[2][NCOL][8];
  __shared__ float4 sb[2][NCOL][8];
  const int q = threadIdx.x & 7, xi = threadIdx.x >> 3;
  int t = blockIdx.x;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int e = blockIdx.y;
  const int col = strip * OCOL - R + xi;
  const bool colin = col >= 0 && col < W;
  const int ybeg = seg * ROWS_PER_SEG, yend = min(H, ybeg + ROWS_PER_SEG);
  const size_t img = (size_t)b * H * W;
  const size_t plane = (size_t)B * H * W * 32;
  const float* Ap = ab + (size_t)(2 * e) * plane;
  const float* Bp = ab + (size_t)(2 * e + 1) * plane;
  float* out = lf + (size_t)e * plane;

  float4 ra[K], rb[K];
  const int r0 = ybeg - R, r1 = yend + R;
  for (int rr = r0; rr < r1; rr += K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int row = rr + k;
      if (row >= r1) break;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (colin && row >= 0 && row < H) {
        const size_t o = (img + (size_t)row * W + col) * 32 + q * 4;
        va = *reinterpret_cast<const float4*>(Ap + o);
        vb = *reinterpret_cast<const float4*>(Bp + o);
      }
      ra[k] = va;
      rb[k] = vb;
      const int orow = row - R;
      if (orow < ybeg) continue;
      float4 ua = make_float4(0.f, 0.f, 0.f, 0.f), ub = ua;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        ua = f4add(ua, ra[j]);
        ub = f4add(ub, rb[j]);
      }
      const int buf = orow & 1;
      sa[buf][xi][q] = ua;
      sb[buf][xi][q] = ub;
      __syncthreads();
      if (xi >= R && xi < NCOL - R && col < W) {
        float4 ba = make_float4(0.f, 0.f, 0.f, 0.f), bb = ba;
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          ba = f4add(ba, sa[buf][xi + j][q]);
          bb = f4add(bb, sb[buf][xi + j][q]);
        }
        const int cy = min(orow + R, H - 1) - max(orow - R, 0) + 1;
        const int cx = min(col + R, W - 1) - max(col - R, 0) + 1;
        const float n = (float)(cy * cx);
        const size_t px = img + (size_t)orow * W + col;
        const float g = guide[px];
        *reinterpret_cast<float4*>(out + px * 32 + q * 4) =
            make_float4(fmaf(ba.x / n, g, bb.x / n), fmaf(ba.y / n, g, bb.y / n), fmaf(ba.z / n, g, bb.z / n),
                        fmaf(ba.w / n, g, bb.w / n));
      }
    }
  }
}

}  // namespace

extern "C" {

int paif_guided_filter_ab_fwd(const float* guide, const float* y, float* ab, float eps0, float eps1, int B, int H, int W,
                              paif_stream_t stream) {
  PAIF_REQUIRE(guide && y && ab && B > 0, PAIF_EINVAL, "guided_filter_ab: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter: H,W must exceed 2r+1 = %d (got %dx%d)", K, H, W);
  const int nstrip = (W + OCOL - 1) / OCOL, nseg = (H + ROWS_PER_SEG - 1) / ROWS_PER_SEG;
  hipLaunchKernelGGL(gf_ab_kernel, dim3(B * nstrip * nseg), dim3(256), 0, paif::as_stream(stream), guide, y, ab, eps0, eps1,
                     B, H, W, nstrip, nseg);
  PAIF_LAUNCH_CHECK("guided_filter_ab");
  return 0;
}

int paif_guided_filter_lf_fwd(const float* guide, const float* ab, float* lf, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(guide && ab && lf && B > 0, PAIF_EINVAL, "guided_filter_lf: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter: H,W must exceed 2r+1 = %d (got %dx%d)", K, H, W);
  const int nstrip = (W + OCOL - 1) / OCOL, nseg = (H + ROWS_PER_SEG - 1) / ROWS_PER_SEG;
  hipLaunchKernelGGL(gf_lf_kernel, dim3(B * nstrip * nseg, 2), dim3(256), 0, paif::as_stream(stream), guide, ab, lf, B, H,
                     W, nstrip, nseg);
  PAIF_LAUNCH_CHECK("guided_filter_lf");
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Fused guided filter (inference path): stage 1 and stage 2 in ONE kernel, the coefficient maps A, b never
// touch HBM (the two-kernel form writes 4 and re-reads 4 x 1.33 thirty-two-channel maps per stream).
// Two-level row streaming: ring 1 = last 9 rows of (y, g) in registers -> vertical sums -> LDS -> horizontal sums
// -> (A_e, b_e) of row i; ring 2 = last 9 rows of (A_e, b_e) in registers -> vertical sums -> LDS -> horizontal sums
// -> LF_e of row i-4.  One eps per blockIdx.y (the register budget holds one (A, b) ring).
// Workgroup = 64 loaded columns x 8 channel quads (512 threads); 56 columns carry valid stage-1 values,
// 48 columns valid outputs.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int FC = 64;             // loaded columns
constexpr int FO = FC - 4 * R;     // 48 output columns
constexpr int PF = 3;                  // rows of (y, g) loads kept in flight ahead of the row being processed

// LDS-only barrier: the rows' output stores and the prefetched loads stay in flight across it (a __syncthreads()
// fence would drain vmcnt; nothing in this kernel communicates through global memory).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Per-pixel guide statistics of the fused kernel, precomputed once per stream (1-channel work: 3 % of the traffic):
//   gs[e][px] = (mean_g, 1 / (var_g + eps_e))  with the border-clipped 9x9 window -- direct sums, fp32.
// In the fused kernel these were recomputed by all 8 channel-quad lanes of every column in every workgroup (17 % of
// its VALU instructions, a fifth of its LDS operations and one of its three divisions per row).
constexpr int GT = 32, GH = GT + 2 * R;   // 32 x 32 output tile, 40 x 40 halo tile
// Output: four [B*H*W] planes -- mean_g, 1/(var_g + eps0), 1/(var_g + eps1), 1/n -- each read as 16-byte column groups by the
// matrix-core kernel (gf_mfma2.hip); block 0 also clears the f16-range flag behind the planes.
__global__ __launch_bounds__(256) void gf_guide_stats_kernel(const float* __restrict__ guide, float* __restrict__ gs, unsigned* __restrict__ flag,
                                                             float eps0, float eps1, int B, int H, int W, int tilesX, int tilesY) {
  __shared__ float sG[GH][GH + 1];
  __shared__ float sH[GH][GT + 1], sH2[GH][GT + 1];
  int t = blockIdx.x;
  if (t == 0 && threadIdx.x == 0) *flag = 0u;
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int x0 = tx * GT - R, y0 = ty * GT - R;
  const float* base = guide + (size_t)b * H * W;
  for (int i = threadIdx.x; i < GH * GH; i += 256) {
    const int r = i / GH, c = i - r * GH;
    const int gy = y0 + r, gx = x0 + c;
    const float v = base[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];   // unconditional load, zero by select
    sG[r][c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? v : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GH * GT; i += 256) {      // horizontal 9-sums of g and g^2
    const int r = i / GT, c = i - r * GT;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { const float g = sG[r][c + j]; s += g; ss = fmaf(g, g, ss); }
    sH[r][c] = s; sH2[r][c] = ss;
  }
  __syncthreads();
  const size_t npix = (size_t)B * H * W;
  for (int i = threadIdx.x; i < GT * GT; i += 256) {      // vertical 9-sums -> statistics
    const int r = i / GT, c = i - r * GT;
    const int yy = ty * GT + r, xx = tx * GT + c;
    if (yy >= H || xx >= W) continue;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { s += sH[r + j][c]; ss += sH2[r + j][c]; }
    const int cy = min(yy + R, H - 1) - max(yy - R, 0) + 1;
    const int cx = min(xx + R, W - 1) - max(xx - R, 0) + 1;
    const float rn = 1.0f / (float)(cy * cx);
    const float mg = s * rn;
    const float var = ss * rn - mg * mg;
    const size_t pix = (size_t)b * H * W + (size_t)yy * W + xx;
    gs[pix] = mg;
    gs[npix + pix] = 1.0f / (var + eps0);
    gs[2 * npix + pix] = 1.0f / (var + eps1);
    gs[3 * npix + pix] = rn;
  }
}

// Generic over the per-thread channel vector V (float4 or float2) and NL = vectors per pixel handled by one workgroup
// (blockIdx.z walks the 32 / (VW * NL) channel groups).  Measured at B=8, 480x640 (one launch, both eps), V = float4:
// NL = 8 (one 512-thread workgroup per CU) 0.844 ms, NL = 4 (2 x 256 threads) 0.912 ms, NL = 2 0.947 ms -- splitting the
// barrier group does not help.  PMC (profiles/r02_pmc_fusion.txt): waves are PARKED (s_waitcnt / barrier) 46 % of their
// cycles, issue-stalled 21 %, issuing 33 %; with ~250 VGPRs per thread only 8 waves fit a CU.  V = float2 halves the
// register rings (16 waves per CU) at the price of repeating the per-column scalar work for twice as many threads.
// PAIF_GF_FORM selects the form for A/B runs: "4x8" (default), "4x4", "4x2", "2x16".
template <typename V> struct VecOps;
template <> struct VecOps<float4> {
  static constexpr int VW = 4;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct VecOps<float2> {
  static constexpr int VW = 2;
  static __device__ __forceinline__ float2 zero() { return make_float2(0.f, 0.f); }
};
template <typename V> __device__ __forceinline__ V vadd(V a, V b) {
  V r; float* pr = reinterpret_cast<float*>(&r); const float* pa = reinterpret_cast<const float*>(&a); const float* pb = reinterpret_cast<const float*>(&b);
#pragma unroll
  for (int i = 0; i < VecOps<V>::VW; ++i) pr[i] = pa[i] + pb[i];
  return r;
}
template <typename V> __device__ __forceinline__ V vfma(float s, V a, V c) {
  V r; float* pr = reinterpret_cast<float*>(&r); const float* pa = reinterpret_cast<const float*>(&a); const float* pc = reinterpret_cast<const float*>(&c);
#pragma unroll
  for (int i = 0; i < VecOps<V>::VW; ++i) pr[i] = fmaf(s, pa[i], pc[i]);
  return r;
}

template <typename V, int NL>
__global__ __launch_bounds__(FC * NL) void gf_fused_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                           const float* __restrict__ gs, const unsigned* __restrict__ only_if, float* __restrict__ lf,
                                                           int B, int H, int W, int nstrip, int nseg, int frows, int out_bf16) {
  constexpr int VW = VecOps<V>::VW;
  // out_bf16 == 3 (round 6): as 2 (fp16 high-frequency output), with y an IEEE fp16 map (the stem's 16-bit twin)
  const bool y_f16 = out_bf16 == 3;
  if (y_f16) out_bf16 = 2;
  const unsigned short* const y16 = reinterpret_cast<const unsigned short*>(y) + (blockIdx.z * NL + threadIdx.x % NL) * VW;
  unsigned short* const lf16 = reinterpret_cast<unsigned short*>(lf);   // out_bf16 (1 bf16 LF, 2 fp16 HF = y - LF): the same element indices in a 16-bit buffer
  if (only_if != nullptr && *only_if == 0u) return;      // fallback launch behind the matrix-core kernel: runs only when it raised the flag
  __shared__ V s_a[2][FC][NL];
  __shared__ V s_b[2][FC][NL];
  const int xi = threadIdx.x / NL;
  const int ql = threadIdx.x - xi * NL;                       // vector slot in the LDS rows; channel = (blockIdx.z * NL + ql) * VW
  y += (blockIdx.z * NL + ql) * VW;
  lf += (blockIdx.z * NL + ql) * VW;
  auto ldy = [&](size_t px) -> V {                            // the lane's VW channels of pixel px (launch-uniform branch)
    if (y_f16) {
      V r;
      float* pr = reinterpret_cast<float*>(&r);
#pragma unroll
      for (int i = 0; i < VW; ++i) pr[i] = (float)__builtin_bit_cast(_Float16, y16[px * 32 + i]);
      return r;
    }
    return *reinterpret_cast<const V*>(y + px * 32);
  };
  int t = blockIdx.x;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int e = blockIdx.y;
  const int col = strip * FO - 2 * R + xi;
  const bool colin = col >= 0 && col < W;
  const int ybeg = seg * frows, yend = min(H, ybeg + frows);   // frows = output rows per workgroup (chosen at launch)
  const size_t img = (size_t)b * H * W;
  float* out = lf + (size_t)e * ((size_t)B * H * W * 32);
  unsigned short* out16 = lf16 + (blockIdx.z * NL + ql) * VW + (size_t)e * ((size_t)B * H * W * 32);
  const float* gmean = gs;                                            // planes: mean_g, 1/(var+eps0), 1/(var+eps1), 1/n
  const float* grden = gs + (size_t)(1 + e) * ((size_t)B * H * W);

  // 1/n of the border-clipped window, n = cy * cx: cx is fixed per thread and cy takes the values 5..9 -- the five
  // reciprocals are formed once (exact divisions, same values as before) instead of one division per row and stage
  const int cx = min(col + R, W - 1) - max(col - R, 0) + 1;
  float rnt[R + 1];
#pragma unroll
  for (int i = 0; i <= R; ++i) rnt[i] = 1.0f / (float)((K - i) * max(cx, 1));
  auto rn_of = [&](int row) -> float {
    const int miss = K - (min(row + R, H - 1) - max(row - R, 0) + 1);    // rows clipped away: 0..4
    float r = rnt[0];
#pragma unroll
    for (int i = 1; i <= R; ++i) r = miss == i ? rnt[i] : r;
    return r;
  };

  // ring 1: the last 9 rows of y and of the guide.  ring 2 holds the (A, b) rows as 3-row partial sums:
  //   tA[j] = A(i) + A(i+1) + A(i+2);   a 9-row window = three of them 3 rows apart (4 adds per row instead of 8, still a
  //   plain sum of the nine terms -- no running add/subtract, no drift)
  V ry[K], tA[K], tB[K];
  float rg[K];
  V a1 = VecOps<V>::zero(), a2 = a1, b1 = a1, b2 = a1;   // the two previous (A, b) rows
#pragma unroll
  for (int k = 0; k < K; ++k) { tA[k] = VecOps<V>::zero(); tB[k] = tA[k]; }
  const int r0 = ybeg - 2 * R, r1 = yend + 2 * R;   // streamed input rows [r0, r1)
  // Loads are unconditional on a clamped address (a load under a divergent branch makes hipcc wait for it at the
  // join); out-of-image values are zeroed when they enter the ring.
  const int colc = min(max(col, 0), W - 1);
  V pv[PF];
  float pg[PF];
  float2 ps[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const size_t px = img + (size_t)min(max(r0 + p, 0), H - 1) * W + colc;
    pv[p] = ldy(px);
    pg[p] = guide[px];
    const size_t sx = img + (size_t)min(max(r0 + p - R, 0), H - 1) * W + colc;
    ps[p] = make_float2(gmean[sx], grden[sx]);
  }
  for (int rr = r0; rr < r1; rr += K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int row = rr + k;
      if (row >= r1) break;   // block-uniform
      const bool in = colin && row >= 0 && row < H;
      ry[k] = in ? pv[k % PF] : VecOps<V>::zero();
      rg[k] = in ? pg[k % PF] : 0.f;
      const float2 st = ps[k % PF];             // (mean_g, 1/(var_g + eps)) of (row - R, col)
      {
        const size_t px = img + (size_t)min(max(row + PF, 0), H - 1) * W + colc;
        pv[k % PF] = ldy(px);
        pg[k % PF] = guide[px];
        const size_t sx = img + (size_t)min(max(row + PF - R, 0), H - 1) * W + colc;
        ps[k % PF] = make_float2(gmean[sx], grden[sx]);
      }
      const int irow = row - R;               // stage-1 row whose 9-row window is complete
      if (irow < ybeg - R) continue;          // block-uniform
      // ---- stage 1: vertical sums -> LDS -> horizontal sums -> (A, b) of (irow, col) ----
      {
        V vy = VecOps<V>::zero(), vgy = vy;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          vy = vadd(vy, ry[j]);
          vgy = vfma(rg[j], ry[j], vgy);
        }
        s_a[0][xi][ql] = vy;
        s_b[0][xi][ql] = vgy;
      }
      lds_barrier();
      V A = VecOps<V>::zero(), Bc = A;
      if (xi >= R && xi < FC - R && colin && irow >= 0 && irow < H) {   // outside the image the coefficients are zero padding
        V by = VecOps<V>::zero(), bgy = by;
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          by = vadd(by, s_a[0][xi + j][ql]);
          bgy = vadd(bgy, s_b[0][xi + j][ql]);
        }
        const float rn = rn_of(irow);
        const float mg = st.x, rden = st.y;
        const float* pby = reinterpret_cast<const float*>(&by);
        const float* pbgy = reinterpret_cast<const float*>(&bgy);
        float* pA = reinterpret_cast<float*>(&A);
        float* pB = reinterpret_cast<float*>(&Bc);
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          const float my = pby[i] * rn;
          pA[i] = (pbgy[i] * rn - mg * my) * rden;
          pB[i] = my - pA[i] * mg;
        }
      }
      // 3-row partial sums of ring 2: slot k <- rows (irow-2, irow-1, irow)
      tA[k] = vadd(vadd(a2, a1), A);
      tB[k] = vadd(vadd(b2, b1), Bc);
      a2 = a1; a1 = A; b2 = b1; b1 = Bc;
      const int orow = irow - R;              // output row whose 9-row (A, b) window is complete
      // the first ring-2 window is complete once stage-1 rows ybeg-R .. ybeg+R are in
      if (orow < ybeg) { lds_barrier(); continue; }   // block-uniform; the barrier keeps s_*[0] safe for the next row
      // ---- stage 2: vertical sums of (A, b) -> LDS -> horizontal sums -> LF ----
      {
        // window rows irow-8 .. irow = partial sums ending at irow, irow-3, irow-6 (slots k, k-3, k-6 mod 9)
        s_a[1][xi][ql] = vadd(vadd(tA[k], tA[(k + 6) % K]), tA[(k + 3) % K]);
        s_b[1][xi][ql] = vadd(vadd(tB[k], tB[(k + 6) % K]), tB[(k + 3) % K]);
      }
      lds_barrier();
      if (xi >= 2 * R && xi < FC - 2 * R && col < W) {
        V ba = VecOps<V>::zero(), bb = ba;
#pragma unroll
        for (int j = -R; j <= R; ++j) {
          ba = vadd(ba, s_a[1][xi + j][ql]);
          bb = vadd(bb, s_b[1][xi + j][ql]);
        }
        const float rn = rn_of(orow);
        const size_t px = img + (size_t)orow * W + col;
        const float g0 = rg[(k + 1) % K];   // guide(orow, col): the oldest ring-1 row (row - 2R)
        V o;
        float* po = reinterpret_cast<float*>(&o);
        const float* pa = reinterpret_cast<const float*>(&ba);
        const float* pb = reinterpret_cast<const float*>(&bb);
#pragma unroll
        for (int i = 0; i < VW; ++i) po[i] = fmaf(pa[i] * rn, g0, pb[i] * rn);
        if (out_bf16 == 2) {   // launch-uniform: the fp16 configuration's high-frequency map (y re-read: this is the rare fallback path)
          const V yv = ldy(px);
          const float* py_ = reinterpret_cast<const float*>(&yv);
#pragma unroll
          for (int i = 0; i < VW; ++i) out16[px * 32 + i] = __builtin_bit_cast(unsigned short, (_Float16)(py_[i] - po[i]));
        } else if (out_bf16) {
#pragma unroll
          for (int i = 0; i < VW; ++i) out16[px * 32 + i] = __builtin_bit_cast(unsigned short, (__bf16)po[i]);
        } else {
          paif::store_nt(out + px * 32, o);
        }
      }
    }
  }
}

}  // namespace

namespace paif_gf2 {
template <int OM>
__global__ void gf2_kernel(const float* __restrict__ guide, const float* __restrict__ y, const float* __restrict__ planes, float* __restrict__ lf,
                           unsigned* __restrict__ flag, int B, int H, int W, int nstrip, int nslots, int rows_per_slot, int total_rows);
extern const int kStripOut, kThreads;      // output columns per strip and threads per workgroup of the build (8 waves: 48, 512; 12 waves: 80, 768)
extern template __global__ void gf2_kernel<0>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
extern template __global__ void gf2_kernel<1>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
extern template __global__ void gf2_kernel<2>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
extern template __global__ void gf2_kernel<3>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
}
namespace paif_gf2w12 {      // gf_mfma2_w12.hip: the 12-wave build, output modes 2 and 3
template <int OM>
__global__ void gf2_kernel(const float* __restrict__ guide, const float* __restrict__ y, const float* __restrict__ planes, float* __restrict__ lf,
                           unsigned* __restrict__ flag, int B, int H, int W, int nstrip, int nslots, int rows_per_slot, int total_rows);
extern const int kStripOut, kThreads;
extern template __global__ void gf2_kernel<2>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
extern template __global__ void gf2_kernel<3>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
}

// The statistics launch alone: the reverse pass (gf_backward.hip) reads the same planes.  workspace: as below.
extern "C" __attribute__((visibility("hidden"))) int paifi_gf_guide_stats(const float* guide, float* workspace, float eps0, float eps1, int B,
                                                                          int H, int W, paif_stream_t stream) {
  const size_t npix = (size_t)B * H * W;
  const int gtx = (W + GT - 1) / GT, gty = (H + GT - 1) / GT;
  hipLaunchKernelGGL(gf_guide_stats_kernel, dim3(B * gtx * gty), dim3(256), 0, paif::as_stream(stream), guide, workspace,
                     reinterpret_cast<unsigned*>(workspace + 4 * npix), eps0, eps1, B, H, W, gtx, gty);
  PAIF_LAUNCH_CHECK("guided_filter(stats)");
  return 0;
}

// workspace: four per-pixel planes + one 256-byte line holding the f16-range flag
extern "C" size_t paif_guided_filter_fused_workspace_floats(int B, int H, int W) { return (size_t)4 * B * H * W + 64; }

static int gf_fused_launch(const float* guide, const float* y, float* lf, float eps0, float eps1, float* workspace, int B, int H, int W,
                           int out_bf16, paif_stream_t stream) {
  PAIF_REQUIRE(guide && y && lf && workspace && B > 0, PAIF_EINVAL, "guided_filter_fused: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter: H,W must exceed 2r+1 = %d (got %dx%d)", K, H, W);
  const int nstrip = (W + FO - 1) / FO;
  // engine: "mfma2" (default; horizontal box sums on the matrix cores, gf_mfma2.hip) or "valu" (the all-VALU kernel below, also the
  // fallback the matrix-core engine's f16-range flag selects).  PAIF_GF_ENGINE / PAIF_GF_FORM are A/B knobs.
  const char* eng = getenv("PAIF_GF_ENGINE");           // read per call: the tests run every engine in one process
  PAIF_REQUIRE(!(eng && !strcmp(eng, "mfma")), PAIF_ENOSUP,
               "guided_filter: the round-3 matrix-core engine (PAIF_GF_ENGINE=mfma) left the library in round 6 (tools/parked/gf_mfma.hip)");
  int engine = (eng && !strcmp(eng, "valu")) ? 0 : 2;
  // the round-4 engine addresses rows with wrapping 32-bit per-lane offsets that the buffer range check filters: every per-image
  // byte size must stay below 2^31 - 2^20
  // fp16 high-frequency outputs: the 12-wave build (96-column strips, three waves per SIMD); PAIF_GF_NW=8 keeps the 8-wave build (A/B runs)
  const char* nwe = getenv("PAIF_GF_NW");
  PAIF_REQUIRE(!nwe || !strcmp(nwe, "8") || !strcmp(nwe, "12"), PAIF_EINVAL, "PAIF_GF_NW must be 8 or 12 (got '%s')", nwe);
  const bool w12 = out_bf16 >= 2 && !(nwe && !strcmp(nwe, "8"));
  const int strip_out = w12 ? paif_gf2w12::kStripOut : paif_gf2::kStripOut, threads2 = w12 ? paif_gf2w12::kThreads : paif_gf2::kThreads;
  const int nstrip2 = (W + strip_out - 1) / strip_out;      // the matrix-core engine's strips
  if (engine == 2 && ((size_t)(B + 1) * H * W * 128 >= 0x7FF00000ull || (size_t)B * nstrip2 * H >= 0x7FFFFFFFull)) engine = 0;   // the all-VALU kernel takes any size
  static const int form = [] {
    const char* e = getenv("PAIF_GF_FORM");
    if (!e) return 0;
    return !strcmp(e, "4x4") ? 1 : !strcmp(e, "4x2") ? 2 : !strcmp(e, "2x16") ? 3 : 0;
  }();
  const int zgroups = form == 0 ? 1 : form == 1 ? 2 : form == 2 ? 4 : 1;
  const int resident = form == 0 ? 1 : form == 1 ? 2 : form == 2 ? 4 : 1;
  hipStream_t st = paif::as_stream(stream);
  const size_t npix = (size_t)B * H * W;
  unsigned* flag = reinterpret_cast<unsigned*>(workspace + 4 * npix);
  const int gtx = (W + GT - 1) / GT, gty = (H + GT - 1) / GT;
  hipLaunchKernelGGL(gf_guide_stats_kernel, dim3(B * gtx * gty), dim3(256), 0, st, guide, workspace, flag, eps0, eps1, B, H, W, gtx, gty);
  PAIF_LAUNCH_CHECK("guided_filter_fused(stats)");
  // Rows per workgroup: one workgroup per CU is resident (8 waves at <= 256 VGPRs), a workgroup costs (rows + warm-up) row
  // iterations: pick the split with the fewest rounds x iterations.
  auto pick = [&](int warm, int res, int zg, int& nseg) {
    long best = -1;
    nseg = 1;
    for (int n = 1; n <= 16 && n <= H; ++n) {
      const int rows = (H + n - 1) / n;
      const long blocks = (long)B * nstrip * n * 2 * zg;
      const long cost = ((blocks + 256 * res - 1) / (256 * res)) * (rows + warm);
      if (best < 0 || cost < best) { best = cost; nseg = n; }
    }
  };
  if (engine == 2) {
    // one 8-wave workgroup per CU (two waves per SIMD), ONE round: the B * nstrip full-height strips laid end to end and cut into
    // equal runs of rows, one per workgroup pair (the two channel halves, 8 block ids apart: same XCD); runs of >= 96 rows
    static const int cus = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return n > 1 ? n : 2;
    }();
    const long total_rows = (long)B * nstrip2 * H;
    int nslots = (int)(total_rows / 96);
    nslots = nslots < 1 ? 1 : (nslots > cus / 2 ? cus / 2 : nslots);
    const int rows_per_slot = (int)((total_rows + nslots - 1) / nslots);
    const int grid = (nslots + 7) / 8 * 16;
    if (w12 && out_bf16 == 3)
      hipLaunchKernelGGL((paif_gf2w12::gf2_kernel<3>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    else if (w12)
      hipLaunchKernelGGL((paif_gf2w12::gf2_kernel<2>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    else if (out_bf16 == 3)
      hipLaunchKernelGGL((paif_gf2::gf2_kernel<3>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    else if (out_bf16 == 2)
      hipLaunchKernelGGL((paif_gf2::gf2_kernel<2>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    else if (out_bf16)
      hipLaunchKernelGGL((paif_gf2::gf2_kernel<1>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    else
      hipLaunchKernelGGL((paif_gf2::gf2_kernel<0>), dim3(grid), dim3(threads2), 0, st, guide, y, workspace, lf, flag, B, H, W, nstrip2, nslots,
                         rows_per_slot, (int)total_rows);
    PAIF_LAUNCH_CHECK("guided_filter_fused(mfma2)");
  }
  int nseg;
  pick(4 * R, resident, zgroups, nseg);
  const int frows = (H + nseg - 1) / nseg;
  const unsigned* only_if = engine == 2 ? flag : nullptr;
  const dim3 grid(B * nstrip * nseg, 2, zgroups);
  if (form == 0) hipLaunchKernelGGL((gf_fused_kernel<float4, 8>), grid, dim3(FC * 8), 0, st, guide, y, workspace, only_if, lf, B, H, W, nstrip, nseg, frows, out_bf16);
  else if (form == 1) hipLaunchKernelGGL((gf_fused_kernel<float4, 4>), grid, dim3(FC * 4), 0, st, guide, y, workspace, only_if, lf, B, H, W, nstrip, nseg, frows, out_bf16);
  else if (form == 2) hipLaunchKernelGGL((gf_fused_kernel<float4, 2>), grid, dim3(FC * 2), 0, st, guide, y, workspace, only_if, lf, B, H, W, nstrip, nseg, frows, out_bf16);
  else hipLaunchKernelGGL((gf_fused_kernel<float2, 16>), grid, dim3(FC * 16), 0, st, guide, y, workspace, only_if, lf, B, H, W, nstrip, nseg, frows, out_bf16);
  PAIF_LAUNCH_CHECK("guided_filter_fused");
  return 0;
}

extern "C" int paif_guided_filter_fused_fwd(const float* guide, const float* y, float* lf, float eps0, float eps1, float* workspace, int B,
                                            int H, int W, paif_stream_t stream) {
  return gf_fused_launch(guide, y, lf, eps0, eps1, workspace, B, H, W, 0, stream);
}

// the fp16 configuration (round 5): the two HIGH-frequency maps HF_e = y - LF_e written as IEEE fp16 (`hf`: [2][B,H,W,32] unsigned short data)
extern "C" int paif_guided_filter_fused_fwd_hf16(const float* guide, const float* y, float* hf, float eps0, float eps1, float* workspace, int B,
                                                 int H, int W, paif_stream_t stream) {
  return gf_fused_launch(guide, y, hf, eps0, eps1, workspace, B, H, W, 2, stream);
}

// round 6: as paif_guided_filter_fused_fwd_hf16 with y READ as IEEE fp16 (`y16`: [B,H,W,32] unsigned short data, the stem's 16-bit twin)
extern "C" int paif_guided_filter_fused_fwd_hf16_y16(const float* guide, const float* y16, float* hf, float eps0, float eps1, float* workspace,
                                                     int B, int H, int W, paif_stream_t stream) {
  return gf_fused_launch(guide, y16, hf, eps0, eps1, workspace, B, H, W, 3, stream);
}

// same, the two low-frequency maps written as bf16 (`lf`: [2][B,H,W,32] unsigned short data): the bf16 configuration
extern "C" int paif_guided_filter_fused_fwd_bf16(const float* guide, const float* y, float* lf, float eps0, float eps1, float* workspace, int B,
                                                 int H, int W, paif_stream_t stream) {
  return gf_fused_launch(guide, y, lf, eps0, eps1, workspace, B, H, W, 1, stream);
}
