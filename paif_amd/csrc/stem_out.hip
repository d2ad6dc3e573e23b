// stem_out of the fusion network in the bf16 inference forward (core/model_fusion_auto.py:616-620, :640):
//   conv3x3 32->16 (no bias, no activation) -> conv3x3 16->1 -> PReLU -> tanh
// as ONE kernel.  The two convs are linear with nothing in between, so away from the image border they are one 5x5 conv 32->1 with
//   W5[e][c] = sum_{c16} sum_{a + d = e} w2[c16][a] * w1[c16][c][d]            (a, d: 3x3 tap indices; e: 5x5 tap index)
// (the 16-channel map between them -- written and read back as 79 MB at B=8 480x640 -- never exists).  With one output channel an
// implicit GEMM over output channels would waste the matrix cores, so the TAPS are the M dimension:
//   G[e][q] = sum_c W5[e][c] * x[q][c]   for every pixel q of the tile + 2-pixel halo     (v_mfma_f32_32x32x16_bf16: M = 25 taps of 32,
//                                                                                         N = 32 pixels, K = 32 channels in 2 steps)
//   z[p]    = sum_e G[e][p + e - 2]                                                      (25 shifted LDS reads per output pixel)
// x is bf16 storage (exact operand); W5 is taken as three bf16 pieces (hi + mid + lo = the fp32 value to 2^-25): 6 MFMAs per 32 pixels,
// fp32 accumulate -- the arithmetic of the fp32 two-conv path up to summation order, not a bf16-weight approximation.
// Border: the reference zero-pads the 16-channel map, so on the outermost pixel ring the 5x5 form counts conv1 outputs "outside" the
// image that the reference does not.  Those positions see only the outermost image row / column, through one row / column of w1: the
// surplus of a top-row pixel is a 5-tap row conv  C(p) = sum_ex Wtop[ex][c] x(0, px + ex - 2)[c],  Wtop[ex][c] = sum_{c16, ax + dx = ex}
// w2[c16][0][ax] w1[c16][c][2][dx]  (likewise bottom / left / right; a corner counts one position twice: a 1-tap term).  The main kernel
// leaves the ring pixels as raw 5x5 sums; stem_out_ring_kernel subtracts the surplus (fp32 weights, 160-352 products per pixel) and
// applies PReLU + tanh.
#include <stdint.h>
#include <type_traits>

#include "paif_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 32;                 // output tile
constexpr int HW_ = TW + 4, HH_ = TH + 4;      // tile + 2-pixel halo: 36 x 12
constexpr int NPX = HW_ * HH_;                 // 432 halo pixels
constexpr int NG = (NPX + 31) / 32;            // 14 groups of 32 pixels
constexpr int GS = 456;                        // floats per tap plane: >= 32 * NG, and 4 * GS % 64 == 32 (the two lane halves of a store hit disjoint banks)
constexpr int NTAP = 25;

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// wpk[ks][piece][lane][8 bf16]: lane (m = lane & 31 = 5x5 tap, hh = lane >> 5) holds W5[m][16 ks + 8 hh + j], j = 0..7, as piece 0 / 1 / 2 =
// hi / mid / lo; taps 25..31 are zero rows
__global__ void stem_out_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, unsigned short* __restrict__ wpk) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (ks, lane, j)
  if (idx >= 2 * 64 * 8) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, ks = idx >> 9;
  const int m = lane & 31, hh = lane >> 5, c = 16 * ks + 8 * hh + j;
  float v = 0.f;
  if (m < NTAP) {
    const int ey = m / 5, ex = m - ey * 5;
    for (int c16 = 0; c16 < 16; ++c16)             // fixed order: deterministic
      for (int ay = 0; ay < 3; ++ay)
        for (int ax = 0; ax < 3; ++ax) {
          const int dy = ey - ay, dx = ex - ax;
          if (dy < 0 || dy > 2 || dx < 0 || dx > 2) continue;
          v = fmaf(w2[(c16 * 3 + ay) * 3 + ax], w1[((c16 * 32 + c) * 3 + dy) * 3 + dx], v);
        }
  }
  const __bf16 hi = (__bf16)v;
  const float r1 = v - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mid);
  wpk[((ks * 3 + 0) * 64 + lane) * 8 + j] = __builtin_bit_cast(unsigned short, hi);
  wpk[((ks * 3 + 1) * 64 + lane) * 8 + j] = __builtin_bit_cast(unsigned short, mid);
  wpk[((ks * 3 + 2) * 64 + lane) * 8 + j] = __builtin_bit_cast(unsigned short, lo);
}

// ring weights, fp32: wr[side][e][c] (side 0 top, 1 bottom, 2 left, 3 right; e = 0..4 along the edge) then wr[640 + corner * 32 + c]
// (corner 0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right)
constexpr int RING_FLOATS = 4 * 5 * 32 + 4 * 32;
__global__ void stem_out_ring_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ wr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= RING_FLOATS) return;
  const int c = idx & 31;
  float v = 0.f;
  if (idx < 640) {
    const int e = (idx >> 5) % 5, side = idx / 160;
    for (int c16 = 0; c16 < 16; ++c16)
      for (int a = 0; a < 3; ++a) {
        const int d = e - a;
        if (d < 0 || d > 2) continue;
        // the excluded w2 row / column (0 or 2) and the w1 row / column through which it sees the image edge (2 or 0)
        const float f2 = side == 0 ? w2[(c16 * 3 + 0) * 3 + a] : side == 1 ? w2[(c16 * 3 + 2) * 3 + a] : side == 2 ? w2[(c16 * 3 + a) * 3 + 0]
                                                                                                              : w2[(c16 * 3 + a) * 3 + 2];
        const float* q = w1 + (size_t)(c16 * 32 + c) * 9;
        const float f1 = side == 0 ? q[2 * 3 + d] : side == 1 ? q[0 * 3 + d] : side == 2 ? q[d * 3 + 2] : q[d * 3 + 0];
        v = fmaf(f2, f1, v);
      }
  } else {
    const int corner = (idx - 640) >> 5;
    const int ay = corner < 2 ? 0 : 2, ax = (corner & 1) ? 2 : 0;
    for (int c16 = 0; c16 < 16; ++c16) v = fmaf(w2[(c16 * 3 + ay) * 3 + ax], w1[((size_t)(c16 * 32 + c) * 3 + (2 - ay)) * 3 + (2 - ax)], v);
  }
  wr[idx] = v;
}

// F32IN (round 4: the fp32-storage forward): x is an fp32 map, taken as bf16 hi + lo (the fp32 value to 2^-17, the operand split of every
// dense conv of that path): x_hi meets the three weight pieces, x_lo the upper two -- 10 MFMAs per 32 pixels instead of 6.
template <bool F32IN>
__global__ __launch_bounds__(256) void stem_out_fused_kernel(const void* __restrict__ xv, const uint4* __restrict__ wpk,
                                                             const float* __restrict__ prelu, float* __restrict__ fused, int B, int H,
                                                             int W, int tilesX, int tilesY, unsigned* __restrict__ flag) {
  typedef typename std::conditional<F32IN, float, unsigned short>::type XT;
  const XT* __restrict__ x = reinterpret_cast<const XT*>(xv);
  __shared__ float G[NTAP * GS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, n = lane & 31;
  int t = paif::xcd_remap(blockIdx.x, gridDim.x);
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int x0 = tx * TW, y0 = ty * TH;
  bf16x8 wa[2][3];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) wa[ks][pc] = __builtin_bit_cast(bf16x8, wpk[(ks * 3 + pc) * 64 + lane]);
  const XT* img = x + (size_t)b * H * W * 32;
  // a wave takes groups wave, wave + 4, wave + 8, wave + 12: all its loads are issued before the first MFMA (unconditional, clamped;
  // group slots >= NG re-read the last group and are not stored)
  constexpr int GPW = (NG + 3) / 4;
  constexpr int NV = F32IN ? 2 : 1;                                 // 16-byte pieces per 8 channels
  uint4 v0[GPW][NV], v1[GPW][NV];
  int hps[GPW];
  bool oks[GPW];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int g = min(wave + 4 * i, NG - 1);
    const int hp = g * 32 + n;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int gy = y0 - 2 + hy, gx = x0 - 2 + hx;
    hps[i] = hp;
    oks[i] = hp < NPX && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t off = ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 32 + hh * 8;
#pragma unroll
    for (int h = 0; h < NV; ++h) {
      v0[i][h] = *reinterpret_cast<const uint4*>(img + off + 4 * h);            // channels 8 hh .. 8 hh + 7           (k step 0)
      v1[i][h] = *reinterpret_cast<const uint4*>(img + off + 16 + 4 * h);       // channels 16 + 8 hh .. 16 + 8 hh + 7 (k step 1)
    }
  }
  auto split = [](const uint4& a, const uint4& c, bf16x8& hi, bf16x8& lo) {    // 8 fp32 values -> bf16 hi + lo
    const float f[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                        __uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w)};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = (__bf16)f[j];
      lo[j] = (__bf16)(f[j] - (float)hi[j]);
    }
  };
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    if (wave + 4 * i >= NG) break;                                   // wave-uniform
    if (!oks[i]) {                                                   // zero padding of the image / unused slots of the last group
#pragma unroll
      for (int h = 0; h < NV; ++h) v0[i][h] = v1[i][h] = make_uint4(0u, 0u, 0u, 0u);
    }
    const int hp = hps[i];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (F32IN) {
      bf16x8 h0, l0, h1, l1;
      split(v0[i][0], v0[i][NV - 1], h0, l0);
      split(v1[i][0], v1[i][NV - 1], h1, l1);
      // smallest terms first: x_lo * W_mid, x_hi * W_lo, x_lo * W_hi, x_hi * W_mid, x_hi * W_hi  (x_lo * W_lo ~ 2^-34 is dropped)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][1], l0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][1], l1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][2], h0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][2], h1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][0], l0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][0], l1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][1], h0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][1], h1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][0], h0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][0], h1, acc, 0, 0, 0);
    } else {
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, v0[i][0]), b1 = __builtin_bit_cast(bf16x8, v1[i][0]);
#pragma unroll
      for (int pc = 2; pc >= 0; --pc) {                                  // smallest pieces first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][pc], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][pc], b1, acc, 0, 0, 0);
      }
    }
    // D[m][n]: this lane holds, for pixel n, the taps m = (r & 3) + 8 (r >> 2) + 4 hh: r = 0..11 are taps < 24 for both halves, r = 12 is
    // tap 24 (hh = 0) or 28 (unused)
#pragma unroll
    for (int r = 0; r < 12; ++r) G[((r & 3) + 8 * (r >> 2) + 4 * hh) * GS + hp] = acc[r];
    if (hh == 0) G[24 * GS + hp] = acc[12];
  }
  __syncthreads();
  const int oy = tid >> 5, ox = tid & 31;
  float z = 0.f;
#pragma unroll
  for (int dy = 0; dy < 5; ++dy)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) z += G[(dy * 5 + dx) * GS + (oy + dy) * HW_ + ox + dx];
  const int py = y0 + oy, px = x0 + ox;
  const bool ring = py == 0 || py == H - 1 || px == 0 || px == W - 1;           // finished by stem_out_ring_kernel
  if (py < H && px < W) fused[((size_t)b * H + py) * W + px] = ring ? z : tanhf(paif::prelu_f(z, *prelu));
  // range guard of the fp16 storage mode (paif_stem_out_fwd_f32_guard): tanh maps an overflowed trunk to a FINITE +-1, so the
  // pre-activation is tested here -- one compare per pixel, an atomic only on the first non-finite value a wave sees
  if (flag != nullptr && py < H && px < W && !(fabsf(z) <= 3.0e38f)) atomicOr(flag, 1u);
}

// The outermost pixel ring: raw 5x5 sum (left in `fused` by the kernel above) minus the surplus terms, then PReLU + tanh.
// 8 lanes per pixel, lane q = channels 4q .. 4q + 3.
template <bool F32IN>
__global__ __launch_bounds__(256) void stem_out_ring_kernel(const void* __restrict__ xv, const float* __restrict__ wr,
                                                            const float* __restrict__ prelu, float* __restrict__ fused, int B, int H, int W,
                                                            unsigned* __restrict__ flag) {
  typedef typename std::conditional<F32IN, float, unsigned short>::type XT;
  const XT* __restrict__ x = reinterpret_cast<const XT*>(xv);
  const int q = threadIdx.x & 7;
  const int id = blockIdx.x * 32 + (threadIdx.x >> 3);
  const int per = 2 * W + 2 * (H - 2);
  const bool live = id < B * per;
  const int idc = live ? id : 0;
  const int b = idc / per, k = idc - b * per;
  int py, px;
  if (k < W) { py = 0; px = k; }
  else if (k < 2 * W) { py = H - 1; px = k - W; }
  else { const int j = k - 2 * W; py = 1 + (j >> 1); px = (j & 1) ? W - 1 : 0; }
  const XT* img = x + (size_t)b * H * W * 32 + q * 4;
  auto dot = [&](const float* w, int yy, int xx) -> float {      // sum over this lane's 4 channels of w[c] * x(yy, xx)[c]; 0 outside the image
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.f;
    const float4 w4 = *reinterpret_cast<const float4*>(w + q * 4);
    if constexpr (F32IN) {
      const float4 xf = *reinterpret_cast<const float4*>(img + ((size_t)yy * W + xx) * 32);
      return fmaf(xf.w, w4.w, fmaf(xf.z, w4.z, fmaf(xf.y, w4.y, xf.x * w4.x)));
    }
    const uint2 v = *reinterpret_cast<const uint2*>(img + ((size_t)yy * W + xx) * 32);
    float sacc = __uint_as_float(v.x << 16) * w4.x;
    sacc = fmaf(__uint_as_float(v.x & 0xffff0000u), w4.y, sacc);
    sacc = fmaf(__uint_as_float(v.y << 16), w4.z, sacc);
    return fmaf(__uint_as_float(v.y & 0xffff0000u), w4.w, sacc);
  };
  float c = 0.f;
  const bool top = py == 0, bot = py == H - 1, lft = px == 0, rgt = px == W - 1;
  if (top) for (int e = 0; e < 5; ++e) c += dot(wr + (0 * 5 + e) * 32, 0, px + e - 2);
  if (bot) for (int e = 0; e < 5; ++e) c += dot(wr + (1 * 5 + e) * 32, H - 1, px + e - 2);
  if (lft) for (int e = 0; e < 5; ++e) c += dot(wr + (2 * 5 + e) * 32, py + e - 2, 0);
  if (rgt) for (int e = 0; e < 5; ++e) c += dot(wr + (3 * 5 + e) * 32, py + e - 2, W - 1);
  if (top && lft) c -= dot(wr + 640 + 0 * 32, 0, 0);              // the position counted by both edges
  if (top && rgt) c -= dot(wr + 640 + 1 * 32, 0, W - 1);
  if (bot && lft) c -= dot(wr + 640 + 2 * 32, H - 1, 0);
  if (bot && rgt) c -= dot(wr + 640 + 3 * 32, H - 1, W - 1);
  c += __shfl_xor(c, 1);
  c += __shfl_xor(c, 2);
  c += __shfl_xor(c, 4);
  if (live && q == 0) {
    float* o = fused + ((size_t)b * H + py) * W + px;
    const float z = *o - c;
    if (flag != nullptr && !(fabsf(z) <= 3.0e38f)) atomicOr(flag, 1u);
    *o = tanhf(paif::prelu_f(z, *prelu));
  }
}

}  // namespace

extern "C" {

int paif_stem_out_pack_floats(void) { return 2 * 3 * 64 * 8 / 2 + RING_FLOATS; }   // the MFMA operand (bf16 pairs counted as floats: 1,536) + the ring weights

int paif_stem_out_pack(const float* w1, const float* w2, float* wpk, paif_stream_t stream) {
  PAIF_REQUIRE(w1 && w2 && wpk, PAIF_EINVAL, "stem_out_pack: null pointer");
  hipLaunchKernelGGL(stem_out_pack_kernel, dim3(4), dim3(256), 0, paif::as_stream(stream), w1, w2, reinterpret_cast<unsigned short*>(wpk));
  PAIF_LAUNCH_CHECK("stem_out_pack");
  hipLaunchKernelGGL(stem_out_ring_pack_kernel, dim3((RING_FLOATS + 255) / 256), dim3(256), 0, paif::as_stream(stream), w1, w2, wpk + 1536);
  PAIF_LAUNCH_CHECK("stem_out_ring_pack");
  return 0;
}

static int stem_out_launch(const void* x, bool f32in, const float* wpk, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream,
                           unsigned* flag = nullptr) {
  PAIF_REQUIRE(x && wpk && prelu && fused && B > 0, PAIF_EINVAL, "stem_out: bad arguments");
  PAIF_REQUIRE(H >= 3 && W >= 3, PAIF_ENOSUP, "stem_out: %dx%d is smaller than 3x3", H, W);
  const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
  PAIF_REQUIRE((size_t)B * tilesX * tilesY < ((size_t)1 << 31), PAIF_EINVAL, "stem_out: %dx%dx%d is too large for one launch", B, H, W);
  hipStream_t st = paif::as_stream(stream);
  const dim3 grid((unsigned)(B * tilesX * tilesY));
  const int ring = B * (2 * W + 2 * (H - 2));
  const dim3 rgrid((unsigned)((ring + 31) / 32));
  if (f32in) {
    hipLaunchKernelGGL(stem_out_fused_kernel<true>, grid, dim3(256), 0, st, x, reinterpret_cast<const uint4*>(wpk), prelu, fused, B, H, W, tilesX, tilesY, flag);
    PAIF_LAUNCH_CHECK("stem_out(f32)");
    hipLaunchKernelGGL(stem_out_ring_kernel<true>, rgrid, dim3(256), 0, st, x, wpk + 1536, prelu, fused, B, H, W, flag);
  } else {
    hipLaunchKernelGGL(stem_out_fused_kernel<false>, grid, dim3(256), 0, st, x, reinterpret_cast<const uint4*>(wpk), prelu, fused, B, H, W, tilesX, tilesY, flag);
    PAIF_LAUNCH_CHECK("stem_out(bf16)");
    hipLaunchKernelGGL(stem_out_ring_kernel<false>, rgrid, dim3(256), 0, st, x, wpk + 1536, prelu, fused, B, H, W, flag);
  }
  PAIF_LAUNCH_CHECK("stem_out ring");
  return 0;
}

int paif_stem_out_fwd_bf16(const float* x, const float* wpk, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream) {
  return stem_out_launch(x, false, wpk, prelu, fused, B, H, W, stream);
}

int paif_stem_out_fwd_f32(const float* x, const float* wpk, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream) {
  return stem_out_launch(x, true, wpk, prelu, fused, B, H, W, stream);
}

// fp16 storage mode's range guard (round 6): as paif_stem_out_fwd_f32 / _bf16 (f32in selects), and *flag |= 1 when any pre-tanh value is
// inf / NaN.  Every 16-bit map of the forward reaches this kernel's input through convs and residual adds, which keep a non-finite
// value non-finite; tanh would hide it (tanh(inf) = 1).  flag: one 32-bit word in device memory, cleared by the caller.
int paif_stem_out_fwd_guard(const float* x, int f32in, const float* wpk, const float* prelu, float* fused, unsigned* flag, int B, int H, int W,
                            paif_stream_t stream) {
  PAIF_REQUIRE(flag, PAIF_EINVAL, "stem_out_guard: null flag");
  return stem_out_launch(x, f32in != 0, wpk, prelu, fused, B, H, W, stream, flag);
}

}  // extern "C"
