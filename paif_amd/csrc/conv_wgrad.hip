// Dense-conv weight gradient (first kernel of the training step, DESIGN.md "Plan for the training step" item 2):
//   dW[co][s*32+ci][ky][kx] = sum_{b,y,x} dAcc[b,y,x,co] * X_s[b, y + ky*d - P, x + kx*d - P, ci]        (zero outside the image)
//   dAcc = dOut * alpha * act'(z) * scale[co]      (z = saved pre-activation of the forward epilogue; act' of PReLU / ReLU)
// for the forward conv of conv_mfma.hip (`nn.Conv2d` inside BasicConv / RDB / ResidualModule ..., operations_m.py:435-464).
//
// v_mfma_f32_32x32x2_f32 consumes the NHWC layout directly: with K = 2 consecutive pixels of a row,
//   A[i = co][k] = dAcc[pixel k][co]   -> lane (co = l & 31, k = l >> 5) loads one dword; a wave-load is 2 px x 128 B contiguous
//   B[k][j = ci] = X[pixel k + tap][ci] -> the same, shifted by the tap
// so a wave streams pixel pairs and keeps one 32 x 32 accumulator per horizontal tap kx; blockIdx.y selects (source, ky).
// A workgroup covers 8 image rows (4 waves x 2 rows); its 4 waves are reduced through LDS and written as one slab
// [block][source*KH + ky][kx][co][ci]; a second pass sums the slabs in block order (deterministic) into the PyTorch layout.
// Exact fp32 (the gradient-parity tolerance of the reference's own fp32 run is what this has to meet).
#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 8;

struct WgradArgs {
  const float* src[3];
  const float* dout;
  const float* z;        // saved pre-activation (act 1/2) or NULL
  const float* scale;    // per-cout forward scale (folded BN) or NULL
  const float* prelu;    // slope (act 1)
  float* slabs;
  float alpha;
  int act, kh, dil, nsrc, B, H, W, nrb;   // nrb = row blocks per image
};

template <int KH>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ float red[KH][16][64];      // waves 1..3 park their accumulators in turn; wave 0 adds them in wave order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = lane >> 5, ch = lane & 31;
  const int s = blockIdx.y / KH, ky = blockIdx.y - s * KH;
  const int b = blockIdx.x / a.nrb, rb = blockIdx.x - b * a.nrb;
  const int P = a.dil * (KH - 1) / 2;
  const float* X = a.src[s];
  const float sc = (a.scale ? a.scale[ch] : 1.f) * a.alpha;
  const float slope = a.act == 1 ? *a.prelu : 0.f;
  f32x16 acc[KH];
#pragma unroll
  for (int t = 0; t < KH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int rr = wave; rr < ROWS; rr += 4) {
    const int y = rb * ROWS + rr;
    if (y >= a.H) break;                       // wave-uniform
    const int ys = y + ky * a.dil - P;         // source row of this (output row, ky)
    if (ys < 0 || ys >= a.H) continue;         // zero padding: contributes nothing (wave-uniform)
    const size_t orow = ((size_t)b * a.H + y) * a.W, srow = ((size_t)b * a.H + ys) * a.W;
    for (int x0 = 0; x0 < a.W; x0 += 2) {
      const int x = x0 + k;
      // unconditional loads on clamped columns, zero by select
      const int xc = min(x, a.W - 1);
      float d = a.dout[(orow + xc) * 32 + ch];
      if (a.act) {
        const float zv = a.z[(orow + xc) * 32 + ch];
        d *= a.act == 1 ? (zv >= 0.f ? 1.f : slope) : (zv > 0.f ? 1.f : 0.f);
      }
      d = x < a.W ? d * sc : 0.f;
#pragma unroll
      for (int t = 0; t < KH; ++t) {
        const int xs = x + t * a.dil - P;
        const float v = X[(srow + min(max(xs, 0), a.W - 1)) * 32 + ch];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, (xs >= 0 && xs < a.W) ? v : 0.f, acc[t], 0, 0, 0);
      }
    }
  }
  // ---- reduce the 4 waves (fixed order) and write the slab ----
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[t][r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[t][r][lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* slab = a.slabs + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * KH * 1024;
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * k;      // C/D layout of the 32x32 MFMA: row = co, column = lane & 31 = ci
        slab[(t * 32 + co) * 32 + ch] = acc[t][r];
      }
  }
}

// dw[co][s*32+ci][ky][kx] = sum over blocks of slab[blk][s*KH+ky][kx][co][ci]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int nblk, int nz,
                                                                int kh, int nsrc, int cout, int accumulate) {
  const int total = cout * nsrc * 32 * kh * kh;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int tmp = i;
    const int kx = tmp % kh; tmp /= kh;
    const int ky = tmp % kh; tmp /= kh;
    const int cin = tmp % (nsrc * 32);
    const int co = tmp / (nsrc * 32);
    const int s = cin >> 5, ci = cin & 31;
    const size_t off = ((size_t)(s * kh + ky) * kh + kx) * 1024 + (size_t)co * 32 + ci;
    float v = 0.f;
    for (int blk = 0; blk < nblk; ++blk) v += slabs[((size_t)blk * nz) * kh * 1024 + off];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

}  // namespace

extern "C" size_t paif_conv2d_wgrad_workspace_floats(int nsrc, int kh, int B, int H) {
  return (size_t)B * ((H + ROWS - 1) / ROWS) * nsrc * kh * kh * 1024;
}

extern "C" int paif_conv2d_wgrad(const float* const* src, int nsrc, const float* dout, const float* z, const float* scale,
                                 const float* prelu, int act, float alpha, int kh, int dil, float* workspace, float* dw, int cout,
                                 int accumulate, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(src && dout && workspace && dw, PAIF_EINVAL, "conv2d_wgrad: null pointer");
  PAIF_REQUIRE(nsrc >= 1 && nsrc <= 3, PAIF_EINVAL, "conv2d_wgrad: nsrc=%d", nsrc);
  PAIF_REQUIRE(B > 0 && H > 0 && W > 0, PAIF_EINVAL, "conv2d_wgrad: empty shape");
  PAIF_REQUIRE(act >= 0 && act <= 2 && (!act || z) && (act != 1 || prelu), PAIF_EINVAL, "conv2d_wgrad: act=%d needs z (and the slope)", act);
  PAIF_REQUIRE(dil == 1 || dil == 2, PAIF_ENOSUP, "conv2d_wgrad: dil=%d", dil);
  PAIF_REQUIRE(cout >= 1 && cout <= 32, PAIF_EINVAL, "conv2d_wgrad: cout=%d (dout always has 32 channels; rows >= cout are dropped)", cout);
  WgradArgs a;
  for (int s = 0; s < 3; ++s) a.src[s] = s < nsrc ? src[s] : nullptr;
  for (int s = 0; s < nsrc; ++s) PAIF_REQUIRE(a.src[s], PAIF_EINVAL, "conv2d_wgrad: src[%d] null", s);
  a.dout = dout; a.z = z; a.scale = scale; a.prelu = prelu; a.slabs = workspace; a.alpha = alpha;
  a.act = act; a.kh = kh; a.dil = dil; a.nsrc = nsrc; a.B = B; a.H = H; a.W = W; a.nrb = (H + ROWS - 1) / ROWS;
  const dim3 grid(B * a.nrb, nsrc * kh);
  hipStream_t st = paif::as_stream(stream);
  switch (kh) {
    case 1: hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL(conv_wgrad_kernel<5>, grid, dim3(256), 0, st, a); break;
    case 7: hipLaunchKernelGGL(conv_wgrad_kernel<7>, grid, dim3(256), 0, st, a); break;
    default:
      paif::set_error("conv2d_wgrad: kernel size %d not built", kh);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("conv2d_wgrad");
  const int total = cout * nsrc * 32 * kh * kh;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, dw, B * a.nrb, nsrc * kh, kh, nsrc,
                     cout, accumulate);
  PAIF_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}
