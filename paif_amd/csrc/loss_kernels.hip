// Fusion loss of the training API: Fusionloss_grad2 = L1(mask, fused) + 1.1 * (1 - SSIM_11x11(fused, mask))
// (core/loss.py:490-502; pytorch_ssim/__init__.py:8-43,70-78: Gaussian window sigma 1.5, zero padding 5, mean of the map).
// Forward value and the gradient w.r.t. the generated image; the parameter-gradient kernels of the training step are
// not built.
//
// One workgroup = a 16 x 16 pixel tile of one image; the 26 x 26 halo tiles of both images sit in LDS (zero padded);
// a thread forms the five windowed sums (x, y, x^2, y^2, xy) of its pixel with the separable weights g[i]*g[j] given by
// the host (computed there in fp32 exactly as the reference builds its window) and the SSIM / |y - x| terms; the
// workgroup's two sums go to partial[2 * block] in a fixed lane order (no float atomics; the host adds the partials).
#include "paif_common.h"

namespace {

constexpr int T = 16, WS = 11, PADW = WS / 2, HT = T + 2 * PADW;   // 26

__global__ __launch_bounds__(256) void ssim_l1_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ g, float* __restrict__ partial, int H, int W,
                                                      int tilesX, int tilesY) {
  __shared__ float sx[HT][HT + 1], sy[HT][HT + 1];
  __shared__ float sg[WS];
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int x0 = tx * T - PADW, y0 = ty * T - PADW;
  const size_t img = (size_t)b * H * W;
  if (tid < WS) sg[tid] = g[tid];
  for (int i = tid; i < HT * HT; i += 256) {
    const int r = i / HT, c = i - r * HT;
    const int gy = min(max(y0 + r, 0), H - 1), gx = min(max(x0 + c, 0), W - 1);   // unconditional load, zero by select
    const bool in = y0 + r >= 0 && y0 + r < H && x0 + c >= 0 && x0 + c < W;
    const float vx = x[img + (size_t)gy * W + gx], vy = y[img + (size_t)gy * W + gx];
    sx[r][c] = in ? vx : 0.f;
    sy[r][c] = in ? vy : 0.f;
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int py = ty * T + ly, px = tx * T + lx;
  float s_ssim = 0.f, s_l1 = 0.f;
  if (py < H && px < W) {
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int i = 0; i < WS; ++i) {
      const float gi = sg[i];
#pragma unroll
      for (int j = 0; j < WS; ++j) {
        const float w = gi * sg[j];   // the reference's 2-D window is the fp32 outer product of the 1-D one
        const float a = sx[ly + i][lx + j], c = sy[ly + i][lx + j];
        m1 = fmaf(w, a, m1); m2 = fmaf(w, c, m2);
        e11 = fmaf(w, a * a, e11); e22 = fmaf(w, c * c, e22); e12 = fmaf(w, a * c, e12);
      }
    }
    const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
    const float s1 = e11 - m11, s2 = e22 - m22, s12 = e12 - m12;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    s_ssim = ((2.f * m12 + C1) * (2.f * s12 + C2)) / ((m11 + m22 + C1) * (s1 + s2 + C2));
    s_l1 = fabsf(sy[ly + PADW][lx + PADW] - sx[ly + PADW][lx + PADW]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s_ssim += __shfl_xor(s_ssim, m);
    s_l1 += __shfl_xor(s_l1, m);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = s_ssim; red[1][tid >> 6] = s_l1; }
  __syncthreads();
  if (tid == 0) {
    partial[2 * (size_t)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * (size_t)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// ---- gradient w.r.t. x of  L = k_l1 * sum|y - x| + k_ss * sum(1 - S)   (k_* carry the upstream gradient and 1/N) ----
// pass 1: per pixel p the partials of S_p w.r.t. its windowed means  (mu1 = E[x], e11 = E[x^2], e12 = E[xy]):
//   A1 = 2 mu1 mu2 + C1, A2 = 2 s12 + C2, B1 = mu1^2 + mu2^2 + C1, B2 = s1 + s2 + C2,  S = A1 A2 / (B1 B2)
//   dS/dmu1 = 2 mu2 (A2 - A1) / (B1 B2) - 2 mu1 S (1/B1 - 1/B2),   dS/de11 = -S / B2,   dS/de12 = 2 A1 / (B1 B2)
__global__ __launch_bounds__(256) void ssim_bwd1_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ g, float* __restrict__ abc, int B, int H, int W,
                                                        int tilesX, int tilesY) {
  __shared__ float sx[HT][HT + 1], sy[HT][HT + 1];
  __shared__ float sg[WS];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int x0 = tx * T - PADW, y0 = ty * T - PADW;
  const size_t img = (size_t)b * H * W, plane = (size_t)B * H * W;
  if (tid < WS) sg[tid] = g[tid];
  for (int i = tid; i < HT * HT; i += 256) {
    const int r = i / HT, c = i - r * HT;
    const int gy = min(max(y0 + r, 0), H - 1), gx = min(max(x0 + c, 0), W - 1);
    const bool in = y0 + r >= 0 && y0 + r < H && x0 + c >= 0 && x0 + c < W;
    const float vx = x[img + (size_t)gy * W + gx], vy = y[img + (size_t)gy * W + gx];
    sx[r][c] = in ? vx : 0.f;
    sy[r][c] = in ? vy : 0.f;
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int py = ty * T + ly, px = tx * T + lx;
  if (py >= H || px >= W) return;
  float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
  for (int i = 0; i < WS; ++i) {
    const float gi = sg[i];
#pragma unroll
    for (int j = 0; j < WS; ++j) {
      const float w = gi * sg[j];
      const float a = sx[ly + i][lx + j], c = sy[ly + i][lx + j];
      m1 = fmaf(w, a, m1); m2 = fmaf(w, c, m2);
      e11 = fmaf(w, a * a, e11); e22 = fmaf(w, c * c, e22); e12 = fmaf(w, a * c, e12);
    }
  }
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * (e12 - m1 * m2) + C2;
  const float B1 = m1 * m1 + m2 * m2 + C1, B2 = (e11 - m1 * m1) + (e22 - m2 * m2) + C2;
  const float inv = 1.f / (B1 * B2), S = A1 * A2 * inv;
  const size_t o = img + (size_t)py * W + px;
  abc[o] = 2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S * (1.f / B1 - 1.f / B2);
  abc[plane + o] = -S / B2;
  abc[2 * plane + o] = 2.f * A1 * inv;
}

// pass 2: dx[q] = -k_ss * ( sum_p w[p-q] a_p + 2 x_q sum_p w[p-q] b_p + y_q sum_p w[p-q] c_p ) - k_l1 * sign(y_q - x_q)
__global__ __launch_bounds__(256) void ssim_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ g, const float* __restrict__ abc,
                                                        const float* __restrict__ k, float* __restrict__ dx, int B, int H, int W,
                                                        int tilesX, int tilesY) {
  __shared__ float sa[HT][HT + 1], sb[HT][HT + 1], sc[HT][HT + 1];
  __shared__ float sg[WS];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tilesX; t /= tilesX;
  const int ty = t % tilesY;
  const int b = t / tilesY;
  const int x0 = tx * T - PADW, y0 = ty * T - PADW;
  const size_t img = (size_t)b * H * W, plane = (size_t)B * H * W;
  if (tid < WS) sg[tid] = g[tid];
  for (int i = tid; i < HT * HT; i += 256) {
    const int r = i / HT, c = i - r * HT;
    const int gy = min(max(y0 + r, 0), H - 1), gx = min(max(x0 + c, 0), W - 1);
    const bool in = y0 + r >= 0 && y0 + r < H && x0 + c >= 0 && x0 + c < W;
    const size_t o = img + (size_t)gy * W + gx;
    const float va = abc[o], vb = abc[plane + o], vc = abc[2 * plane + o];
    sa[r][c] = in ? va : 0.f;
    sb[r][c] = in ? vb : 0.f;
    sc[r][c] = in ? vc : 0.f;
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int py = ty * T + ly, px = tx * T + lx;
  if (py >= H || px >= W) return;
  float wa = 0.f, wb = 0.f, wc = 0.f;
  for (int i = 0; i < WS; ++i) {
    const float gi = sg[i];
#pragma unroll
    for (int j = 0; j < WS; ++j) {
      const float w = gi * sg[j];   // symmetric window: correlation = convolution
      wa = fmaf(w, sa[ly + i][lx + j], wa);
      wb = fmaf(w, sb[ly + i][lx + j], wb);
      wc = fmaf(w, sc[ly + i][lx + j], wc);
    }
  }
  const size_t o = img + (size_t)py * W + px;
  const float xv = x[o], yv = y[o];
  const float d = yv - xv;
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  dx[o] = -k[1] * (wa + 2.f * xv * wb + yv * wc) - k[0] * sgn;
}

}  // namespace

extern "C" int paif_ssim_l1_bwd_input(const float* x, const float* y, const float* window1d, const float* k, float* abc_scratch,
                                      float* dx, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && y && window1d && k && abc_scratch && dx && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "ssim_l1_bwd: bad arguments");
  const int tilesX = (W + T - 1) / T, tilesY = (H + T - 1) / T;
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(ssim_bwd1_kernel, dim3(B * tilesX * tilesY), dim3(256), 0, st, x, y, window1d, abc_scratch, B, H, W, tilesX, tilesY);
  PAIF_LAUNCH_CHECK("ssim_bwd1");
  hipLaunchKernelGGL(ssim_bwd2_kernel, dim3(B * tilesX * tilesY), dim3(256), 0, st, x, y, window1d, abc_scratch, k, dx, B, H, W, tilesX,
                     tilesY);
  PAIF_LAUNCH_CHECK("ssim_bwd2");
  return 0;
}

namespace {
// out[0] = sum(partial[2i]) * inv_n (mean SSIM), out[1] = sum(partial[2i+1]) * inv_n (mean |y - x|); block order, double
__global__ void ssim_l1_finish_kernel(const float* __restrict__ partial, int nblk, float inv_n, float* __restrict__ out) {
  __shared__ double sl[2][64];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) { a += partial[2 * (size_t)i]; b += partial[2 * (size_t)i + 1]; }
  sl[0][threadIdx.x] = a; sl[1][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += sl[threadIdx.x][i];
    out[threadIdx.x] = (float)(t * (double)inv_n);
  }
}
}  // namespace

extern "C" int paif_ssim_l1_blocks(int B, int H, int W) { return B * ((H + T - 1) / T) * ((W + T - 1) / T); }

extern "C" int paif_ssim_l1_fwd(const float* x, const float* y, const float* window1d, float* partial, float* means, int B, int H, int W,
                                paif_stream_t stream) {
  PAIF_REQUIRE(x && y && window1d && partial && means && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "ssim_l1: bad arguments");
  const int tilesX = (W + T - 1) / T, tilesY = (H + T - 1) / T;
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(ssim_l1_kernel, dim3(B * tilesX * tilesY), dim3(256), 0, st, x, y, window1d, partial, H, W, tilesX, tilesY);
  PAIF_LAUNCH_CHECK("ssim_l1");
  hipLaunchKernelGGL(ssim_l1_finish_kernel, dim3(1), dim3(64), 0, st, partial, B * tilesX * tilesY,
                     (float)(1.0 / ((double)B * H * W)), means);
  PAIF_LAUNCH_CHECK("ssim_l1_finish");
  return 0;
}


// ---------------------------------------------------------------------------------------------------
// Image-space attack losses (attack/attack.py:75-100, 132-133, 216-218: nn.MSELoss / nn.L1Loss on trans_format(fused, vis) against
// X_fusion, "mean" reduction, the target broadcast over the channels when it has one) -- the glue of the single-modality attacks
// that neither entry script calls, as HIP kernels so that those attacks are all-HIP too.
// ---------------------------------------------------------------------------------------------------
namespace {

// out[b][c][i] = x[b][c][i] * scale[c] + shift[c]   (NCHW planes; trans_format = this on the SegFormer-normalised image, and its adjoint)
__global__ __launch_bounds__(256) void channel_affine_nchw_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ out, int C, size_t HW,
                                                                  size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    out[i] = fmaf(x[i], scale[c], shift ? shift[c] : 0.f);
  }
}

constexpr int IL_BLOCKS = 1024;
// kind 0: (a - t)^2, kind 1: |a - t|; per-block partial sums in a fixed order, finished by one thread (deterministic)
__global__ __launch_bounds__(256) void image_loss_kernel(const float* __restrict__ a, const float* __restrict__ t, int kind, int C, int Ct,
                                                         size_t HW, size_t total, float* __restrict__ partial) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / ((size_t)C * HW), r = i - b * (size_t)C * HW;
    const size_t ti = Ct == C ? i : b * HW + (r % HW);
    const float d = a[i] - t[ti];
    s += kind == 0 ? d * d : fabsf(d);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  __shared__ float sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ void image_loss_finish_kernel(const float* __restrict__ partial, int n, float inv_n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    *loss = s * inv_n;
  }
}
// da = g * d(mean loss)/da: kind 0: 2 (a - t) / n, kind 1: sign(a - t) / n   (g: the upstream scalar gradient x the caller's sign)
__global__ __launch_bounds__(256) void image_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ t, int kind, int C, int Ct,
                                                             size_t HW, size_t total, float g_over_n, float* __restrict__ da) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / ((size_t)C * HW), r = i - b * (size_t)C * HW;
    const size_t ti = Ct == C ? i : b * HW + (r % HW);
    const float d = a[i] - t[ti];
    da[i] = kind == 0 ? 2.f * d * g_over_n : (d > 0.f ? g_over_n : (d < 0.f ? -g_over_n : 0.f));
  }
}

inline int il_grid(size_t total) {
  const size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > (size_t)IL_BLOCKS ? (size_t)IL_BLOCKS : g));
}

}  // namespace

extern "C" int paif_channel_affine_nchw_fwd(const float* x, const float* scale, const float* shift, float* out, int B, int C, int H, int W,
                                            paif_stream_t stream) {
  PAIF_REQUIRE(x && scale && out && B > 0 && C > 0 && H > 0 && W > 0, PAIF_EINVAL, "channel_affine_nchw: bad arguments");
  const size_t HW = (size_t)H * W, total = (size_t)B * C * HW;
  hipLaunchKernelGGL(channel_affine_nchw_kernel, dim3(il_grid(total)), dim3(256), 0, paif::as_stream(stream), x, scale, shift, out, C, HW, total);
  PAIF_LAUNCH_CHECK("channel_affine_nchw");
  return 0;
}

extern "C" int paif_image_loss_blocks(void) { return IL_BLOCKS; }

extern "C" int paif_image_loss_fwd(const float* a, const float* target, int kind, int B, int C, int Ct, int H, int W, float* partial, float* loss,
                                   paif_stream_t stream) {
  PAIF_REQUIRE(a && target && partial && loss && B > 0 && C > 0 && H > 0 && W > 0, PAIF_EINVAL, "image_loss: bad arguments");
  PAIF_REQUIRE((kind == 0 || kind == 1) && (Ct == C || Ct == 1), PAIF_EINVAL, "image_loss: kind=%d, target channels %d vs %d", kind, Ct, C);
  const size_t HW = (size_t)H * W, total = (size_t)B * C * HW;
  const int g = il_grid(total);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(image_loss_kernel, dim3(g), dim3(256), 0, st, a, target, kind, C, Ct, HW, total, partial);
  PAIF_LAUNCH_CHECK("image_loss");
  hipLaunchKernelGGL(image_loss_finish_kernel, dim3(1), dim3(64), 0, st, partial, g, 1.0f / (float)total, loss);
  PAIF_LAUNCH_CHECK("image_loss_finish");
  return 0;
}

extern "C" int paif_image_loss_bwd(const float* a, const float* target, int kind, int B, int C, int Ct, int H, int W, float g, float* da,
                                   paif_stream_t stream) {
  PAIF_REQUIRE(a && target && da && B > 0 && C > 0 && H > 0 && W > 0, PAIF_EINVAL, "image_loss_bwd: bad arguments");
  PAIF_REQUIRE((kind == 0 || kind == 1) && (Ct == C || Ct == 1), PAIF_EINVAL, "image_loss_bwd: kind=%d, target channels %d vs %d", kind, Ct, C);
  const size_t HW = (size_t)H * W, total = (size_t)B * C * HW;
  hipLaunchKernelGGL(image_loss_bwd_kernel, dim3(il_grid(total)), dim3(256), 0, paif::as_stream(stream), a, target, kind, C, Ct, HW, total,
                     g / (float)total, da);
  PAIF_LAUNCH_CHECK("image_loss_bwd");
  return 0;
}
