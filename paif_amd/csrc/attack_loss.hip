// The loss glue of the attack variants on bilinearly upsampled logits (attack/attack.py:447-499: attack_way = 'PGD' |
// 'segPGD' | 'cosPGD'; 'newPGD' multiplies the PGD loss by cos/cos == 1 and is served by way 0), forward value and the
// gradient w.r.t. the FULL-RESOLUTION logits, as two HIP kernels -- so that the PGD loop of every variant runs without a
// single ATen op.
//
//   o[c]   = bilinear(logits)[c] at the label's resolution (F.interpolate, align_corners=False)
//   nll    = logsumexp(o) - o[label]                      on valid pixels (label != ignore)
//   way 0  loss = mean_valid(nll)
//   way 1  segPGD (:449-464): pred = max_c o (the VALUE); t = (pred == label) compares that float with the integer label, as
//          the reference does;  loss = (1-l) * CE(t * o) + l * CE((1-t) * o),  CE(0 * o) = log(C) per valid pixel
//          -> per pixel  w * nll + (1 - w') ... written out below with w_true = 1-l, w_false = l
//   way 2  cosPGD (:465-471): loss = cos(pred, label) * CE(o, label),  cos over ALL pixels (ignored ones included, label as
//          a float), cos = <p,l> / sqrt(max(<p,p><l,l>, 1e-16))  (ATen cosine_similarity, eps 1e-8)
// Reductions: per-block partials in a fixed lane order, then ONE wave sums the blocks in a fixed order (no float atomics).
#include <math.h>

#include "paif_common.h"

namespace {

constexpr int AL_MAXC = 32;
constexpr int AL_NPART = 6;   // weighted nll sum, #valid, <p,l>, <p,p>, <l,l>, #labels outside [0,C) that are not ignore_index

inline int al_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 65535 ? (b ? b : 1) : 65535);
}

__device__ __forceinline__ void al_src_index(float scale, int o, int isz, int& i0, int& i1, float& l1) {
  float f = scale * ((float)o + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  i0 = (int)f;
  i1 = i0 + (i0 < isz - 1 ? 1 : 0);
  l1 = f - (float)i0;
}

// coef (device, written by the finish kernel, read by the backward kernel):
//   [0] loss  [1] #valid  [2] plain CE  [3] cos  [4] a = d loss / d nll_pixel base (1/Nv, times cos for way 2)
//   [5] bl = CE / (|p||l|)   [6] bp = CE * cos / |p|^2        (way 2:  d loss / d pred_i = bl * label_i - bp * pred_i)
//   [7] number of labels outside [0, C) that are not ignore_index (0 for a well-formed mask)
template <bool BWD>
__global__ __launch_bounds__(256) void attack_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                          float* __restrict__ partial, float* __restrict__ dfull,
                                                          const float* __restrict__ coef, int way, float w_true, float w_false,
                                                          float upstream, int B, int IH, int IW, int C, int OH, int OW, int ignore, int CP) {
  const size_t total = (size_t)B * OH * OW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const float logC = logf((float)C);
  float acc[AL_NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float a = 0.f, bl = 0.f, bp = 0.f;
  if (BWD) { a = coef[4] * upstream; bl = coef[5] * upstream; bp = coef[6] * upstream; }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    size_t t = i / OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    const long long lab = label[i];
    // a label outside [0, C) that is not ignore_index (torch's cross entropy raises on it): never used as an index -- the pixel is
    // dropped like an ignored one and COUNTED (coef[7]); the attack entry points validate the label range once per call on the host
    const bool bad = lab != (long long)ignore && (lab < 0 || lab >= (long long)C);
    const bool valid = lab != (long long)ignore && !bad;
    if (!BWD && bad) acc[5] += 1.f;
    if (!valid && way != 2) {
      if (BWD) for (int c = 0; c < CP; ++c) dfull[i * CP + c] = 0.f;
      continue;
    }
    int y0, y1, x0, x1; float ly, lx;
    al_src_index(sy, oy, IH, y0, y1, ly);
    al_src_index(sx, ox, IW, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = logits + (((size_t)b * IH + y0) * IW + x0) * C;
    const float* p01 = logits + (((size_t)b * IH + y0) * IW + x1) * C;
    const float* p10 = logits + (((size_t)b * IH + y1) * IW + x0) * C;
    const float* p11 = logits + (((size_t)b * IH + y1) * IW + x1) * C;
#define PAIF_INTERP(c) (hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]))
    float mx = -INFINITY;
    int amax = 0;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
      const float v = PAIF_INTERP(c);
      if (v > mx) { mx = v; amax = c; }          // first maximum, like torch.max(dim)
    }
    float se = 0.f;
#pragma unroll 1
    for (int c = 0; c < C; ++c) se += expf(PAIF_INTERP(c) - mx);
    const float lse = mx + logf(se);
    const float flab = (float)lab;
    const bool tmask = mx == flab;               // segPGD's `pred == label`
    const float w = way == 1 ? (tmask ? w_true : w_false) : 1.f;
    if (!BWD) {
      if (valid) {
        const float nll = lse - PAIF_INTERP((int)lab);
        acc[0] += way == 1 ? (w * nll + (tmask ? w_false : w_true) * logC) : nll;
        acc[1] += 1.f;
      }
      if (way == 2) {
        acc[2] = fmaf(mx, flab, acc[2]);
        acc[3] = fmaf(mx, mx, acc[3]);
        acc[4] = fmaf(flab, flab, acc[4]);
      }
    } else {
      const float g = valid ? a * w : 0.f;
      const float gmax = way == 2 ? (bl * flab - bp * mx) : 0.f;
#pragma unroll 1
      for (int c = 0; c < CP; ++c) {
        float d = 0.f;
        if (c < C) {
          if (valid) d = g * (expf(PAIF_INTERP(c) - lse) - (c == (int)lab ? 1.f : 0.f));
          if (c == amax) d += gmax;
        }
        dfull[i * CP + c] = d;
      }
    }
#undef PAIF_INTERP
  }
  if (!BWD) {
    __shared__ float s[AL_NPART][4];
#pragma unroll
    for (int k = 0; k < AL_NPART; ++k) {
      float v = acc[k];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if ((threadIdx.x & 63) == 0) s[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < AL_NPART) partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (s[threadIdx.x][0] + s[threadIdx.x][1]) + (s[threadIdx.x][2] + s[threadIdx.x][3]);
  }
}

__global__ void attack_loss_finish_kernel(const float* __restrict__ partial, int nblk, int way, float* __restrict__ coef) {
  double acc[AL_NPART];
#pragma unroll
  for (int k = 0; k < AL_NPART; ++k) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) v += (double)partial[(size_t)k * nblk + i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    acc[k] = v;
  }
  if (threadIdx.x == 0) {
    const float nv = (float)acc[1];
    const float mean = (float)(acc[0] / acc[1]);        // way 0 / 2: the plain CE; way 1: the weighted segPGD loss
    float loss = mean, cosv = 1.f, a = 1.f / nv, bl = 0.f, bp = 0.f;
    if (way == 2) {
      const float w12 = (float)acc[2], w1 = (float)acc[3], w2 = (float)acc[4];
      const float n12 = sqrtf(fmaxf(w1 * w2, 1e-16f));
      cosv = w12 / n12;
      loss = cosv * mean;
      a = cosv / nv;
      bl = mean / n12;                      // d cos / d p_i = l_i / n12 - cos * p_i / <p,p>   (clamp inactive)
      bp = mean * cosv / w1;
    }
    coef[0] = loss; coef[1] = nv; coef[2] = mean; coef[3] = cosv; coef[4] = a; coef[5] = bl; coef[6] = bp; coef[7] = (float)acc[5];
  }
}

}  // namespace

extern "C" {

int paif_attack_loss_blocks(int B, int OH, int OW) { return al_grid((size_t)B * OH * OW); }

int paif_attack_loss_fwd(const float* logits, const long long* label, float* partial, float* coef, int way, float w_true, float w_false,
                         int B, int IH, int IW, int C, int OH, int OW, int ignore_index, paif_stream_t stream) {
  PAIF_REQUIRE(logits && label && partial && coef && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, PAIF_EINVAL, "attack_loss_fwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C <= AL_MAXC, PAIF_ENOSUP, "attack_loss_fwd: C=%d (max %d)", C, AL_MAXC);
  PAIF_REQUIRE(way >= 0 && way <= 2, PAIF_EINVAL, "attack_loss_fwd: way %d (0 PGD, 1 segPGD, 2 cosPGD)", way);
  const int nblk = paif_attack_loss_blocks(B, OH, OW);
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(attack_loss_kernel<false>, dim3(nblk), dim3(256), 0, st, logits, label, partial, (float*)nullptr, (const float*)nullptr,
                     way, w_true, w_false, 1.f, B, IH, IW, C, OH, OW, ignore_index, 0);
  PAIF_LAUNCH_CHECK("attack_loss_fwd");
  hipLaunchKernelGGL(attack_loss_finish_kernel, dim3(1), dim3(64), 0, st, partial, nblk, way, coef);
  PAIF_LAUNCH_CHECK("attack_loss_finish");
  return 0;
}

int paif_attack_loss_bwd(const float* logits, const long long* label, const float* coef, float* dfull, int way, float w_true, float w_false,
                         float upstream, int B, int IH, int IW, int C, int OH, int OW, int ignore_index, int CP, paif_stream_t stream) {
  PAIF_REQUIRE(logits && label && coef && dfull && B > 0, PAIF_EINVAL, "attack_loss_bwd: bad arguments");
  PAIF_REQUIRE(C > 0 && C <= AL_MAXC && CP >= C && CP % 4 == 0, PAIF_ENOSUP, "attack_loss_bwd: C=%d CP=%d", C, CP);
  PAIF_REQUIRE(way >= 0 && way <= 2, PAIF_EINVAL, "attack_loss_bwd: way %d", way);
  hipLaunchKernelGGL(attack_loss_kernel<true>, dim3(paif_attack_loss_blocks(B, OH, OW)), dim3(256), 0, paif::as_stream(stream), logits, label,
                     (float*)nullptr, dfull, coef, way, w_true, w_false, upstream, B, IH, IW, C, OH, OW, ignore_index, CP);
  PAIF_LAUNCH_CHECK("attack_loss_bwd");
  return 0;
}

}  // extern "C"
