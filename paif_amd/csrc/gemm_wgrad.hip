// Linear-layer weight gradient (training step, DESIGN.md "Plan for the training step" item 3):
//   dW[n][k] = sum_m dY[m][n] * X[m][k]          db[n] = sum_m dY[m][n]
// for y = x W^T + b of the MiT blocks / head MLPs (core/mix_transformer.py:22-25,66-69; core/segformer_head.py:19).
// dY is the gradient at the GEMM output (the activation derivative, where the forward fused one, is the caller's).
//
// Same operand trick as conv_wgrad.hip: with K = 2 consecutive tokens, v_mfma_f32_32x32x2_f32 takes
//   A[i = n][kk] = dY[token kk][n0 + i],   B[kk][j = k] = X[token kk][k0 + j]
// straight from the row-major activations (a wave-load is 2 tokens x 128 B).  A workgroup owns a 32 (n) x 128 (k) tile of
// dW (4 waves x 32 columns of k, all reading the same dY strip) and a slice of the tokens (blockIdx.z); the slices'
// partial tiles are summed in slice order by a second pass (deterministic).  Exact fp32.
#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Operands go through a wave-private LDS region in their natural [token][channel] layout: the MFMA takes ONE dword per lane per
// operand (lane (kk, c) = channel c of token 2j + kk), and fetching those dwords straight from global memory made every MFMA wait
// on a 256-B wave load (the kernel ran at the CU's L1 rate, ~23 TFLOP/s).  Staged with coalesced float4 loads (8 tokens x 128 B
// per wave instruction), the same dword is a conflict-free ds_read_b32.
constexpr int GW_CH = 32;     // tokens per staged chunk

// NT x KT = 32-row n tiles x 32-column k tiles per wave (1 or 2 each: a 64 x 64 wave tile reads every operand dword from LDS for
// two MFMAs and halves how often the dY / X strips are fetched -- with 32 x 32 tiles the X strip of a 1280 x 320 weight was read
// 40 times through L2).  The NEXT chunk's global loads are issued before the MFMAs of the current one (software pipeline): in the
// serial form every 16 MFMAs waited a full memory latency and the kernel ran at a sixth of the fp32 matrix peak.
template <int NT, int KT>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                         float* __restrict__ part_w, float* __restrict__ part_b, int M, int N, int K,
                                                         int mper) {
  extern __shared__ __align__(16) float lds_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * (32 * NT), k0 = blockIdx.y * (128 * KT) + wave * (32 * KT);
  const int mbeg = blockIdx.z * mper, mend = min(M, mbeg + mper);
  float* sD = lds_dyn + wave * ((NT + KT) * GW_CH * 32);      // NT tiles of [token][32]
  float* sX = sD + NT * GW_CH * 32;                            // KT tiles of [token][32]
  const int tl = lane >> 3, q = lane & 7;                      // staging role: token lane, channel quad
  bool nvec[NT], kvec[KT];
#pragma unroll
  for (int t = 0; t < NT; ++t) nvec[t] = (N % 4 == 0 && lddy % 4 == 0 && n0 + 32 * t + 4 * q + 4 <= N);
#pragma unroll
  for (int t = 0; t < KT; ++t) kvec[t] = (K % 4 == 0 && ldx % 4 == 0 && k0 + 32 * t + 4 * q + 4 <= K);
  f32x16 acc[NT][KT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;
  float bsum[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) bsum[u] = 0.f;

  float4 pd[NT][GW_CH / 8], px[KT][GW_CH / 8];
  auto gload = [&](int m0) {
#pragma unroll
    for (int it = 0; it < GW_CH / 8; ++it) {
      const int m = m0 + it * 8 + tl;
      const int mm = min(m, M - 1);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int nc = n0 + 32 * u + 4 * q;
        float4 d;
        if (nvec[u]) d = *reinterpret_cast<const float4*>(dy + (size_t)mm * lddy + nc);
        else {
          const float* r_ = dy + (size_t)mm * lddy;
          d = make_float4(r_[min(nc, N - 1)], r_[min(nc + 1, N - 1)], r_[min(nc + 2, N - 1)], r_[min(nc + 3, N - 1)]);
        }
        if (m >= mend) d = make_float4(0.f, 0.f, 0.f, 0.f);    // tokens past the slice contribute nothing
        pd[u][it] = d;
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int kc = k0 + 32 * t + 4 * q;
        if (kvec[t]) px[t][it] = *reinterpret_cast<const float4*>(x + (size_t)mm * ldx + kc);
        else {
          const float* r_ = x + (size_t)mm * ldx;
          px[t][it] = make_float4(r_[min(kc, K - 1)], r_[min(kc + 1, K - 1)], r_[min(kc + 2, K - 1)], r_[min(kc + 3, K - 1)]);
        }
      }
    }
  };
  if (mbeg < mend) gload(mbeg);
  for (int m0 = mbeg; m0 < mend; m0 += GW_CH) {
#pragma unroll
    for (int it = 0; it < GW_CH / 8; ++it) {
#pragma unroll
      for (int u = 0; u < NT; ++u) *reinterpret_cast<float4*>(sD + u * (GW_CH * 32) + (it * 8 + tl) * 32 + 4 * q) = pd[u][it];
#pragma unroll
      for (int t = 0; t < KT; ++t) *reinterpret_cast<float4*>(sX + t * (GW_CH * 32) + (it * 8 + tl) * 32 + 4 * q) = px[t][it];
    }
    __builtin_amdgcn_wave_barrier();          // wave-private region, LDS operations of one wave complete in order:
    asm volatile("" ::: "memory");            // only the compiler must not reorder across the phases
    if (m0 + GW_CH < mend) gload(m0 + GW_CH); // wave-uniform; in flight during the MFMAs below
#pragma unroll 4
    for (int j = 0; j < GW_CH / 2; ++j) {
      float a[NT], b[KT];
#pragma unroll
      for (int u = 0; u < NT; ++u) { a[u] = sD[u * (GW_CH * 32) + (2 * j + kk) * 32 + c]; bsum[u] += a[u]; }
#pragma unroll
      for (int t = 0; t < KT; ++t) b[t] = sX[t * (GW_CH * 32) + (2 * j + kk) * 32 + c];
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[t], acc[u][t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  float* pw = part_w + (size_t)blockIdx.z * N * K;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * kk;     // C/D layout: row = n, column = lane & 31 = k
        const int kc = k0 + 32 * t + c;
        if (nn < N && kc < K) pw[(size_t)nn * K + kc] = acc[u][t][r];
      }
  if (part_b && blockIdx.y == 0 && wave == 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float bs = bsum[u] + __shfl_xor(bsum[u], 32);                  // the two token halves of the pair
      if (kk == 0 && n0 + 32 * u + c < N) part_b[(size_t)blockIdx.z * N + n0 + 32 * u + c] = bs;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// fp16-pair form (round 5; precision PAIF_CONV_F16X3): the exact kernel above runs at the fp32 matrix rate (16 of a 16-bit MFMA's cycles per
// product).  Here dY and X are staged as two IEEE fp16 pieces each (hi = rn(v), lo = rn(v - hi): 22 significant bits; dY multiplied by the
// exact power of two `gscale` first -- gradients of 1e-7 are far below fp16's range -- and the accumulators by its inverse at the end) into a
// TRANSPOSED wave-private LDS image [channel][piece][token]: the 8 consecutive tokens a lane contributes to v_mfma_f32_32x32x16_f16
// (contraction over tokens) are one ds_read_b128.  Three MFMAs per 16 tokens and tile pair instead of eight fp32 ones.
// ---------------------------------------------------------------------------------------------------------------------------
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
constexpr int GH_ROW = GW_CH * 2 + 16;            // bytes per (channel, piece) row: 32 tokens x fp16 + pad (80 B: conflict-free b128 over 32 channels)
constexpr int GH_TILE = 2 * 32 * GH_ROW;          // one [32 channels][2 pieces] tile

template <int NT, int KT>
__global__ __launch_bounds__(256) void gemm_wgrad_h_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ part_w, float* __restrict__ part_b, int M, int N, int K,
                                                           int mper, float gscale) {
  extern __shared__ __align__(16) char lds_h[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, c = lane & 31;
  const int n0 = blockIdx.x * (32 * NT), k0 = blockIdx.y * (128 * KT) + wave * (32 * KT);
  const int mbeg = blockIdx.z * mper, mend = min(M, mbeg + mper);
  char* sD = lds_h + wave * ((NT + KT) * GH_TILE);
  char* sX = sD + NT * GH_TILE;
  const int tl = lane >> 3, q = lane & 7;                      // staging role: token lane, channel quad
  bool nvec[NT], kvec[KT];
#pragma unroll
  for (int t = 0; t < NT; ++t) nvec[t] = (N % 4 == 0 && lddy % 4 == 0 && n0 + 32 * t + 4 * q + 4 <= N);
#pragma unroll
  for (int t = 0; t < KT; ++t) kvec[t] = (K % 4 == 0 && ldx % 4 == 0 && k0 + 32 * t + 4 * q + 4 <= K);
  f32x16 acc[NT][KT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;
  float4 bsum[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) bsum[u] = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 pd[NT][GW_CH / 8], px[KT][GW_CH / 8];
  auto gload = [&](int m0) {
#pragma unroll
    for (int it = 0; it < GW_CH / 8; ++it) {
      const int m = m0 + it * 8 + tl;
      const int mm = min(m, M - 1);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int nc = n0 + 32 * u + 4 * q;
        float4 d;
        if (nvec[u]) d = *reinterpret_cast<const float4*>(dy + (size_t)mm * lddy + nc);
        else {
          const float* r_ = dy + (size_t)mm * lddy;
          d = make_float4(nc < N ? r_[nc] : 0.f, nc + 1 < N ? r_[nc + 1] : 0.f, nc + 2 < N ? r_[nc + 2] : 0.f, nc + 3 < N ? r_[nc + 3] : 0.f);
        }
        if (m >= mend) d = make_float4(0.f, 0.f, 0.f, 0.f);    // tokens past the slice contribute nothing
        pd[u][it] = d;
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int kc = k0 + 32 * t + 4 * q;
        float4 v;
        if (kvec[t]) v = *reinterpret_cast<const float4*>(x + (size_t)mm * ldx + kc);
        else {
          const float* r_ = x + (size_t)mm * ldx;
          v = make_float4(r_[min(kc, K - 1)], r_[min(kc + 1, K - 1)], r_[min(kc + 2, K - 1)], r_[min(kc + 3, K - 1)]);
        }
        if (m >= mend) v = make_float4(0.f, 0.f, 0.f, 0.f);    // (0 x anything finite: also keeps a padded token's garbage out)
        px[t][it] = v;
      }
    }
  };
  // one float4 (4 channels of one token) -> hi | lo fp16 pieces, scattered into the transposed image as 2-byte stores
  auto put = [&](char* tile, int token, float4 v) {
    const uint2 hi = paif::f32_to_f16x4(v);
    const float4 hf = paif::f16x4_to_f32(hi);
    const uint2 lo = paif::f32_to_f16x4(make_float4(v.x - hf.x, v.y - hf.y, v.z - hf.z, v.w - hf.w));
    char* p0 = tile + (4 * q) * GH_ROW + token * 2;
    *reinterpret_cast<unsigned short*>(p0) = (unsigned short)hi.x;
    *reinterpret_cast<unsigned short*>(p0 + GH_ROW) = (unsigned short)(hi.x >> 16);
    *reinterpret_cast<unsigned short*>(p0 + 2 * GH_ROW) = (unsigned short)hi.y;
    *reinterpret_cast<unsigned short*>(p0 + 3 * GH_ROW) = (unsigned short)(hi.y >> 16);
    char* p1 = p0 + 32 * GH_ROW;
    *reinterpret_cast<unsigned short*>(p1) = (unsigned short)lo.x;
    *reinterpret_cast<unsigned short*>(p1 + GH_ROW) = (unsigned short)(lo.x >> 16);
    *reinterpret_cast<unsigned short*>(p1 + 2 * GH_ROW) = (unsigned short)lo.y;
    *reinterpret_cast<unsigned short*>(p1 + 3 * GH_ROW) = (unsigned short)(lo.y >> 16);
  };
  if (mbeg < mend) gload(mbeg);
  for (int m0 = mbeg; m0 < mend; m0 += GW_CH) {
#pragma unroll
    for (int it = 0; it < GW_CH / 8; ++it) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const float4 d = pd[u][it];
        bsum[u].x += d.x; bsum[u].y += d.y; bsum[u].z += d.z; bsum[u].w += d.w;
        put(sD + u * GH_TILE, it * 8 + tl, make_float4(d.x * gscale, d.y * gscale, d.z * gscale, d.w * gscale));
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) put(sX + t * GH_TILE, it * 8 + tl, px[t][it]);
    }
    __builtin_amdgcn_wave_barrier();          // wave-private region, LDS operations of one wave complete in order:
    asm volatile("" ::: "memory");            // only the compiler must not reorder across the phases
    if (m0 + GW_CH < mend) gload(m0 + GW_CH); // wave-uniform; in flight during the MFMAs below
#pragma unroll
    for (int st = 0; st < GW_CH / 16; ++st) {
      wf16x8 ah[NT], al[NT], bh[KT], bl[KT];
      const int off = c * GH_ROW + st * 32 + g * 16;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        ah[u] = *reinterpret_cast<const wf16x8*>(sD + u * GH_TILE + off);
        al[u] = *reinterpret_cast<const wf16x8*>(sD + u * GH_TILE + 32 * GH_ROW + off);
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        bh[t] = *reinterpret_cast<const wf16x8*>(sX + t * GH_TILE + off);
        bl[t] = *reinterpret_cast<const wf16x8*>(sX + t * GH_TILE + 32 * GH_ROW + off);
      }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u], bh[t], acc[u][t], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bl[t], acc[u][t], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], bh[t], acc[u][t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  const float inv = 1.0f / gscale;
  float* pw = part_w + (size_t)blockIdx.z * N * K;
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * g;      // C/D layout: row = n, column = lane & 31 = k
        const int kc = k0 + 32 * t + c;
        if (nn < N && kc < K) pw[(size_t)nn * K + kc] = acc[u][t][r] * inv;
      }
  if (part_b && blockIdx.y == 0 && wave == 0) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      float4 b = bsum[u];
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {       // over the 8 token lanes
        b.x += __shfl_xor(b.x, m); b.y += __shfl_xor(b.y, m); b.z += __shfl_xor(b.z, m); b.w += __shfl_xor(b.w, m);
      }
      if (tl == 0) {
        const int nb = n0 + 32 * u + 4 * q;
        float* pb = part_b + (size_t)blockIdx.z * N;
        if (nb < N) pb[nb] = b.x;
        if (nb + 1 < N) pb[nb + 1] = b.y;
        if (nb + 2 < N) pb[nb + 2] = b.z;
        if (nb + 3 < N) pb[nb + 3] = b.w;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_wgrad_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                                float* __restrict__ dw, float* __restrict__ db, int splits, int N,
                                                                int K, int accumulate) {
  const size_t total = (size_t)N * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total + (db ? N : 0); i += (size_t)gridDim.x * 256) {
    float v = 0.f;
    if (i < total) {
      for (int s = 0; s < splits; ++s) v += part_w[(size_t)s * total + i];
      dw[i] = accumulate ? dw[i] + v : v;
    } else {
      const size_t n = i - total;
      for (int s = 0; s < splits; ++s) v += part_b[(size_t)s * N + n];
      db[n] = accumulate ? db[n] + v : v;
    }
  }
}

}  // namespace

// token slices: enough workgroups to fill the chip, at least 256 tokens per slice, at most 64 slices
extern "C" int paif_gemm_wgrad_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const int kw = K >= 256 ? 256 : 128;   // k columns per workgroup (two 32-column tiles per wave from K = 256)
  const int nw = N >= 64 ? 64 : 32;      // n rows per workgroup (two 32-row tiles per wave from N = 64)
  const long tiles = (long)((N + nw - 1) / nw) * ((K + kw - 1) / kw);
  long s = (2048 + tiles - 1) / tiles;
  const long maxs = (M + 255) / 256;
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

extern "C" int paif_gemm_wgrad_p(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                                 float* workspace, int accumulate, int precision, float gscale, paif_stream_t stream);

extern "C" int paif_gemm_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                               float* workspace, int accumulate, paif_stream_t stream) {
  return paif_gemm_wgrad_p(dy, lddy, x, ldx, dw, db, M, N, K, splits, workspace, accumulate, 0, 1.0f, stream);
}

// precision 0: exact fp32 MFMA (= paif_gemm_wgrad); PAIF_CONV_F16X3: fp16 pairs, dY scaled by the power of two gscale (undone inside)
extern "C" int paif_gemm_wgrad_p(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                                 float* workspace, int accumulate, int precision, float gscale, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 6, PAIF_EINVAL, "gemm_wgrad: precision=%d", precision);
  PAIF_REQUIRE(precision == 0 || (gscale > 0.f && ldexpf(1.f, ilogbf(gscale)) == gscale), PAIF_EINVAL, "gemm_wgrad: gscale=%g is not a power of two", gscale);
  PAIF_REQUIRE(dy && x && dw && workspace, PAIF_EINVAL, "gemm_wgrad: null pointer");
  PAIF_REQUIRE(M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K, PAIF_EINVAL, "gemm_wgrad: shape %dx%dx%d ld %d/%d", M, N, K, lddy, ldx);
  PAIF_REQUIRE(splits >= 1 && splits <= 64, PAIF_EINVAL, "gemm_wgrad: splits=%d", splits);
  int mper = (M + splits - 1) / splits;
  mper += mper & 1;                                           // token pairs never straddle a slice
  float* part_w = workspace;
  float* part_b = db ? workspace + (size_t)splits * N * K : nullptr;
  hipStream_t st = paif::as_stream(stream);
  const int NT = N >= 64 ? 2 : 1, KT = K >= 256 ? 2 : 1;
  const dim3 grid((N + 32 * NT - 1) / (32 * NT), (K + 128 * KT - 1) / (128 * KT), splits);
  const size_t lds_bytes = (size_t)4 * (NT + KT) * GW_CH * 32 * sizeof(float);   // <= 64 KB
#define PAIF_GW_LAUNCH(NT_, KT_) \
  hipLaunchKernelGGL((gemm_wgrad_kernel<NT_, KT_>), grid, dim3(256), lds_bytes, st, dy, lddy, x, ldx, part_w, part_b, M, N, K, mper)
#define PAIF_GH_LAUNCH(NT_, KT_)                                                                                                          \
  do {                                                                                                                                    \
    const int hb = 4 * (NT_ + KT_) * GH_TILE;                                                                                             \
    if (hb > 64 * 1024) {                                                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wgrad_h_kernel<NT_, KT_>),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, hb);                                                 \
      if (e != hipSuccess) { paif::set_error("gemm_wgrad(f16x3): LDS %d: %s", hb, hipGetErrorString(e)); return (int)e; }                \
    }                                                                                                                                     \
    hipLaunchKernelGGL((gemm_wgrad_h_kernel<NT_, KT_>), grid, dim3(256), hb, st, dy, lddy, x, ldx, part_w, part_b, M, N, K, mper, gscale); \
  } while (0)
  if (precision == 6) {
    if (NT == 2 && KT == 2) PAIF_GH_LAUNCH(2, 2);
    else if (NT == 2) PAIF_GH_LAUNCH(2, 1);
    else if (KT == 2) PAIF_GH_LAUNCH(1, 2);
    else PAIF_GH_LAUNCH(1, 1);
  } else if (NT == 2 && KT == 2) PAIF_GW_LAUNCH(2, 2);
  else if (NT == 2) PAIF_GW_LAUNCH(2, 1);
  else if (KT == 2) PAIF_GW_LAUNCH(1, 2);
  else PAIF_GW_LAUNCH(1, 1);
#undef PAIF_GW_LAUNCH
#undef PAIF_GH_LAUNCH
  PAIF_LAUNCH_CHECK("gemm_wgrad");
  const size_t total = (size_t)N * K + (db ? N : 0);
  const int rb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gemm_wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, part_w, part_b, dw, db, splits, N, K, accumulate);
  PAIF_LAUNCH_CHECK("gemm_wgrad_reduce");
  return 0;
}
