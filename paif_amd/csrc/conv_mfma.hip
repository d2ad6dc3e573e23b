// Dense k x k convolution (stride 1, same padding, Cout <= 32) as an fp32 implicit GEMM on the
// gfx950 matrix cores: v_mfma_f32_32x32x2_f32 (exact f32 = a k-ordered fmaf chain, 64 FLOP/clk/SIMD).
//
//   GEMM view per workgroup:  M = 8 rows x 32 columns of output pixels (8 "segments" of 32 px),
//                             N = 32 output channels, K = kh*kh*cin per source, sources looped.
//   A operand (pixels x k): halo tile of one NHWC source staged in LDS, pixel stride padded to
//       cin+4 dwords so that a wave's ds_read_b128 (4 consecutive channels per lane, 16-lane groups)
//       is bank-conflict-free ((36*p) mod 64 and (20*p) mod 64 hit 16 distinct 4-bank slots).
//   B operand (k x cout): pre-packed weights, one float4 per lane per (tap, channel-octet), read
//       straight from global memory (<= 200 KB per layer, L2 resident), prefetched one tap ahead.
//   One ds_read_b128 + a quarter of a global_load_dwordx4 feed 4 MFMAs (256 cycles of matrix pipe).
//   D (32 px x 32 cout per segment): lane (h = lane>>5, n = lane&31) holds cout n of pixels
//       (r&3) + 8*(r>>2) + 4*h, r = 0..15 -> every store instruction writes two full 128-B NHWC lines.
//   Epilogue fused: per-channel affine (folded BN / bias), PReLU/ReLU, alpha, up to 3 residual
//       tensors, optional per-block channel sums (ECA average pool).
//
// Replaces: operations_m.py BasicConv / nn.Conv2d inside ResidualDenseBlock (:435-449),
// ResidualModule (:451-464), ECABasicBlock (:368-393), DilConv's 1x1 (:501);
// core/model_fusion_auto.py conv1x1_lf/hf (:501-502), stem_out.0 (:616).
#include <stdlib.h>

#include "conv_dma.h"
#include "paif_common.h"
#include <type_traits>

#ifndef PAIF_UB_HOOKS
#define PAIF_UB_HOOKS 3   // same, gradient-hook kernels (each staged element also carries the saved pre-activation)
#endif
#ifndef PAIF_UB
#define PAIF_UB 11  // staged float4 loads in flight per lane (forward kernels): the whole 3x3 halo tile in one batch
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PAIF_TH
#define PAIF_TH 8
#endif
constexpr int TH = PAIF_TH;    // tile rows  (one segment per row)
constexpr int TW = 32;   // tile cols  (= MFMA M)
constexpr int NTHREADS = 256;
constexpr int SEGS_PER_WAVE = TH / 4;

struct ConvArgs {
  const float* src[3];
  const float* res[3];
  const float4* wpk;
  const float* in_prelu;
  const float* scale;
  const float* shift;
  const float* prelu;
  float* out;
  float* pool_partial;
  float* cpool;            // optional: fused ChannelPool of the output map (paif_conv_desc.cpool; cout = 32), pre-offset, 4 floats per pixel
  float* aux_out;          // optional: pre-activation z = acc*scale+shift (saved for the backward pass)
  const float* in_aux;     // in_act 3/4: pre-activation of the layer whose gradient is being propagated
  const float* in_scale;   // in_act 3/4/5: per-input-channel factor (folded BN scale of that layer) or NULL
  const float* epi_aux;    // epi_dact: tensor whose sign gates the output (d PReLU / d ReLU)
  float alpha, in_alpha;
  int nsrc, in_act, act, cout, epi_dact;
  int B, H, W, tilesX, tilesY, nblk;
  int reverse;             // 1: tiles are walked from the end of each XCD range (serpentine order across consecutive layers)
  int st;                  // activation storage (PAIF_ST_*): 0 fp32/fp32, 1 bf16/bf16, 2 fp32 in / bf16 out, 3 fp16/fp16,
                           // 4 fp16 in (residual maps too) / fp32 out
  int wl0;                 // 1: plain 16-bit weights (PAIF_CONV_BF16 / PAIF_CONV_F16): the kernels' storage codes 4 / 5 / 6 / 7 (+ 8 for fp16)
};

// Transform applied to a staged float4 (channels 4q..4q+3 of one pixel):
//   0 none | 1 PReLU(slope) | 2 ReLU                                   (forward pre-activations)
//   3 v * in_alpha * in_scale[c] * (aux >= 0 ? 1 : slope)              (dgrad through PReLU(BN(.))*alpha)
//   4 v * in_alpha * in_scale[c] * (aux > 0)                           (dgrad through ReLU)
//   5 v * in_alpha * in_scale[c]                                       (dgrad through an affine only)
template <bool HOOKS>
__device__ __forceinline__ float4 stage_xform(const ConvArgs& a, float4 v, float4 aux, float slope, int q) {
  if (a.in_act == 0) return v;
  if (a.in_act == 1) {
    v.x = paif::prelu_f(v.x, slope); v.y = paif::prelu_f(v.y, slope);
    v.z = paif::prelu_f(v.z, slope); v.w = paif::prelu_f(v.w, slope);
    return v;
  }
  if (a.in_act == 2) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
  }
  if (!HOOKS) return v;
  float4 f = make_float4(a.in_alpha, a.in_alpha, a.in_alpha, a.in_alpha);
  if (a.in_scale) {
    const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + q * 4);
    f.x *= s4.x; f.y *= s4.y; f.z *= s4.z; f.w *= s4.w;
  }
  if (a.in_act == 3) {
    f.x *= aux.x >= 0.f ? 1.f : slope; f.y *= aux.y >= 0.f ? 1.f : slope;
    f.z *= aux.z >= 0.f ? 1.f : slope; f.w *= aux.w >= 0.f ? 1.f : slope;
  } else if (a.in_act == 4) {
    f.x *= aux.x > 0.f ? 1.f : 0.f; f.y *= aux.y > 0.f ? 1.f : 0.f;
    f.z *= aux.z > 0.f ? 1.f : 0.f; f.w *= aux.w > 0.f ? 1.f : 0.f;
  }
  return make_float4(v.x * f.x, v.y * f.y, v.z * f.z, v.w * f.w);
}

// Fused epilogue of one wave: y = act(acc*scale+shift)*alpha (+res0 +res1 +res2), NHWC store.
// Wide epilogue through LDS (guide T21: a row-per-lane dword epilogue is store-ISSUE bound): each wave parks its
// accumulator tiles in a private LDS region as [pixel][channel] and re-reads them as float4 per (pixel, channel quad),
// so residual loads and output stores are 16 B per lane / 1 KiB per wave-instruction (8 instead of 32 per tensor).
// Ablation: the dword epilogue was 25 % of the 3x3 kernel (profiles/r01_ablation_conv3x3_bf16x3.txt).
// Must be called after a __syncthreads() that retires every read of the staged tile (the region is reused).
// Per-launch epilogue constants of one lane (channel quad q = lane & 7).  Loaded through pointer SELECTS, not
// branches: a load under a branch makes hipcc wait for it at the join, which put three serial L2 round trips
// (scale, shift, slope) on every tile's critical path.
struct EpiParams {
  float4 sc, sh;
  float slope;
};
__device__ const float k_ones[32] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                     1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
__device__ const float k_zeros[32] = {};

template <bool HOOKS>
__device__ __forceinline__ EpiParams load_epi_params(const ConvArgs& a, int lane) {
  const int q = lane & 7;
  EpiParams e;
  const bool want_slope = a.act == 1 || (HOOKS && a.epi_dact == 1);
  const float* pp = want_slope ? a.prelu : k_zeros;
  if (a.cout == 32) {   // launch-uniform
    const float* sp = a.scale ? a.scale : k_ones;
    const float* hp = a.shift ? a.shift : k_zeros;
    e.sc = *reinterpret_cast<const float4*>(sp + 4 * q);
    e.sh = *reinterpret_cast<const float4*>(hp + 4 * q);
  } else {
    float scv[4], shv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * q + j;
      scv[j] = (a.scale && c < a.cout) ? a.scale[c] : 1.f;
      shv[j] = (a.shift && c < a.cout) ? a.shift[c] : 0.f;
    }
    e.sc = make_float4(scv[0], scv[1], scv[2], scv[3]);
    e.sh = make_float4(shv[0], shv[1], shv[2], shv[3]);
  }
  e.slope = *pp;
  return e;
}

// CH: elements per lane whose residual loads are in flight together (NITER = all, the default; the persistent kernel,
// which holds a prefetched halo tile in registers across its epilogue, uses half)
template <bool FULL, bool HOOKS, int NRES, int SEGS, int CH = SEGS * 32 * 8 / 64, int BFO = 0, int BFR = BFO>
__device__ __forceinline__ float4 epilogue_lds_n(const ConvArgs& a, const EpiParams& ep_par, const f32x16 (&acc)[SEGS],
                                                 float* lds, int b, int y0, int x0, int wave, int lane) {
  float* ep = lds + wave * (SEGS * 32 * 32);
  const int h = lane >> 5, n = lane & 31;
#pragma unroll
  for (int sg = 0; sg < SEGS; ++sg)
#pragma unroll
    for (int r = 0; r < 16; ++r) ep[(sg * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = acc[sg][r];
  // same wave wrote and reads: LDS operations of one wave complete in order (the compiler's lgkmcnt covers the RAW)
  const int q = lane & 7;
  const bool qvalid = FULL || (4 * q < a.cout);
  const float4 sc = ep_par.sc, sh = ep_par.sh;
  const float slope = ep_par.slope;
  constexpr int nres = NRES;   // compile-time: the residual loads are straight-line, not one branch per load
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int NITER = SEGS * 32 * 8 / 64;
  static_assert(NITER % CH == 0, "chunk must divide the element count");
  float4 r0[CH], r1[CH], r2[CH], ea[CH];
  // element (it): pixel it*8 + lane/8 of the wave's SEGS*32 pixels, channel quad q.
  // FULL tiles: (uniform row base) + (compile-time step) + (32-bit lane offset), so the address arithmetic is a
  // handful of scalar ops per row instead of 64-bit vector multiplies per element (VALU issue shares the SIMD
  // with the MFMA stream).
  const unsigned lane_off = (unsigned)(lane >> 3) * 32u + 4u * (unsigned)q;
  auto locate = [&](int it, size_t& off) -> bool {
    if constexpr (FULL) {
      const int sg = it >> 2;
      const size_t rowbase = ((size_t)(b * a.H + y0 + wave * SEGS + sg) * a.W + x0) * 32;
      off = rowbase + (size_t)((it & 3) * 256) + lane_off;
      return true;
    } else {
      const int pix = it * 8 + (lane >> 3);
      const int sg = pix >> 5, px = pix & 31;
      const int y = y0 + wave * SEGS + sg, x = x0 + px;
      off = ((size_t)(b * a.H + y) * a.W + x) * a.cout + 4 * q;
      return qvalid && y < a.H && x < a.W;
    }
  };
#pragma unroll
  for (int it0 = 0; it0 < NITER; it0 += CH) {
#pragma unroll
  for (int it = it0; it < it0 + CH; ++it) {
    size_t off;
    const bool ok = locate(it, off);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    r0[it - it0] = (nres > 0 && ok) ? paif::ldq_nt<BFR>(a.res[0], off) : z4;
    r1[it - it0] = (nres > 1 && ok) ? paif::ldq_nt<BFR>(a.res[1], off) : z4;
    r2[it - it0] = (nres > 2 && ok) ? paif::ldq_nt<BFR>(a.res[2], off) : z4;
    ea[it - it0] = (HOOKS && a.epi_dact && ok) ? *reinterpret_cast<const float4*>(a.epi_aux + off) : z4;
  }
#pragma unroll
  for (int it = it0; it < it0 + CH; ++it) {
    size_t off;
    const bool ok = locate(it, off);
    const int pix = it * 8 + (lane >> 3);
    float4 v = *reinterpret_cast<const float4*>(ep + pix * 32 + 4 * q);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (HOOKS && a.aux_out && ok) *reinterpret_cast<float4*>(a.aux_out + off) = v;
    if (a.act == 1) {
      v.x = paif::prelu_f(v.x, slope); v.y = paif::prelu_f(v.y, slope); v.z = paif::prelu_f(v.z, slope); v.w = paif::prelu_f(v.w, slope);
    } else if (a.act == 2) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
    if (HOOKS && a.epi_dact == 1) {
      v.x *= ea[it - it0].x >= 0.f ? 1.f : slope; v.y *= ea[it - it0].y >= 0.f ? 1.f : slope;
      v.z *= ea[it - it0].z >= 0.f ? 1.f : slope; v.w *= ea[it - it0].w >= 0.f ? 1.f : slope;
    } else if (HOOKS && a.epi_dact == 2) {
      v.x *= ea[it - it0].x > 0.f ? 1.f : 0.f; v.y *= ea[it - it0].y > 0.f ? 1.f : 0.f;
      v.z *= ea[it - it0].z > 0.f ? 1.f : 0.f; v.w *= ea[it - it0].w > 0.f ? 1.f : 0.f;
    }
    if (nres > 0) { v.x += r0[it - it0].x; v.y += r0[it - it0].y; v.z += r0[it - it0].z; v.w += r0[it - it0].w; }
    if (nres > 1) { v.x += r1[it - it0].x; v.y += r1[it - it0].y; v.z += r1[it - it0].z; v.w += r1[it - it0].w; }
    if (nres > 2) { v.x += r2[it - it0].x; v.y += r2[it - it0].y; v.z += r2[it - it0].z; v.w += r2[it - it0].w; }
    if (ok) {
      paif::stq_nt<BFO>(a.out, off, v);
      psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w;
    }
    if (a.cpool) {     // launch-uniform: ChannelPool of the output (the 8 lanes of a pixel; channel_pool2_kernel's summation tree)
      float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), sm = (v.x + v.y) + (v.z + v.w);
      paif::pix8_max_sum(mx, sm);
      if (ok && q == 0) *reinterpret_cast<float2*>(a.cpool + (off >> 5) * 4) = make_float2(mx, sm * (1.0f / 32.0f));     // cout = 32: off / 32 = the pixel
    }
  }
  }
  return psum;   // per-lane partial channel sums of quad q (for the ECA pool)
}

template <bool FULL, bool HOOKS, int SEGS = SEGS_PER_WAVE, int CH = SEGS * 32 * 8 / 64, int BFO = 0, int BFR = BFO>
__device__ __forceinline__ float4 epilogue_lds(const ConvArgs& a, const EpiParams& ep_par, const f32x16 (&acc)[SEGS], float* lds,
                                               int b, int y0, int x0, int wave, int lane) {
  // residuals are packed from index 0; the count is launch-uniform
  if (!a.res[0]) return epilogue_lds_n<FULL, HOOKS, 0, SEGS, CH, BFO, BFR>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  if (!a.res[1]) return epilogue_lds_n<FULL, HOOKS, 1, SEGS, CH, BFO, BFR>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  if (!a.res[2]) return epilogue_lds_n<FULL, HOOKS, 2, SEGS, CH, BFO, BFR>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  return epilogue_lds_n<FULL, HOOKS, 3, SEGS, CH, BFO, BFR>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
}

template <int KH, int DIL, int CIN, bool HOOKS>
__global__ __launch_bounds__(NTHREADS, HOOKS ? 2 : 3) void conv_mfma_f32(ConvArgs a) {
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr int TWH = TW + 2 * P;
  constexpr int THH = TH + 2 * P;
  constexpr int PS = CIN + 4;          // pixel stride in dwords
  constexpr int QPP = CIN / 4;         // float4 per pixel
  constexpr int NO = CIN / 8;          // channel octets
  constexpr int NTAP = KH * KH;
  extern __shared__ __align__(16) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5;
  const int p = lane & 31;

  const int tile_id = paif::xcd_remap(blockIdx.x, a.nblk, a.reverse);  // (b, ty, tx) row-major
  int t = tile_id;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b = t / a.tilesY;
  const int x0 = tx * TW, y0 = ty * TH;

  f32x16 acc[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  float in_slope = 0.f;
  if (a.in_act == 1 || (HOOKS && a.in_act == 3)) in_slope = *a.in_prelu;

  // per-lane LDS read base (dwords): pixel (row = wave*SEGS + s, col = p), channel offset 4*h
  int abase[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s) abase[s] = ((wave * SEGS_PER_WAVE + s) * TWH + p) * PS + 4 * h;

  for (int s = 0; s < a.nsrc; ++s) {
    if (s > 0) __syncthreads();  // all waves finished reading the previous tile
    // ---- stage the halo tile of source s: THH x TWH pixels x CIN channels, zero padded ----------
    const float* src = a.src[s];
    // loads are issued UB at a time before the first LDS write so that UB global loads are in flight
    // per lane (one-at-a-time staging is latency-bound: ~1 us per dependent HBM load)
    constexpr int TOTAL = THH * TWH * QPP;
    constexpr int NIT = (TOTAL + NTHREADS - 1) / NTHREADS;
    constexpr int UB = HOOKS ? PAIF_UB_HOOKS : PAIF_UB;
#pragma unroll
    for (int i0 = 0; i0 < NIT; i0 += UB) {
      float4 v[UB], xa[UB];
      int dst[UB];
      bool inb[UB];
      size_t go[UB];
      // Loads are UNCONDITIONAL on clamped indices and padding is applied afterwards: a global load under an
      // exec-masked branch is followed by s_waitcnt vmcnt(0) at the join, which serialised the UB loads (the asm had
      // load / wait / load / wait).  Out-of-tile slots re-read the tile's last element and are never written.
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = min(tid + (i0 + u) * NTHREADS, TOTAL - 1);
        const int pix = idx / QPP, q = idx - pix * QPP;
        const int tyy = pix / TWH, txx = pix - tyy * TWH;
        const int gy = y0 - P + tyy, gx = x0 - P + txx;
        dst[u] = (i0 + u < NIT && tid + (i0 + u) * NTHREADS < TOTAL) ? pix * PS + q * 4 : -1;
        inb[u] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        go[u] = ((size_t)(b * a.H + gyc) * a.W + gxc) * CIN + q * 4;
        v[u] = *reinterpret_cast<const float4*>(src + go[u]);
        xa[u] = v[u];
      }
      if (HOOKS && (a.in_act == 3 || a.in_act == 4)) {   // launch-uniform
#pragma unroll
        for (int u = 0; u < UB; ++u) xa[u] = *reinterpret_cast<const float4*>(a.in_aux + go[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (!inb[u]) v[u] = xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (dst[u] >= 0) {
          const float4 t4 = stage_xform<HOOKS>(a, v[u], xa[u], in_slope, (dst[u] % PS) >> 2);
          *reinterpret_cast<float4*>(lds + dst[u]) = t4;
        }
      }
    }
    __syncthreads();

    // ---- K loop: taps x channel octets, B prefetched one tap ahead -----------------------------
    const float4* wsrc = a.wpk + (size_t)s * NTAP * NO * 64 + lane;
    float4 bcur[NO], bnxt[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) bcur[o] = wsrc[o * 64];
#pragma unroll 1
    for (int tap = 0; tap < NTAP; ++tap) {
      if (tap + 1 < NTAP) {
#pragma unroll
        for (int o = 0; o < NO; ++o) bnxt[o] = wsrc[((tap + 1) * NO + o) * 64];
      }
      const int dy = tap / KH, dx = tap - dy * KH;
      const int toff = (dy * DIL * TWH + dx * DIL) * PS;
#pragma unroll
      for (int o = 0; o < NO; ++o) {
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) {
          const float4 av = *reinterpret_cast<const float4*>(lds + abase[sg] + toff + 8 * o);
          acc[sg] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bcur[o].x, acc[sg], 0, 0, 0);
          acc[sg] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bcur[o].y, acc[sg], 0, 0, 0);
          acc[sg] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bcur[o].z, acc[sg], 0, 0, 0);
          acc[sg] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bcur[o].w, acc[sg], 0, 0, 0);
        }
      }
#pragma unroll
      for (int o = 0; o < NO; ++o) bcur[o] = bnxt[o];
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------
  const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W) && (a.cout == 32);  // block-uniform
  const EpiParams ep_par = load_epi_params<HOOKS>(a, lane);   // in flight across the barrier
  __syncthreads();  // every wave has finished reading the staged tile: its LDS is reused by the epilogue
  float4 ps;
  if (full) ps = epilogue_lds<true, HOOKS>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  else ps = epilogue_lds<false, HOOKS>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  if (a.pool_partial) {
    // lanes with equal (lane & 7) hold the same channel quad for different pixels: reduce over the 8 pixel lanes,
    // then over the 4 waves through LDS (fixed order -> deterministic)
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
      ps.x += __shfl_xor(ps.x, m); ps.y += __shfl_xor(ps.y, m); ps.z += __shfl_xor(ps.z, m); ps.w += __shfl_xor(ps.w, m);
    }
    __syncthreads();  // the per-wave epilogue regions are no longer needed
    if (lane < 8) *reinterpret_cast<float4*>(lds + wave * 32 + lane * 4) = ps;
    __syncthreads();
    if (tid < 32) a.pool_partial[(size_t)tile_id * 32 + tid] = lds[tid] + lds[32 + tid] + lds[64 + tid] + lds[96 + tid];
  }
}


// ---------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: every fp32 operand x is split into hi = bf16(x) and lo = bf16(x - hi)
// and the product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (the dropped lo*lo term is 2^-16 relative to the product).  3 bf16 MFMAs of K=16 replace 8 fp32 MFMAs of
// K=2: 5.3x less matrix-pipe time at ~1e-5 relative accuracy, which keeps the fp32 parity tolerance while
// moving the k x k convs from the fp32-MFMA roof to (nearly) the HBM roof.
//   LDS pixel record (144 B): 32 x bf16 hi | 32 x bf16 lo | 16 B pad (conflict-free ds_read_b128);
//   A fragment: lane (p = pixel, hh) reads 8 consecutive channels (16 B) of its pixel: channels 16*ks + 8*hh + j;
//   B fragment: packed weights, lane (n = cout, hh) holds w[n][16*ks + 8*hh + j][tap], hi and lo streams.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// NP = 3 ("bf16x6", fp32 storage only): every operand as THREE bf16 pieces (hi + mid + lo = the fp32 value to 2^-27) and the six
// products down to 2^-25 relative -- fp32-level parity at 6 bf16 MFMAs per product (the exact fp32 MFMA costs 16 of their cycles).
// PF = 1 ("f16x3", fp32 storage): the two pieces are IEEE fp16 (22 significant bits per operand instead of 16), the
// weight pack carries fp16 pieces of 2^8 * w (exact scale, undone on the accumulators): ~2^-21.5 per product for O(1) data at three
// MFMAs -- the forward passes of the attack loops (gemm_mfma.hip has the same form and the reasoning about fp16's exponent range).
template <int KH, int DIL, bool HOOKS, int ST = 0, int NP = 2, int PF = 0>
#ifndef PAIF_LB
#define PAIF_LB 3
#endif
// two workgroups per CU (256 VGPRs) for the gradient-hook kernels and the wide halos (k >= 5, dilation 2: at three the
// staging batch spills 84-220 B/lane and the kernels measure 5-25 % slower), three for the 1x1 / 3x3 dilation-1 kernels
__device__ __forceinline__ void conv_mfma_split_body(const ConvArgs& a) {
  constexpr int CIN = 32;
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr int TWH = TW + 2 * P;
  constexpr int THH = TH + 2 * P;
  static_assert(NP == 2 || (NP == 3 && ST == 0), "the three-piece split is built for fp32 storage");
  static_assert(PF == 0 || (NP == 2 && ST == 0), "fp16 pairs: fp32 storage");
  constexpr int FM = PF == 1 ? 2 : paif::st_fmt16(ST);     // 16-bit format of the MFMA operands (1 bf16, 2 fp16)
  constexpr int PSB = NP == 3 ? 208 : 144;   // pixel record in bytes: NP x 64 + 16 (13 / 9 sixteen-byte slots: odd -> conflict-free b128)
  constexpr int QPP = CIN / 4;
  constexpr int NKS = CIN / 16;        // K=16 steps per tap
  constexpr int NTAP = KH * KH;
  extern __shared__ __align__(16) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hh = lane >> 5;
  const int p = lane & 31;

  const int tile_id = paif::xcd_remap(blockIdx.x, a.nblk, a.reverse);
  int t = tile_id;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b = t / a.tilesY;
  const int x0 = tx * TW, y0 = ty * TH;

  f32x16 acc[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  float in_slope = 0.f;
  if (a.in_act == 1 || (HOOKS && a.in_act == 3)) in_slope = *a.in_prelu;

  int abase[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s) abase[s] = ((wave * SEGS_PER_WAVE + s) * TWH + p) * PSB + 16 * hh;

  for (int s = 0; s < a.nsrc; ++s) {
    if (s > 0) __syncthreads();
    const float* src = a.src[s];
    constexpr int TOTAL = THH * TWH * QPP;
    constexpr int NIT = (TOTAL + NTHREADS - 1) / NTHREADS;
    constexpr int UB = HOOKS ? PAIF_UB_HOOKS : PAIF_UB;
#pragma unroll
    for (int i0 = 0; i0 < NIT; i0 += UB) {
      float4 v[UB], xa[UB];
      int dst[UB];
      bool inb[UB];
      size_t go[UB];
      // Loads are UNCONDITIONAL on clamped indices and padding is applied afterwards: a global load under an
      // exec-masked branch is followed by s_waitcnt vmcnt(0) at the join, which serialised the UB loads (the asm had
      // load / wait / load / wait).  Out-of-tile slots re-read the tile's last element and are never written.
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int idx = min(tid + (i0 + u) * NTHREADS, TOTAL - 1);
        const int pix = idx / QPP, q = idx - pix * QPP;
        const int tyy = pix / TWH, txx = pix - tyy * TWH;
        const int gy = y0 - P + tyy, gx = x0 - P + txx;
        dst[u] = (i0 + u < NIT && tid + (i0 + u) * NTHREADS < TOTAL) ? pix * PSB + q * 8 : -1;
        inb[u] = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        go[u] = ((size_t)(b * a.H + gyc) * a.W + gxc) * CIN + q * 4;
        v[u] = paif::ldq<paif::st_in(ST)>(src, go[u]);
        xa[u] = v[u];
      }
      if (HOOKS && (a.in_act == 3 || a.in_act == 4)) {   // launch-uniform
#pragma unroll
        for (int u = 0; u < UB; ++u) xa[u] = *reinterpret_cast<const float4*>(a.in_aux + go[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (!inb[u]) v[u] = xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (dst[u] >= 0) {
          const float4 t4 = stage_xform<HOOKS>(a, v[u], xa[u], in_slope, (dst[u] % PSB) >> 3);
          if constexpr (paif::st_f16(ST)) {   // fp16 maps: the staged operand is ONE fp16 value (stored as such, or rounded after the input PReLU)
            *reinterpret_cast<uint2*>(ldsb + dst[u]) = paif::f32_to_f16x4(t4);
            continue;
          }
          if constexpr (PF == 1) {             // fp16 hi | lo pieces of the fp32 value
            const uint2 hi = paif::f32_to_f16x4(t4);
            const float4 hf = paif::f16x4_to_f32(hi);
            *reinterpret_cast<uint2*>(ldsb + dst[u]) = hi;
            *reinterpret_cast<uint2*>(ldsb + dst[u] + 64) = paif::f32_to_f16x4(make_float4(t4.x - hf.x, t4.y - hf.y, t4.z - hf.z, t4.w - hf.w));
            continue;
          }
          const __bf16 hx = (__bf16)t4.x, hy = (__bf16)t4.y, hz = (__bf16)t4.z, hw = (__bf16)t4.w;
          uint2 hi, lo;
          hi.x = (unsigned)__builtin_bit_cast(unsigned short, hx) | ((unsigned)__builtin_bit_cast(unsigned short, hy) << 16);
          hi.y = (unsigned)__builtin_bit_cast(unsigned short, hz) | ((unsigned)__builtin_bit_cast(unsigned short, hw) << 16);
          lo.x = pack_bf16(t4.x - (float)hx, t4.y - (float)hy);
          lo.y = pack_bf16(t4.z - (float)hz, t4.w - (float)hw);
          *reinterpret_cast<uint2*>(ldsb + dst[u]) = hi;
          if constexpr (NP == 3) {      // lo computed above is the MIDDLE piece; the third: what is left of the residual
            const float rx = (t4.x - (float)hx) - __uint_as_float(lo.x << 16), ry = (t4.y - (float)hy) - __uint_as_float(lo.x & 0xffff0000u);
            const float rz = (t4.z - (float)hz) - __uint_as_float(lo.y << 16), rw = (t4.w - (float)hw) - __uint_as_float(lo.y & 0xffff0000u);
            *reinterpret_cast<uint2*>(ldsb + dst[u] + 64) = lo;
            *reinterpret_cast<uint2*>(ldsb + dst[u] + 128) = make_uint2(pack_bf16(rx, ry), pack_bf16(rz, rw));
          } else if constexpr (!paif::st_lo0(ST)) {
            *reinterpret_cast<uint2*>(ldsb + dst[u] + 64) = lo;
          }
        }
      }
    }
    __syncthreads();

    const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpk) + (size_t)s * NTAP * NKS * NP * 64 + lane;
    // B operand: a statically indexed 3-slot register ring, two taps ahead.  A rolled loop with `bcur = bnxt` copies
    // a pending load and therefore waits for the prefetch at the end of every tap (it only ever overlapped one tap's 12
    // MFMAs with the L2 latency).  The tap loop runs in groups of 3 (slot = position in the group; every k*k is 3n or
    // 3n+1, the tail tap lands on slot 0); k = 3 unrolls fully, k >= 5 keeps the groups in a rolled outer loop
    // (full unrolling of 49 taps costs minutes of compile time per instantiation).
    static_assert(NTAP % 3 == 0 || NTAP % 3 == 1, "tap count must be 3n or 3n+1");
    constexpr int BRING = NTAP >= 3 ? 3 : 1;
    uint4 bw[BRING][NKS * NP];
    auto fetch = [&](int tap, int slot) {
#pragma unroll
      for (int i = 0; i < NKS * NP; ++i)
        if (!(paif::st_wl0(ST) && (i & 1))) bw[slot][i] = wsrc[(tap * NKS * NP + i) * 64];
    };
    // A operand of one K=16 step (hi and lo halves of this wave's two row segments), double-buffered in registers: the
    // reads of step i+1 are issued before the MFMAs of step i and pinned there with sched_barrier.  Left to itself the
    // scheduler sinks every ds_read (and the B prefetch) to just above its first use to save registers, and each MFMA
    // group then eats the full LDS / L2 latency (measured: 9.9k -> 5.6k cycles per 3x3 source on the MFMA phase).
    struct AStep { bf16x8 h[SEGS_PER_WAVE], l[SEGS_PER_WAVE], t[NP == 3 ? SEGS_PER_WAVE : 1]; };
    auto readA = [&](AStep& A, int tap, int ks) {
      const int dy = tap / KH, dx = tap - dy * KH;
      const int toff = (dy * DIL * TWH + dx * DIL) * PSB + 32 * ks;
#pragma unroll
      for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) {
        A.h[sg] = *reinterpret_cast<const bf16x8*>(ldsb + abase[sg] + toff);
        if constexpr (!paif::st_lo0(ST)) A.l[sg] = *reinterpret_cast<const bf16x8*>(ldsb + abase[sg] + toff + 64);
        if constexpr (NP == 3) A.t[sg] = *reinterpret_cast<const bf16x8*>(ldsb + abase[sg] + toff + 128);
      }
    };
    auto mma_step = [&](const AStep& A, int slot, int ks) {
      if constexpr (NP == 3) {     // h = hi, l = mid, t = lo of both operands; smallest products first
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, bw[slot][3 * ks]), b1 = __builtin_bit_cast(bf16x8, bw[slot][3 * ks + 1]),
                     b2 = __builtin_bit_cast(bf16x8, bw[slot][3 * ks + 2]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.h[sg], b2, acc[sg]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.t[sg], b0, acc[sg]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.l[sg], b1, acc[sg]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.h[sg], b1, acc[sg]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.l[sg], b0, acc[sg]);
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.h[sg], b0, acc[sg]);
        return;
      }
      const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[slot][2 * ks]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[slot][2 * ks + 1]);
      if constexpr (!paif::st_lo0(ST)) {
#pragma unroll
        for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.l[sg], bh, acc[sg]);
      }
      if constexpr (!paif::st_wl0(ST)) {
#pragma unroll
      for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.h[sg], bl, acc[sg]);
      }
#pragma unroll
      for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = paif::mfma16<FM>(A.h[sg], bh, acc[sg]);
    };
    static_assert(NKS == 2, "the A double buffer alternates on the K step");
    AStep A[2];
    // tap `tap` from ring slot `slot`; afterwards the slot is refilled with tap + 3 (two taps of MFMAs ahead of its use)
    auto mma_tap = [&](int tap, int slot) {
      readA(A[1], tap, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(A[0], slot, 0);
      __builtin_amdgcn_sched_barrier(0);
      // k <= 3: `tap` is a compile-time constant after unrolling and the tails fold away; k >= 5 (rolled groups): clamped,
      // unconditional (a branch around loads would make every wait count conservative)
      if constexpr (NTAP <= 9) {
        if (tap + 1 < NTAP) readA(A[0], tap + 1, 0);
      } else {
        readA(A[0], min(tap + 1, NTAP - 1), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_step(A[1], slot, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NTAP <= 9) {
        if (tap + 3 < NTAP) fetch(tap + 3, slot);
      } else {
        fetch(min(tap + 3, NTAP - 1), slot);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    fetch(0, 0);
    if constexpr (NTAP > 1) {
      fetch(1, 1);
      fetch(2, 2);
    }
    readA(A[0], 0, 0);
    if constexpr (NTAP == 1) {
      mma_tap(0, 0);
    } else {
      constexpr int NGRP = NTAP / 3;
      if constexpr (NTAP <= 9) {
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          mma_tap(3 * g, 0);
          mma_tap(3 * g + 1, 1);
          mma_tap(3 * g + 2, 2);
        }
      } else {
#pragma unroll 1
        for (int g = 0; g < NGRP; ++g) {
          mma_tap(3 * g, 0);
          mma_tap(3 * g + 1, 1);
          mma_tap(3 * g + 2, 2);
        }
      }
      if constexpr (NTAP % 3 == 1) mma_tap(NTAP - 1, 0);
    }
  }

  if constexpr (PF == 1) {   // undo the 2^8 of the weight pack (exact)
#pragma unroll
    for (int s = 0; s < SEGS_PER_WAVE; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] *= 1.0f / 256.f;
  }
  const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W) && (a.cout == 32);  // block-uniform
  const EpiParams ep_par = load_epi_params<HOOKS>(a, lane);   // in flight across the barrier
  __syncthreads();  // every wave has finished reading the staged tile: its LDS is reused by the epilogue
  float4 ps;
  if (full) ps = epilogue_lds<true, HOOKS, SEGS_PER_WAVE, SEGS_PER_WAVE * 32 * 8 / 64, paif::st_out(ST), paif::st_res(ST)>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  else ps = epilogue_lds<false, HOOKS, SEGS_PER_WAVE, SEGS_PER_WAVE * 32 * 8 / 64, paif::st_out(ST), paif::st_res(ST)>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  if (a.pool_partial) {
    // lanes with equal (lane & 7) hold the same channel quad for different pixels: reduce over the 8 pixel lanes,
    // then over the 4 waves through LDS (fixed order -> deterministic)
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
      ps.x += __shfl_xor(ps.x, m); ps.y += __shfl_xor(ps.y, m); ps.z += __shfl_xor(ps.z, m); ps.w += __shfl_xor(ps.w, m);
    }
    __syncthreads();  // the per-wave epilogue regions are no longer needed
    if (lane < 8) *reinterpret_cast<float4*>(lds + wave * 32 + lane * 4) = ps;
    __syncthreads();
    if (tid < 32) a.pool_partial[(size_t)tile_id * 32 + tid] = lds[tid] + lds[32 + tid] + lds[64 + tid] + lds[96 + tid];
  }
}

// One kernel name per arithmetic of the tile conv (round 6: rocprofv3 rows say what ran): two bf16 pieces (3 MFMAs per product), three
// bf16 pieces (6), two IEEE fp16 pieces (3, fp32-level); ST = storage code (paif_common.h)
#define PAIF_CONV_LB(HOOKS, KH, DIL, NP) __launch_bounds__(NTHREADS, ((HOOKS) || (KH) >= 5 || (DIL) == 2 || (NP) == 3) ? 2 : PAIF_LB)
template <int KH, int DIL, bool HOOKS, int ST = 0>
__global__ PAIF_CONV_LB(HOOKS, KH, DIL, 2) void conv_mfma_bf16x3(ConvArgs a) { conv_mfma_split_body<KH, DIL, HOOKS, ST, 2, 0>(a); }
template <int KH, int DIL, bool HOOKS, int ST = 0>
__global__ PAIF_CONV_LB(HOOKS, KH, DIL, 3) void conv_mfma_bf16x6(ConvArgs a) { conv_mfma_split_body<KH, DIL, HOOKS, ST, 3, 0>(a); }
template <int KH, int DIL, bool HOOKS, int ST = 0>
__global__ PAIF_CONV_LB(HOOKS, KH, DIL, 2) void conv_mfma_f16x3(ConvArgs a) { conv_mfma_split_body<KH, DIL, HOOKS, ST, 2, 1>(a); }

template <int KH, int DIL, bool HOOKS, int ST = 0, int NP = 2, int PF = 0>
int launch_bf16x3_h(const ConvArgs& a, hipStream_t st) {
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr size_t tile_bytes = (size_t)(TH + 2 * P) * (TW + 2 * P) * (NP == 3 ? 208 : 144);
  constexpr size_t epi_bytes = (size_t)TH * 32 * 32 * 4;
  constexpr size_t lds_bytes = tile_bytes > epi_bytes ? tile_bytes : epi_bytes;
  static_assert(lds_bytes <= 160 * 1024, "tile does not fit LDS");
  auto kern = [] {
    if constexpr (PF == 1) return &conv_mfma_f16x3<KH, DIL, HOOKS, ST>;
    else if constexpr (NP == 3) return &conv_mfma_bf16x6<KH, DIL, HOOKS, ST>;
    else return &conv_mfma_bf16x3<KH, DIL, HOOKS, ST>;
  }();
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      paif::set_error("conv2d(bf16x3): cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(NTHREADS), lds_bytes, st, a);
  PAIF_LAUNCH_CHECK("conv2d(bf16x3)");
  return 0;
}


// ---------------------------------------------------------------------------------------------------
// Wave-specialised persistent form of the split-bf16 kernel (forward, no hooks, no in-activation, cout == 32).
// The plain kernel runs load -> LDS -> MFMA -> store as serial phases of one workgroup and relies on 3
// workgroups per CU drifting apart to overlap them.  Here the overlap is structural: one workgroup per CU,
// 12 waves in three roles, one wave of each role per SIMD:
//   LOADERS  (waves 8-11): stream (tile, source) stages HBM -> registers -> split-bf16 -> LDS tile[stage & 1];
//            the loads of stage g+1 are issued before stage g is converted (two register sets, counted vmcnt);
//   MFMA     (waves 0-3):  run the MFMA loop on tile[(g-1) & 1]; their only memory traffic is the B-operand
//            ring from L2 (loads only); after a tile's last source they park the accumulators in LDS;
//   STORERS  (waves 4-7):  pick the parked tile up, apply scale/shift/activation/alpha, add the residual maps
//            and store -- the only waves that mix loads and stores.
// Why three roles (cycle-stamped timeline + ablations of the two-role form):
//   * hipcc's wait-count pass treats vmcnt as out-of-order once a wave has loads AND stores pending and then
//     waits vmcnt(0): a wave that loads B and stores outputs stalls on its own stores, a wave that loads tiles
//     and stores outputs loses its prefetch.  Keeping every role single-kind keeps the waits counted.
//   * VMEM issue back-pressure (store queue behind the HBM read stream) stalled the MFMA wave for 2.6 us per
//     tile when it issued the stores itself.
//   * MFMA and VALU of the waves of one SIMD share its issue port: the loader is written for instruction count
//     (per-lane tile coordinates computed once per launch; med3 + mad + shift per load with 32-bit offsets from a
//     scalar base; the hi/lo split is 10 packed instructions per float4) and the MFMA waves run no address math.
// A step has two LDS-only barriers (no vmcnt drain): phase A = load/convert | MFMA | store, phase B = park.
// Tiles are dealt so that the 32 workgroups of one XCD walk a contiguous tile range (halo rows hit that XCD's L2).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// fp32 x4 -> (hi bf16 x4, lo bf16 x4): v_cvt_pk_bf16_f32 / shift / and / v_pk_add_f32 (10 instructions)
__device__ __forceinline__ void split_bf16x4(float4 t, uint2& hi, uint2& lo) {
  const f32x2 a = {t.x, t.y}, b = {t.z, t.w};
  const unsigned ua = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
  const unsigned ub = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2));
  const f32x2 ra = {t.x - __uint_as_float(ua << 16), t.y - __uint_as_float(ua & 0xffff0000u)};
  const f32x2 rb = {t.z - __uint_as_float(ub << 16), t.w - __uint_as_float(ub & 0xffff0000u)};
  hi = make_uint2(ua, ub);
  lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(ra, bf16x2)),
                  __builtin_bit_cast(unsigned, __builtin_convertvector(rb, bf16x2)));
}


// A staged 4-channel quad as it travels in registers between the global load and the LDS write: the fp32 quad, or the
// bf16 quad of a bf16-stored map (half the prefetch registers; its split is the identity: hi = the stored bits, lo = 0)
template <int BF> struct RawQ { typedef float4 T; };
template <> struct RawQ<1> { typedef uint2 T; };
template <> struct RawQ<2> { typedef uint2 T; };   // fp16 quad
template <int BF> __device__ __forceinline__ typename RawQ<BF>::T ldraw(const char* p) {
  return *reinterpret_cast<const typename RawQ<BF>::T*>(p);
}
template <int BF> __device__ __forceinline__ typename RawQ<BF>::T ldraw_nt(const char* p) {
  if constexpr (BF) {
    typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
    const u32x2_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(p));
    return make_uint2(v.x, v.y);
  } else {
    return paif::load_nt(reinterpret_cast<const float*>(p));
  }
}
__device__ __forceinline__ void split_raw(float4 t, uint2& hi, uint2& lo) { split_bf16x4(t, hi, lo); }
__device__ __forceinline__ void split_raw(uint2 t, uint2& hi, uint2& lo) { hi = t; lo = make_uint2(0u, 0u); }
__device__ __forceinline__ float4 raw_zero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ uint2 raw_zero(uint2) { return make_uint2(0u, 0u); }
// ReLU of a staged quad.  bf16 / fp16 pairs (both sign-magnitude): as signed 16-bit integers every negative value (sign bit) is below 0 -- one packed max
__device__ __forceinline__ float4 raw_relu(float4 t) { return make_float4(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f)); }
__device__ __forceinline__ uint2 raw_relu(uint2 t) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 z = {0, 0};
  const s16x2 a = __builtin_elementwise_max(__builtin_bit_cast(s16x2, t.x), z), b = __builtin_elementwise_max(__builtin_bit_cast(s16x2, t.y), z);
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

// ---------------------------------------------------------------------------------------------------
// Multi-source form of the tile-per-workgroup split-bf16 kernel (forward, no hooks, no in-activation, cout == 32):
// the sources of the virtual concat (RDB conv2 / conv3: 2 and 3 sources) are software-pipelined INSIDE one instruction
// stream.  What made that impossible in the plain kernel is the in-order vmcnt: a B-operand fetch issued inside the MFMA
// loop waits for every older load, so a register prefetch of the next source's halo tile stalled each B fetch behind it.
// Here the WHOLE B operand of a source (9 taps x 4 x 16 B per lane = 144 VGPRs) is resident before its MFMA loop starts:
// Here the B operand runs PAIF_MS_DEPTH taps ahead over the flattened (source, tap) sequence in a register ring:
//   prologue : B[0..D) loads, A(0) loads -> split-bf16 -> LDS
//   source s : issue A(s+1) loads (registers) ; for each tap: MFMAs from LDS + ring slot, refill the slot with tap g+D ;
//              LDS barrier ; A(s+1) -> split-bf16 -> LDS ; LDS barrier
// The first D taps of a source were fetched before the A(s+1) prefetch was issued, so they never wait behind it; the
// later taps do, D taps (>= 2000 cycles of MFMA) after it was issued -- by then the tile has arrived.
// ---------------------------------------------------------------------------------------------------
// Round 4: ring depth and occupancy per configuration (measured inside bench.py, fp32 maps, B=8 480x640; each a same-box A/B):
//   two sources  : depth 2, THREE workgroups per CU (168 VGPRs; 3 x 49 KB of LDS)  298 -> 271 us -- with two sources the exposed load of
//                  the first tile and the epilogue are 53 % of a workgroup's time (round-2 stamps), two workgroups cannot cover that;
//   three sources: depth 3, two workgroups (3 do not fit without spilling the loop: 455 us)  432 -> 424 us;
//   bf16-stored inputs / outputs (ST != 0) and one source keep depth 6, two workgroups (not re-measured).
// PAIF_MS_DEPTH / PAIF_MS_WGS override both for A/B builds (tools/build_variant.sh).
template <int NSRC, int ST> struct MsCfg {
#ifdef PAIF_MS_DEPTH
  static constexpr int D = PAIF_MS_DEPTH;
#else
  static constexpr int D = (ST == 0 && NSRC == 2) ? 2 : (ST == 0 && NSRC == 3) ? 3 : 6;
#endif
#ifdef PAIF_MS_WGS
  static constexpr int WGS = PAIF_MS_WGS;
#else
  static constexpr int WGS = (ST == 0 && NSRC == 2) ? 3 : 2;
#endif
};
#ifndef PAIF_MS_STAMP
#define PAIF_MS_STAMP(i)   // tools/microbench/conv_ms_trace.hip defines these to record per-phase clock stamps
#define PAIF_TRACE_DECL
#define PAIF_TRACE(i)
#define PAIF_TRACE_END
#endif
template <int KH, int DIL, int NSRC, int ST = 0>
__global__ __launch_bounds__(NTHREADS, (MsCfg<NSRC, ST>::WGS)) void conv_bf16x3_ms(ConvArgs a) {
  constexpr int BFI = paif::st_in(ST);
  constexpr unsigned ES = BFI ? 2u : 4u;               // bytes per stored input element
  typedef typename RawQ<BFI>::T raw_t;
  constexpr int CIN = 32;
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr int TWH = TW + 2 * P;
  constexpr int THH = TH + 2 * P;
  constexpr int PSB = 144;             // pixel record in bytes
  constexpr int QPP = CIN / 4;
  constexpr int NKS = CIN / 16;        // K=16 steps per tap
  constexpr int NTAP = KH * KH;
  constexpr int TOTAL = THH * TWH * QPP;
  constexpr int NIT = (TOTAL + NTHREADS - 1) / NTHREADS;
  extern __shared__ __align__(16) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hh = lane >> 5;
  const int p = lane & 31;

  const int tile_id = paif::xcd_remap(blockIdx.x, a.nblk, a.reverse);
  int t = tile_id;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b = t / a.tilesY;
  const int x0 = tx * TW, y0 = ty * TH;

  f32x16 acc[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  int abase[SEGS_PER_WAVE];
#pragma unroll
  for (int s = 0; s < SEGS_PER_WAVE; ++s) abase[s] = ((wave * SEGS_PER_WAVE + s) * TWH + p) * PSB + 16 * hh;

  // staging slot u of this thread: element idx = tid + u * NTHREADS of the halo tile (pixel-major, 8 channel quads per
  // pixel).  The same 32-bit byte offsets / LDS addresses / padding bits serve every source (identical NHWC-32 shapes).
  unsigned goff[NIT];
  unsigned padmask = 0;
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int idx = min(tid + u * NTHREADS, TOTAL - 1);
    const int pix = idx / QPP, q = idx - pix * QPP;
    const int tyy = pix / TWH, txx = pix - tyy * TWH;
    const int gy = y0 - P + tyy, gx = x0 - P + txx;
    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
    goff[u] = (unsigned)(((b * a.H + gyc) * a.W + gxc) * CIN + q * 4) * ES;
    if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) padmask |= 1u << u;
  }
  constexpr bool LAST_PARTIAL = (TOTAL % NTHREADS) != 0;
  const bool last_valid = tid + (NIT - 1) * NTHREADS < TOTAL;
  auto issueA = [&](const float* __restrict__ src, raw_t (&v)[NIT]) {
    const char* base = reinterpret_cast<const char*>(src);
#pragma unroll
    for (int u = 0; u < NIT; ++u) v[u] = ldraw<BFI>(base + goff[u]);   // unconditional, clamped
  };
  auto convertA = [&](raw_t (&v)[NIT]) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      raw_t t4 = v[u];
      if (padmask & (1u << u)) t4 = raw_zero(t4);      // zero padding by select
      uint2 hi, lo;
      split_raw(t4, hi, lo);
      if (!LAST_PARTIAL || u + 1 < NIT || last_valid) {
        // LDS address recomputed (QPP = 8: a shift, a multiply-add, a mask) instead of kept: 11 registers of a kernel that sits at its limit
        const int idx = min(tid + u * NTHREADS, TOTAL - 1);
        char* d = ldsb + (idx >> 3) * PSB + (idx & 7) * 8;
        *reinterpret_cast<uint2*>(d) = hi;
        if constexpr (!paif::st_lo0(ST)) *reinterpret_cast<uint2*>(d + 64) = lo;
      }
    }
  };

  // B ring: D taps (of the flattened source x tap sequence) resident per lane
  constexpr int D = MsCfg<NSRC, ST>::D;
  constexpr int NG = NSRC * NTAP;
  uint4 bw[D][NKS * 2];
  // uniform (SGPR) base per load + one 32-bit lane offset: with a per-lane 64-bit pointer the loop-invariant address
  // of every one of the NG*4 loads is hoisted out of the tile loop into VGPR pairs (hundreds of bytes of spills)
  const unsigned lane16 = (unsigned)lane * 16u;
  auto fetchB = [&](int g) {
#pragma unroll
    for (int i = 0; i < NKS * 2; ++i)
      if (!(paif::st_wl0(ST) && (i & 1)))
        bw[g % D][i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.wpk) + (size_t)(g * NKS * 2 + i) * 1024 + lane16);
  };
  // A operand of one K=16 step: hi and lo halves of this wave's two row segments.  The reads of step i+1 are issued
  // before the MFMAs of step i and pinned there (sched_barrier): left to itself the scheduler sinks every ds_read to
  // just above its first use and each MFMA group then eats the LDS latency.
  struct AStep { bf16x8 h[SEGS_PER_WAVE], l[SEGS_PER_WAVE]; };
  constexpr int NSTEP = NTAP * NKS;
  auto readA = [&](AStep& A, int step) {
    const int tap = step / NKS, ks = step - tap * NKS;
    const int dy = tap / KH, dx = tap - dy * KH;
    const int toff = (dy * DIL * TWH + dx * DIL) * PSB + 32 * ks;
#pragma unroll
    for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) {
      A.h[sg] = *reinterpret_cast<const bf16x8*>(ldsb + abase[sg] + toff);
      if constexpr (!paif::st_lo0(ST)) A.l[sg] = *reinterpret_cast<const bf16x8*>(ldsb + abase[sg] + toff + 64);
    }
  };
  auto mma_step = [&](const AStep& A, int g, int ks) {
    const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[g % D][2 * ks]);
    const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[g % D][2 * ks + 1]);
    if constexpr (!paif::st_lo0(ST)) {
#pragma unroll
      for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l[sg], bh, acc[sg], 0, 0, 0);
    }
    if constexpr (!paif::st_wl0(ST)) {
#pragma unroll
    for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[sg], bl, acc[sg], 0, 0, 0);
    }
#pragma unroll
    for (int sg = 0; sg < SEGS_PER_WAVE; ++sg) acc[sg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[sg], bh, acc[sg], 0, 0, 0);
  };
  // one source: 18 steps, A double-buffered in registers, B slot of a tap refilled (D taps ahead) after its last step
  auto mma_source = [&](int s) {
    AStep A[2];
    readA(A[0], 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int tap = step / NKS, ks = step - tap * NKS;
      const int g = s * NTAP + tap;
      if (step + 1 < NSTEP) readA(A[(step + 1) & 1], step + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(A[step & 1], g, ks);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == NKS - 1 && g + D < NG) {
        fetchB(g + D);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: the first D taps of B, then A(0) ----
  PAIF_MS_STAMP(0);
#pragma unroll
  for (int g = 0; g < D; ++g) fetchB(g);
  {
    raw_t v0[NIT];
    issueA(a.src[0], v0);
    convertA(v0);
  }
  PAIF_MS_STAMP(1);
  lds_barrier();
  PAIF_MS_STAMP(2);
#pragma unroll
  for (int s = 0; s < NSRC; ++s) {
    raw_t vn[NIT];
    if (s + 1 < NSRC) issueA(a.src[s + 1], vn);          // the next source's halo tile: in flight during this source's MFMA loop
    __builtin_amdgcn_sched_barrier(0);
    mma_source(s);
    PAIF_MS_STAMP(3 + 4 * s);
    if (s + 1 < NSRC) {
      lds_barrier();                                     // every wave has finished reading tile s
      PAIF_MS_STAMP(4 + 4 * s);
      convertA(vn);
      PAIF_MS_STAMP(5 + 4 * s);
      lds_barrier();
      PAIF_MS_STAMP(6 + 4 * s);
    }
  }

  const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W) && (a.cout == 32);  // block-uniform
  const EpiParams ep_par = load_epi_params<false>(a, lane);
  __syncthreads();  // every wave has finished reading the staged tile: its LDS is reused by the epilogue
  float4 ps;
  if (full) ps = epilogue_lds<true, false, SEGS_PER_WAVE, SEGS_PER_WAVE * 32 * 8 / 64, paif::st_out(ST)>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  else ps = epilogue_lds<false, false, SEGS_PER_WAVE, SEGS_PER_WAVE * 32 * 8 / 64, paif::st_out(ST)>(a, ep_par, acc, lds, b, y0, x0, wave, lane);
  (void)ps;
  PAIF_MS_STAMP(15);
}

// ---------------------------------------------------------------------------------------------------
// Persistent 3x3 kernel with the B operand of a whole source resident in registers.
//
// What paces the tile-per-workgroup kernels is not a roof but latency they cannot cover: nothing is in flight during
// their prologue wait, last source and epilogue, and a B fetch issued inside the MFMA loop returns in order behind
// whatever halo-tile prefetch the same wave has outstanding (vmcnt is in-order), so a register prefetch of the next
// halo tile stalls the MFMA loop a few taps later.  Here:
//   * one 512-thread workgroup per CU walks its tiles; wave w owns output row w of the 8x32 tile (one 32x32 accumulator),
//     which leaves room for the B operand of one source -- 9 taps x 4 x 16 B per lane = 144 VGPRs -- to stay resident.
//     With one source it is loaded once per launch; with 2-3 sources tap t's registers are refilled with the next
//     source's tap t right after its last use, i.e. a whole source (>= 3.4k MFMA cycles) before they are needed, so it
//     does not matter that the refill returns behind the halo prefetch: the next source needs that tile anyway.
//   * the (tile, source) sequence is one flattened software pipeline: while the MFMAs of a unit run from one LDS
//     buffer, the halo tile of the next unit (next source, or source 0 of the next tile) is in flight into registers;
//     it is split into bf16 hi/lo and written to the OTHER buffer after the MFMAs -- one barrier per unit, and every
//     workgroup has a 44.5 KB halo tile outstanding all the time.
//   * all loads are buffer loads (resource in SGPRs + 32-bit lane offset): with per-lane 64-bit pointers the tile loop's
//     invariant addresses were hoisted into VGPR pairs and spilled.
// Tiles are dealt so that the workgroups of one XCD walk one contiguous eighth of the tile list side by side (halo
// rows and columns shared between neighbouring tiles hit that XCD's L2).
// ---------------------------------------------------------------------------------------------------
#ifndef PAIF_RES_ROWS
#define PAIF_RES_ROWS 4
#endif

template <int KH, int DIL, int NSRC, int RT, int ST = 0>
__global__ __launch_bounds__(RT * 64, RT == 4 ? 2 : 1) void conv_bf16x3_res(ConvArgs a, int ntiles, int tilesY) {
  constexpr int BFI = paif::st_in(ST);
  constexpr unsigned ES = BFI ? 2u : 4u;               // bytes per stored input element
  typedef typename std::conditional<BFI != 0, unsigned __attribute__((ext_vector_type(2))), u32x4>::type rawv_t;
  constexpr int RES_THREADS = RT * 64;   // one wave per tile row
  constexpr int TH = RT;                 // shadows the file-wide tile height
  constexpr int CIN = 32;
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr int TWH = TW + 2 * P;
  constexpr int THH = TH + 2 * P;
  constexpr int PSB = 144;             // pixel record in bytes
  constexpr int QPP = CIN / 4;
  constexpr int NKS = CIN / 16;        // K=16 steps per tap
  constexpr int NTAP = KH * KH;
  constexpr int TOTAL = THH * TWH * QPP;
  constexpr int NIT = (TOTAL + RES_THREADS - 1) / RES_THREADS;
  constexpr int TILE_BYTES = THH * TWH * PSB;
  extern __shared__ __align__(16) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  float* epi = reinterpret_cast<float*>(ldsb + 2 * TILE_BYTES);   // RT waves x [32 px][32 ch] fp32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int hh = lane >> 5;
  const int p = lane & 31;

  // this workgroup's tiles: XCD x (hardware deals workgroups round-robin over the 8 XCDs) owns tiles [x*tpx, (x+1)*tpx)
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int t_beg = xcd * tpx, t_end = min(ntiles, (xcd + 1) * tpx);
  if (t_beg + wg >= t_end) return;     // block-uniform
  // logical position p = wg, wg + nwg, ... of this XCD's range; tile = t_beg + p, or counted from the end when reversed
  int pos = wg;
  int tile = a.reverse ? t_end - 1 - pos : t_beg + pos;

  f32x16 acc[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  const int abase = (wave * TWH + p) * PSB + 16 * hh;

  float in_slope = 0.f;                // in-activation as PReLU: ReLU = slope 0
  if (a.in_act == 1) in_slope = *a.in_prelu;
  constexpr unsigned RSRC_W3 = 0x00020000u;
  __amdgpu_buffer_rsrc_t rsrc[NSRC];
#pragma unroll
  for (int s = 0; s < NSRC; ++s)
    rsrc[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src[s]), 0, a.B * a.H * a.W * (int)(CIN * ES), RSRC_W3);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wpk), 0, NSRC * NTAP * NKS * 2 * 1024, RSRC_W3);

  // staging slot u of this thread: element tid + u * RES_THREADS of the halo tile (pixel-major, 8 channel quads per pixel).
  // Slot geometry is recomputed from an opaque copy of tid where it is used (a handful of VALU instructions per tile):
  // as tile-loop invariants the 3 x NIT values would be hoisted and held in registers through every MFMA phase.
  constexpr bool LAST_PARTIAL = (TOTAL % RES_THREADS) != 0;
  auto slot_pix = [&](int t, int u) { return min(t + u * RES_THREADS, TOTAL - 1) / QPP; };

  unsigned goff[NIT];                  // 32-bit byte offsets into a source, clamped into the image
  unsigned padmask = 0;                // slots that are zero padding
  auto locate = [&](int t, int& b, int& y0, int& x0) {
    const int tx = t % a.tilesX;
    t /= a.tilesX;
    const int ty = t % tilesY;
    b = t / tilesY; y0 = ty * TH; x0 = tx * TW;
  };
  auto setup = [&](int b, int y0, int x0) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const unsigned qoff = (unsigned)(t & (QPP - 1)) * (4u * ES);   // RES_THREADS % QPP == 0: the channel quad is the same in every slot
    padmask = 0;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int pix = slot_pix(t, u);
      const int tyy = pix / TWH, txx = pix - tyy * TWH;
      const int gy = y0 - P + tyy, gx = x0 - P + txx;
      const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
      goff[u] = (unsigned)((b * a.H + gyc) * a.W + gxc) * (CIN * ES) + qoff;
      if (gy != gyc || gx != gxc) padmask |= 1u << u;
    }
    // materialise the mask here: otherwise the compiler defers it into the conversion as compares of gy/gyc/gx/gxc and
    // keeps those 4 x NIT values alive through the MFMA phase
    asm volatile("" : "+v"(padmask));
  };
  auto issueA = [&](int s, rawv_t (&v)[NIT]) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {   // unconditional, clamped
      if constexpr (BFI) v[u] = __builtin_amdgcn_raw_buffer_load_b64(rsrc[s], goff[u], 0, 0);
      else v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc[s], goff[u], 0, 0);
    }
  };
  auto convert_n = [&](rawv_t (&v)[NIT], char* buf, auto with_act) {
    int t = tid;
    asm volatile("" : "+v"(t));
    const bool last_valid = t + (NIT - 1) * RES_THREADS < TOTAL;
    const int q8 = (t & (QPP - 1)) * 8;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int dstu = slot_pix(t, u) * PSB + q8;
      float4 t4;
      if constexpr (BFI) t4 = paif::bf16x4_to_f32(make_uint2(v[u][0], v[u][1]));
      else t4 = make_float4(__uint_as_float(v[u][0]), __uint_as_float(v[u][1]), __uint_as_float(v[u][2]), __uint_as_float(v[u][3]));
      if constexpr (decltype(with_act)::value) {
        t4.x = paif::prelu_f(t4.x, in_slope); t4.y = paif::prelu_f(t4.y, in_slope);
        t4.z = paif::prelu_f(t4.z, in_slope); t4.w = paif::prelu_f(t4.w, in_slope);
      }
      if (padmask & (1u << u)) t4 = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding by select
      uint2 hi, lo;
      split_bf16x4(t4, hi, lo);
      if (!LAST_PARTIAL || u + 1 < NIT || last_valid) {
        *reinterpret_cast<uint2*>(buf + dstu) = hi;
        if constexpr (!paif::st_lo0(ST)) *reinterpret_cast<uint2*>(buf + dstu + 64) = lo;
      }
    }
  };
  auto convertA = [&](rawv_t (&v)[NIT], char* buf) {
    if (a.in_act) convert_n(v, buf, std::true_type{});    // launch-uniform: one branch per tile, not one per element
    else convert_n(v, buf, std::false_type{});
  };

  // the B operand of one source; with plain bf16 weights (half the registers: 72 per source) of EVERY source -- loaded once per
  // launch, no refills behind the halo prefetch
  constexpr int NB = paif::st_wl0(ST) ? NSRC : 1;
  u32x4 bw[NB][NTAP][NKS * 2];
  const unsigned lane16 = (unsigned)lane * 16u;
  auto fetchB = [&](int s, int tap) {
#pragma unroll
    for (int i = 0; i < NKS * 2; ++i)
      if (!(paif::st_wl0(ST) && (i & 1)))
        bw[NB > 1 ? s : 0][tap][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, ((s * NTAP + tap) * NKS * 2 + i) * 1024, 0);
  };
  struct AStep { bf16x8 h, l; };
  constexpr int NSTEP = NTAP * NKS;
  auto readA = [&](AStep& A, const char* buf, int step) {
    const int tap = step / NKS, ks = step - tap * NKS;
    const int dy = tap / KH, dx = tap - dy * KH;
    const int toff = (dy * DIL * TWH + dx * DIL) * PSB + 32 * ks;
    A.h = *reinterpret_cast<const bf16x8*>(buf + abase + toff);
    if constexpr (!paif::st_lo0(ST)) A.l = *reinterpret_cast<const bf16x8*>(buf + abase + toff + 64);
  };
  // one source from `buf`: the A operand runs two K steps (>= 192 MFMA cycles) ahead in a 3-slot register ring, issue
  // order pinned (see conv_mfma_bf16x3): one 3-MFMA dependent chain per wave leaves only that much time per step
  auto mma_source = [&](int s, const char* buf) {
    AStep A[3];
    readA(A[0], buf, 0);
    readA(A[1], buf, 1);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int tap = step / NKS, ks = step - tap * NKS;
      if (step + 2 < NSTEP) readA(A[(step + 2) % 3], buf, step + 2);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[NB > 1 ? s : 0][tap][2 * ks]);
      if constexpr (!paif::st_lo0(ST)) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[step % 3].l, bh, acc[0], 0, 0, 0);
      if constexpr (!paif::st_wl0(ST)) {
        const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[0][tap][2 * ks + 1]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[step % 3].h, bl, acc[0], 0, 0, 0);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[step % 3].h, bh, acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (NSRC > 1 && NB == 1 && ks == NKS - 1) fetchB((s + 1) % NSRC, tap);    // this tap's registers: next source, same tap
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  int b, y0, x0;
  locate(tile, b, y0, x0);
  setup(b, y0, x0);
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) fetchB(s, tap);
  {
    rawv_t v0[NIT];
    issueA(0, v0);
    convertA(v0, ldsb);
  }
  // the epilogue constants of a lane (scale / shift quad, slope) wait in LDS, not in 9 registers
  float* epp = epi + RT * 1024 + RT * 32;
  if (tid < 64) {
    const EpiParams e0 = load_epi_params<false>(a, lane);
    *reinterpret_cast<float4*>(epp + lane * 4) = e0.sc;
    *reinterpret_cast<float4*>(epp + 256 + lane * 4) = e0.sh;
    if (lane == 0) epp[512] = e0.slope;
  }
  lds_barrier();
  PAIF_TRACE_DECL
  int par = 0;                         // LDS buffer of the current unit
  for (;;) {
    const bool more = t_beg + pos + nwg < t_end;
    const int tile_next = a.reverse ? tile - nwg : tile + nwg;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
      rawv_t vn[NIT];
      int bn = b, yn = y0, xn = x0;
      if (s == NSRC - 1) {             // the next unit is source 0 of the next tile (the last tile re-fetches itself: unused)
        locate(more ? tile_next : tile, bn, yn, xn);
        setup(bn, yn, xn);
      }
      issueA((s + 1) % NSRC, vn);
      __builtin_amdgcn_sched_barrier(0);
      PAIF_TRACE(0);
      mma_source(s, ldsb + par * TILE_BYTES);
      PAIF_TRACE(1);
      if (s == NSRC - 1) {
        const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W);   // block-uniform
        // opaque copies: handed the plain ids, the epilogue's loop-invariant per-lane address arithmetic is hoisted
        // in front of the tile loop and lives (spilled) through every MFMA phase
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e), "+v"(wave_e));
        EpiParams ep_par;
        ep_par.sc = *reinterpret_cast<const float4*>(epp + lane_e * 4);
        ep_par.sh = *reinterpret_cast<const float4*>(epp + 256 + lane_e * 4);
        ep_par.slope = epp[512];
        float4 ps;                     // private per-wave LDS region: no barrier between the MFMAs and the epilogue
        if (full) ps = epilogue_lds<true, false, 1, 2, paif::st_out(ST)>(a, ep_par, acc, epi, b, y0, x0, wave_e, lane_e);
        else ps = epilogue_lds<false, false, 1, 2, paif::st_out(ST)>(a, ep_par, acc, epi, b, y0, x0, wave_e, lane_e);
        if (a.pool_partial) {          // launch-uniform; fixed-order reduction -> deterministic
#pragma unroll
          for (int m = 8; m < 64; m <<= 1) {
            ps.x += __shfl_xor(ps.x, m); ps.y += __shfl_xor(ps.y, m); ps.z += __shfl_xor(ps.z, m); ps.w += __shfl_xor(ps.w, m);
          }
          if (lane < 8) *reinterpret_cast<float4*>(epi + RT * 1024 + wave * 32 + lane * 4) = ps;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        PAIF_TRACE(2);
      }
      if (!(s == NSRC - 1 && !more)) convertA(vn, ldsb + (par ^ 1) * TILE_BYTES);   // block-uniform
      PAIF_TRACE(3);
      lds_barrier();
      PAIF_TRACE(4);
      if (s == NSRC - 1) {
        if (a.pool_partial && tid < 32) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < RT; ++w) t += epi[RT * 1024 + w * 32 + tid];
          a.pool_partial[(size_t)tile * 32 + tid] = t;
        }
        b = bn; y0 = yn; x0 = xn;
      }
      par ^= 1;
    }
    if (!more) {
      PAIF_TRACE_END
      return;
    }
    tile = tile_next;
    pos += nwg;
  }
}

template <int KH, int DIL, int NSRC, int ST = 0>
int launch_bf16x3_res(const ConvArgs& a, hipStream_t st) {
  constexpr int RT = PAIF_RES_ROWS;
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr size_t lds_bytes = 2 * (size_t)(RT + 2 * P) * (TW + 2 * P) * 144 + RT * 32 * 32 * 4 + RT * 32 * 4 + 520 * 4;
  static_assert(lds_bytes * (RT == 4 ? 2 : 1) <= 160 * 1024, "two halo-tile buffers + the epilogue regions do not fit LDS");
  static bool raised = false;   // once per instantiation (one device per process)
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_res<KH, DIL, NSRC, RT, ST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      paif::set_error("conv2d(bf16x3 resident): cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    raised = true;
  }
  const int tilesY = (a.H + RT - 1) / RT;
  hipLaunchKernelGGL((conv_bf16x3_res<KH, DIL, NSRC, RT, ST>), dim3(RT == 4 ? 512 : 256), dim3(RT * 64), lds_bytes, st, a,
                     a.B * tilesY * a.tilesX, tilesY);
  PAIF_LAUNCH_CHECK("conv2d(bf16x3 resident)");
  return 0;
}

template <int KH, int DIL, int NSRC, int ST = 0>
int launch_bf16x3_ms(const ConvArgs& a, hipStream_t st) {
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr size_t tile_bytes = (size_t)(TH + 2 * P) * (TW + 2 * P) * 144;
  constexpr size_t epi_bytes = (size_t)TH * 32 * 32 * 4;
  constexpr size_t lds_bytes = tile_bytes > epi_bytes ? tile_bytes : epi_bytes;
  static_assert(lds_bytes <= 64 * 1024, "ms form: expected a tile under 64 KiB");
  hipLaunchKernelGGL((conv_bf16x3_ms<KH, DIL, NSRC, ST>), dim3(a.nblk), dim3(NTHREADS), lds_bytes, st, a);
  PAIF_LAUNCH_CHECK("conv2d(bf16x3 ms)");
  return 0;
}

static inline bool ms_enabled() {
  static const bool on = [] {
    const char* e = getenv("PAIF_CONV_MS");   // PAIF_CONV_MS=0 keeps multi-source convs on the plain kernel (A/B runs)
    return !(e && e[0] == '0');
  }();
  return on;
}


constexpr int WS_THREADS = 12 * 64;

template <int KH, int DIL, int ST, bool RELU>
__device__ __forceinline__ void conv_ws_body(const ConvArgs& a, int ntiles) {
  constexpr int BFI = paif::st_in(ST), BFO = paif::st_out(ST), BFR = paif::st_res(ST), FM = paif::st_fmt16(ST);
  static_assert(!paif::st_f16(ST) || BFI != 0, "fp16 storage: the persistent form takes fp16 sources, staged as stored");
  constexpr unsigned ES = BFI ? 2u : 4u;               // bytes per stored input element
  typedef typename RawQ<BFI>::T raw_t;
  static_assert(TH == 8, "the MFMA-wave mapping assumes 8-row tiles");
  constexpr int CIN = 32;
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr int TWH = TW + 2 * P;
  constexpr int THH = TH + 2 * P;
  constexpr int PSB = 144;
  constexpr int QPP = CIN / 4;
  constexpr int NKS = CIN / 16;
  constexpr int NTAP = KH * KH;
  constexpr int NB = NTAP * NKS * 2;   // uint4 of B per lane per source
  constexpr int TILE_BYTES = THH * TWH * PSB;
  extern __shared__ __align__(16) float lds[];
  char* ldsb = reinterpret_cast<char*>(lds);
  float* outbuf = reinterpret_cast<float*>(ldsb + 2 * TILE_BYTES);   // [8 rows][32 px][32 ch] fp32

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar role branch

  // tile schedule: XCD x owns tiles [x*per, (x+1)*per); its workgroups interleave over that range
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per = (ntiles + 7) >> 3;
  const int tbeg = xcd * per + slot, tend = min(ntiles, (xcd + 1) * per);
  const int cnt = tbeg < tend ? (tend - tbeg + nslot - 1) / nslot : 0;
  const int nsrc = a.nsrc;
  const int S = cnt * nsrc;            // stages; steps g = 0 .. S+1, two barriers each
  auto tile_of = [&](int i, int& b, int& y0, int& x0) {
    int t = tbeg + i * nslot;
    x0 = (t % a.tilesX) * TW;
    t /= a.tilesX;
    y0 = (t % a.tilesY) * TH;
    b = t / a.tilesY;
  };

  if (wave >= 8) {
    // ------------------------------- loaders -------------------------------
    const int pt = tid - 8 * 64;
    const unsigned q16 = (unsigned)(pt & 7) * (4u * ES);
    constexpr int TOTAL = THH * TWH * QPP;
    constexpr int NIT = (TOTAL + 255) / 256;
    // per-lane tile coordinates of its NIT elements: fixed for the whole launch
    int tyy[NIT], txx[NIT];
    unsigned ldo[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int pix = min(pt + u * 256, TOTAL - 1) >> 3;
      tyy[u] = pix / TWH;
      txx[u] = pix - tyy[u] * TWH;
      ldo[u] = (unsigned)pix * PSB + (unsigned)(pt & 7) * 8u;
    }
    const bool live_last = pt + (NIT - 1) * 256 < TOTAL;
    // Every load is unconditional on a clamped stage / address (a load under a branch makes hipcc wait at the
    // join); the in-image mask travels with the register set and zero padding is applied at the LDS write.
    auto issue = [&](int st, raw_t (&v)[NIT], unsigned& mask) {
      st = min(st, S - 1);
      const int i = st / nsrc, s = st - i * nsrc;
      int b, y0, x0;
      tile_of(i, b, y0, x0);
      const char* base = reinterpret_cast<const char*>(a.src[s]) + (size_t)b * a.H * a.W * (CIN * ES);
      const int y0p = y0 - P, x0p = x0 - P;
      unsigned m = 0;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int gy = y0p + tyy[u], gx = x0p + txx[u];
        const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
        const unsigned off = (unsigned)(gyc * a.W + gxc) * (CIN * ES) + q16;   // < 4 GiB per image: checked at launch
        v[u] = KH == 1 ? ldraw_nt<BFI>(base + off) : ldraw<BFI>(base + off);
        m |= (gy == gyc && gx == gxc) ? (1u << u) : 0u;
      }
      mask = m;
    };
    auto commit = [&](int st, const raw_t (&v)[NIT], unsigned mask) {
      char* buf = ldsb + (st & 1) * TILE_BYTES;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        uint2 hi, lo;
        if constexpr (RELU) split_raw(raw_relu(v[u]), hi, lo);   // in_act = ReLU: the composed DilConv (conv_bf16x3_wsr)
        else split_raw(v[u], hi, lo);
        if (!((mask >> u) & 1u)) hi = lo = make_uint2(0u, 0u);
        if (u < NIT - 1 || live_last) {
          *reinterpret_cast<uint2*>(buf + ldo[u]) = hi;
          if constexpr (!paif::st_lo0(ST)) *reinterpret_cast<uint2*>(buf + ldo[u] + 64) = lo;
        }
      }
    };
    if (S > 0) {
      // straight-line pair body + peeled odd tail: with a conditional second half the wait-count pass must assume
      // set B is still pending at the loop header and drains vmcnt(0) there, which kills the prefetch
      raw_t va[NIT], vb[NIT];
      unsigned ma, mb;
      issue(0, va, ma);
      const int npair = S >> 1;
      for (int k = 0; k < npair; ++k) {
        const int st = 2 * k;
        issue(st + 1, vb, mb);
        commit(st, va, ma);
        lds_barrier();
        lds_barrier();
        issue(st + 2, va, ma);
        commit(st + 1, vb, mb);
        lds_barrier();
        lds_barrier();
      }
      if (S & 1) {
        commit(S - 1, va, ma);
        lds_barrier();
        lds_barrier();
      }
    }
    lds_barrier();   // steps S and S+1
    lds_barrier();
    lds_barrier();
    lds_barrier();
  } else if (wave < 4) {
    // ------------------------------- MFMA waves -------------------------------
    const int hh = lane >> 5, p = lane & 31;
    int abase[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) abase[sg] = ((wave * 2 + sg) * TWH + p) * PSB + 16 * hh;
    // park address of acc[sg][r]: pixel (row = wave*2+sg, px = (r&3) + 8*(r>>2) + 4*hh), channel p
    const int pbase = (wave * 2 * 32 + 4 * hh) * 32 + p;
    f32x16 acc[2];
    // B-operand ring: 3 taps in registers (4 x 16 B per lane per tap), refilled from L2 three taps ahead and running
    // straight into the next stage's taps (NTAP % 3 == 0 keeps the slots static).  The L2 hits queue behind the
    // loaders' HBM stream in the CU's memory pipeline (~3.4k cycles per B load under load, cycle-stamped), which is
    // what still paces this loop; a 6-tap ring (two statically indexed stage phases) spills at 168 registers.
    constexpr int BR = NTAP % 3 == 0 ? 3 : 1;
    static_assert(NTAP % BR == 0, "ring slots must stay static across stages");
    uint4 bw[BR][NKS * 2];
    const uint4* wbase = reinterpret_cast<const uint4*>(a.wpk) + lane;
#pragma unroll
    for (int t = 0; t < BR; ++t)
#pragma unroll
      for (int j = 0; j < NKS * 2; ++j)
        if (!(paif::st_wl0(ST) && (j & 1))) bw[t][j] = wbase[(t * NKS * 2 + j) * 64];
    int s = 0;
    for (int g = 0; g <= S + 1; ++g) {
      const bool work = g >= 1 && g <= S;
      const bool last = s == nsrc - 1;
      if (work) {
        const char* buf = ldsb + ((g - 1) & 1) * TILE_BYTES;
        const int ns = last ? 0 : s + 1;
        const uint4* wcur = wbase + (size_t)s * NB * 64;
        const uint4* wnxt = wbase + (size_t)ns * NB * 64;   // always a valid source: the refill needs no branch
        if (s == 0) {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sg][r] = 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          const int dy = tap / KH, dx = tap - dy * KH;
          const int toff = (dy * DIL * TWH + dx * DIL) * PSB;
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bw[tap % BR][ks * 2]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, bw[tap % BR][ks * 2 + 1]);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(buf + abase[sg] + toff + 32 * ks);
              if constexpr (!paif::st_lo0(ST)) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(buf + abase[sg] + toff + 64 + 32 * ks);
                acc[sg] = paif::mfma16<FM>(al, bh, acc[sg]);
              }
              if constexpr (!paif::st_wl0(ST)) {
              acc[sg] = paif::mfma16<FM>(ah, bl, acc[sg]);
              }
              acc[sg] = paif::mfma16<FM>(ah, bh, acc[sg]);
            }
          }
          {
            const int nt = tap + BR;
            const uint4* wsrc = nt < NTAP ? wcur + (size_t)nt * NKS * 2 * 64 : wnxt + (size_t)(nt - NTAP) * NKS * 2 * 64;
#pragma unroll
            for (int j = 0; j < NKS * 2; ++j)
              if (!(paif::st_wl0(ST) && (j & 1))) bw[tap % BR][j] = wsrc[j * 64];
          }
        }
      }
      lds_barrier();
      if (work) {
        if (last) {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
#pragma unroll
            for (int r = 0; r < 16; ++r) outbuf[pbase + (sg * 32 + (r & 3) + 8 * (r >> 2)) * 32] = acc[sg][r];
        }
        s = last ? 0 : s + 1;
      }
      lds_barrier();
    }
  } else {
    // ------------------------------- storers -------------------------------
    const int t = tid - 4 * 64;
    const int q = t & 7, px = t >> 3;
    const float4 sc = *reinterpret_cast<const float4*>((a.scale ? a.scale : k_ones) + 4 * q);
    const float4 sh = *reinterpret_cast<const float4*>((a.shift ? a.shift : k_zeros) + 4 * q);
    const float slope = *(a.act == 1 ? a.prelu : k_zeros);
    const int nres = a.res[0] ? (a.res[1] ? (a.res[2] ? 3 : 2) : 1) : 0;
    const unsigned lane_off = (unsigned)px * 32u + 4u * (unsigned)q;
    // The residual maps of tile e+1 are requested right after tile e has been stored, so they are in flight for a
    // whole tile period (a storer mixes loads and stores, its waits are vmcnt(0) anyway -- but everything it waits
    // for is a step old).  Loads are unconditional on clamped coordinates; the store is predicated.
    auto run = [&](auto nres_c) {
      constexpr int NR = decltype(nres_c)::value;
      float4 r[NR > 0 ? NR : 1][8];
      auto request = [&](int i) {
        int b, y0, x0;
        tile_of(min(i, cnt - 1), b, y0, x0);
        const size_t base = ((size_t)(b * a.H + y0) * a.W + x0) * 32;
        const unsigned lo = x0 + px < a.W ? lane_off : 0u;
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int yy = min(y0 + j, a.H - 1) - y0;
            r[k][j] = paif::ldq_nt<BFR>(a.res[k], base + (size_t)yy * a.W * 32 + lo);
          }
      };
      if (cnt > 0) request(0);
      // a tile parked in phase B of step g-1 (stage g-2 was the last source of its tile) is stored in phase A of step g
      int s2 = 0, i2 = 0;   // source index / tile index of stage g-2
      for (int g = 0; g <= S + 1; ++g) {
        if (g >= 2) {
          if (s2 == nsrc - 1) {
            int b, y0, x0;
            tile_of(i2, b, y0, x0);
            float4 o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = *reinterpret_cast<const float4*>(outbuf + (j * 32 + px) * 32 + 4 * q);
            const size_t base = ((size_t)(b * a.H + y0) * a.W + x0) * 32;
            const bool colok = x0 + px < a.W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float4 v = o[j];
              v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
              if (a.act == 1) {
                v.x = paif::prelu_f(v.x, slope); v.y = paif::prelu_f(v.y, slope);
                v.z = paif::prelu_f(v.z, slope); v.w = paif::prelu_f(v.w, slope);
              } else if (a.act == 2) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              }
              v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
#pragma unroll
              for (int k = 0; k < NR; ++k) { v.x += r[k][j].x; v.y += r[k][j].y; v.z += r[k][j].z; v.w += r[k][j].w; }
              if (colok && y0 + j < a.H)
                paif::stq_nt<BFO>(a.out, base + (size_t)j * a.W * 32 + lane_off, v);
              if (a.cpool) {   // launch-uniform: the fused ChannelPool of the output (8 lanes per pixel, channel_pool2_kernel's tree)
                float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), sm = (v.x + v.y) + (v.z + v.w);
                paif::pix8_max_sum(mx, sm);
                if (colok && y0 + j < a.H && q == 0)
                  *reinterpret_cast<float2*>(a.cpool + ((base + (size_t)j * a.W * 32) >> 5) * 4 + (size_t)px * 4) = make_float2(mx, sm * (1.0f / 32.0f));
              }
            }
            if (NR > 0) request(i2 + 1);
          }
          if (++s2 == nsrc) { s2 = 0; ++i2; }
        }
        lds_barrier();
        lds_barrier();
      }
    };
    if (nres == 0) run(std::integral_constant<int, 0>{});
    else if (nres == 1) run(std::integral_constant<int, 1>{});
    else if (nres == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
  }
}

template <int KH, int DIL, int ST = 0>
__global__ __launch_bounds__(WS_THREADS, 1) void conv_bf16x3_ws(ConvArgs a, int ntiles) { conv_ws_body<KH, DIL, ST, false>(a, ntiles); }
// the same kernel with ReLU on the loaded tile (in_act = 2): DilConv of the bf16 forward as one dense dilated conv
template <int KH, int DIL, int ST = 0>
__global__ __launch_bounds__(WS_THREADS, 1) void conv_bf16x3_wsr(ConvArgs a, int ntiles) { conv_ws_body<KH, DIL, ST, true>(a, ntiles); }

template <int KH, int DIL, int ST = 0>
int launch_bf16x3_ws(const ConvArgs& a, hipStream_t st) {
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr size_t lds_bytes = 2 * (size_t)(TH + 2 * P) * (TW + 2 * P) * 144 + 8 * 32 * 32 * 4;
  static_assert(lds_bytes <= 160 * 1024, "two tile buffers + the parked tile do not fit LDS");
  static bool raised = false;   // once per instantiation (one device per process)
  constexpr bool HAS_RELU = KH == 3 && DIL == 2 && paif::st_in(ST) != 0 && paif::st_low(ST) != 7;   // the ReLU form is built for the 16-bit-stored dilated 3x3
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_ws<KH, DIL, ST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if constexpr (HAS_RELU) {
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_wsr<KH, DIL, ST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    }
    if (e != hipSuccess) {
      paif::set_error("conv2d(bf16x3 ws): cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    raised = true;
  }
  if (a.in_act == 2) {
    if constexpr (HAS_RELU) {
      hipLaunchKernelGGL((conv_bf16x3_wsr<KH, DIL, ST>), dim3(256), dim3(WS_THREADS), lds_bytes, st, a, a.nblk);
      PAIF_LAUNCH_CHECK("conv2d(bf16x3 wsr)");
      return 0;
    }
    paif::set_error("conv2d(bf16x3 ws): input ReLU is built for the bf16-stored dilation-2 3x3 only");
    return PAIF_ENOSUP;
  }
  hipLaunchKernelGGL((conv_bf16x3_ws<KH, DIL, ST>), dim3(256), dim3(WS_THREADS), lds_bytes, st, a, a.nblk);
  PAIF_LAUNCH_CHECK("conv2d(bf16x3 ws)");
  return 0;
}

// the loaders address a source image with 32-bit byte offsets; in-activations and the ECA pool stay on the plain kernel
static inline bool ws_enabled() {
  static const bool on = [] {
    const char* e = getenv("PAIF_CONV_WS");   // PAIF_CONV_WS=0 selects the plain kernel everywhere (A/B runs)
    return !(e && e[0] == '0');
  }();
  return on;
}
// 16-bit maps in AND out (bf16: 1, fp16: 3) / fp16 maps anywhere
static inline bool st_h16(const ConvArgs& a) { return a.st == 1 || a.st == 3; }
static inline bool st_is_f16(const ConvArgs& a) { return a.st >= 3; }
static inline bool ws_eligible(const ConvArgs& a) {
  return ws_enabled() && a.nblk >= 1024 && !a.pool_partial && a.cout == 32 && (a.in_act == 0 || (a.in_act == 2 && st_h16(a))) &&
         (size_t)a.H * a.W * 128 < ((size_t)1 << 32);
}

// Resident-B persistent 3x3 (conv_bf16x3_res): one source, enough tiles to amortise the pipeline fill of 512 workgroups.
// Measured against the tile-per-workgroup kernel at the bench shape: 165 vs 188 us.  With 2-3 sources its B refills queue
// behind the halo prefetch of the same wave and stall the MFMA phase (400 / 544 us vs 346 / 444 us for conv_bf16x3_ms),
// so those stay on the multi-source kernel.
static inline bool res_eligible(const ConvArgs& a) {
  static const bool on = [] {
    const char* e = getenv("PAIF_CONV_RES");  // PAIF_CONV_RES=0: tile-per-workgroup kernels everywhere (A/B runs)
    return !(e && e[0] == '0');
  }();
  static const int ms_res = [] {
    const char* e = getenv("PAIF_CONV_RES_NSRC");   // experiment: largest source count the resident form takes with plain bf16 weights
    return e ? atoi(e) : 1;
  }();
  return on && !st_is_f16(a) && (a.nsrc == 1 || (a.wl0 && a.st == 1 && a.in_act == 0 && a.nsrc <= ms_res)) && a.nblk >= 2048 && a.cout == 32 && a.in_act <= 2 && (PAIF_RES_ROWS == 8 || !a.pool_partial) &&
         (size_t)a.B * a.H * a.W * 128 < ((size_t)1 << 31);   // buffer resources: 32-bit byte counts and offsets
}

static inline bool needs_hooks(const ConvArgs& a) { return a.in_act >= 3 || a.aux_out || a.epi_dact; }

// Persistent wave-specialised form (needs several tiles per CU to amortise its pipeline fill).  Measured per
// configuration against the tile-per-workgroup kernel (tools/conv_bench.py, B=8 480x640, same box,
// profiles/r01_conv_ws_study.txt).  Since the plain kernel stages with unconditional loads (all 11 in flight) it is
// the faster one for every 3x3 dilation-1 configuration (e.g. 1 source 191 vs 200 us, 3 sources 479 vs 567 us); the
// persistent form keeps the pure streams: 1x1 without residual maps (119 vs 125, 171 vs 177, 232 vs 237 us -- the
// device's copy rate) and, for bf16-stored maps, the dilation-2 3x3 with one source (222 vs 236 us).
static inline bool takes_ws(const ConvArgs& a, int kh, int dil) {
#if PAIF_TH == 8
  if (needs_hooks(a) || kh > 3 || !ws_eligible(a)) return false;
  if (a.st == 4) return dil == 2 && a.nsrc == 1 && a.in_act == 0;  // fp16 in / fp32 out: the last conv of the fp16 forward
  if (kh == 1) return !a.res[0] && a.in_act == 0;
  // dilation-2 3x3, one source (the composed DilConv; in_act none or ReLU): bf16-stored maps only.  With fp32 maps the tile-per-workgroup
  // kernel is the faster one INSIDE the forward (round 4, three alternating bench.py runs each: 6.154 vs 6.173 ms per step); the 222 vs
  // 236 us that chose the persistent form in round 1 were micro-benchmark times on random inputs, i.e. at the power-capped clock.
  return dil == 2 && a.nsrc == 1 && st_h16(a);
#else
  return false;
#endif
}

// which split-bf16 kernel a launch takes (one place: the dispatcher and paif_conv2d_kernel_name use it)
enum ConvVariant { CV_PLAIN, CV_HOOKS, CV_WS, CV_RES, CV_MS, CV_DMA };
static inline int res_count(const ConvArgs& a) { return a.res[0] ? (a.res[1] ? (a.res[2] ? 3 : 2) : 1) : 0; }
static inline ConvVariant bf16x3_variant(const ConvArgs& a, int kh, int dil) {
  if (needs_hooks(a)) return CV_HOOKS;
  // bf16 maps + plain bf16 weights, 3x3 dilation 1, 32 -> 32 per source: the LDS-DMA kernel (conv_dma.hip)
  if (dil == 1 && st_h16(a) && a.wl0 && a.in_act == 0 && !a.pool_partial &&
      ((kh == 3 && a.cout == 32 && paif_conv_dma::eligible(a.nsrc, res_count(a), a.B, a.H, a.W, a.alpha)) ||
       (kh == 3 && a.cout == 16 && paif_conv_dma::eligible16(a.nsrc, res_count(a), a.B, a.H, a.W, a.alpha)) ||
       (kh == 7 && a.cout == 32 && paif_conv_dma::eligible7(a.nsrc, res_count(a), a.B, a.H, a.W, a.alpha))))
    return CV_DMA;
  // round 6: 3x3 dilation 2 behind an input ReLU (the composed DilConv), 16-bit maps in and out, one source, 1 or 3 residual maps
  if (kh == 3 && dil == 2 && st_h16(a) && a.wl0 && a.in_act == 2 && !a.pool_partial && a.cout == 32 &&
      paif_conv_dma::eligible_d2(a.nsrc, res_count(a), a.B, a.H, a.W, a.alpha))
    return CV_DMA;
  if (takes_ws(a, kh, dil)) return CV_WS;
  if (kh == 3 && dil == 1 && !st_is_f16(a)) {    // (fp16 maps: LDS-DMA, persistent or tile-per-workgroup kernel only)
    if (res_eligible(a)) return CV_RES;
    if (ms_enabled() && a.nsrc >= 2 && a.cout == 32 && a.in_act == 0 && !a.pool_partial &&
        (size_t)a.B * a.H * a.W * 128 < ((size_t)1 << 32))   // 32-bit byte offsets into a source
      return CV_MS;
  }
  return CV_PLAIN;
}

// the fused ChannelPool (ConvArgs::cpool): every tile kernel of this file through the shared LDS epilogue / the persistent form's storers
// (cout = 32, no gradient hooks); the LDS-DMA kernel for the source / residual counts it instantiates
static inline bool variant_can_cpool(const ConvArgs& a, int kh, int dil) {
  if (a.cout != 32 || needs_hooks(a)) return false;
  if (bf16x3_variant(a, kh, dil) == CV_DMA) return paif_conv_dma::can_cpool(a.nsrc, res_count(a), kh, a.cout, dil);
  return true;
}

template <int KH, int DIL, int ST>
int launch_bf16x3_st(const ConvArgs& a, hipStream_t st) {
  switch (bf16x3_variant(a, KH, DIL)) {
    case CV_DMA: {
      paif_conv_dma::Args d{};
      for (int s = 0; s < 3; ++s) { d.src[s] = a.src[s]; d.res[s] = a.res[s]; }
      d.wpk = a.wpk; d.scale = a.scale; d.shift = a.shift; d.prelu = a.prelu; d.out = a.out; d.alpha = a.alpha;
      d.nsrc = a.nsrc; d.nres = res_count(a); d.act = a.act; d.B = a.B; d.H = a.H; d.W = a.W; d.reverse = a.reverse; d.kh = KH; d.cout = a.cout;
      d.f16 = paif::st_f16(ST) ? 1 : 0;
      d.cpool = a.cpool;
      d.dil = DIL; d.in_relu = a.in_act == 2 ? 1 : 0;
      return paif_conv_dma::launch(d, st);
    }
    case CV_HOOKS:
      if constexpr (ST == 0) return launch_bf16x3_h<KH, DIL, true>(a, st);
      paif::set_error("conv2d: the gradient hooks (in_act >= 3, aux_out, epi_dact) are built for fp32 storage only");
      return PAIF_ENOSUP;
    case CV_WS:
#if PAIF_TH == 8
      if constexpr (KH <= 3) return launch_bf16x3_ws<KH, DIL, ST>(a, st);
#endif
      break;
    case CV_RES:
      if constexpr (KH == 3 && DIL == 1 && !paif::st_f16(ST)) {
        if constexpr (ST == 4) {
          if (a.nsrc == 2) return launch_bf16x3_res<3, 1, 2, ST>(a, st);
          if (a.nsrc == 3) return launch_bf16x3_res<3, 1, 3, ST>(a, st);
        }
        return launch_bf16x3_res<3, 1, 1, ST>(a, st);
      }
      break;
    case CV_MS:
      if constexpr (KH == 3 && DIL == 1 && !paif::st_f16(ST)) return a.nsrc == 2 ? launch_bf16x3_ms<3, 1, 2, ST>(a, st) : launch_bf16x3_ms<3, 1, 3, ST>(a, st);
      break;
    default: break;
  }
  return launch_bf16x3_h<KH, DIL, false, ST>(a, st);
}

// bf16 storage is built for the kernel shapes of the inference forward (1x1, 3x3 dil 1 / 2, 7x7); fp32 in / bf16 out only for
// the 1x1 behind the fp32 guided-filter block
// kernel storage code (template argument ST, paif_common.h) of a launch
static inline int kernel_st(const ConvArgs& a) {
  if (a.st == 4) return 15;                                     // fp16 in / fp32 out (plain fp16 weights)
  const int st = a.st == 3 ? 1 : a.st;                          // fp16 in / out -> the bf16 codes' 1
  const int base = st == 1 ? (a.in_act == 1 ? 3 : 1) : st;
  const int code = (a.wl0 && base) ? base + 3 : base;
  return a.st >= 3 ? code + 8 : code;
}

// three-piece split (PAIF_CONV_BF16X6): the tile-per-workgroup kernel, with or without the gradient hooks; fp32 storage
template <int KH, int DIL>
int launch_bf16x6(const ConvArgs& a, hipStream_t st) {
  if (a.st != 0) {
    paif::set_error("conv2d: precision bf16x6 is built for fp32-stored maps");
    return PAIF_ENOSUP;
  }
  return needs_hooks(a) ? launch_bf16x3_h<KH, DIL, true, 0, 3>(a, st) : launch_bf16x3_h<KH, DIL, false, 0, 3>(a, st);
}

// fp16 pairs (PAIF_CONV_F16X3): the tile-per-workgroup kernel, forward form, fp32 storage
template <int KH, int DIL>
int launch_f16x3(const ConvArgs& a, hipStream_t st) {
  if (a.st != 0) {
    paif::set_error("conv2d: precision f16x3 is built for fp32-stored maps");
    return PAIF_ENOSUP;
  }
  // (a saved pre-activation -- aux_out, the taped forward -- and the dgrad staging / epilogue modes take the hook kernel; a caller that
  // sends GRADIENTS through this arithmetic scales them into fp16's exponent range first: ops.attack_grad_scale)
  return needs_hooks(a) ? launch_bf16x3_h<KH, DIL, true, 0, 2, 1>(a, st) : launch_bf16x3_h<KH, DIL, false, 0, 2, 1>(a, st);
}

template <int KH, int DIL>
int launch_bf16x3(const ConvArgs& a, hipStream_t st) {
  const int code = kernel_st(a);
  if (code == 0) return launch_bf16x3_st<KH, DIL, 0>(a, st);
  if constexpr ((KH == 1 || KH == 3 || KH == 7) && (DIL == 1 || (KH == 3 && DIL == 2))) {
    switch (code) {
      case 1: return launch_bf16x3_st<KH, DIL, 1>(a, st);
      case 3: return launch_bf16x3_st<KH, DIL, 3>(a, st);
      case 4: return launch_bf16x3_st<KH, DIL, 4>(a, st);
      case 6: return launch_bf16x3_st<KH, DIL, 6>(a, st);
      default: break;
    }
    if constexpr (KH == 1) {
      if (code == 2) return launch_bf16x3_st<KH, DIL, 2>(a, st);
      if (code == 5) return launch_bf16x3_st<KH, DIL, 5>(a, st);
    }
    // fp16 maps (round 5): one fp16 MFMA per product (12; 14 behind an input PReLU); the folded 1x1 behind the guided filter with
    // fp16 hi + lo weights (9); the dilation-2 3x3 that writes the forward's last map as fp32 (15)
    switch (code) {
      case 12: return launch_bf16x3_st<KH, DIL, 12>(a, st);
      case 14: return launch_bf16x3_st<KH, DIL, 14>(a, st);
      default: break;
    }
    if constexpr (KH == 1) {
      if (code == 9) return launch_bf16x3_st<KH, DIL, 9>(a, st);
    }
    if constexpr (KH == 3 && DIL == 2) {
      if (code == 15) return launch_bf16x3_st<KH, DIL, 15>(a, st);
    }
  }
  paif::set_error("conv2d: storage %d is not built for the %dx%d dilation-%d kernel", a.st, KH, KH, DIL);
  return PAIF_ENOSUP;
}

// w [cout][nsrc*32][kh][kh] fp32 -> wpk[src][tap][ks][hi|lo][64 lanes][8 bf16]
__global__ void pack_weight_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int cout, int nsrc,
                                          int kh) {
  const int ntap = kh * kh;
  const int total = nsrc * ntap * 2 * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1, ks = (idx >> 10) & 1;
    int rest = idx >> 11;
    const int tap = rest % ntap;
    const int s = rest / ntap;
    const int n = lane & 31, hh = lane >> 5;
    const int c = s * 32 + 16 * ks + 8 * hh + j;
    const float v = (n < cout) ? w[((size_t)n * (nsrc * 32) + c) * ntap + tap] : 0.f;
    const __bf16 hi = (__bf16)v;
    const __bf16 out = part == 0 ? hi : (__bf16)(v - (float)hi);
    wpk[idx] = __builtin_bit_cast(unsigned short, out);
  }
}

// depthwise k x k followed by a 1x1 (operations_m.py:494-506 DilConv) as ONE dense k x k conv: W[co][ci][tap] = pw[co][ci] * dw[ci][tap]
__global__ void compose_dw_pw_kernel(const float* __restrict__ dw, const float* __restrict__ pw, float* __restrict__ out, int cout,
                                     int cin, int ntap) {
  const int total = cout * cin * ntap;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int tap = idx % ntap, ci = (idx / ntap) % cin, co = idx / (ntap * cin);
    out[idx] = pw[co * cin + ci] * dw[ci * ntap + tap];
  }
}

// a k x k conv followed by a 1x1 with nothing in between (operations_m.py:451-464 ResidualModule: conv3x3 dil 2 -> conv1x1 -> BN -> PReLU)
// as ONE k x k conv: W[co][ci][tap] = sum_m pw[co][m] * w[m][ci][tap]  (exact composition: the 1x1 has no padding); summed in double,
// rounded to fp32 once
__global__ void compose_pw_conv_kernel(const float* __restrict__ pw, const float* __restrict__ w, float* __restrict__ out, int cout, int cmid,
                                       int cin, int ntap) {
  const int total = cout * cin * ntap;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int rest = idx % (cin * ntap), co = idx / (cin * ntap);
    double acc = 0.0;
    for (int m = 0; m < cmid; ++m) acc = fma((double)pw[co * cmid + m], (double)w[(size_t)m * cin * ntap + rest], acc);
    out[idx] = (float)acc;
  }
}

// three-piece form (bf16x6): wpk[src][tap][ks][hi|mid|lo][64 lanes][8 bf16]
__device__ __forceinline__ unsigned short bf16x6_piece(float v, int part) {
  const __bf16 hi = (__bf16)v;
  const float r1 = v - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 out = part == 0 ? hi : (part == 1 ? mid : (__bf16)(r1 - (float)mid));
  return __builtin_bit_cast(unsigned short, out);
}
__global__ void pack_weight_bf16x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int cout, int nsrc, int kh) {
  const int ntap = kh * kh;
  const int total = nsrc * ntap * 2 * 3 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int rest = idx >> 9;
    const int part = rest % 3; rest /= 3;
    const int ks = rest & 1; rest >>= 1;
    const int tap = rest % ntap;
    const int s = rest / ntap;
    const int n = lane & 31, hh = lane >> 5;
    const int c = s * 32 + 16 * ks + 8 * hh + j;
    const float v = (n < cout) ? w[((size_t)n * (nsrc * 32) + c) * ntap + tap] : 0.f;
    wpk[idx] = bf16x6_piece(v, part);
  }
}
__global__ void pack_decomp1x1_bf16x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk) {
  const int total = 3 * 2 * 3 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int rest = idx >> 9;
    const int part = rest % 3; rest /= 3;
    const int ks = rest & 1;
    const int s = rest >> 1;
    const int n = lane & 31, hh = lane >> 5;
    const int c = 16 * ks + 8 * hh + j;
    const float* wn = w + n * 128;
    float v;
    if (s == 0) v = wn[64 + c] + wn[96 + c];
    else if (s == 1) v = wn[c] - wn[64 + c];
    else v = wn[32 + c] - wn[96 + c];
    wpk[idx] = bf16x6_piece(v, part);
  }
}

// decomposition 1x1 (see pack_decomp1x1_kernel) in the bf16x3 stream layout
__global__ void pack_decomp1x1_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk) {
  const int total = 3 * 2 * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1, ks = (idx >> 10) & 1, s = idx >> 11;
    const int n = lane & 31, hh = lane >> 5;
    const int c = 16 * ks + 8 * hh + j;
    const float* wn = w + n * 128;
    float v;
    if (s == 0) v = wn[64 + c] + wn[96 + c];
    else if (s == 1) v = wn[c] - wn[64 + c];
    else v = wn[32 + c] - wn[96 + c];
    const __bf16 hi = (__bf16)v;
    const __bf16 out = part == 0 ? hi : (__bf16)(v - (float)hi);
    wpk[idx] = __builtin_bit_cast(unsigned short, out);
  }
}

// fp16 storage (round 5): the same stream layout [src][tap][ks][hi|lo][64 lanes][8] with fp16 pieces (hi = rn(v), lo = rn(v - hi): 22
// significant bits); precision PAIF_CONV_F16 reads the hi pieces only
__device__ __forceinline__ unsigned short f16x2_piece(float v, int part) {
  const _Float16 hi = (_Float16)v;
  const _Float16 out = part == 0 ? hi : (_Float16)(v - (float)hi);
  return __builtin_bit_cast(unsigned short, out);
}
__global__ void pack_weight_f16x2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk, int cout, int nsrc, int kh) {
  const int ntap = kh * kh;
  const int total = nsrc * ntap * 2 * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1, ks = (idx >> 10) & 1;
    int rest = idx >> 11;
    const int tap = rest % ntap;
    const int s = rest / ntap;
    const int n = lane & 31, hh = lane >> 5;
    const int c = s * 32 + 16 * ks + 8 * hh + j;
    const float v = (n < cout) ? w[((size_t)n * (nsrc * 32) + c) * ntap + tap] : 0.f;
    wpk[idx] = f16x2_piece(v, part);
  }
}
// decomposition 1x1 over the sources (x, HF1, HF2) with HF_i = x - LF_i written by the guided filter (fp16 forward):
//   W . [LF1, LF2, HF1, HF2] = (Wl1 + Wl2) x + (Wh1 - Wl1) HF1 + (Wh2 - Wl2) HF2      (w rows: [Wl1 | Wl2 | Wh1 | Wh2], 32 columns each)
__global__ void pack_decomp1x1_hf_f16x2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpk) {
  const int total = 3 * 2 * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1, ks = (idx >> 10) & 1, s = idx >> 11;
    const int n = lane & 31, hh = lane >> 5;
    const int c = 16 * ks + 8 * hh + j;
    const float* wn = w + n * 128;
    float v;
    if (s == 0) v = wn[c] + wn[32 + c];
    else if (s == 1) v = wn[64 + c] - wn[c];
    else v = wn[96 + c] - wn[32 + c];
    wpk[idx] = f16x2_piece(v, part);
  }
}

template <int KH, int DIL, int CIN, bool HOOKS>
int launch_h(const ConvArgs& a, hipStream_t st) {
  constexpr int P = DIL * (KH - 1) / 2;
  constexpr size_t tile_bytes = (size_t)(TH + 2 * P) * (TW + 2 * P) * (CIN + 4) * 4;
  constexpr size_t epi_bytes = (size_t)TH * 32 * 32 * 4;   // the LDS-transposed epilogue reuses the tile's LDS
  constexpr size_t lds_bytes = tile_bytes > epi_bytes ? tile_bytes : epi_bytes;
  static_assert(lds_bytes <= 160 * 1024, "tile does not fit LDS");
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_f32<KH, DIL, CIN, HOOKS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      paif::set_error("conv2d: cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL((conv_mfma_f32<KH, DIL, CIN, HOOKS>), dim3(a.nblk), dim3(NTHREADS), lds_bytes, st, a);
  PAIF_LAUNCH_CHECK("conv2d");
  return 0;
}

template <int KH, int DIL, int CIN>
int launch(const ConvArgs& a, hipStream_t st) {
  return needs_hooks(a) ? launch_h<KH, DIL, CIN, true>(a, st) : launch_h<KH, DIL, CIN, false>(a, st);
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cout, int nsrc, int cin,
                                   int kh) {
  const int ntap = kh * kh, no = cin / 8;
  const int total = nsrc * ntap * no * 256;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int i = idx & 3, lane = (idx >> 2) & 63;
    int rest = idx >> 8;
    const int o = rest % no; rest /= no;
    const int tap = rest % ntap;
    const int s = rest / ntap;
    const int n = lane & 31, h = lane >> 5;
    const int c = s * cin + 8 * o + 4 * h + i;
    wpk[idx] = (n < cout) ? w[((size_t)n * (nsrc * cin) + c) * ntap + tap] : 0.f;
  }
}

// w [32][128] over cat[LF1, LF2, x-LF1, x-LF2]  ->  1x1 over sources (x, LF1, LF2)
__global__ void pack_decomp1x1_kernel(const float* __restrict__ w, float* __restrict__ wpk) {
  const int total = 3 * 4 * 256;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int i = idx & 3, lane = (idx >> 2) & 63;
    const int o = (idx >> 8) & 3, s = idx >> 10;
    const int n = lane & 31, h = lane >> 5;
    const int c = 8 * o + 4 * h + i;
    const float* wn = w + n * 128;
    float v;
    if (s == 0) v = wn[64 + c] + wn[96 + c];
    else if (s == 1) v = wn[c] - wn[64 + c];
    else v = wn[32 + c] - wn[96 + c];
    wpk[idx] = v;
  }
}

__global__ void bn_fold_kernel(const float* g, const float* bta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = g[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = bta[c] - mean[c] * s;
  }
}

}  // namespace

extern "C" {

int paif_conv2d_blocks(int B, int H, int W) { return B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }

size_t paif_conv_wpk_floats(int nsrc, int cin, int kh) { return (size_t)nsrc * kh * kh * (cin / 8) * 256; }

int paif_conv2d_is_persistent(const paif_conv_desc* d, int B, int H, int W) {
  if (!d || (d->precision != PAIF_CONV_BF16X3 && d->precision != PAIF_CONV_BF16 && d->precision != PAIF_CONV_F16 && d->precision != PAIF_CONV_F16X2) ||
      d->cin != 32 || B <= 0 || H <= 0 || W <= 0)
    return 0;
  ConvArgs a{};
  for (int s = 0; s < 3; ++s) a.res[s] = d->res[s];
  a.pool_partial = d->pool_partial; a.nsrc = d->nsrc; a.in_act = d->in_act; a.cout = d->cout;
  a.aux_out = d->aux_out; a.epi_dact = d->epi_dact; a.st = d->storage;
  a.H = H; a.W = W;
  a.nblk = B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  return takes_ws(a, d->kh, d->dil) ? 1 : 0;
}

int paif_conv2d_can_cpool(const paif_conv_desc* d, int B, int H, int W) {
  if (!d || d->cout != 32 || d->cin != 32 || B <= 0 || H <= 0 || W <= 0 || d->precision == PAIF_CONV_F32 || d->precision == PAIF_CONV_BF16X6 || d->precision == PAIF_CONV_F16X3) return 0;
  ConvArgs a{};
  for (int s = 0; s < 3; ++s) a.res[s] = d->res[s];
  a.pool_partial = d->pool_partial; a.nsrc = d->nsrc; a.in_act = d->in_act; a.cout = d->cout;
  a.aux_out = d->aux_out; a.epi_dact = d->epi_dact;
  a.B = B; a.H = H; a.W = W;
  a.nblk = B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  a.st = d->storage; a.wl0 = (d->precision == PAIF_CONV_BF16 || d->precision == PAIF_CONV_F16) ? 1 : 0; a.alpha = d->alpha;
  return variant_can_cpool(a, d->kh, d->dil) ? 1 : 0;
}

int paif_conv2d_kernel_name(const paif_conv_desc* d, int B, int H, int W, char* buf, int buflen) {
  PAIF_REQUIRE(d && buf && buflen > 0 && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "conv2d_kernel_name: bad arguments");
  ConvArgs a{};
  for (int s = 0; s < 3; ++s) a.res[s] = d->res[s];
  a.pool_partial = d->pool_partial; a.nsrc = d->nsrc; a.in_act = d->in_act; a.cout = d->cout;
  a.aux_out = d->aux_out; a.epi_dact = d->epi_dact;
  a.B = B; a.H = H; a.W = W;
  a.nblk = B * ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  a.st = d->storage; a.wl0 = (d->precision == PAIF_CONV_BF16 || d->precision == PAIF_CONV_F16) ? 1 : 0; a.alpha = d->alpha;
  const int code = kernel_st(a);
  if (d->precision == PAIF_CONV_BF16X6 && d->cin == 32) {
    snprintf(buf, buflen, "conv_mfma_bf16x6<%d, %d, %s, 0>", d->kh, d->dil, needs_hooks(a) ? "true" : "false");
    return 0;
  }
  if (d->precision == PAIF_CONV_F16X3 && d->cin == 32) {
    snprintf(buf, buflen, "conv_mfma_f16x3<%d, %d, %s, 0>", d->kh, d->dil, needs_hooks(a) ? "true" : "false");
    return 0;
  }
  if (d->precision != PAIF_CONV_BF16X3 && d->precision != PAIF_CONV_BF16 && d->precision != PAIF_CONV_F16 && d->precision != PAIF_CONV_F16X2) {
    snprintf(buf, buflen, "conv_mfma_f32<%d, %d, %d, %s>", d->kh, d->dil, d->cin, needs_hooks(a) ? "true" : "false");
    return 0;
  }
  if (d->cin != 32) {   // the folded decomposition conv (cin 16) runs on the exact-fp32 kernel
    snprintf(buf, buflen, "conv_mfma_f32<%d, %d, %d, %s>", d->kh, d->dil, d->cin, needs_hooks(a) ? "true" : "false");
    return 0;
  }
  // the names rocprofv3 prints: every template argument, the storage code last
  switch (bf16x3_variant(a, d->kh, d->dil)) {
    case CV_DMA:
      if (d->kh == 7) snprintf(buf, buflen, "conv7x7_h16_dma<%d>", a.st >= 3 ? 2 : 1);
      else snprintf(buf, buflen, "conv3x3_h16_dma<%d, %d, %d, %s, %d, %d>", d->nsrc, res_count(a), a.st >= 3 ? 2 : 1, d->cpool ? "true" : "false",
                    d->dil, d->dil == 2 ? 2 : 0);
      break;
    case CV_WS: snprintf(buf, buflen, "conv_bf16x3_ws%s<%d, %d, %d>", d->in_act == 2 ? "r" : "", d->kh, d->dil, code); break;
    case CV_RES: snprintf(buf, buflen, "conv_bf16x3_res<%d, %d, %d, %d, %d>", d->kh, d->dil, d->nsrc, PAIF_RES_ROWS, code); break;
    case CV_MS: snprintf(buf, buflen, "conv_bf16x3_ms<%d, %d, %d, %d>", d->kh, d->dil, d->nsrc, code); break;
    case CV_HOOKS: snprintf(buf, buflen, "conv_mfma_bf16x3<%d, %d, true, %d>", d->kh, d->dil, code); break;
    default: snprintf(buf, buflen, "conv_mfma_bf16x3<%d, %d, false, %d>", d->kh, d->dil, code); break;
  }
  return 0;
}

int paif_conv2d_fwd(const paif_conv_desc* d, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(d && d->out && d->wpk, PAIF_EINVAL, "conv2d: null descriptor/out/wpk");
  PAIF_REQUIRE(d->nsrc >= 1 && d->nsrc <= 3, PAIF_EINVAL, "conv2d: nsrc=%d", d->nsrc);
  for (int s = 0; s < d->nsrc; ++s) PAIF_REQUIRE(d->src[s], PAIF_EINVAL, "conv2d: src[%d] null", s);
  PAIF_REQUIRE(d->cout == 32 || d->cout == 16, PAIF_ENOSUP, "conv2d: cout=%d", d->cout);
  PAIF_REQUIRE(B > 0 && H > 0 && W > 0, PAIF_EINVAL, "conv2d: empty shape %dx%dx%d", B, H, W);
  PAIF_REQUIRE(d->in_act != 1 || d->in_prelu, PAIF_EINVAL, "conv2d: in_act=PReLU without slope");
  PAIF_REQUIRE(d->act != 1 || d->prelu, PAIF_EINVAL, "conv2d: act=PReLU without slope");
  ConvArgs a;
  for (int s = 0; s < 3; ++s) {
    a.src[s] = s < d->nsrc ? d->src[s] : nullptr;
    a.res[s] = d->res[s];
  }
  a.wpk = reinterpret_cast<const float4*>(d->wpk);
  a.in_prelu = d->in_prelu; a.scale = d->scale; a.shift = d->shift; a.prelu = d->prelu;
  a.out = d->out; a.pool_partial = d->pool_partial; a.cpool = d->cpool; a.alpha = d->alpha;
  a.nsrc = d->nsrc; a.in_act = d->in_act; a.act = d->act; a.cout = d->cout;
  a.aux_out = d->aux_out; a.in_aux = d->in_aux; a.in_scale = d->in_scale; a.in_alpha = d->in_alpha;
  a.epi_aux = d->epi_aux; a.epi_dact = d->epi_dact;
  PAIF_REQUIRE(d->in_act >= 0 && d->in_act <= 5, PAIF_EINVAL, "conv2d: in_act=%d", d->in_act);
  PAIF_REQUIRE(!(d->in_act == 3 || d->in_act == 4) || (d->in_aux && d->nsrc == 1), PAIF_EINVAL,
               "conv2d: in_act=%d needs in_aux and a single source", d->in_act);
  PAIF_REQUIRE(d->in_act != 3 || d->in_prelu, PAIF_EINVAL, "conv2d: in_act=3 without slope");
  PAIF_REQUIRE(d->epi_dact >= 0 && d->epi_dact <= 2 && (!d->epi_dact || d->epi_aux), PAIF_EINVAL, "conv2d: epi_dact=%d", d->epi_dact);
  PAIF_REQUIRE(d->epi_dact != 1 || d->prelu, PAIF_EINVAL, "conv2d: epi_dact=PReLU without slope");
  a.B = B; a.H = H; a.W = W;
  a.tilesX = (W + TW - 1) / TW; a.tilesY = (H + TH - 1) / TH; a.nblk = B * a.tilesX * a.tilesY;
  a.reverse = d->reverse_tiles ? 1 : 0;
  a.st = d->storage;
  a.wl0 = (d->precision == PAIF_CONV_BF16 || d->precision == PAIF_CONV_F16) ? 1 : 0;
  PAIF_REQUIRE(d->storage >= 0 && d->storage <= 4, PAIF_EINVAL, "conv2d: storage=%d", d->storage);
  const bool st_bf = d->storage == PAIF_ST_BF16 || d->storage == PAIF_ST_F32_BF16, st_hf = d->storage == PAIF_ST_F16 || d->storage == PAIF_ST_F16_F32;
  PAIF_REQUIRE(!st_bf || ((d->precision == PAIF_CONV_BF16X3 || d->precision == PAIF_CONV_BF16) && d->cin == 32), PAIF_ENOSUP,
               "conv2d: bf16 storage needs the split-bf16 kernels (cin 32)");
  PAIF_REQUIRE(!st_hf || ((d->precision == PAIF_CONV_F16 || d->precision == PAIF_CONV_F16X2) && d->cin == 32), PAIF_ENOSUP,
               "conv2d: fp16 storage takes precision F16 / F16X2 (cin 32)");
  PAIF_REQUIRE(d->precision != PAIF_CONV_BF16 || st_bf, PAIF_ENOSUP, "conv2d: precision bf16 is built for bf16-stored maps only");
  PAIF_REQUIRE((d->precision != PAIF_CONV_F16 && d->precision != PAIF_CONV_F16X2) || st_hf, PAIF_ENOSUP,
               "conv2d: precision fp16 is built for fp16-stored maps only");
  PAIF_REQUIRE(d->storage != PAIF_ST_F16_F32 || d->precision == PAIF_CONV_F16, PAIF_ENOSUP, "conv2d: fp16 in / fp32 out takes plain fp16 weights");
  hipStream_t st = paif::as_stream(stream);
  PAIF_REQUIRE(!d->cpool || paif_conv2d_can_cpool(d, B, H, W), PAIF_ENOSUP, "conv2d: no fused ChannelPool for this descriptor (paif_conv2d_can_cpool)");
  PAIF_REQUIRE(d->cin == 32 || d->cin == 16, PAIF_ENOSUP, "conv2d: cin=%d", d->cin);
  PAIF_REQUIRE(d->precision == PAIF_CONV_F32 || d->precision == PAIF_CONV_BF16X3 || d->precision == PAIF_CONV_BF16 ||
                   d->precision == PAIF_CONV_BF16X6 || d->precision == PAIF_CONV_F16 || d->precision == PAIF_CONV_F16X2 ||
                   d->precision == PAIF_CONV_F16X3, PAIF_EINVAL,
               "conv2d: precision=%d", d->precision);
  PAIF_REQUIRE((d->precision != PAIF_CONV_BF16X6 && d->precision != PAIF_CONV_F16X3) || d->cin == 32, PAIF_ENOSUP, "conv2d: bf16x6 / f16x3 need 32-channel sources");
  const int key = d->kh * 100 + d->dil * 10 + (d->cin == 32 ? 0 : 1);
  if (d->precision == PAIF_CONV_F16X3) {
    switch (key) {
      case 110: return launch_f16x3<1, 1>(a, st);
      case 310: return launch_f16x3<3, 1>(a, st);
      case 320: return launch_f16x3<3, 2>(a, st);
      case 510: return launch_f16x3<5, 1>(a, st);
      case 710: return launch_f16x3<7, 1>(a, st);
      default: break;
    }
    paif::set_error("conv2d(f16x3): kernel %dx%d dil %d cin %d not built", d->kh, d->kh, d->dil, d->cin);
    return PAIF_ENOSUP;
  }
  if (d->precision == PAIF_CONV_BF16X6) {
    switch (key) {
      case 110: return launch_bf16x6<1, 1>(a, st);
      case 310: return launch_bf16x6<3, 1>(a, st);
      case 320: return launch_bf16x6<3, 2>(a, st);
      case 510: return launch_bf16x6<5, 1>(a, st);
      case 710: return launch_bf16x6<7, 1>(a, st);
      default: break;
    }
    paif::set_error("conv2d(bf16x6): kernel %dx%d dil %d cin %d not built", d->kh, d->kh, d->dil, d->cin);
    return PAIF_ENOSUP;
  }
  if (d->precision == PAIF_CONV_BF16X3 || d->precision == PAIF_CONV_BF16 || d->precision == PAIF_CONV_F16 || d->precision == PAIF_CONV_F16X2) {
    switch (key) {
      case 110: return launch_bf16x3<1, 1>(a, st);
      case 310: return launch_bf16x3<3, 1>(a, st);
      case 320: return launch_bf16x3<3, 2>(a, st);
      case 510: return launch_bf16x3<5, 1>(a, st);
      case 520: return launch_bf16x3<5, 2>(a, st);
      case 710: return launch_bf16x3<7, 1>(a, st);
      case 720: return launch_bf16x3<7, 2>(a, st);
      default: break;
    }
    paif::set_error("conv2d(bf16x3): kernel %dx%d dil %d cin %d not built", d->kh, d->kh, d->dil, d->cin);
    return PAIF_ENOSUP;
  }
  switch (key) {
    case 110: return launch<1, 1, 32>(a, st);
    case 310: return launch<3, 1, 32>(a, st);
    case 320: return launch<3, 2, 32>(a, st);
    case 510: return launch<5, 1, 32>(a, st);
    case 520: return launch<5, 2, 32>(a, st);
    case 710: return launch<7, 1, 32>(a, st);
    case 720: return launch<7, 2, 32>(a, st);
    case 311: return launch<3, 1, 16>(a, st);
    default: break;
  }
  paif::set_error("conv2d: kernel %dx%d dil %d cin %d not built", d->kh, d->kh, d->dil, d->cin);
  return PAIF_ENOSUP;
}

int paif_pack_conv_weight_bf16x3(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_conv_weight_bf16x3: null pointer");
  PAIF_REQUIRE(cout >= 1 && cout <= 32 && nsrc >= 1 && nsrc <= 3 && kh >= 1 && kh <= 7, PAIF_ENOSUP,
               "pack_conv_weight_bf16x3: cout=%d nsrc=%d kh=%d", cout, nsrc, kh);
  const int total = nsrc * kh * kh * 2048;
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk), cout, nsrc, kh);
  PAIF_LAUNCH_CHECK("pack_conv_weight_bf16x3");
  return 0;
}

int paif_pack_conv_weight_f16x2(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_conv_weight_f16x2: null pointer");
  PAIF_REQUIRE(cout >= 1 && cout <= 32 && nsrc >= 1 && nsrc <= 3 && kh >= 1 && kh <= 7, PAIF_ENOSUP,
               "pack_conv_weight_f16x2: cout=%d nsrc=%d kh=%d", cout, nsrc, kh);
  const int total = nsrc * kh * kh * 2048;
  hipLaunchKernelGGL(pack_weight_f16x2_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk), cout, nsrc, kh);
  PAIF_LAUNCH_CHECK("pack_conv_weight_f16x2");
  return 0;
}

int paif_pack_decomp1x1_hf_weight_f16x2(const float* w, float* wpk, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_decomp1x1_hf_weight_f16x2: null pointer");
  hipLaunchKernelGGL(pack_decomp1x1_hf_f16x2_kernel, dim3(24), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk));
  PAIF_LAUNCH_CHECK("pack_decomp1x1_hf_weight_f16x2");
  return 0;
}

int paif_pack_conv_weight_bf16x6(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_conv_weight_bf16x6: null pointer");
  PAIF_REQUIRE(cout >= 1 && cout <= 32 && nsrc >= 1 && nsrc <= 3 && kh >= 1 && kh <= 7, PAIF_ENOSUP,
               "pack_conv_weight_bf16x6: cout=%d nsrc=%d kh=%d", cout, nsrc, kh);
  const int total = nsrc * kh * kh * 3072;
  hipLaunchKernelGGL(pack_weight_bf16x6_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk), cout, nsrc, kh);
  PAIF_LAUNCH_CHECK("pack_conv_weight_bf16x6");
  return 0;
}

int paif_compose_dw_pw_weight(const float* dw, const float* pw, float* out, int cout, int cin, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(dw && pw && out && cout >= 1 && cin >= 1 && kh >= 1 && kh <= 7, PAIF_EINVAL, "compose_dw_pw_weight: bad arguments");
  const int total = cout * cin * kh * kh;
  hipLaunchKernelGGL(compose_dw_pw_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), dw, pw, out, cout, cin, kh * kh);
  PAIF_LAUNCH_CHECK("compose_dw_pw_weight");
  return 0;
}

int paif_compose_pw_conv_weight(const float* pw, const float* w, float* out, int cout, int cmid, int cin, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(pw && w && out && cout >= 1 && cmid >= 1 && cin >= 1 && kh >= 1 && kh <= 7, PAIF_EINVAL, "compose_pw_conv_weight: bad arguments");
  const int total = cout * cin * kh * kh;
  hipLaunchKernelGGL(compose_pw_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), pw, w, out, cout, cmid, cin, kh * kh);
  PAIF_LAUNCH_CHECK("compose_pw_conv_weight");
  return 0;
}

int paif_pack_decomp1x1_weight_bf16x6(const float* w, float* wpk, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_decomp1x1_weight_bf16x6: null pointer");
  hipLaunchKernelGGL(pack_decomp1x1_bf16x6_kernel, dim3(36), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk));
  PAIF_LAUNCH_CHECK("pack_decomp1x1_weight_bf16x6");
  return 0;
}

int paif_pack_decomp1x1_weight_bf16x3(const float* w, float* wpk, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_decomp1x1_weight_bf16x3: null pointer");
  hipLaunchKernelGGL(pack_decomp1x1_bf16x3_kernel, dim3(24), dim3(256), 0, paif::as_stream(stream), w,
                     reinterpret_cast<unsigned short*>(wpk));
  PAIF_LAUNCH_CHECK("pack_decomp1x1_weight_bf16x3");
  return 0;
}

int paif_pack_conv_weight(const float* w, float* wpk, int cout, int nsrc, int cin, int kh, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_conv_weight: null pointer");
  PAIF_REQUIRE(cout >= 1 && cout <= 32 && nsrc >= 1 && nsrc <= 3 && (cin == 32 || cin == 16) && kh >= 1 && kh <= 7,
               PAIF_ENOSUP, "pack_conv_weight: cout=%d nsrc=%d cin=%d kh=%d", cout, nsrc, cin, kh);
  const int total = (int)paif_conv_wpk_floats(nsrc, cin, kh);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, paif::as_stream(stream), w, wpk, cout,
                     nsrc, cin, kh);
  PAIF_LAUNCH_CHECK("pack_conv_weight");
  return 0;
}

int paif_pack_decomp1x1_weight(const float* w, float* wpk, paif_stream_t stream) {
  PAIF_REQUIRE(w && wpk, PAIF_EINVAL, "pack_decomp1x1_weight: null pointer");
  hipLaunchKernelGGL(pack_decomp1x1_kernel, dim3(12), dim3(256), 0, paif::as_stream(stream), w, wpk);
  PAIF_LAUNCH_CHECK("pack_decomp1x1_weight");
  return 0;
}

int paif_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int C, paif_stream_t stream) {
  PAIF_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, PAIF_EINVAL, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, paif::as_stream(stream), gamma, beta, mean, var,
                     eps, scale, shift, C);
  PAIF_LAUNCH_CHECK("bn_fold");
  return 0;
}

}  // extern "C"
