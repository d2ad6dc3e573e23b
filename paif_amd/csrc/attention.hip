// Fused spatial-reduction attention of the MixTransformer encoder (core/mix_transformer.py:93-115):
//   out[b, n, hd*D + :] = softmax_k( q[b,n,hd,:] . k[b,:,hd,:]^T * D^-0.5 ) @ v[b,:,hd,:]
// The spatial-reduction conv leaves only Nk = (H/sr)(W/sr) keys (300 at 480x640 for every stage), so the
// whole K and V of one (batch, head) live in LDS (Nk*D*8 B <= 160 KiB) and each wave streams 32 queries
// through fp32 MFMA (v_mfma_f32_32x32x2_f32) with an online softmax -- no score matrix in HBM.
// One workgroup per CU (the K/V image fills LDS) of 8 waves = 256 queries: two waves per SIMD, so the softmax VALU
// section of one overlaps the MFMAs of the other, and the K/V staging is paid once per 256 queries.
//
// Orientation ("key on the register, query on the lane"):
//   S^T tile [32 keys x 32 queries] = K_tile . Q^T   (A = K rows from LDS, B = Q fragment in registers)
//     -> lane (h, j) holds, for query j, keys (r&3)+8(r>>2)+4h of the tile: the softmax row reduction is
//        in-register plus ONE cross-half exchange (lane ^ 32);
//   O^T [D x 32 queries] += V_tile^T . P^T            (B = the exponentiated accumulator registers as they
//     stand: register r of half h is key c0(r)+4h; A = V[key][dim = lane&31] read row-wise from LDS).
// K is stored with a 16-byte-chunk XOR swizzle (chunk ^ (key & 15)) so the 16 keys of a ds_read_b128 lane
// group hit 16 distinct bank slots without padding.
#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
  const float* q; const float* kv; float* out; float* lse;
  int B, N, Nk, C, heads;
  float scale;
};

template <int D>
__global__ __launch_bounds__(512) void sr_attention_kernel(AttnArgs a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int NT = 512;          // threads: 8 waves x 32 queries
  constexpr int CH = D / 4;        // 16-byte chunks per row
  constexpr int NO = D / 8;        // k-octets of the QK^T contraction
  constexpr int DT = D / 32;       // 32-wide dim tiles of O
  float* Ks = lds;                 // [Nk][D] swizzled
  float* Vs = lds + (size_t)a.Nk * D;  // [Nk][D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk;
  const float* kvb = a.kv + (size_t)b * Nk * 2 * C + hd * D;

  // ---- stage K (swizzled) and V of this (batch, head) ----
  for (int idx = tid; idx < Nk * CH; idx += NT) {
    const int key = idx / CH, c = idx - key * CH;
    const float4 kk = *reinterpret_cast<const float4*>(kvb + (size_t)key * 2 * C + c * 4);
    const float4 vv = *reinterpret_cast<const float4*>(kvb + (size_t)key * 2 * C + C + c * 4);
    *reinterpret_cast<float4*>(Ks + key * D + ((c ^ (key & (CH - 1))) << 2)) = kk;
    *reinterpret_cast<float4*>(Vs + key * D + c * 4) = vv;
  }
  __syncthreads();

  const int q0 = blockIdx.x * (NT / 2) + wave * 32;
  if (q0 >= a.N) return;  // whole wave out of range (no barriers below)
  const int qi = min(q0 + p, a.N - 1);  // clamp: lanes past N compute a duplicate and do not store
  const float* qrow = a.q + ((size_t)b * a.N + qi) * C + hd * D;
  float4 qf[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) qf[o] = *reinterpret_cast<const float4*>(qrow + 8 * o + 4 * h);

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntile = (Nk + 31) / 32;
  for (int t = 0; t < ntile; ++t) {
    // ---- S^T tile = K_tile . Q^T ----
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const int key_a = min(t * 32 + p, Nk - 1);
    const float* krow = Ks + key_a * D;
    const int sw = key_a & (CH - 1);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float4 kf = *reinterpret_cast<const float4*>(krow + (((2 * o + h) ^ sw) << 2));
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[o].x, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[o].y, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[o].z, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[o].w, st, 0, 0, 0);
    }
    // ---- online softmax over this tile's keys (register r <-> key t*32 + (r&3)+8(r>>2)+4h) ----
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      st[r] = (key < Nk) ? st[r] * a.scale : -INFINITY;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);          // finite: every tile has >= 1 valid key
    const float alpha = expf(m_run - m_new);       // exp(-inf) = 0 on the first tile
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_new);                 // masked keys: exp(-inf) = 0
      ls += st[r];
    }
    ls += __shfl_xor(ls, 32);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
    // ---- O^T += V_tile^T . P^T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, Nk - 1);  // P = 0 beyond Nk
      const float* vrow = Vs + key * D + p;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * dt], st[r], oacc[dt], 0, 0, 0);
    }
  }

  // ---- normalise and store: lane (h, j) holds dims 32*dt + 8*g + 4*h + (0..3) of query j ----
  if (a.lse && h == 0 && q0 + p < a.N) a.lse[((size_t)b * a.heads + hd) * a.N + q0 + p] = m_run + logf(l_run);
  if (q0 + p < a.N) {
    const float inv = 1.0f / l_run;
    float* orow = a.out + ((size_t)b * a.N + q0 + p) * C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * h) =
            make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
  }
}

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t st) {
  const size_t lds_bytes = (size_t)a.Nk * D * 8;
  if (lds_bytes > 160 * 1024) {
    paif::set_error("sr_attention: Nk=%d keys x D=%d need %zu B of LDS (> 160 KiB); key tiling through LDS is not built", a.Nk, D,
                    lds_bytes);
    return PAIF_ENOSUP;
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sr_attention_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      paif::set_error("sr_attention: cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL((sr_attention_kernel<D>), dim3((a.N + 255) / 256, a.heads, a.B), dim3(512), lds_bytes, st, a);
  PAIF_LAUNCH_CHECK("sr_attention");
  return 0;
}

}  // namespace

extern "C" int paif_sr_attention_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C,
                                     int heads, paif_stream_t stream) {
  PAIF_REQUIRE(q && kv && out && B > 0 && N > 0 && Nk > 0 && heads > 0, PAIF_EINVAL, "sr_attention: bad arguments");
  PAIF_REQUIRE(C % heads == 0, PAIF_EINVAL, "sr_attention: C=%d not divisible by heads=%d", C, heads);
  const int D = C / heads;
  AttnArgs a;
  a.q = q; a.kv = kv; a.out = out; a.lse = lse; a.B = B; a.N = N; a.Nk = Nk; a.C = C; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  hipStream_t st = paif::as_stream(stream);
  if (D == 64) return launch_attn<64>(a, st);
  if (D == 32) return launch_attn<32>(a, st);
  paif::set_error("sr_attention: head dim %d not built (32 and 64 are)", D);
  return PAIF_ENOSUP;
}
