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

// ---------------------------------------------------------------------------------------------------
// Split-bf16 form (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate, ~1e-5 relative products): the exact
// kernel above is bound by the fp32 matrix pipe (stage 1 at B=16: 23.6 GFLOP per launch = 150 us at 157 TF, measured 145 us);
// this one needs 5.3x less matrix-pipe time and becomes softmax-VALU bound.  Same orientation, same online softmax.
//   K  in LDS as [key][D] bf16 hi | lo (D*4 B per key, 16-byte chunks XOR-swizzled with key & 15): A operand of S^T = K . Q^T
//   V  in LDS TRANSPOSED, [dim][key slot] bf16 hi | lo, records padded by 16 B (conflict-free b128): A operand of O^T += V^T . P^T.  The key
//      slots of a 32-key tile are permuted so that the 8 slots a lane half supplies to a K=16 step are exactly the keys its
//      accumulator registers hold (slot 16*(r>>3) + 8*h + (r&7) <-> key (r&3) + 8*(r>>2) + 4*h): P goes from the softmax
//      registers into the B operand without any cross-lane exchange.
// LDS: Nk*D*4 + D*(ceil32(Nk)*4 + 16) bytes (159,744 B at Nk = 300, D = 64).
// ---------------------------------------------------------------------------------------------------
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));

using paif::splitN;
using paif::mfma_pieces;

// NP: bf16 pieces per operand.  KC (multiple of 32): keys per LDS chunk -- K (krows = min(KC, Nk) row records of NP x D bf16) and
// V^T (D records of NP x KC slots) of a chunk are staged together; the online softmax carries over chunk boundaries as it does
// over key tiles.  Two pieces x 300 keys x 64 dims fit at once (159,744 B); three pieces take chunks of 160 keys.
template <int D, int NP, int PF = 0>
__device__ __forceinline__ void sr_attention_split_body(const AttnArgs& a, int KC, int krows) {
  extern __shared__ __align__(16) char ldsc[];
  constexpr int NT = 512;
  constexpr int NO = D / 16;       // K=16 steps of the QK^T contraction
  constexpr int DT = D / 32;       // 32-wide dim tiles of O
  constexpr int KREC = D * 2 * NP; // bytes per key record: piece 0 | piece 1 (| piece 2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk;
  const int VREC = KC * 2 * NP + 16;   // bytes per dim record of V^T: NP x KC slots | pad
  char* Ks = ldsc;
  char* Vt = ldsc + (size_t)krows * KREC;
  const float* kvb = a.kv + (size_t)b * Nk * 2 * C + hd * D;

  const int q0 = blockIdx.x * (NT / 2) + wave * 32;
  const bool wave_live = q0 < a.N;      // dead waves still stage and hit the barriers
  const int qi = min(q0 + p, a.N - 1);  // clamp: lanes past N compute a duplicate and do not store
  const float* qrow = a.q + ((size_t)b * a.N + qi) * C + hd * D;
  abf16x8 qp[NO][NP];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float4 q0v = *reinterpret_cast<const float4*>(qrow + 16 * o + 8 * h);
    const float4 q1v = *reinterpret_cast<const float4*>(qrow + 16 * o + 8 * h + 4);
    const float q8[8] = {q0v.x, q0v.y, q0v.z, q0v.w, q1v.x, q1v.y, q1v.z, q1v.w};
    splitN<NP, PF>(q8, qp[o]);
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int c0 = 0; c0 < Nk; c0 += KC) {
    const int kc = min(KC, Nk - c0);
    const int kcp = (kc + 31) & ~31;
    if (c0 > 0) __syncthreads();           // previous chunk fully consumed
    // ---- stage K (row-major, split) and V (transposed, split, key slots permuted) of the chunk ----
    for (int idx = tid; idx < kc * (D / 8); idx += NT) {
      const int key = idx / (D / 8), c = idx - key * (D / 8);       // c: 8-dim group
      const float* src = kvb + (size_t)(c0 + key) * 2 * C + c * 8;
      const float4 k0 = *reinterpret_cast<const float4*>(src);
      const float4 k1 = *reinterpret_cast<const float4*>(src + 4);
      const float kv8[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      abf16x8 pc[NP];
      splitN<NP, PF>(kv8, pc);
      // record: piece q at q * 2D bytes; 16-byte chunks swizzled inside each piece (D/8 is 4 or 8 chunks: key & (D/8 - 1))
      const int sw = key & (D / 8 - 1);
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<abf16x8*>(Ks + (size_t)key * KREC + q * D * 2 + ((c ^ sw) << 4)) = pc[q];
    }
    for (int idx = tid; idx < kcp * (D / 4); idx += NT) {
      const int key = idx / (D / 4), q4 = idx - key * (D / 4);
      float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);                    // slots past the last key hold zeros (their P is zero anyway)
      if (key < kc) vv = *reinterpret_cast<const float4*>(kvb + (size_t)(c0 + key) * 2 * C + C + q4 * 4);
      const int kk = key & 31, t = key >> 5;
      const int hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
      const int slot = t * 32 + (r >> 3) * 16 + 8 * hh + (r & 7);
      const float vf[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int dim = q4 * 4 + i;
        float rr = vf[i];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          float back;
          *reinterpret_cast<__bf16*>(Vt + (size_t)dim * VREC + q * KC * 2 + slot * 2) = paif::piece16<PF>(rr, back);
          if (q + 1 < NP) rr -= back;
        }
      }
    }
    __syncthreads();
    if (!wave_live) continue;

    const int ntile = kcp >> 5;
    for (int t = 0; t < ntile; ++t) {
      // ---- S^T tile = K_tile . Q^T ----
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const int key_a = min(t * 32 + p, kc - 1);
      const char* krow = Ks + (size_t)key_a * KREC;
      const int sw = key_a & (D / 8 - 1);
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int c = 2 * o + h;
        abf16x8 kp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) kp[q] = *reinterpret_cast<const abf16x8*>(krow + q * D * 2 + ((c ^ sw) << 4));
        mfma_pieces<NP, PF>(st, kp, qp[o]);
      }
      // ---- online softmax over this tile's keys (register r <-> key c0 + t*32 + (r&3)+8(r>>2)+4h) ----
      float mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        st[r] = (key < kc) ? st[r] * a.scale : -INFINITY;
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
      // ---- O^T += V_tile^T . P^T : two K=16 steps, step s takes registers 8s .. 8s+7 of both lane halves ----
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float p8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p8[i] = st[8 * s2 + i];
        abf16x8 pp[NP];
        splitN<NP, PF>(p8, pp);
        const int soff = (t * 32 + s2 * 16 + 8 * h) * 2;   // byte offset of this lane half's 8 slots
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int dim = 32 * dt + p;
          abf16x8 vp[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) vp[q] = *reinterpret_cast<const abf16x8*>(Vt + (size_t)dim * VREC + q * KC * 2 + soff);
          mfma_pieces<NP, PF>(oacc[dt], vp, pp);
        }
      }
    }
  }

  // ---- normalise and store: lane (h, j) holds dims 32*dt + 8*g + 4*h + (0..3) of query j ----
  if (!wave_live) return;
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

// one kernel name per arithmetic (round 6: rocprofv3 rows need no template-argument decoding): two bf16 pieces (3 MFMAs per product),
// three bf16 pieces (6), two IEEE fp16 pieces (3, fp32-level)
template <int D> __global__ __launch_bounds__(512) void sr_attention_bf16x3_kernel(AttnArgs a, int KC, int krows) { sr_attention_split_body<D, 2, 0>(a, KC, krows); }
template <int D> __global__ __launch_bounds__(512) void sr_attention_bf16x6_kernel(AttnArgs a, int KC, int krows) { sr_attention_split_body<D, 3, 0>(a, KC, krows); }
template <int D> __global__ __launch_bounds__(512) void sr_attention_f16x3_kernel(AttnArgs a, int KC, int krows) { sr_attention_split_body<D, 2, 1>(a, KC, krows); }

template <int D, int NP, int PF = 0>
int launch_attn_split(const AttnArgs& a, hipStream_t st) {
  auto kern = [] {
    if constexpr (PF == 1) return &sr_attention_f16x3_kernel<D>;
    else if constexpr (NP == 3) return &sr_attention_bf16x6_kernel<D>;
    else return &sr_attention_bf16x3_kernel<D>;
  }();
  // the largest chunk (multiple of 32 keys, at most ceil32(Nk)) whose K rows + V^T records fit the 160 KiB of LDS
  int KC = (a.Nk + 31) & ~31;
  size_t lds_bytes = 0;
  for (; KC >= 32; KC -= 32) {
    const int krows = KC < a.Nk ? KC : a.Nk;
    lds_bytes = (size_t)krows * D * 2 * NP + (size_t)D * (KC * 2 * NP + 16);
    if (lds_bytes <= 160 * 1024) break;
  }
  if (KC < 32) { paif::set_error("sr_attention(split): no key chunk fits LDS"); return PAIF_ENOSUP; }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      paif::set_error("sr_attention(split): cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
  }
  hipLaunchKernelGGL(kern, dim3((a.N + 255) / 256, a.heads, a.B), dim3(512), lds_bytes, st, a, KC,
                     KC < a.Nk ? KC : a.Nk);
  PAIF_LAUNCH_CHECK("sr_attention(split)");
  return 0;
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
                                     int heads, paif_stream_t stream);

extern "C" int paif_sr_attention_split_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C,
                                          int heads, int precision, paif_stream_t stream) {
  PAIF_REQUIRE(q && kv && out && B > 0 && N > 0 && Nk > 0 && heads > 0, PAIF_EINVAL, "sr_attention(split): bad arguments");
  PAIF_REQUIRE(C % heads == 0, PAIF_EINVAL, "sr_attention(split): C=%d not divisible by heads=%d", C, heads);
  PAIF_REQUIRE(precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "sr_attention(split): precision=%d", precision);
  const int D = C / heads;
  AttnArgs a;
  a.q = q; a.kv = kv; a.out = out; a.lse = lse; a.B = B; a.N = N; a.Nk = Nk; a.C = C; a.heads = heads;
  a.scale = 1.0f / sqrtf((float)D);
  hipStream_t st = paif::as_stream(stream);
  if (D == 64) return precision == 3 ? launch_attn_split<64, 3>(a, st) : precision == 6 ? launch_attn_split<64, 2, 1>(a, st) : launch_attn_split<64, 2>(a, st);
  if (D == 32) return precision == 3 ? launch_attn_split<32, 3>(a, st) : precision == 6 ? launch_attn_split<32, 2, 1>(a, st) : launch_attn_split<32, 2>(a, st);
  paif::set_error("sr_attention(split): head dim %d not built (32 and 64 are)", D);
  return PAIF_ENOSUP;
}

extern "C" int paif_sr_attention_bf16x3_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C,
                                            int heads, paif_stream_t stream) {
  return paif_sr_attention_split_fwd(q, kv, out, lse, B, N, Nk, C, heads, 1, stream);
}

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
