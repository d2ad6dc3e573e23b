// Backward of the fused spatial-reduction attention (attention.hip) w.r.t. q and kv, for the PGD loop.
// With P = softmax(scale * Q K^T) (recomputed from the saved log-sum-exp), O = P V:
//   delta_i = dO_i . O_i ;  dP = dO V^T ;  dS = P o (dP - delta) * scale
//   dQ = dS K ;  dK = dS^T Q ;  dV = P^T dO
// Two fp32-MFMA kernels, each in the orientation whose accumulators feed the next product directly
// (no lane movement, no score matrix in HBM):
//   dq kernel  -- "query on the lane" (as the forward): S^T and dP^T tiles, then dQ^T += K^T . dS^T with the
//                 dS^T accumulator registers as the B operand.  Also writes delta.
//   dkv kernel -- "key on the lane": one wave per 32-key tile keeps its K and V fragments in registers and
//                 walks a chunk of query tiles staged in LDS: S and dP tiles, then dV^T += dO^T . P and
//                 dK^T += Q^T . dS with the accumulator registers as B operands.  Partial dK/dV per query
//                 chunk go to a slab that a fixed-order reduction sums (deterministic, no float atomics).
#include <type_traits>

#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnBwdArgs {
  const float* q; const float* kv; const float* o; const float* dout; const float* lse;
  float* delta; float* dq; float* dkv_partial;
  int B, N, Nk, C, heads, chunk_tiles, nchunk;
  float scale;
};

// swizzled row-major [rows][D] image: 16-byte chunk c of row r lives at chunk (c ^ (r & (CH-1)))
template <int D>
__device__ __forceinline__ int swz(int row, int chunk) { return row * D + ((chunk ^ (row & (D / 4 - 1))) << 2); }

// ---------------------------------------------------------------------------------------------
template <int D>
// 8 waves (256 queries) per workgroup, like the forward kernel: two waves per SIMD, K/V staged once per 256 queries
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnBwdArgs a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int CH = D / 4, NO = D / 8, DT = D / 32;
  float* Ks = lds;
  float* Vs = lds + (size_t)a.Nk * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk;
  const float* kvb = a.kv + (size_t)b * Nk * 2 * C + hd * D;
  for (int idx = tid; idx < Nk * CH; idx += 512) {
    const int key = idx / CH, c = idx - key * CH;
    *reinterpret_cast<float4*>(Ks + swz<D>(key, c)) = *reinterpret_cast<const float4*>(kvb + (size_t)key * 2 * C + c * 4);
    *reinterpret_cast<float4*>(Vs + swz<D>(key, c)) = *reinterpret_cast<const float4*>(kvb + (size_t)key * 2 * C + C + c * 4);
  }
  __syncthreads();

  const int q0 = blockIdx.x * 256 + wave * 32;
  if (q0 >= a.N) return;
  const int qi = min(q0 + p, a.N - 1);
  const size_t rowoff = ((size_t)b * a.N + qi) * C + hd * D;
  float4 qf[NO], dof[NO];
  float dpart = 0.f;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    qf[o] = *reinterpret_cast<const float4*>(a.q + rowoff + 8 * o + 4 * h);
    dof[o] = *reinterpret_cast<const float4*>(a.dout + rowoff + 8 * o + 4 * h);
    const float4 of = *reinterpret_cast<const float4*>(a.o + rowoff + 8 * o + 4 * h);
    dpart += (dof[o].x * of.x + dof[o].y * of.y) + (dof[o].z * of.z + dof[o].w * of.w);
  }
  const float delta = dpart + __shfl_xor(dpart, 32);
  const size_t stat = ((size_t)b * a.heads + hd) * a.N + qi;
  const float lse = a.lse[stat];
  if (h == 0 && q0 + p < a.N) a.delta[stat] = delta;

  f32x16 dqacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[t][r] = 0.f;

  const int ntile = (Nk + 31) / 32;
  for (int t = 0; t < ntile; ++t) {
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
    const int key_a = min(t * 32 + p, Nk - 1);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float4 kf = *reinterpret_cast<const float4*>(Ks + swz<D>(key_a, 2 * o + h));
      const float4 vf = *reinterpret_cast<const float4*>(Vs + swz<D>(key_a, 2 * o + h));
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[o].x, st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, dof[o].x, dp, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[o].y, st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, dof[o].y, dp, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[o].z, st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, dof[o].z, dp, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[o].w, st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, dof[o].w, dp, 0, 0, 0);
    }
    // dS^T = P^T o (dP^T - delta) * scale     (register r <-> key t*32 + (r&3)+8(r>>2)+4h)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float pr = (key < Nk) ? expf(st[r] * a.scale - lse) : 0.f;
      st[r] = pr * (dp[r] - delta) * a.scale;
    }
    // dQ^T[dim][query] += K^T[dim][key] . dS^T[key][query]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, Nk - 1);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int dim = 32 * dt + p;
        dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[swz<D>(key, dim >> 2) + (dim & 3)], st[r], dqacc[dt], 0, 0, 0);
      }
    }
  }
  if (q0 + p < a.N) {
    float* orow = a.dq + ((size_t)b * a.N + q0 + p) * C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * h) =
            make_float4(dqacc[dt][4 * g], dqacc[dt][4 * g + 1], dqacc[dt][4 * g + 2], dqacc[dt][4 * g + 3]);
  }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 form of the dq kernel (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate): the exact kernel
// runs at the fp32 matrix-pipe roof (6*N*Nk*D FLOP at 157 TF).  Same orientation.  The three A operands -- K and V row-major,
// K transposed for dQ^T += K^T . dS^T -- are staged as bf16 hi | lo for a CHUNK of keys at a time (all three for 300 keys x 64
// dims do not fit LDS; the backward has no online softmax, so key chunks are independent).  K^T uses the forward kernel's
// key-slot permutation: the dS^T accumulator registers are the B operand as they stand.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bbf16x8 __attribute__((ext_vector_type(8)));
using paif::splitN;
using paif::mfma_pieces;

// NP: bf16 pieces per operand (2: "bf16x3", three products; 3: "bf16x6", six products -- fp32-level, the attack loops' arithmetic)
template <int D, int NP>
struct DqCfg {
  static constexpr int KC = D == 64 ? (NP == 3 ? 128 : 160) : (NP == 3 ? 160 : 320);   // keys per LDS chunk
  static constexpr int KREC = D * 2 * NP;                  // row-major record: NP pieces of D bf16
  static constexpr int TREC = KC * 2 * NP + 16;            // transposed record: NP x KC slots | pad
  static constexpr size_t lds_bytes = (size_t)2 * KC * KREC + (size_t)D * TREC;
  static_assert(lds_bytes <= 160 * 1024, "dq(split): key chunk does not fit LDS");
};

template <int D, int NP, int PF = 0>
__device__ __forceinline__ void attn_bwd_dq_split_body(const AttnBwdArgs& a) {
  extern __shared__ __align__(16) char ldsc[];
  constexpr int NT = 512;
  constexpr int NO = D / 16, DT = D / 32;
  constexpr int KC = DqCfg<D, NP>::KC, KREC = DqCfg<D, NP>::KREC, TREC = DqCfg<D, NP>::TREC;
  char* Kr = ldsc;
  char* Vr = ldsc + KC * KREC;
  char* Kt = ldsc + 2 * KC * KREC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk;
  const float* kvb = a.kv + (size_t)b * Nk * 2 * C + hd * D;

  const int q0 = blockIdx.x * (NT / 2) + wave * 32;
  const bool wave_live = q0 < a.N;          // dead waves still stage and hit the barriers
  const int qi = min(q0 + p, a.N - 1);
  const size_t rowoff = ((size_t)b * a.N + qi) * C + hd * D;
  bbf16x8 qp[NO][NP], dop[NO][NP];
  float dpart = 0.f;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    float q8[8], d8[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 qv = *reinterpret_cast<const float4*>(a.q + rowoff + 16 * o + 8 * h + 4 * j);
      const float4 dv = *reinterpret_cast<const float4*>(a.dout + rowoff + 16 * o + 8 * h + 4 * j);
      const float4 ov = *reinterpret_cast<const float4*>(a.o + rowoff + 16 * o + 8 * h + 4 * j);
      q8[4 * j] = qv.x; q8[4 * j + 1] = qv.y; q8[4 * j + 2] = qv.z; q8[4 * j + 3] = qv.w;
      d8[4 * j] = dv.x; d8[4 * j + 1] = dv.y; d8[4 * j + 2] = dv.z; d8[4 * j + 3] = dv.w;
      dpart += (dv.x * ov.x + dv.y * ov.y) + (dv.z * ov.z + dv.w * ov.w);
    }
    splitN<NP, PF>(q8, qp[o]);
    splitN<NP, PF>(d8, dop[o]);
  }
  const float delta = dpart + __shfl_xor(dpart, 32);
  const size_t stat = ((size_t)b * a.heads + hd) * a.N + qi;
  const float lse = a.lse[stat];
  if (wave_live && h == 0 && q0 + p < a.N) a.delta[stat] = delta;

  f32x16 dqacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[t][r] = 0.f;

  for (int c0 = 0; c0 < Nk; c0 += KC) {
    const int kc = min(KC, Nk - c0);
    const int kcp = (kc + 31) & ~31;
    if (c0 > 0) __syncthreads();           // previous chunk fully consumed
    // ---- stage the chunk: K and V row-major (split), K transposed (split, key slots permuted, zero beyond the last key) ----
    for (int idx = tid; idx < kcp * (D / 8); idx += NT) {
      const int key = idx / (D / 8), c = idx - key * (D / 8);
      const bool live = key < kc;
      const float* krow = kvb + (size_t)(c0 + min(key, kc - 1)) * 2 * C + c * 8;
      const float4 k0 = *reinterpret_cast<const float4*>(krow), k1 = *reinterpret_cast<const float4*>(krow + 4);
      const float4 v0 = *reinterpret_cast<const float4*>(krow + C), v1 = *reinterpret_cast<const float4*>(krow + C + 4);
      const float k8[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      const float v8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      bbf16x8 kp[NP], vp[NP];
      splitN<NP, PF>(k8, kp);
      splitN<NP, PF>(v8, vp);
      if (live) {
        const int sw = key & (D / 8 - 1);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          *reinterpret_cast<bbf16x8*>(Kr + key * KREC + q * D * 2 + ((c ^ sw) << 4)) = kp[q];
          *reinterpret_cast<bbf16x8*>(Vr + key * KREC + q * D * 2 + ((c ^ sw) << 4)) = vp[q];
        }
      }
      const int kk = key & 31, tt = key >> 5;
      const int hh = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
      const int slot = tt * 32 + (r >> 3) * 16 + 8 * hh + (r & 7);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dim = c * 8 + i;
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<__bf16*>(Kt + dim * TREC + q * KC * 2 + slot * 2) = live ? kp[q][i] : (__bf16)0.f;
      }
    }
    __syncthreads();

    const int ntile = kcp >> 5;
    for (int t = 0; t < ntile; ++t) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
      const int key_a = min(t * 32 + p, kc - 1);
      const char* krow = Kr + key_a * KREC;
      const char* vrow = Vr + key_a * KREC;
      const int sw = key_a & (D / 8 - 1);
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int cc = ((2 * o + h) ^ sw) << 4;
        bbf16x8 kp[NP], vp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          kp[q] = *reinterpret_cast<const bbf16x8*>(krow + q * D * 2 + cc);
          vp[q] = *reinterpret_cast<const bbf16x8*>(vrow + q * D * 2 + cc);
        }
        mfma_pieces<NP, PF>(st, kp, qp[o]);
        mfma_pieces<NP, PF>(dp, vp, dop[o]);
      }
      // dS^T = P^T o (dP^T - delta) * scale     (register r <-> key c0 + t*32 + (r&3)+8(r>>2)+4h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float pr = (key < kc) ? expf(st[r] * a.scale - lse) : 0.f;
        st[r] = pr * (dp[r] - delta) * a.scale;
      }
      // dQ^T[dim][query] += K^T[dim][key slot] . dS^T[key slot][query]: two K=16 steps, step s2 = registers 8*s2 .. 8*s2+7
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float d8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d8[i] = st[8 * s2 + i];
        bbf16x8 dsp[NP];
        splitN<NP, PF>(d8, dsp);
        const int soff = (t * 32 + s2 * 16 + 8 * h) * 2;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int dim = 32 * dt + p;
          bbf16x8 tp[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) tp[q] = *reinterpret_cast<const bbf16x8*>(Kt + dim * TREC + q * KC * 2 + soff);
          mfma_pieces<NP, PF>(dqacc[dt], tp, dsp);
        }
      }
    }
  }
  if (wave_live && q0 + p < a.N) {
    float* orow = a.dq + ((size_t)b * a.N + q0 + p) * C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * h) =
            make_float4(dqacc[dt][4 * g], dqacc[dt][4 * g + 1], dqacc[dt][4 * g + 2], dqacc[dt][4 * g + 3]);
  }
}

// one kernel name per arithmetic (see attention.hip)
template <int D> __global__ __launch_bounds__(512) void attn_bwd_dq_bf16x3_kernel(AttnBwdArgs a) { attn_bwd_dq_split_body<D, 2, 0>(a); }
template <int D> __global__ __launch_bounds__(512) void attn_bwd_dq_bf16x6_kernel(AttnBwdArgs a) { attn_bwd_dq_split_body<D, 3, 0>(a); }
template <int D> __global__ __launch_bounds__(512) void attn_bwd_dq_f16x3_kernel(AttnBwdArgs a) { attn_bwd_dq_split_body<D, 2, 1>(a); }

template <int D, int NP, int PF = 0>
int launch_dq_split(const AttnBwdArgs& a, hipStream_t st) {
  constexpr size_t lds_bytes = DqCfg<D, NP>::lds_bytes;
  auto kern = [] {
    if constexpr (PF == 1) return &attn_bwd_dq_f16x3_kernel<D>;
    else if constexpr (NP == 3) return &attn_bwd_dq_bf16x6_kernel<D>;
    else return &attn_bwd_dq_bf16x3_kernel<D>;
  }();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) { paif::set_error("sr_attention_bwd(dq split): LDS %zu: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(kern, dim3((a.N + 255) / 256, a.heads, a.B), dim3(512), lds_bytes, st, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// grid (nchunk * ngroups, heads, B); one wave per key tile, KG_TILES key tiles (waves) per workgroup
constexpr int KG_TILES = 5;
template <int D>
__global__ __launch_bounds__(64 * KG_TILES) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
  constexpr int CH = D / 4, NO = D / 8, DT = D / 32;
  __shared__ __align__(16) float Qs[32 * D];
  __shared__ __align__(16) float Ds[32 * D];
  __shared__ float stat[64];  // lse[32], delta[32]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int nthreads = blockDim.x;
  const int ngroups = ((a.Nk + 31) / 32 + KG_TILES - 1) / KG_TILES;
  const int chunk = blockIdx.x / ngroups, kg = blockIdx.x - chunk * ngroups;
  const int wave = kg * KG_TILES + (tid >> 6);   // key tile of this wave (may be past the last tile: idle but barrier-safe)
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk, N = a.N;

  // this wave's key tile: K and V fragments as B operands (lane (h, j) = key j, dims 8o+4h+i)
  const int key_b = min(wave * 32 + p, Nk - 1);
  const float* krow = a.kv + ((size_t)b * Nk + key_b) * 2 * C + hd * D;
  float4 kf[NO], vf[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    kf[o] = *reinterpret_cast<const float4*>(krow + 8 * o + 4 * h);
    vf[o] = *reinterpret_cast<const float4*>(krow + C + 8 * o + 4 * h);
  }
  const bool keyvalid = wave * 32 + p < Nk;

  f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[t][r] = 0.f; dvacc[t][r] = 0.f; }

  const int tile0 = chunk * a.chunk_tiles;
  const int ntq = (N + 31) / 32;
  for (int qt = tile0; qt < min(ntq, tile0 + a.chunk_tiles); ++qt) {
    __syncthreads();  // previous tile consumed
    for (int idx = tid; idx < 32 * CH; idx += nthreads) {
      const int row = idx / CH, c = idx - row * CH;
      const int qi = min(qt * 32 + row, N - 1);
      const size_t off = ((size_t)b * N + qi) * C + hd * D + c * 4;
      *reinterpret_cast<float4*>(Qs + swz<D>(row, c)) = *reinterpret_cast<const float4*>(a.q + off);
      *reinterpret_cast<float4*>(Ds + swz<D>(row, c)) = *reinterpret_cast<const float4*>(a.dout + off);
    }
    if (tid < 64) {
      const int row = tid & 31;
      const int qi = qt * 32 + row;
      const size_t so = ((size_t)b * a.heads + hd) * N + min(qi, N - 1);
      // invalid query rows get lse = +inf -> P = exp(-inf) = 0 -> no contribution
      stat[tid] = tid < 32 ? (qi < N ? a.lse[so] : INFINITY) : a.delta[so];
    }
    __syncthreads();

    // S = Q_tile . K^T and dP = dO_tile . V^T : rows (registers) = queries, lanes = keys
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float4 qa = *reinterpret_cast<const float4*>(Qs + swz<D>(p, 2 * o + h));
      const float4 da = *reinterpret_cast<const float4*>(Ds + swz<D>(p, 2 * o + h));
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.x, kf[o].x, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.x, vf[o].x, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.y, kf[o].y, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.y, vf[o].y, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.z, kf[o].z, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.z, vf[o].z, dp, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa.w, kf[o].w, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da.w, vf[o].w, dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = (r & 3) + 8 * (r >> 2) + 4 * h;   // query row of this register
      const float pr = keyvalid ? expf(s[r] * a.scale - stat[qr]) : 0.f;
      s[r] = pr;                                        // P
      dp[r] = pr * (dp[r] - stat[32 + qr]) * a.scale;   // dS
    }
    // dV^T[dim][key] += dO^T[dim][query] . P[query][key];  dK^T[dim][key] += Q^T[dim][query] . dS[query][key]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int dim = 32 * dt + p;
        const int off = swz<D>(qr, dim >> 2) + (dim & 3);
        dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[off], s[r], dvacc[dt], 0, 0, 0);
        dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[off], dp[r], dkacc[dt], 0, 0, 0);
      }
    }
  }

  // partial slab [nchunk][B][Nk][2C]: lane (h, j = key) holds dims 32*dt + 8*g + 4*h + (0..3)
  if (keyvalid) {
    float* base = a.dkv_partial + (((size_t)chunk * a.B + b) * Nk + wave * 32 + p) * 2 * C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(base + 32 * dt + 8 * g + 4 * h) =
            make_float4(dkacc[dt][4 * g], dkacc[dt][4 * g + 1], dkacc[dt][4 * g + 2], dkacc[dt][4 * g + 3]);
        *reinterpret_cast<float4*>(base + C + 32 * dt + 8 * g + 4 * h) =
            make_float4(dvacc[dt][4 * g], dvacc[dt][4 * g + 1], dvacc[dt][4 * g + 2], dvacc[dt][4 * g + 3]);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 form of the dkv kernel (same orientation: one wave per 32-key tile, K / V fragments in registers as B operands,
// query tiles staged in LDS).  A query tile is staged four ways, all as bf16 hi | lo: Q and dO row-major (A operands of
// S = Q . K^T and dP = dO . V^T) and transposed with the query-slot permutation (A operands of dK^T += Q^T . dS and
// dV^T += dO^T . P), so that the P / dS accumulator registers are the B operands as they stand.
// ---------------------------------------------------------------------------------------------
template <int D, int NP, int PF = 0>
__device__ __forceinline__ void attn_bwd_dkv_split_body(const AttnBwdArgs& a) {
  constexpr int NO = D / 16, DT = D / 32;
  constexpr int KREC = D * 2 * NP;         // row-major record: NP pieces of D bf16
  constexpr int TREC = 32 * 2 * NP + 16;   // transposed record: NP x 32 slots | pad
  __shared__ __align__(16) char Qr[32 * KREC];
  __shared__ __align__(16) char Dr[32 * KREC];
  __shared__ __align__(16) char Qt[D * TREC];
  __shared__ __align__(16) char Dt[D * TREC];
  __shared__ float stat[64];  // lse[32], delta[32]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int nthreads = blockDim.x;
  const int ngroups = ((a.Nk + 31) / 32 + KG_TILES - 1) / KG_TILES;
  const int chunk = blockIdx.x / ngroups, kg = blockIdx.x - chunk * ngroups;
  const int wave = kg * KG_TILES + (tid >> 6);   // key tile of this wave (may be past the last tile: idle but barrier-safe)
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, Nk = a.Nk, N = a.N;

  // this wave's key tile: K and V fragments as B operands (lane (h, j) = key j, dims 16o + 8h + i), split once
  const int key_b = min(wave * 32 + p, Nk - 1);
  const float* krow = a.kv + ((size_t)b * Nk + key_b) * 2 * C + hd * D;
  bbf16x8 kp[NO][NP], vp[NO][NP];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    float k8[8], v8[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 kv4 = *reinterpret_cast<const float4*>(krow + 16 * o + 8 * h + 4 * j);
      const float4 vv4 = *reinterpret_cast<const float4*>(krow + C + 16 * o + 8 * h + 4 * j);
      k8[4 * j] = kv4.x; k8[4 * j + 1] = kv4.y; k8[4 * j + 2] = kv4.z; k8[4 * j + 3] = kv4.w;
      v8[4 * j] = vv4.x; v8[4 * j + 1] = vv4.y; v8[4 * j + 2] = vv4.z; v8[4 * j + 3] = vv4.w;
    }
    splitN<NP, PF>(k8, kp[o]);
    splitN<NP, PF>(v8, vp[o]);
  }
  const bool keyvalid = wave * 32 + p < Nk;

  f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[t][r] = 0.f; dvacc[t][r] = 0.f; }

  const int tile0 = chunk * a.chunk_tiles;
  const int ntq = (N + 31) / 32;
  for (int qt = tile0; qt < min(ntq, tile0 + a.chunk_tiles); ++qt) {
    __syncthreads();  // previous tile consumed
    for (int idx = tid; idx < 2 * 32 * (D / 8); idx += nthreads) {
      const int which = idx / (32 * (D / 8));            // 0: Q, 1: dO
      const int rem = idx - which * (32 * (D / 8));
      const int row = rem / (D / 8), c = rem - row * (D / 8);
      const int qi = min(qt * 32 + row, N - 1);
      const float* src = (which ? a.dout : a.q) + ((size_t)b * N + qi) * C + hd * D + c * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
      const float x8[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      bbf16x8 xp[NP];
      splitN<NP, PF>(x8, xp);
      char* rm = which ? Dr : Qr;
      char* tr = which ? Dt : Qt;
      const int sw = row & (D / 8 - 1);
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bbf16x8*>(rm + row * KREC + q * D * 2 + ((c ^ sw) << 4)) = xp[q];
      const int hh = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
      const int slot = (r >> 3) * 16 + 8 * hh + (r & 7);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dim = c * 8 + i;
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<__bf16*>(tr + dim * TREC + q * 64 + slot * 2) = xp[q][i];
      }
    }
    if (tid < 64) {
      const int row = tid & 31;
      const int qi = qt * 32 + row;
      const size_t so = ((size_t)b * a.heads + hd) * N + min(qi, N - 1);
      // invalid query rows get lse = +inf -> P = exp(-inf) = 0 -> no contribution
      stat[tid] = tid < 32 ? (qi < N ? a.lse[so] : INFINITY) : a.delta[so];
    }
    __syncthreads();

    // S = Q_tile . K^T and dP = dO_tile . V^T : rows (registers) = queries, lanes = keys
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const int sw = p & (D / 8 - 1);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const int cc = ((2 * o + h) ^ sw) << 4;
      bbf16x8 qp[NP], dop[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        qp[q] = *reinterpret_cast<const bbf16x8*>(Qr + p * KREC + q * D * 2 + cc);
        dop[q] = *reinterpret_cast<const bbf16x8*>(Dr + p * KREC + q * D * 2 + cc);
      }
      mfma_pieces<NP, PF>(s, qp, kp[o]);
      mfma_pieces<NP, PF>(dp, dop, vp[o]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = (r & 3) + 8 * (r >> 2) + 4 * h;   // query row of this register
      const float pr = keyvalid ? expf(s[r] * a.scale - stat[qr]) : 0.f;
      s[r] = pr;                                        // P
      dp[r] = pr * (dp[r] - stat[32 + qr]) * a.scale;   // dS
    }
    // dV^T[dim][key] += dO^T[dim][query slot] . P[query slot][key];  dK^T[dim][key] += Q^T[dim][query slot] . dS[query slot][key]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float p8[8], d8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { p8[i] = s[8 * s2 + i]; d8[i] = dp[8 * s2 + i]; }
      bbf16x8 pp[NP], dsp[NP];
      splitN<NP, PF>(p8, pp);
      splitN<NP, PF>(d8, dsp);
      const int soff = (s2 * 16 + 8 * h) * 2;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int dim = 32 * dt + p;
        bbf16x8 op[NP], tp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          op[q] = *reinterpret_cast<const bbf16x8*>(Dt + dim * TREC + q * 64 + soff);
          tp[q] = *reinterpret_cast<const bbf16x8*>(Qt + dim * TREC + q * 64 + soff);
        }
        mfma_pieces<NP, PF>(dvacc[dt], op, pp);
        mfma_pieces<NP, PF>(dkacc[dt], tp, dsp);
      }
    }
  }

  // partial slab [nchunk][B][Nk][2C]: lane (h, j = key) holds dims 32*dt + 8*g + 4*h + (0..3)
  if (keyvalid) {
    float* base = a.dkv_partial + (((size_t)chunk * a.B + b) * Nk + wave * 32 + p) * 2 * C + hd * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(base + 32 * dt + 8 * g + 4 * h) =
            make_float4(dkacc[dt][4 * g], dkacc[dt][4 * g + 1], dkacc[dt][4 * g + 2], dkacc[dt][4 * g + 3]);
        *reinterpret_cast<float4*>(base + C + 32 * dt + 8 * g + 4 * h) =
            make_float4(dvacc[dt][4 * g], dvacc[dt][4 * g + 1], dvacc[dt][4 * g + 2], dvacc[dt][4 * g + 3]);
      }
  }
}

template <int D> __global__ __launch_bounds__(64 * KG_TILES) void attn_bwd_dkv_bf16x3_kernel(AttnBwdArgs a) { attn_bwd_dkv_split_body<D, 2, 0>(a); }
template <int D> __global__ __launch_bounds__(64 * KG_TILES) void attn_bwd_dkv_bf16x6_kernel(AttnBwdArgs a) { attn_bwd_dkv_split_body<D, 3, 0>(a); }
template <int D> __global__ __launch_bounds__(64 * KG_TILES) void attn_bwd_dkv_f16x3_kernel(AttnBwdArgs a) { attn_bwd_dkv_split_body<D, 2, 1>(a); }

__global__ void reduce_slabs_kernel(const float4* __restrict__ partial, float4* __restrict__ out, size_t n4, int nchunk) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 s = partial[i];
    for (int c = 1; c < nchunk; ++c) {
      const float4 v = partial[(size_t)c * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = s;
  }
}

}  // namespace

extern "C" {

// number of query chunks the dkv kernel uses for (B, N, heads): sizes the partial slab
int paif_sr_attention_bwd_chunks(int B, int N, int heads) {
  const int ntq = (N + 31) / 32;
  int nchunk = (512 + B * heads - 1) / (B * heads);  // x key groups (2 at Nk = 300) -> ~4 workgroups per CU
  if (nchunk > ntq) nchunk = ntq;
  if (nchunk < 1) nchunk = 1;
  const int chunk_tiles = (ntq + nchunk - 1) / nchunk;
  return (ntq + chunk_tiles - 1) / chunk_tiles;
}

int paif_sr_attention_bwd_input_p(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                  float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C, int heads,
                                  int precision, paif_stream_t stream);

int paif_sr_attention_bwd_input(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C, int heads,
                                paif_stream_t stream) {
  return paif_sr_attention_bwd_input_p(q, kv, o, dout, lse, delta, dq, dkv, dkv_partial, B, N, Nk, C, heads, 0, stream);
}

int paif_sr_attention_bwd_input_p(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                  float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C, int heads,
                                  int precision, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "sr_attention_bwd: precision=%d", precision);
  PAIF_REQUIRE(q && kv && o && dout && lse && delta && dq && dkv && dkv_partial, PAIF_EINVAL, "sr_attention_bwd: null pointer");
  PAIF_REQUIRE(B > 0 && N > 0 && Nk > 0 && heads > 0 && C % heads == 0, PAIF_EINVAL, "sr_attention_bwd: bad shape");
  const int D = C / heads;
  PAIF_REQUIRE(D == 64 || D == 32, PAIF_ENOSUP, "sr_attention_bwd: head dim %d not built", D);
  PAIF_REQUIRE(precision != 0 || Nk <= 320, PAIF_ENOSUP, "sr_attention_bwd: Nk=%d > 320 keys not built for the exact kernels (the split forms chunk the keys)", Nk);
  AttnBwdArgs a;
  a.q = q; a.kv = kv; a.o = o; a.dout = dout; a.lse = lse; a.delta = delta; a.dq = dq; a.dkv_partial = dkv_partial;
  a.B = B; a.N = N; a.Nk = Nk; a.C = C; a.heads = heads; a.scale = 1.0f / sqrtf((float)D);
  const int ntq = (N + 31) / 32;
  a.nchunk = paif_sr_attention_bwd_chunks(B, N, heads);
  a.chunk_tiles = (ntq + a.nchunk - 1) / a.nchunk;
  hipStream_t st = paif::as_stream(stream);
  const size_t lds_bytes = (size_t)Nk * D * 8;
  const int ntile = (Nk + 31) / 32;
  const int ngroups = (ntile + KG_TILES - 1) / KG_TILES;
  const int wpb = ntile < KG_TILES ? ntile : KG_TILES;
  const dim3 kvgrid(a.nchunk * ngroups, heads, B), kvblk(64 * wpb);
  auto run = [&](auto dtag) -> int {
    constexpr int DD = decltype(dtag)::value;
    if (precision == 0) {
      if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<DD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { paif::set_error("sr_attention_bwd: LDS %zu: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
      }
      hipLaunchKernelGGL(attn_bwd_dq_kernel<DD>, dim3((N + 255) / 256, heads, B), dim3(512), lds_bytes, st, a);
    } else {
      const int rc = precision == 3 ? launch_dq_split<DD, 3>(a, st) : precision == 6 ? launch_dq_split<DD, 2, 1>(a, st) : launch_dq_split<DD, 2>(a, st);
      if (rc) return rc;
    }
    PAIF_LAUNCH_CHECK("sr_attention_bwd(dq)");
    if (precision == 3) hipLaunchKernelGGL(attn_bwd_dkv_bf16x6_kernel<DD>, kvgrid, kvblk, 0, st, a);
    else if (precision == 6) hipLaunchKernelGGL(attn_bwd_dkv_f16x3_kernel<DD>, kvgrid, kvblk, 0, st, a);
    else if (precision == 1) hipLaunchKernelGGL(attn_bwd_dkv_bf16x3_kernel<DD>, kvgrid, kvblk, 0, st, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<DD>, kvgrid, kvblk, 0, st, a);
    return 0;
  };
  const int rc = D == 64 ? run(std::integral_constant<int, 64>{}) : run(std::integral_constant<int, 32>{});
  if (rc) return rc;
  PAIF_LAUNCH_CHECK("sr_attention_bwd(dkv)");
  const size_t n4 = (size_t)B * Nk * 2 * C / 4;
  size_t g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)g), dim3(256), 0, st, reinterpret_cast<const float4*>(dkv_partial),
                     reinterpret_cast<float4*>(dkv), n4, a.nchunk);
  PAIF_LAUNCH_CHECK("sr_attention_bwd(reduce)");
  return 0;
}

}  // extern "C"
