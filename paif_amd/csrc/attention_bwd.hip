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

int paif_sr_attention_bwd_input(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C, int heads,
                                paif_stream_t stream) {
  PAIF_REQUIRE(q && kv && o && dout && lse && delta && dq && dkv && dkv_partial, PAIF_EINVAL, "sr_attention_bwd: null pointer");
  PAIF_REQUIRE(B > 0 && N > 0 && Nk > 0 && heads > 0 && C % heads == 0, PAIF_EINVAL, "sr_attention_bwd: bad shape");
  const int D = C / heads;
  PAIF_REQUIRE(D == 64 || D == 32, PAIF_ENOSUP, "sr_attention_bwd: head dim %d not built", D);
  PAIF_REQUIRE(Nk <= 320, PAIF_ENOSUP, "sr_attention_bwd: Nk=%d > 320 keys not built", Nk);
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
  if (D == 64) {
    if (lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) { paif::set_error("sr_attention_bwd: LDS %zu: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, dim3((N + 255) / 256, heads, B), dim3(512), lds_bytes, st, a);
    PAIF_LAUNCH_CHECK("sr_attention_bwd(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, dim3(a.nchunk * ngroups, heads, B), dim3(64 * wpb), 0, st, a);
  } else {
    if (lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<32>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) { paif::set_error("sr_attention_bwd: LDS %zu: %s", lds_bytes, hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, dim3((N + 255) / 256, heads, B), dim3(512), lds_bytes, st, a);
    PAIF_LAUNCH_CHECK("sr_attention_bwd(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, dim3(a.nchunk * ngroups, heads, B), dim3(64 * wpb), 0, st, a);
  }
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
