// fp32 GEMM on the gfx950 matrix cores for the SegFormer linears / 1x1 convs / im2col'ed convs:
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )         (torch nn.Linear / F.linear weight layout)
// v_mfma_f32_32x32x2_f32: exact f32 (k-ordered fmaf chain) at the fp32 vector rate.
//
//   workgroup tile 128 (M) x 64 (N) x 32 (K), 4 waves, wave w owns rows [32w, 32w+32) x all 64 columns
//   (2 accumulator tiles).  A and W tiles staged in LDS with row stride 36 dwords (conflict-free
//   ds_read_b128, see conv_mfma.hip); one b128 A-read + two b128 W-reads feed 8 MFMAs.
//   D layout: lane (h, n) holds column n of rows (r&3)+8(r>>2)+4h -> each store writes 128-B row segments.
//   Epilogue: y = acc*scale[n] + shift[n] (bias or folded BN; NULL -> 1/0), GELU(erf)/ReLU, + res[m, n].
//   K must be a multiple of 32 (callers pad); M, N arbitrary (masked).
//
// Replaces nn.Linear in core/mix_transformer.py (Mlp :22-25, Attention :66-69, :74 sr conv via im2col),
// core/segformer_head.py MLP.proj (:19), linear_fuse 1x1 conv + BN + ReLU (:50-55), linear_pred (:57),
// OverlapPatchEmbed.proj via im2col (core/mix_transformer.py:168-169).
#include <stdint.h>

#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 64, BK = 32, LDS_STRIDE = BK + 4;

struct GemmArgs {
  const float* A; const float* W; const float* scale; const float* shift; const float* res; float* C;
  const float* a_mask; const float* a_scale;
  int M, N, K, lda, ldc, ldres, act;
  int tilesN, nblk;
  int kper;            // k extent of one split (= K without split-K); blockIdx.y selects the split
  float* partial;      // split-K: raw accumulators go to partial[split][M][N] (no epilogue); else NULL
  int wide;            // 1: N, ldc, ldres multiples of 4 and C / res 16-byte aligned -> float4 epilogue through LDS
  // GATHER form (paif_gemm_conv_fwd): A is an NHWC map [gB, gH, gW, gC]; row m = output pixel (b, oy, ox) of a gk x gk conv with
  // stride gs and padding gpad, column index (ky * gk + kx) * gC + c -- the im2col matrix, never materialised
  int gk, gs, gpad, gH, gW, gC, gOH, gOW;
  // SCATTER form (paif_gemm_col2im_fwd): C is an NHWC map [B, sH, sW, sC]; row m = patch (b, oy, ox) of a NON-OVERLAPPING sk x sk / stride sk
  // tiling, column n = (ky * sk + kx) * sC + c -- the col2im of the dgrad GEMM's output, written by the epilogue (0 = off)
  int sk, sH, sW, sC, sOH, sOW;
};

// address of output element (m, n): row-major, or its pixel / channel in the SCATTER form (n .. n + 3 share a tap: sC % 4 == 0)
__device__ __forceinline__ float* c_addr(const GemmArgs& a, int m, int n) {
  if (!a.sk) return a.C + (size_t)m * a.ldc + n;
  const int ox = m % a.sOW, t = m / a.sOW, oy = t % a.sOH, b = t / a.sOH;
  const int tap = n / a.sC, cc = n - tap * a.sC, ky = tap / a.sk, kx = tap - ky * a.sk;
  return a.C + ((size_t)(b * a.sH + oy * a.sk + ky) * a.sW + ox * a.sk + kx) * a.sC + cc;
}

__device__ __forceinline__ float gelu_erf(float x) { return paif::gelu_erf_fast(x); }   // paif_common.h: x * Phi(x), one polynomial + v_exp_f32

// split-K slab store / fused epilogue of one workgroup tile (shared by all three kernels).
// Wide form: a row-per-lane dword epilogue is store-ISSUE bound (32 store instructions of 256 B per wave; the stage-1 GEMMs
// wrote their 315 MB outputs at 1.5 TB/s).  Each wave parks one 32x32 accumulator tile at a time in its private slice of
// `park` (the A-operand LDS of the finished k loop, 36-float rows) and re-reads it as float4 per (row, column quad): scale /
// shift / residual become float4 loads and a store instruction writes eight full 128-byte row segments.
// Must be called after a __syncthreads() that retires every read of the staged operands.
__device__ __forceinline__ void gemm_finish(const GemmArgs& a, const f32x16 (&acc)[2], float* park, int m0, int n0, int wave, int h, int p) {
  if (a.partial) {   // split-K: raw partial sums; scale / activation / residual are applied by the reduction pass
    float* slab = a.partial + (size_t)blockIdx.y * a.M * a.N;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = n0 + 32 * t + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < a.M && n < a.N) slab[(size_t)m * a.N + n] = acc[t][r];
      }
    }
    return;
  }
  if (a.wide) {      // launch-uniform
    float* ep = park + wave * (32 * 36);
    const int lane = h * 32 + p;
    const int c4 = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + p] = acc[t][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave wrote and reads: LDS operations complete in order
      const int n = n0 + 32 * t + 4 * c4;
      const bool nok = n < a.N;   // N % 4 == 0: a quad is all in or all out
      const int nc = nok ? n : 0;
      const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 rv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min(m0 + wave * 32 + it * 8 + rsub, a.M - 1);   // unconditional on a clamped row
        rv[it] = a.res ? *reinterpret_cast<const float4*>(a.res + (size_t)m * a.ldres + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        const int m = m0 + wave * 32 + row;
        float4 v = *reinterpret_cast<const float4*>(ep + row * 36 + 4 * c4);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (a.act == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (a.act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.res) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
        if (nok && m < a.M) *reinterpret_cast<float4*>(c_addr(a, m, n)) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile's reads are done before the next one is parked
    }
    return;
  }
  // ---- scalar epilogue (odd N / leading dimensions: e.g. the 9-class prediction layer) ----
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + 32 * t + p;
    if (n >= a.N) continue;
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float sh = a.shift ? a.shift[n] : 0.f;
    float rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      rv[r] = (a.res && m < a.M) ? a.res[(size_t)m * a.ldres + n] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < a.M) {
        float v = acc[t][r] * sc + sh;
        if (a.act == 1) v = gelu_erf(v);
        else if (a.act == 2) v = fmaxf(v, 0.f);
        if (a.res) v += rv[r];
        *c_addr(a, m, n) = v;
      }
    }
  }
}

// Serial form (load -> wait -> LDS -> MFMA per k tile), kept for short k loops (K <= 160: at most 5 tiles): those GEMMs
// are bound by the A / C streams, what matters is that every resident workgroup has its loads out as early as possible,
// and the 88-register serial body measured 10-20 % faster there than the pipelined one (which wins 20-45 % at K >= 320).
// GATHER (paif_gemm_conv_fwd, exact arithmetic): A is gathered element-wise from an NHWC map with ANY channel count (the 3-channel
// input of OverlapPatchEmbed 1, core/mix_transformer.py:168: 7x7 taps, stride 4): column k = (ky * gk + kx) * gC + c decoded once per
// workgroup into an LDS table; columns >= gk*gk*gC (the padding up to K) and taps outside the map read as 0.
template <bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_f32_serial(GemmArgs a) {
  __shared__ __align__(16) float sA[BM * LDS_STRIDE];
  __shared__ __align__(16) float sW[BN * LDS_STRIDE];
  __shared__ int ktab[GATHER ? 160 : 1];   // (ky << 20) | (kx << 10) | c, or -1 for a padding column
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  if constexpr (GATHER) {
    if (tid < a.K) {
      const int kk = tid / a.gC, c = tid - kk * a.gC, ky = kk / a.gk, kx = kk - ky * a.gk;
      ktab[tid] = ky < a.gk ? (ky << 20) | (kx << 10) | c : -1;
    }
    __syncthreads();
  }
  // consecutive workgroups share the same rows of A (same m-tile, different n-tile) -> L2 reuse of A
  // XCD-aware tile order (round 4): the tilesN tiles that share 128 rows of A run on ONE XCD, so its L2 serves the re-reads
  // (round-robin dispatch put them on tilesN different XCDs: A came out of the Infinity Cache tilesN times).  Measured on the
  // large split-bf16 shapes: 436 -> 399 us (153600 x 256 x 1024), 106 -> 94 us (38400 x 512 x 512); configs[2]: within the box spread.
  const int bid = paif::xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % a.tilesN, tm = bid / a.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging map: A tile = 128 rows x 8 float4 -> 4 per thread; W tile = 64 rows x 8 float4 -> 2 per thread
  const int srow = tid >> 3, sq = tid & 7;
  const int abase = (wave * 32 + p) * LDS_STRIDE + 4 * h;
  const int wbase = p * LDS_STRIDE + 4 * h;

  int giy[GATHER ? 4 : 1], gix[GATHER ? 4 : 1];
  unsigned gpix[GATHER ? 4 : 1];
  if constexpr (GATHER) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + srow + 32 * i, a.M - 1);
      const int ox = m % a.gOW, t = m / a.gOW, oy = t % a.gOH, b = t / a.gOH;
      giy[i] = oy * a.gs - a.gpad; gix[i] = ox * a.gs - a.gpad;
      gpix[i] = (unsigned)b * (unsigned)(a.gH * a.gW);
    }
  }
  const int kbeg = blockIdx.y * a.kper, kend = kbeg + a.kper;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    float4 va[4], vw[2];
    if constexpr (GATHER) {
      int kt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) kt[j] = ktab[k0 + sq * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // unconditional load of a clamped address, then select (no load under a data-dependent branch)
          const int iy = giy[i] + (kt[j] >> 20), ix = gix[i] + ((kt[j] >> 10) & 1023);
          const bool ok = kt[j] >= 0 && (unsigned)iy < (unsigned)a.gH && (unsigned)ix < (unsigned)a.gW;
          const unsigned off = ok ? (gpix[i] + (unsigned)(iy * a.gW + ix)) * (unsigned)a.gC + (unsigned)(kt[j] & 1023) : 0u;
          const float v = a.A[off];
          e[j] = ok ? v : 0.f;
        }
        va[i] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
    if constexpr (!GATHER) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        va[i] = (m < a.M) ? *reinterpret_cast<const float4*>(a.A + (size_t)m * a.lda + k0 + sq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.a_mask && m < a.M) {  // dgrad prologue: ReLU mask and per-column scale (folded BatchNorm)
          const float4 mk = *reinterpret_cast<const float4*>(a.a_mask + (size_t)m * a.lda + k0 + sq * 4);
          va[i].x = mk.x > 0.f ? va[i].x : 0.f; va[i].y = mk.y > 0.f ? va[i].y : 0.f;
          va[i].z = mk.z > 0.f ? va[i].z : 0.f; va[i].w = mk.w > 0.f ? va[i].w : 0.f;
        }
        if (a.a_scale) {
          const float4 sc4 = *reinterpret_cast<const float4*>(a.a_scale + k0 + sq * 4);
          va[i].x *= sc4.x; va[i].y *= sc4.y; va[i].z *= sc4.z; va[i].w *= sc4.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + srow + 32 * i;
      vw[i] = (n < a.N) ? *reinterpret_cast<const float4*>(a.W + (size_t)n * a.K + k0 + sq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (k0 > kbeg) __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(sA + (srow + 32 * i) * LDS_STRIDE + sq * 4) = va[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sW + (srow + 32 * i) * LDS_STRIDE + sq * 4) = vw[i];
    __syncthreads();
#pragma unroll
    for (int o = 0; o < BK / 8; ++o) {
      const float4 av = *reinterpret_cast<const float4*>(sA + abase + 8 * o);
      const float4 w0 = *reinterpret_cast<const float4*>(sW + wbase + 8 * o);
      const float4 w1 = *reinterpret_cast<const float4*>(sW + wbase + 32 * LDS_STRIDE + 8 * o);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w1.w, acc[1], 0, 0, 0);
    }
  }

  __syncthreads();   // every wave has finished reading the last k tile: its A region becomes the epilogue park
  gemm_finish(a, acc, sA, m0, n0, wave, h, p);
}

// (at most 4 waves per SIMD: aiming at 5 the register allocator spills one prefetched float4 through scratch every k tile)
template <bool MASKED>   // MASKED: dgrad prologue (ReLU mask and / or per-column scale on the A operand)
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 4))) void gemm_mfma_f32(GemmArgs a) {
  __shared__ __align__(16) float sA[BM * LDS_STRIDE];
  __shared__ __align__(16) float sW[BN * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
  // consecutive workgroups share the same rows of A (same m-tile, different n-tile) -> L2 reuse of A
  // XCD-aware tile order, see gemm_mfma_f32_serial
  const int bid = paif::xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % a.tilesN, tm = bid / a.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging map: A tile = 128 rows x 8 float4 -> 4 per thread; W tile = 64 rows x 8 float4 -> 2 per thread
  const int srow = tid >> 3, sq = tid & 7;
  const int abase = (wave * 32 + p) * LDS_STRIDE + 4 * h;
  const int wbase = p * LDS_STRIDE + 4 * h;

  // Software pipeline over the k tiles: the global loads of tile k+1 are issued right after tile k has been written to
  // LDS and stay in flight during its 32 MFMAs (the serial load -> wait -> LDS -> MFMA form left the matrix pipe idle
  // for a full memory latency per k tile and relied on the second workgroup of the CU to fill it).  Loads are
  // unconditional on clamped rows (rows >= M / columns >= N are never stored).
  unsigned aoff[4], woff[2];           // element offsets (M * lda and N * K are checked < 2^32 at launch)
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)min(m0 + srow + 32 * i, a.M - 1) * (unsigned)a.lda + sq * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) woff[i] = (unsigned)min(n0 + srow + 32 * i, a.N - 1) * (unsigned)a.K + sq * 4;
  const bool masked = MASKED && a.a_mask != nullptr, scaled = MASKED && a.a_scale != nullptr;   // launch-uniform

  float4 va[4], vw[2], vm[MASKED ? 4 : 1], vs;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) va[i] = *reinterpret_cast<const float4*>(a.A + (aoff[i] + (unsigned)k0));
#pragma unroll
    for (int i = 0; i < 2; ++i) vw[i] = *reinterpret_cast<const float4*>(a.W + (woff[i] + (unsigned)k0));
    if constexpr (MASKED) {
      if (masked) {
#pragma unroll
        for (int i = 0; i < 4; ++i) vm[i] = *reinterpret_cast<const float4*>(a.a_mask + (aoff[i] + (unsigned)k0));
      }
      if (scaled) vs = *reinterpret_cast<const float4*>(a.a_scale + k0 + sq * 4);
    }
  };

  const int kbeg = blockIdx.y * a.kper, kend = kbeg + a.kper;
  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (k0 > kbeg) __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 t = va[i];
      if constexpr (MASKED) {   // dgrad prologue: ReLU mask and per-column scale (folded BatchNorm)
        if (masked) {
          t.x = vm[i].x > 0.f ? t.x : 0.f; t.y = vm[i].y > 0.f ? t.y : 0.f;
          t.z = vm[i].z > 0.f ? t.z : 0.f; t.w = vm[i].w > 0.f ? t.w : 0.f;
        }
        if (scaled) { t.x *= vs.x; t.y *= vs.y; t.z *= vs.z; t.w *= vs.w; }
      }
      *reinterpret_cast<float4*>(sA + (srow + 32 * i) * LDS_STRIDE + sq * 4) = t;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(sW + (srow + 32 * i) * LDS_STRIDE + sq * 4) = vw[i];
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);   // block-uniform
    // operands of step o+1 are read before the 8 MFMAs of step o and pinned there (see conv_mfma.hip: left alone the
    // scheduler sinks every ds_read to just above its first use)
    float4 av[2], w0[2], w1[2];
    av[0] = *reinterpret_cast<const float4*>(sA + abase);
    w0[0] = *reinterpret_cast<const float4*>(sW + wbase);
    w1[0] = *reinterpret_cast<const float4*>(sW + wbase + 32 * LDS_STRIDE);
#pragma unroll
    for (int o = 0; o < BK / 8; ++o) {
      const int c = o & 1, n = c ^ 1;
      if (o + 1 < BK / 8) {
        av[n] = *reinterpret_cast<const float4*>(sA + abase + 8 * (o + 1));
        w0[n] = *reinterpret_cast<const float4*>(sW + wbase + 8 * (o + 1));
        w1[n] = *reinterpret_cast<const float4*>(sW + wbase + 32 * LDS_STRIDE + 8 * (o + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, w0[c].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, w1[c].x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, w0[c].y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, w1[c].y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, w0[c].z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, w1[c].z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, w0[c].w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, w1[c].w, acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  __syncthreads();   // every wave has finished reading the last k tile: its A region becomes the epilogue park
  gemm_finish(a, acc, sA, m0, n0, wave, h, p);
}


// ---------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") GEMM: same tiling, operands split into bf16 hi + bf16 lo at staging time and multiplied as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, ~1e-5 relative) -- 5.3x less matrix-pipe time
// than the exact fp32 MFMA.  LDS row record (144 B): 32 x bf16 hi | 32 x bf16 lo | 16 B pad (conflict-free b128).
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_store(char* dst, float4 t4) {
  const __bf16 hx = (__bf16)t4.x, hy = (__bf16)t4.y, hz = (__bf16)t4.z, hw = (__bf16)t4.w;
  const __bf16 lx = (__bf16)(t4.x - (float)hx), ly = (__bf16)(t4.y - (float)hy), lz = (__bf16)(t4.z - (float)hz),
               lw = (__bf16)(t4.w - (float)hw);
  uint2 hi, lo;
  hi.x = (unsigned)__builtin_bit_cast(unsigned short, hx) | ((unsigned)__builtin_bit_cast(unsigned short, hy) << 16);
  hi.y = (unsigned)__builtin_bit_cast(unsigned short, hz) | ((unsigned)__builtin_bit_cast(unsigned short, hw) << 16);
  lo.x = (unsigned)__builtin_bit_cast(unsigned short, lx) | ((unsigned)__builtin_bit_cast(unsigned short, ly) << 16);
  lo.y = (unsigned)__builtin_bit_cast(unsigned short, lz) | ((unsigned)__builtin_bit_cast(unsigned short, lw) << 16);
  *reinterpret_cast<uint2*>(dst) = hi;
  *reinterpret_cast<uint2*>(dst + 64) = lo;
}

// NP = 3 ("bf16x6"): every operand as three bf16 pieces (hi + mid + lo = the fp32 value to 2^-27), six products down to 2^-25
// relative: fp32-level parity at 6 bf16 MFMAs per product (precision code 3; the attack loops' GEMM arithmetic where K >= 256).
__device__ __forceinline__ void split_store3(char* dst, float4 t4) {
  const __bf16 hx = (__bf16)t4.x, hy = (__bf16)t4.y, hz = (__bf16)t4.z, hw = (__bf16)t4.w;
  const float rx = t4.x - (float)hx, ry = t4.y - (float)hy, rz = t4.z - (float)hz, rw = t4.w - (float)hw;
  const __bf16 mx = (__bf16)rx, my = (__bf16)ry, mz = (__bf16)rz, mw = (__bf16)rw;
  const __bf16 lx = (__bf16)(rx - (float)mx), ly = (__bf16)(ry - (float)my), lz = (__bf16)(rz - (float)mz), lw = (__bf16)(rw - (float)mw);
  auto pk = [](__bf16 a, __bf16 b) { return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16); };
  *reinterpret_cast<uint2*>(dst) = make_uint2(pk(hx, hy), pk(hz, hw));
  *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pk(mx, my), pk(mz, mw));
  *reinterpret_cast<uint2*>(dst + 128) = make_uint2(pk(lx, ly), pk(lz, lw));
}

// PF = 1 ("f16x3"): the two pieces are IEEE fp16 (11 significant bits each: 22 bits per operand, against 16 for two bf16 pieces), the
// three products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: ~2^-21.5 per product -- fp32-level for O(1) data at HALF the MFMAs of the
// three-piece bf16 form.  fp16's narrow exponent is handled by an exact power-of-two scale on the W side (x 2^8, undone on the
// accumulator): the lo piece of a weight of magnitude >= 5e-4 stays a normal fp16; activations are split unscaled (lo pieces of values
// below 0.125 are subnormal: absolute error <= 3e-8 on an operand of an O(1) sum).  |x| must stay below 65504 (it does by orders of
// magnitude on this path; an overflow shows up as inf / NaN, never silently).
constexpr float F16X3_WSCALE = 256.f;
__device__ __forceinline__ void split_store_h(char* dst, float4 t4) {
#ifdef PAIF_GEMM_NOSPLIT   // timing build (wrong results): what the kernel would cost if its operands arrived pre-split (no split arithmetic)
  {
    const unsigned a0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t4.x, t4.y)), a1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t4.z, t4.w));
    *reinterpret_cast<uint2*>(dst) = make_uint2(a0, a1);
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(a0 & 0x03ff03ffu, a1 & 0x03ff03ffu);     // tiny finite "lo" pieces
    return;
  }
#endif
  const _Float16 hx = (_Float16)t4.x, hy = (_Float16)t4.y, hz = (_Float16)t4.z, hw = (_Float16)t4.w;
  const _Float16 lx = (_Float16)(t4.x - (float)hx), ly = (_Float16)(t4.y - (float)hy), lz = (_Float16)(t4.z - (float)hz),
                 lw = (_Float16)(t4.w - (float)hw);
  auto pk = [](_Float16 a, _Float16 b) { return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16); };
  *reinterpret_cast<uint2*>(dst) = make_uint2(pk(hx, hy), pk(hz, hw));
  *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pk(lx, ly), pk(lz, lw));
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool MASKED, int NP = 2, bool GATHER = false, int PF = 0>
__device__ __forceinline__ void gemm_split_body(const GemmArgs& a) {
  static_assert(!(MASKED && GATHER), "the dgrad prologue reads a mask with A's own layout");
  static_assert(PF == 0 || NP == 2, "fp16 pieces come in pairs");
  constexpr int RB = NP == 3 ? 208 : 144;   // bytes per staged row of a 32-wide K tile (shadows the two-piece constant)
  __shared__ __align__(16) char sA[BM * RB];
  __shared__ __align__(16) char sW[BN * RB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, p = lane & 31;
  // XCD-aware tile order, see gemm_mfma_f32_serial
  const int bid = paif::xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % a.tilesN, tm = bid / a.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int srow = tid >> 3, sq = tid & 7;
  const int abase = (wave * 32 + p) * RB + 16 * hh;
  const int wbase = p * RB + 16 * hh;

  // software pipeline over the k tiles and pinned operand reads: see gemm_mfma_f32
  unsigned aoff[4], woff[2];           // element offsets; rows >= M / columns >= N are never stored
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)min(m0 + srow + 32 * i, a.M - 1) * (unsigned)a.lda + sq * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) woff[i] = (unsigned)min(n0 + srow + 32 * i, a.N - 1) * (unsigned)a.K + sq * 4;
  const bool masked = MASKED && a.a_mask != nullptr, scaled = MASKED && a.a_scale != nullptr;   // launch-uniform
  int iy0[GATHER ? 4 : 1], ix0[GATHER ? 4 : 1];   // GATHER: top-left input pixel of the row's patch; aoff = pixel index of its image
  if constexpr (GATHER) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = min(m0 + srow + 32 * i, a.M - 1);
      const int ox = m % a.gOW, t = m / a.gOW, oy = t % a.gOH, b = t / a.gOH;
      iy0[i] = oy * a.gs - a.gpad; ix0[i] = ox * a.gs - a.gpad;
      aoff[i] = (unsigned)b * (unsigned)(a.gH * a.gW);
    }
  }

  float4 va[4], vw[2], vm[MASKED ? 4 : 1], vs;
  auto gload = [&](int k0) {
    if constexpr (GATHER) {   // a 32-wide k tile lies inside one tap (gC % 32 == 0): 128 contiguous bytes of one input pixel, or padding
      const int kk = k0 / a.gC, c0 = k0 - kk * a.gC, ky = kk / a.gk, kx = kk - ky * a.gk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = iy0[i] + ky, ix = ix0[i] + kx;
        const bool ok = (unsigned)iy < (unsigned)a.gH && (unsigned)ix < (unsigned)a.gW;
        const unsigned off = (aoff[i] + (unsigned)(iy * a.gW + ix)) * (unsigned)a.gC + (unsigned)(c0 + sq * 4);
        va[i] = ok ? *reinterpret_cast<const float4*>(a.A + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) va[i] = *reinterpret_cast<const float4*>(a.A + (aoff[i] + (unsigned)k0));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) vw[i] = *reinterpret_cast<const float4*>(a.W + (woff[i] + (unsigned)k0));
    if constexpr (MASKED) {
      if (masked) {
#pragma unroll
        for (int i = 0; i < 4; ++i) vm[i] = *reinterpret_cast<const float4*>(a.a_mask + (aoff[i] + (unsigned)k0));
      }
      if (scaled) vs = *reinterpret_cast<const float4*>(a.a_scale + k0 + sq * 4);
    }
  };

  const int kbeg = blockIdx.y * a.kper, kend = kbeg + a.kper;   // split-K: blockIdx.y selects the k range
  gload(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (k0 > kbeg) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 t = va[i];
      if constexpr (MASKED) {
        if (masked) {
          t.x = vm[i].x > 0.f ? t.x : 0.f; t.y = vm[i].y > 0.f ? t.y : 0.f;
          t.z = vm[i].z > 0.f ? t.z : 0.f; t.w = vm[i].w > 0.f ? t.w : 0.f;
        }
        if (scaled) { t.x *= vs.x; t.y *= vs.y; t.z *= vs.z; t.w *= vs.w; }
      }
      if constexpr (NP == 3) split_store3(sA + (srow + 32 * i) * RB + sq * 8, t);
      else if constexpr (PF == 1) split_store_h(sA + (srow + 32 * i) * RB + sq * 8, t);
      else split_store(sA + (srow + 32 * i) * RB + sq * 8, t);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (NP == 3) split_store3(sW + (srow + 32 * i) * RB + sq * 8, vw[i]);
      else if constexpr (PF == 1) {
        const float4 w4 = vw[i];
        split_store_h(sW + (srow + 32 * i) * RB + sq * 8, make_float4(w4.x * F16X3_WSCALE, w4.y * F16X3_WSCALE, w4.z * F16X3_WSCALE, w4.w * F16X3_WSCALE));
      } else split_store(sW + (srow + 32 * i) * RB + sq * 8, vw[i]);
    }
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);   // block-uniform
    struct Ops { bf16x8 ah, al, wh[2], wl[2], at, wt[2]; };
    auto rd = [&](Ops& o, int ks) {
      o.ah = *reinterpret_cast<const bf16x8*>(sA + abase + 32 * ks);
      o.al = *reinterpret_cast<const bf16x8*>(sA + abase + 64 + 32 * ks);
      if constexpr (NP == 3) o.at = *reinterpret_cast<const bf16x8*>(sA + abase + 128 + 32 * ks);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        o.wh[t] = *reinterpret_cast<const bf16x8*>(sW + wbase + t * 32 * RB + 32 * ks);
        o.wl[t] = *reinterpret_cast<const bf16x8*>(sW + wbase + t * 32 * RB + 64 + 32 * ks);
        if constexpr (NP == 3) o.wt[t] = *reinterpret_cast<const bf16x8*>(sW + wbase + t * 32 * RB + 128 + 32 * ks);
      }
    };
    Ops o0, o1;
    rd(o0, 0);
    rd(o1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NP == 3) {   // h = hi, l = mid, t = lo; smallest products first
      auto six = [&](const Ops& o) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.ah, o.wt[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.at, o.wh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.al, o.wl[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.ah, o.wl[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.al, o.wh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.ah, o.wh[t], acc[t], 0, 0, 0);
      };
      six(o0);
      __builtin_amdgcn_sched_barrier(0);
      six(o1);
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
    auto mm = [&](const bf16x8& x, const bf16x8& y, f32x16 c) {
      if constexpr (PF == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o0.al, o0.wh[t], acc[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o0.ah, o0.wl[t], acc[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o0.ah, o0.wh[t], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o1.al, o1.wh[t], acc[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o1.ah, o1.wl[t], acc[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = mm(o1.ah, o1.wh[t], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (PF == 1) {   // undo the W-side scale (exact: a power of two)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= 1.0f / F16X3_WSCALE;
  }
  __syncthreads();   // every wave has finished reading the last k tile: its A region becomes the epilogue park
  gemm_finish(a, acc, reinterpret_cast<float*>(sA), m0, n0, wave, hh, p);
}

// The three arithmetics of the split GEMM as kernels of their own (round 6: a rocprofv3 row names what ran, no template-argument decoding):
//   gemm_mfma_bf16x3  two bf16 pieces per operand, 3 MFMAs per product (~2^-16)
//   gemm_mfma_bf16x6  three bf16 pieces, 6 MFMAs (2^-25)
//   gemm_mfma_f16x3   two IEEE fp16 pieces, 3 MFMAs (~2^-21.5; weight side x 2^8)
template <bool MASKED, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_bf16x3(GemmArgs a) { gemm_split_body<MASKED, 2, GATHER, 0>(a); }
template <bool MASKED, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_bf16x6(GemmArgs a) { gemm_split_body<MASKED, 3, GATHER, 0>(a); }
template <bool MASKED, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_f16x3(GemmArgs a) { gemm_split_body<MASKED, 2, GATHER, 1>(a); }

// split-K second pass: C[m][n] = act(scale[n] * sum_s partial[s][m][n] + shift[n]) (+ res), splits summed in index order
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 int act, const float* __restrict__ res, int ldres,
                                                                 float* __restrict__ C, int ldc, int M, int N) {
  const size_t total = (size_t)M * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += partial[(size_t)s * total + i];
    v = v * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
    if (act == 1) v = gelu_erf(v);
    else if (act == 2) v = fmaxf(v, 0.f);
    if (res) v += res[(size_t)m * ldres + n];
    C[(size_t)m * ldc + n] = v;
  }
}

}  // namespace

extern "C" int paif_gemm_masked_fwd(const float* A, int lda, const float* a_mask, const float* a_scale, const float* W,
                                    const float* scale, const float* shift, int act, const float* res, int ldres, float* C,
                                    int ldc, int M, int N, int K, int precision, paif_stream_t stream);

extern "C" int paif_gemm_fwd(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                             const float* res, int ldres, float* C, int ldc, int M, int N, int K, int precision,
                             paif_stream_t stream) {
  return paif_gemm_masked_fwd(A, lda, nullptr, nullptr, W, scale, shift, act, res, ldres, C, ldc, M, N, K, precision, stream);
}

extern "C" int paif_gemm_masked_fwd(const float* A, int lda, const float* a_mask, const float* a_scale, const float* W,
                                    const float* scale, const float* shift, int act, const float* res, int ldres, float* C,
                                    int ldc, int M, int N, int K, int precision, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "gemm: precision=%d", precision);
  PAIF_REQUIRE(A && W && C, PAIF_EINVAL, "gemm: null pointer");
  PAIF_REQUIRE(M > 0 && N > 0 && K > 0, PAIF_EINVAL, "gemm: empty shape %dx%dx%d", M, N, K);
  PAIF_REQUIRE(K % 32 == 0, PAIF_ENOSUP, "gemm: K=%d must be a multiple of 32 (pad the operands)", K);
  PAIF_REQUIRE(lda >= K && ldc >= N && (lda % 4) == 0, PAIF_EINVAL, "gemm: bad leading dimensions lda=%d ldc=%d", lda, ldc);
  PAIF_REQUIRE(act >= 0 && act <= 2, PAIF_EINVAL, "gemm: act=%d", act);
  GemmArgs a;
  a.gk = 0; a.sk = 0;
  a.a_mask = a_mask; a.a_scale = a_scale;
  a.A = A; a.W = W; a.scale = scale; a.shift = shift; a.res = res; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldres = res ? ldres : 0; a.act = act;
  a.tilesN = (N + BN - 1) / BN;
  a.nblk = a.tilesN * ((M + BM - 1) / BM);
  a.kper = K;
  a.partial = nullptr;
  a.wide = (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)C % 16 == 0) && (!res || (ldres % 4 == 0 && (uintptr_t)res % 16 == 0)) &&
           (!scale || (uintptr_t)scale % 16 == 0) && (!shift || (uintptr_t)shift % 16 == 0);
  PAIF_REQUIRE((size_t)M * lda < ((size_t)1 << 32) && (size_t)N * K < ((size_t)1 << 32), PAIF_ENOSUP,
               "gemm: operands of %dx%d (lda %d) / %dx%d elements exceed the 32-bit element offsets", M, K, lda, N, K);
  const bool pro = a_mask || a_scale;
  const dim3 grid(a.nblk), blk(256);
  hipStream_t st = paif::as_stream(stream);
  if (precision == 3) {
    if (pro) hipLaunchKernelGGL((gemm_mfma_bf16x6<true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((gemm_mfma_bf16x6<false>), grid, blk, 0, st, a);
  } else if (precision == 6) {
    if (pro) hipLaunchKernelGGL((gemm_mfma_f16x3<true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((gemm_mfma_f16x3<false>), grid, blk, 0, st, a);
  } else if (precision == 1) {
    if (pro) hipLaunchKernelGGL(gemm_mfma_bf16x3<true>, grid, blk, 0, st, a);
    else hipLaunchKernelGGL(gemm_mfma_bf16x3<false>, grid, blk, 0, st, a);
  } else {
    if (K <= 160) hipLaunchKernelGGL(gemm_mfma_f32_serial<false>, grid, blk, 0, st, a);
    else if (pro) hipLaunchKernelGGL(gemm_mfma_f32<true>, grid, blk, 0, st, a);
    else hipLaunchKernelGGL(gemm_mfma_f32<false>, grid, blk, 0, st, a);
  }
  PAIF_LAUNCH_CHECK("gemm");
  return 0;
}

// Split-K plan for the exact-fp32 GEMM: a 128 x 64 tile grid that leaves most of the 256 CUs idle while each
// workgroup walks a long k loop (MiT stages 3-4 and the SR convs at small batch) is split over k so that about one
// workgroup per CU is in flight; every split keeps at least 2 k-tiles.  Returns 1 when splitting does not pay.
extern "C" int paif_gemm_splitk_plan(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K < 256 || (K % 32) != 0) return 1;
  const int nblk = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
  if (nblk >= 128) return 1;
  int splits = (256 + nblk - 1) / nblk;
  const int ktiles = K / BK;
  if (splits > ktiles / 2) splits = ktiles / 2;
  if (splits > 16) splits = 16;
  while (splits > 1 && ktiles % splits) --splits;   // equal k extents per split
  return splits < 2 ? 1 : splits;
}

extern "C" int paif_gemm_splitk_fwd_p(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                                      const float* res, int ldres, float* C, int ldc, int M, int N, int K, int splits,
                                      float* workspace, int precision, paif_stream_t stream);

extern "C" int paif_gemm_splitk_fwd(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                                    const float* res, int ldres, float* C, int ldc, int M, int N, int K, int splits,
                                    float* workspace, paif_stream_t stream) {
  return paif_gemm_splitk_fwd_p(A, lda, W, scale, shift, act, res, ldres, C, ldc, M, N, K, splits, workspace, 0, stream);
}

extern "C" int paif_gemm_splitk_fwd_p(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                                      const float* res, int ldres, float* C, int ldc, int M, int N, int K, int splits,
                                      float* workspace, int precision, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "gemm_splitk: precision=%d", precision);
  PAIF_REQUIRE(A && W && C && workspace, PAIF_EINVAL, "gemm_splitk: null pointer");
  PAIF_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0, PAIF_EINVAL, "gemm_splitk: shape %dx%dx%d", M, N, K);
  PAIF_REQUIRE(splits >= 2 && (K / BK) % splits == 0, PAIF_EINVAL, "gemm_splitk: splits=%d does not divide %d k-tiles", splits,
               K / BK);
  PAIF_REQUIRE(lda >= K && ldc >= N && (lda % 4) == 0, PAIF_EINVAL, "gemm_splitk: bad leading dimensions lda=%d ldc=%d", lda, ldc);
  PAIF_REQUIRE(act >= 0 && act <= 2, PAIF_EINVAL, "gemm_splitk: act=%d", act);
  GemmArgs a;
  a.gk = 0; a.sk = 0;
  a.a_mask = nullptr; a.a_scale = nullptr;
  a.A = A; a.W = W; a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldres = 0; a.act = 0;
  a.tilesN = (N + BN - 1) / BN;
  a.nblk = a.tilesN * ((M + BM - 1) / BM);
  a.kper = K / splits;
  a.partial = workspace;
  a.wide = 0;
  hipStream_t st = paif::as_stream(stream);
  PAIF_REQUIRE((size_t)M * lda < ((size_t)1 << 32) && (size_t)N * K < ((size_t)1 << 32), PAIF_ENOSUP,
               "gemm_splitk: operands exceed the 32-bit element offsets");
  if (precision == 3) hipLaunchKernelGGL((gemm_mfma_bf16x6<false>), dim3(a.nblk, splits), dim3(256), 0, st, a);
  else if (precision == 6) hipLaunchKernelGGL((gemm_mfma_f16x3<false>), dim3(a.nblk, splits), dim3(256), 0, st, a);
  else if (precision == 1) hipLaunchKernelGGL(gemm_mfma_bf16x3<false>, dim3(a.nblk, splits), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_mfma_f32<false>, dim3(a.nblk, splits), dim3(256), 0, st, a);
  PAIF_LAUNCH_CHECK("gemm_splitk");
  const size_t total = (size_t)M * N;
  const int rblocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(rblocks), dim3(256), 0, st, workspace, splits, scale, shift, act, res,
                     res ? ldres : 0, C, ldc, M, N);
  PAIF_LAUNCH_CHECK("gemm_splitk_reduce");
  return 0;
}

// Strided conv as a GEMM whose A operand is gathered from the NHWC map (OverlapPatchEmbed.proj core/mix_transformer.py:168-169 for the
// stages with Cin % 32 == 0, Attention.sr :74): the im2col matrix of paif_im2col_fwd + paif_gemm_fwd without writing and re-reading it.
// Same k order, same arithmetic: bit-identical to that pair.  x [B,H,W,Cin]; Wt [N, k*k*Cin] (paif_pack_conv_gemm_weight);
// out [B*OH*OW, >= N] row stride ldc.  precision: PAIF_CONV_BF16X3 / PAIF_CONV_BF16X6.  splits > 1 (paif_gemm_splitk_plan of the
// im2col shape): workspace[splits * M * N] floats, partial sums added in split order by the reduction pass.
extern "C" int paif_gemm_conv_fwd(const float* x, int B, int H, int W, int Cin, int k, int stride, int pad, const float* Wt,
                                  const float* scale, const float* shift, int act, const float* res, int ldres, float* out, int ldc,
                                  int N, int precision, int splits, float* workspace, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "gemm_conv: precision=%d", precision);
  PAIF_REQUIRE(x && Wt && out, PAIF_EINVAL, "gemm_conv: null pointer");
  PAIF_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0 && k > 0 && stride > 0 && pad >= 0 && Cin > 0, PAIF_EINVAL, "gemm_conv: bad shape");
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  PAIF_REQUIRE(OH > 0 && OW > 0, PAIF_EINVAL, "gemm_conv: empty output");
  const int M = B * OH * OW;
  const int K = (k * k * Cin + 31) / 32 * 32;   // Wt's row length: the taps padded up to a multiple of 32 (paif_pack_conv_gemm_weight)
  if (precision == 0) {
    // exact fp32 MFMA, element-wise gather through a column table: the short-k form (K <= 160: the 3-channel 7x7 patch embed)
    PAIF_REQUIRE(K <= 160 && k < 1024 && Cin < 1024, PAIF_ENOSUP, "gemm_conv: the exact gathered form is built for K <= 160 (got %d)", K);
    PAIF_REQUIRE(splits == 1, PAIF_EINVAL, "gemm_conv: the exact gathered form does not split k");
  } else {
    PAIF_REQUIRE(Cin % 32 == 0, PAIF_ENOSUP, "gemm_conv: Cin=%d must be a multiple of 32 (a k tile inside one tap)", Cin);
  }
  PAIF_REQUIRE(ldc >= N && act >= 0 && act <= 2, PAIF_EINVAL, "gemm_conv: ldc=%d act=%d", ldc, act);
  PAIF_REQUIRE(splits >= 1 && (K / BK) % splits == 0 && (splits == 1 || workspace), PAIF_EINVAL, "gemm_conv: splits=%d, %d k tiles", splits, K / BK);
  PAIF_REQUIRE((size_t)B * H * W * Cin < ((size_t)1 << 32) && (size_t)N * K < ((size_t)1 << 32), PAIF_ENOSUP, "gemm_conv: operands exceed the 32-bit element offsets");
  GemmArgs a;
  a.a_mask = nullptr; a.a_scale = nullptr;
  a.A = x; a.W = Wt; a.C = out;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = ldc;
  a.tilesN = (N + BN - 1) / BN;
  a.nblk = a.tilesN * ((M + BM - 1) / BM);
  a.kper = K / splits;
  a.sk = 0;
  a.gk = k; a.gs = stride; a.gpad = pad; a.gH = H; a.gW = W; a.gC = Cin; a.gOH = OH; a.gOW = OW;
  hipStream_t st = paif::as_stream(stream);
  if (splits > 1) {
    a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.ldres = 0; a.act = 0; a.partial = workspace; a.wide = 0;
  } else {
    a.scale = scale; a.shift = shift; a.res = res; a.ldres = res ? ldres : 0; a.act = act; a.partial = nullptr;
    a.wide = (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)out % 16 == 0) && (!res || (ldres % 4 == 0 && (uintptr_t)res % 16 == 0)) &&
             (!scale || (uintptr_t)scale % 16 == 0) && (!shift || (uintptr_t)shift % 16 == 0);
  }
  if (precision == 3) hipLaunchKernelGGL((gemm_mfma_bf16x6<false, true>), dim3(a.nblk, splits), dim3(256), 0, st, a);
  else if (precision == 6) hipLaunchKernelGGL((gemm_mfma_f16x3<false, true>), dim3(a.nblk, splits), dim3(256), 0, st, a);
  else if (precision == 1) hipLaunchKernelGGL((gemm_mfma_bf16x3<false, true>), dim3(a.nblk, splits), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_mfma_f32_serial<true>, dim3(a.nblk), dim3(256), 0, st, a);
  PAIF_LAUNCH_CHECK("gemm_conv");
  if (splits > 1) {
    const size_t total = (size_t)M * N;
    const int rblocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(rblocks), dim3(256), 0, st, workspace, splits, scale, shift, act, res,
                       res ? ldres : 0, out, ldc, M, N);
    PAIF_LAUNCH_CHECK("gemm_conv_reduce");
  }
  return 0;
}

// The dgrad of a NON-OVERLAPPING strided conv (Attention.sr core/mix_transformer.py:74: kernel = stride = sr_ratio) as ONE GEMM:
// d_col[M, sr*sr*C] = dY[M, K] . Wt[sr*sr*C, K]^T with the col2im of paif_col2im_fwd done by the epilogue's addresses -- every
// column lands on exactly one input pixel.  dx [B,H,W,C] (H, W multiples of sr: every pixel is written); bit-identical to
// paif_gemm_fwd + paif_col2im_fwd.  precision: any of paif_gemm_fwd's.
extern "C" int paif_gemm_col2im_fwd(const float* dY, int lda, const float* Wt, float* dx, int B, int H, int W, int C, int sr, int K,
                                    int precision, paif_stream_t stream) {
  PAIF_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, PAIF_EINVAL, "gemm_col2im: precision=%d", precision);
  PAIF_REQUIRE(dY && Wt && dx, PAIF_EINVAL, "gemm_col2im: null pointer");
  PAIF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && sr >= 1 && K > 0, PAIF_EINVAL, "gemm_col2im: bad shape");
  PAIF_REQUIRE(H % sr == 0 && W % sr == 0, PAIF_ENOSUP, "gemm_col2im: %dx%d is not a multiple of the stride %d (pixels without a patch: use paif_col2im_fwd)", H, W, sr);
  PAIF_REQUIRE(C % 4 == 0 && K % 32 == 0 && lda >= K && (lda % 4) == 0 && (uintptr_t)dx % 16 == 0, PAIF_EINVAL, "gemm_col2im: C=%d K=%d lda=%d", C, K, lda);
  const int OH = H / sr, OW = W / sr, M = B * OH * OW, N = sr * sr * C;
  PAIF_REQUIRE((size_t)M * lda < ((size_t)1 << 32) && (size_t)N * K < ((size_t)1 << 32), PAIF_ENOSUP, "gemm_col2im: operands exceed the 32-bit element offsets");
  GemmArgs a;
  a.gk = 0;
  a.a_mask = nullptr; a.a_scale = nullptr;
  a.A = dY; a.W = Wt; a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.C = dx;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = N; a.ldres = 0; a.act = 0;
  a.tilesN = (N + BN - 1) / BN;
  a.nblk = a.tilesN * ((M + BM - 1) / BM);
  a.kper = K; a.partial = nullptr; a.wide = 1;
  a.sk = sr; a.sH = H; a.sW = W; a.sC = C; a.sOH = OH; a.sOW = OW;
  const dim3 grid(a.nblk), blk(256);
  hipStream_t st = paif::as_stream(stream);
  if (precision == 3) hipLaunchKernelGGL((gemm_mfma_bf16x6<false>), grid, blk, 0, st, a);
  else if (precision == 6) hipLaunchKernelGGL((gemm_mfma_f16x3<false>), grid, blk, 0, st, a);
  else if (precision == 1) hipLaunchKernelGGL(gemm_mfma_bf16x3<false>, grid, blk, 0, st, a);
  else if (K <= 160) hipLaunchKernelGGL(gemm_mfma_f32_serial<false>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(gemm_mfma_f32<false>, grid, blk, 0, st, a);
  PAIF_LAUNCH_CHECK("gemm_col2im");
  return 0;
}
