// Split-bf16 GEMM, wide-tile form (round 5) for the SegFormer linears whose k loop is long enough to be matrix-pipe work
// (core/mix_transformer.py Mlp :22-25 fc1 / fc2, Attention :66-69 q / kv / proj, :74 sr conv via im2col; segformer_head.py MLP.proj :19):
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  operands split into bf16 pieces at staging time, fp32 accumulate (gemm_mfma.hip).
//
// Why another tiling (tools/gemm_shapes_b16.py, B = 16 mit_b3 shapes): gemm_mfma_bf16x3's wave owns 32 x 64 outputs -- per k step it
// issues 6 ds_read_b128 for 6 MFMAs and every staged element is split (3 vector instructions) for 2 N tiles only; at 4 workgroups
// per CU the LDS pipe, the vector issue and the matrix pipe each sit at 25-40 % and the kernel reaches 0.19-0.24 of the 3-MFMA
// peak (fc1 / fc2 of stage 3: 78 us).  Here
//   * workgroup tile 128 (M) x 64*NT (N), 4 waves as 2 x 2, wave tile 64 x 32*NT: 2 A fragments and NT W fragments feed 6*NT
//     MFMAs per k step ((2 + NT) * 2 reads per 6 * NT MFMAs: 0.47 at NT = 5 instead of 1.0);
//   * NT = 5 (BN = 320) covers a whole stage-3 row: A is read from HBM once, split once;
//   * W arrives PRE-SPLIT (paif_gemm2_pack_weight: [k tile][n][piece][32] bf16, the LDS row record without its pad): staging it is a
//     16-byte load and a ds_write_b128, no vector arithmetic -- split on the fly the 2*NT W rows per thread were 70 % of the ~7.5 vector
//     instructions per MFMA that kept the first version of this kernel at the old one's speed;
//   * the accumulators (32 * NT registers) live in the unified 512-entry register file: one or two waves per SIMD.
#include <stdint.h>

#include <type_traits>

#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;
#ifndef G2_TAIL
#define G2_TAIL 2   // chunks of MFMAs behind the barrier (they cover the first fragment reads of the next tile)
#endif
#ifndef G2_EXP
#define G2_EXP 0    // experiments (timing only, wrong results): 1 no C stores, 2 no MFMAs / fragment reads, 4 no staging
#endif

// compile-time loop: the body sees its index as a constant expression (register-array indices, region boundaries)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct G2Args {
  const float* A; const void* W; const float* scale; const float* shift; const float* res; float* C;
  int M, N, K, lda, ldc, ldres, act;
  int tilesN;
};

__device__ __forceinline__ unsigned pk2(__bf16 a, __bf16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

template <int NP>
__device__ __forceinline__ void split_store(char* dst, float4 t4) {
  const __bf16 hx = (__bf16)t4.x, hy = (__bf16)t4.y, hz = (__bf16)t4.z, hw = (__bf16)t4.w;
  const float rx = t4.x - (float)hx, ry = t4.y - (float)hy, rz = t4.z - (float)hz, rw = t4.w - (float)hw;
  const __bf16 mx = (__bf16)rx, my = (__bf16)ry, mz = (__bf16)rz, mw = (__bf16)rw;
  *reinterpret_cast<uint2*>(dst) = make_uint2(pk2(hx, hy), pk2(hz, hw));
  *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pk2(mx, my), pk2(mz, mw));
  if constexpr (NP == 3) {
    const __bf16 lx = (__bf16)(rx - (float)mx), ly = (__bf16)(ry - (float)my), lz = (__bf16)(rz - (float)mz), lw = (__bf16)(rw - (float)mw);
    *reinterpret_cast<uint2*>(dst + 128) = make_uint2(pk2(lx, ly), pk2(lz, lw));
  }
}

// Staging item of region r: the four A items (split: ~22 vector instructions each) spread evenly among the W items (a copy)
template <int S>
constexpr int item_at(int r) {
  constexpr int step = S / 4;
  if (r % step == 0 && r / step < 4) return r / step;
  int na = (r + step - 1) / step;
  if (na > 4) na = 4;
  return 4 + (r - na);
}

// NT: 32-column tiles per wave (workgroup tile 128 x 64*NT); NP: bf16 pieces per operand (2: three products, 2^-16; 3: six, 2^-25)
//
// Software pipeline (one basic block per k tile, one barrier per k tile, LDS double-buffered):
//   iteration kt runs the MFMAs of tile kt out of buffer kt & 1 in NC = 2 * (3 | 6) * 2 CHUNKS of NT MFMAs (k step, product, A fragment);
//   behind the MFMAs of chunk c the wave splits its share of tile kt + 1 (held in registers since iteration kt - 1) into the other
//   buffer, one staging item (a float4 of A or W: ~12 vector instructions + 2 ds_write_b64) at a time, and re-issues that register's
//   global load for tile kt + 2 -- a whole iteration (~2000 matrix-pipe cycles) ahead of its use;
//   the fragments of k step 1 are read during k step 0; after chunk NC - 3 (all of tile kt + 1 written, all of tile kt read) the
//   barrier, and the first fragments of tile kt + 1 are read behind the last two chunks.
template <int NT, int NP>
__global__ __launch_bounds__(256) void gemm_split2_kernel(G2Args a) {
  constexpr int RB = NP == 3 ? 208 : 144;   // bytes per staged row of a 32-wide k tile: NP x 64 B + 16 B pad (conflict-free b128)
  constexpr int BN = 64 * NT;
  constexpr int ABYTES = BM * RB, WBYTES = BN * RB, BUF = ABYTES + WBYTES;
  constexpr int NQ = NP == 3 ? 6 : 3;       // products per k step
  constexpr int NC = 2 * NQ * 2;            // chunks per k tile
  constexpr int SW = NP * NT;               // 16-byte chunks of the packed W tile per thread and k tile (BN rows x NP x 4 chunks / 256)
  constexpr int S = 4 + SW;                 // staging items per thread and k tile
  constexpr int NCW = NC - G2_TAIL;         // chunks that carry staging items
  constexpr int NM = NC * NT, NMW = NCW * NT;
  extern __shared__ __align__(16) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, p = lane & 31;
  const int wr = wave >> 1, wc = wave & 1;
  const int bid = paif::xcd_remap(blockIdx.x, gridDim.x);   // the tilesN tiles that share 128 rows of A run on one XCD
  const int tn = bid % a.tilesN, tm = bid / a.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int srow = tid >> 3, sq = tid & 7;
  const int abase = (wr * 64 + p) * RB + 16 * hh;
  const int wbase = ABYTES + (wc * 32 * NT + p) * RB + 16 * hh;
  const int sbase = srow * RB + sq * 8;
  unsigned goff[4];                          // element offsets of the A staging items (rows clamped)
#pragma unroll
  for (int i = 0; i < 4; ++i) goff[i] = (unsigned)min(m0 + srow + 32 * i, a.M - 1) * (unsigned)a.lda + sq * 4;
  // packed W: k tile kt at kt * N * NP*64 bytes; this workgroup's BN rows are contiguous in it; chunk j = tid + 256 * i of the tile
  // is 16 bytes of row j / (4 NP) at byte (j % (4 NP)) * 16 of the LDS record
  const char* wsrc = static_cast<const char*>(a.W) + (size_t)n0 * (NP * 64) + tid * 16;
  const size_t wkt = (size_t)a.N * (NP * 64);
  int wdst[SW];
#pragma unroll
  for (int i = 0; i < SW; ++i) {
    const int j = tid + 256 * i;
    wdst[i] = ABYTES + (j / (4 * NP)) * RB + (j % (4 * NP)) * 16;
  }

  // the prefetched 16 bytes cross the loop's back edge as ONE 128-bit value (the empty asm is its only use): split into scalars
  // the allocator gives the four dwords unrelated registers and copies the loaded tuple into them at the back edge -- behind vmcnt(0)
  f32x4 vr[4];
  u32x4 wv[SW];
  auto gload = [&](int s, int kt) {
    if (s < 4) vr[s] = *reinterpret_cast<const f32x4*>(a.A + (goff[s] + (unsigned)(kt * BK)));
    else wv[s - 4] = *reinterpret_cast<const u32x4*>(wsrc + kt * wkt + (s - 4) * 4096);
  };
  auto stage = [&](int s, char* buf) {
    if (G2_EXP & 4) return;
    if (s < 4) {
      f32x4 v = vr[s];
      asm("" : "+v"(v));
      split_store<NP>(buf + 32 * s * RB + sbase, make_float4(v[0], v[1], v[2], v[3]));
    } else {
      u32x4 v = wv[s - 4];
      asm("" : "+v"(v));
      *reinterpret_cast<u32x4*>(buf + wdst[s - 4]) = v;
    }
  };
  bf16x8 af[2][2][NP], wf[2][NT][NP];        // [set][fragment][piece]
  auto rdfrag = [&](int set, const char* buf, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q) af[set][i][q] = *reinterpret_cast<const bf16x8*>(buf + abase + i * 32 * RB + 64 * q + 32 * ks);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < NP; ++q) wf[set][t][q] = *reinterpret_cast<const bf16x8*>(buf + wbase + t * 32 * RB + 64 * q + 32 * ks);
  };

  const int nk = a.K / BK;
#pragma unroll
  for (int s = 0; s < S; ++s) gload(s, 0);
#pragma unroll
  for (int s = 0; s < S; ++s) stage(s, smem);
#pragma unroll
  for (int s = 0; s < S; ++s) gload(s, nk > 1 ? 1 : 0);
  __syncthreads();
  rdfrag(0, smem, 0);

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * BUF;
    char* nxt = smem + ((kt & 1) ^ 1) * BUF;
    const int k2 = min(kt + 2, nk - 1);        // past the end: the last tile again (loaded, never used)
    // REGION r (r < S) = { the global load of item r - 1, MFMAs [r * NMW / S, (r + 1) * NMW / S), the split of item r }: the scheduler
    // interleaves inside a region only.  The load of an item sits in the region AFTER its split: hoisted above the split's reads it
    // gets a second register and a copy at the loop's back edge -- behind s_waitcnt vmcnt(0), i.e. no prefetch at all.
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int c = m / NT, t = m % NT;
      constexpr int ks = c / (2 * NQ), qi = (c / 2) % NQ, i = c & 1;
      static_for<0, S>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if constexpr (m == r * NMW / S) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (r > 0) gload(item_at<S>(r - 1), k2);
          else rdfrag(1, cur, 1);
          stage(item_at<S>(r), nxt);
        }
      });
      if constexpr (m == NMW) {
        __builtin_amdgcn_sched_barrier(0);
        gload(item_at<S>(S - 1), k2);
        __syncthreads();            // tile kt + 1 is complete in `nxt`; every fragment of tile kt has been read
        rdfrag(0, nxt, 0);
      }
      // smallest products first (piece 0 = hi), in gemm_mfma_bf16x3's order: the accumulation sequence of an output is the same
      constexpr int QA3[6] = {0, 2, 1, 0, 1, 0}, QW3[6] = {2, 0, 1, 1, 0, 0}, QA2[3] = {1, 0, 0}, QW2[3] = {0, 1, 0};
      constexpr int qa = NP == 3 ? QA3[qi] : QA2[qi % 3], qw = NP == 3 ? QW3[qi] : QW2[qi % 3];
      if constexpr (!(G2_EXP & 2)) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][qa], wf[ks][t][qw], acc[i][t], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  }

  __syncthreads();   // every wave has finished reading the last k tile: the A region becomes the epilogue park
  // Epilogue (gemm_mfma.hip gemm_finish, wide form): a wave parks one 32 x 32 accumulator tile at a time in its private 32 x 36-float
  // slice and re-reads it as float4 per (row, column quad): scale / shift / residual as float4, stores of full 128-byte row segments.
  float* ep = reinterpret_cast<float*>(smem) + wave * (32 * 36);   // 4 x 4,608 B = the A region of buffer 0
  const int c4 = lane & 7, rsub = lane >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + p] = acc[i][t][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int n = n0 + wc * 32 * NT + 32 * t + 4 * c4;
      const float4 sc = a.scale ? *reinterpret_cast<const float4*>(a.scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 sh = a.shift ? *reinterpret_cast<const float4*>(a.shift + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int mb = m0 + wr * 64 + i * 32;
      float4 rv[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min(mb + it * 8 + rsub, a.M - 1);
        rv[it] = a.res ? *reinterpret_cast<const float4*>(a.res + (size_t)m * a.ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        const int m = mb + row;
        float4 v = *reinterpret_cast<const float4*>(ep + row * 36 + 4 * c4);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (a.act == 1) { v.x = paif::gelu_erf_fast(v.x); v.y = paif::gelu_erf_fast(v.y); v.z = paif::gelu_erf_fast(v.z); v.w = paif::gelu_erf_fast(v.w); }
        else if (a.act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.res) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
        if (m < a.M && !(G2_EXP & 1)) *reinterpret_cast<float4*>(a.C + (size_t)m * a.ldc + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <int NT, int NP>
int launch(const G2Args& a, hipStream_t st) {
  constexpr int RB = NP == 3 ? 208 : 144;
  constexpr int lds = 2 * (BM + 64 * NT) * RB;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split2_kernel<NT, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { paif::set_error("gemm2: cannot raise dynamic LDS to %d: %s", lds, hipGetErrorString(e)); return (int)e; }
  }
  const int nblk = a.tilesN * ((a.M + BM - 1) / BM);
  hipLaunchKernelGGL((gemm_split2_kernel<NT, NP>), dim3(nblk), dim3(256), lds, st, a);
  return 0;
}

// W [N, K] fp32 -> [K / 32][N][NP][32] bf16 pieces (one thread per 4 consecutive k of a row)
template <int NP>
__global__ __launch_bounds__(256) void gemm2_pack_kernel(const float* __restrict__ W, char* __restrict__ out, int N, int K) {
  const size_t total = (size_t)N * (K / 4);
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int n = (int)(e / (K / 4)), k4 = (int)(e % (K / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)n * K + k4);
    split_store<NP>(out + ((size_t)(k4 / 32) * N + n) * (NP * 64) + (k4 % 32) * 2, v);
  }
}

}  // namespace

extern "C" size_t paif_gemm2_packed_bytes(int N, int K, int precision) {
  if (N <= 0 || K <= 0 || (K % 32) != 0 || (precision != 1 && precision != 3)) return 0;
  return (size_t)(K / 32) * N * ((precision == 3 ? 3 : 2) * 64);
}

extern "C" int paif_gemm2_pack_weight(const float* W, void* out, int N, int K, int precision, paif_stream_t stream) {
  PAIF_REQUIRE(W && out, PAIF_EINVAL, "gemm2_pack: null pointer");
  PAIF_REQUIRE(N > 0 && K > 0 && K % 32 == 0, PAIF_EINVAL, "gemm2_pack: shape %dx%d (K must be a multiple of 32)", N, K);
  PAIF_REQUIRE(precision == 1 || precision == 3, PAIF_EINVAL, "gemm2_pack: precision=%d", precision);
  PAIF_REQUIRE((uintptr_t)W % 16 == 0 && (uintptr_t)out % 16 == 0, PAIF_EINVAL, "gemm2_pack: 16-byte alignment");
  const size_t total = (size_t)N * (K / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = paif::as_stream(stream);
  if (precision == 3) hipLaunchKernelGGL(gemm2_pack_kernel<3>, dim3(blocks), dim3(256), 0, st, W, static_cast<char*>(out), N, K);
  else hipLaunchKernelGGL(gemm2_pack_kernel<2>, dim3(blocks), dim3(256), 0, st, W, static_cast<char*>(out), N, K);
  PAIF_LAUNCH_CHECK("gemm2_pack");
  return 0;
}

// Column tiles per wave the wide form would use for N (0: not built for this N -- callers take paif_gemm_fwd)
extern "C" int paif_gemm2_plan(int M, int N, int K, int precision) {
  if (precision != 1 && precision != 3) return 0;
  if (M < 1 || K < 32 || (K % 32) != 0) return 0;
  int nt = 0;
  if (N % 320 == 0) nt = 5;
  else if (N % 256 == 0) nt = 4;
  else if (N % 128 == 0) nt = 2;
  else if (N % 64 == 0) nt = 1;
  if (precision == 3 && nt == 5) nt = 0;   // three pieces x 320 rows do not fit LDS
  return nt;
}

extern "C" int paif_gemm2_fwd(const float* A, int lda, const void* W, const float* scale, const float* shift, int act,
                              const float* res, int ldres, float* C, int ldc, int M, int N, int K, int precision, int nt,
                              paif_stream_t stream) {
  PAIF_REQUIRE(precision == 1 || precision == 3, PAIF_EINVAL, "gemm2: precision=%d", precision);
  PAIF_REQUIRE(A && W && C, PAIF_EINVAL, "gemm2: null pointer");
  PAIF_REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0, PAIF_EINVAL, "gemm2: shape %dx%dx%d", M, N, K);
  PAIF_REQUIRE((nt == 1 || nt == 2 || nt == 4 || nt == 5) && N % (64 * nt) == 0, PAIF_EINVAL, "gemm2: nt=%d does not tile N=%d", nt, N);
  PAIF_REQUIRE(!(precision == 3 && nt == 5), PAIF_ENOSUP, "gemm2: three pieces x 320 columns do not fit LDS");
  PAIF_REQUIRE(lda >= K && ldc >= N && (lda % 4) == 0 && (ldc % 4) == 0 && (uintptr_t)C % 16 == 0 && (uintptr_t)A % 16 == 0, PAIF_EINVAL,
               "gemm2: leading dimensions / alignment lda=%d ldc=%d", lda, ldc);
  PAIF_REQUIRE(!res || (ldres % 4 == 0 && (uintptr_t)res % 16 == 0), PAIF_EINVAL, "gemm2: residual alignment");
  PAIF_REQUIRE((!scale || (uintptr_t)scale % 16 == 0) && (!shift || (uintptr_t)shift % 16 == 0), PAIF_EINVAL, "gemm2: scale / shift alignment");
  PAIF_REQUIRE(act >= 0 && act <= 2, PAIF_EINVAL, "gemm2: act=%d", act);
  PAIF_REQUIRE((size_t)M * lda < ((size_t)1 << 32), PAIF_ENOSUP, "gemm2: A exceeds the 32-bit element offsets");
  PAIF_REQUIRE((uintptr_t)W % 16 == 0, PAIF_EINVAL, "gemm2: packed W alignment");
  G2Args a;
  a.A = A; a.W = W; a.scale = scale; a.shift = shift; a.res = res; a.C = C;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldres = res ? ldres : 0; a.act = act;
  a.tilesN = N / (64 * nt);
  hipStream_t st = paif::as_stream(stream);
  int rc = 0;
  if (precision == 1) {
    if (nt == 1) rc = launch<1, 2>(a, st);
    else if (nt == 2) rc = launch<2, 2>(a, st);
    else if (nt == 4) rc = launch<4, 2>(a, st);
    else rc = launch<5, 2>(a, st);
  } else {
    if (nt == 1) rc = launch<1, 3>(a, st);
    else if (nt == 2) rc = launch<2, 3>(a, st);
    else rc = launch<4, 3>(a, st);
  }
  if (rc) return rc;
  PAIF_LAUNCH_CHECK("gemm2");
  return 0;
}
