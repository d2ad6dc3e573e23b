// Shared helpers for the gfx950 kernels of libpaif_hip.so (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "paif_hip.h"

namespace paif {

void set_error(const char* fmt, ...);

#define PAIF_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      paif::set_error(__VA_ARGS__);     \
      return (code);                    \
    }                                   \
  } while (0)

// Launch check: hipGetLastError after a launch (does not synchronise).
#define PAIF_LAUNCH_CHECK(name)                                            \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      paif::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                     \
    }                                                                      \
  } while (0)

static inline hipStream_t as_stream(paif_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (round-robin
// dispatch, MI355X_MICROARCH "Workgroup dispatch"), so give each XCD a CONTIGUOUS chunk of tiles:
// spatially adjacent tiles (which share halo pixels) then hit the same 4 MiB L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk, int reverse = 0) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int cnt = (xcd < r) ? q + 1 : q;
  return base + (reverse ? cnt - 1 - k : k);   // reverse: each XCD walks its tile range from the end
}

__device__ __forceinline__ float prelu_f(float v, float a) { return v >= 0.f ? v : a * v; }

// GELU (erf form, nn.GELU default: x * Phi(x)) and its derivative, round-4 form.  Phi(-t) = 2^q(t) for t = min(|x|, 5.8) with ONE degree-10
// polynomial q (weighted minimax fit of log2 Phi(-t), weight t * Phi(-t) = the absolute error it causes in GELU; q(0) = -1 exactly), and
// Phi(x) = 1 - Phi(-x) for x >= 0: 10 fma + v_exp_f32 + 4 instead of the ~30 instructions of a two-range erf (both ranges evaluated, one
// selected) -- the MixFFN depthwise-conv + GELU kernel is vector-issue bound (a SIMD issues one vector instruction per ~4 cycles:
// 171 us at 3.7 TB/s on the stage-1 map of a B=16 batch with the erf form).  Accuracy vs float64 over [-8, 8] and N(0, 1.5) samples:
// GELU max |error| 3.8e-7 (the fp32 rounding of x * Phi at |x| = 8), max relative error 9.9e-7 for |x| < 3 -- the erf form lost
// relative accuracy for x < 0 (1 + erf cancels: 1.8e-5); derivative max |error| 1.5e-7.  Beyond 5.8: Phi(-t) is held at 3.3e-9.
__device__ __forceinline__ float phi_neg_tail(float x) {   // Phi(-|x|)
  const float t = fminf(fabsf(x), 5.8f);
  float r = -3.315222873e-08f;
  r = fmaf(r, t, 9.477192170e-07f);
  r = fmaf(r, t, -1.167834977e-05f);
  r = fmaf(r, t, 7.916988962e-05f);
  r = fmaf(r, t, -2.841531522e-04f);
  r = fmaf(r, t, -2.489754232e-07f);
  r = fmaf(r, t, 6.957890404e-03f);
  r = fmaf(r, t, -5.245515152e-02f);
  r = fmaf(r, t, -4.592144081e-01f);
  r = fmaf(r, t, -1.151105125e+00f);
  r = fmaf(r, t, -1.0f);
  return __builtin_amdgcn_exp2f(r);
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float e = phi_neg_tail(x);
  return x * (x >= 0.f ? 1.0f - e : e);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float e = phi_neg_tail(x);
  return (x >= 0.f ? 1.0f - e : e) + x * 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // Phi + x phi
}

// Streaming 16-byte store (global_store_dwordx4 ... nt): the activation maps of the bench shapes (315 MB each) are far
// larger than L2 + MALL, so keeping freshly written lines cached only evicts the halo rows the neighbouring tiles are
// about to re-read.  Measured +1 % on the fusion forward with the conv outputs alone.
#ifdef PAIF_NO_NT   // A/B knob (tools/build_variant.sh nont -DPAIF_NO_NT): plain cache policy everywhere
__device__ __forceinline__ void store_nt(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 load_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void store_nt(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void store_nt(float* p, float v) { *p = v; }
#else
__device__ __forceinline__ void store_nt(float* p, float4 v) {
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  const f32x4_nt vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<f32x4_nt*>(p));
}
// Streaming 16-byte load for maps that are read exactly once per kernel (residuals, pointwise inputs)
__device__ __forceinline__ float4 load_nt(const float* p) {
  typedef float f32x4_nt __attribute__((ext_vector_type(4)));
  const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_nt(float* p, float2 v) {
  typedef float f32x2_nt __attribute__((ext_vector_type(2)));
  const f32x2_nt vv = {v.x, v.y};
  __builtin_nontemporal_store(vv, reinterpret_cast<f32x2_nt*>(p));
}
__device__ __forceinline__ void store_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
#endif
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

// ---------------------------------------------------------------------------------------------
// Activation STORAGE (BASELINE configs[1]: "bf16"): a 32-channel NHWC map is held either as fp32 (BF = 0, the default) or as
// bf16 (BF = 1: 64 bytes per pixel, round-to-nearest-even on store); arithmetic is fp32 / split-bf16 MFMA either way.
// A kernel's storage template parameter ST names the pair: 0 = fp32 in / fp32 out, 1 = bf16 in / bf16 out,
// 2 = fp32 in / bf16 out (the first layer behind the fp32 guided-filter block).  Pointers stay typed `float*` in the
// argument structs; the helpers address `eoff` ELEMENTS from the base in the storage's element size.
// ---------------------------------------------------------------------------------------------
// Kernel storage codes (template argument ST): 0 fp32 in / out | 1 bf16 in / out | 2 fp32 in, bf16 out | 3 (internal) bf16 in / out
// behind an input PReLU.  +3 (4, 5, 6): the same with the WEIGHTS taken as plain bf16 (their split-bf16 low half dropped:
// precision PAIF_CONV_BF16, the plain bf16 MFMA of BASELINE configs[1]).  7: 16-bit in (residual maps too), fp32 OUT, plain weights
// (the last conv of the 16-bit forward: its map feeds the fp32-input stem_out kernel).
// +8 (9, 12, 13, 14, 15): the same codes with the 16-bit format being IEEE fp16 instead of bf16 (round 5: PAIF_ST_F16*; the MFMA is
// v_mfma_f32_32x32x16_f16, the weight pack holds fp16 hi | lo pieces).  Format values of the helpers below: 0 fp32, 1 bf16, 2 fp16.
constexpr bool st_f16(int ST) { return ST >= 8; }
constexpr int st_low(int ST) { return ST & 7; }
constexpr int st_base(int ST) { return st_low(ST) == 7 ? 1 : (st_low(ST) >= 4 ? st_low(ST) - 3 : st_low(ST)); }
constexpr int st_fmt16(int ST) { return st_f16(ST) ? 2 : 1; }
constexpr int st_in(int ST) { return (st_base(ST) == 1 || st_base(ST) == 3) ? st_fmt16(ST) : 0; }
constexpr int st_out(int ST) { return (st_low(ST) >= 1 && st_low(ST) != 7) ? st_fmt16(ST) : 0; }
constexpr int st_res(int ST) { return st_low(ST) == 7 ? st_fmt16(ST) : st_out(ST); }   // storage of the residual maps
constexpr bool st_wl0(int ST) { return st_low(ST) >= 4; }
// ST 1 / 4 / 7: the staged A operand IS the stored 16-bit value (no input activation, or ReLU), so its split low half is exactly
// zero: the kernels drop the lo x W_hi MFMA and the low half's LDS traffic.  ST 3: bf16 in / out with an input PReLU, whose result
// is not a bf16 value -- the full split is kept; ST 6 (plain 16-bit arithmetic) rounds that result to the format (nearest-even) instead.
constexpr bool st_lo0(int ST) { return st_f16(ST) || st_low(ST) == 1 || st_low(ST) == 4 || st_low(ST) == 6 || st_low(ST) == 7; }   // fp16 operands are never split

__device__ __forceinline__ float4 bf16x4_to_f32(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f32_to_bf16x4(float4 v) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_t)));
}
// fp16 (IEEE half) twins of the two conversions: v_cvt_f32_f16 / v_cvt_pk_f16_f32 (round to nearest even)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 f16x4_to_f32(uint2 u) {
  const f32x4v_t v = __builtin_convertvector(__builtin_bit_cast(f16x4_t, u), f32x4v_t);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 f32_to_f16x4(float4 v) {
  const f32x4v_t a = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(uint2, __builtin_convertvector(a, f16x4_t));
}
// 16-bit quad <-> fp32 quad in format F (1 bf16, 2 fp16)
template <int F> __device__ __forceinline__ float4 h4_to_f32(uint2 u) {
  if constexpr (F == 2) return f16x4_to_f32(u);
  else return bf16x4_to_f32(u);
}
template <int F> __device__ __forceinline__ uint2 f32_to_h4(float4 v) {
  if constexpr (F == 2) return f32_to_f16x4(v);
  else return f32_to_bf16x4(v);
}
// max(x, 0) on a packed pair of 16-bit values (the ReLU of a stored map, no conversion)
template <int F> __device__ __forceinline__ unsigned relu_h2(unsigned u) {
  if constexpr (F == 2) {
    const f16x2_t z = {(_Float16)0, (_Float16)0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(f16x2_t, u), z));
  } else {   // bf16: the sign bit of each half decides
    return u & ~((((u >> 15) & 0x00010001u) * 0xffffu));
  }
}

template <int BF> __device__ __forceinline__ float4 ldq(const float* base, size_t eoff) {
  if constexpr (BF) return h4_to_f32<BF>(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + eoff));
  else return *reinterpret_cast<const float4*>(base + eoff);
}
template <int BF> __device__ __forceinline__ float4 ldq_nt(const float* base, size_t eoff) {
  if constexpr (BF) {
    typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
    const u32x2_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(reinterpret_cast<const unsigned short*>(base) + eoff));
    return h4_to_f32<BF>(make_uint2(v.x, v.y));
  } else {
    return load_nt(base + eoff);
  }
}
template <int BF> __device__ __forceinline__ void stq(float* base, size_t eoff, float4 v) {
  if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + eoff) = f32_to_h4<BF>(v);
  else *reinterpret_cast<float4*>(base + eoff) = v;
}
template <int BF> __device__ __forceinline__ void stq_nt(float* base, size_t eoff, float4 v) {
  if constexpr (BF) {
    typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
    const uint2 u = f32_to_h4<BF>(v);
    const u32x2_nt vv = {u.x, u.y};
    __builtin_nontemporal_store(vv, reinterpret_cast<u32x2_nt*>(reinterpret_cast<unsigned short*>(base) + eoff));
  } else {
    store_nt(base + eoff, v);
  }
}

// (max, sum) of a value over the 8 CONSECUTIVE lanes that hold one pixel's channel quads (lane & 7 = quad), result in all 8 lanes, on
// the DPP path (no LDS traffic, unlike __shfl_xor's ds_bpermute): lanes ^1 and ^2 by quad permutes; after those every lane of a quad
// holds its quad's result, so the mirror of the 8-lane half row (lane i <-> 7 - i) supplies the other quad's.  The summation tree is
// channel_pool2_kernel's ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)) up to commutation: bit-equal.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ void pix8_max_sum(float& mx, float& sm) {
  mx = fmaxf(mx, dpp_f<0xB1>(mx)); sm += dpp_f<0xB1>(sm);       // quad_perm [1,0,3,2]
  mx = fmaxf(mx, dpp_f<0x4E>(mx)); sm += dpp_f<0x4E>(sm);       // quad_perm [2,3,0,1]
  mx = fmaxf(mx, dpp_f<0x141>(mx)); sm += dpp_f<0x141>(sm);     // row_half_mirror
}

// One 32x32x16 MFMA on 16-bit operands in format F (1 bf16, 2 fp16); operands travel as 8 packed 16-bit values (uint4 bits)
typedef float mf32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mf16x8 __attribute__((ext_vector_type(8)));
template <int F, typename AV, typename BV> __device__ __forceinline__ mf32x16 mfma16(AV a, BV b, mf32x16 c) {
  static_assert(sizeof(AV) == 16 && sizeof(BV) == 16, "8 x 16-bit operands");
  if constexpr (F == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mf16x8, a), __builtin_bit_cast(mf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mbf16x8, a), __builtin_bit_cast(mbf16x8, b), c, 0, 0, 0);
}

// ---- split-bf16 operands with a selectable number of pieces (attention*.hip; gemm_mfma.hip / gemm_split2.hip carry their own staging forms) ----
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// v -> NP pieces: piece 0 = rn(v), piece q = rn(v - the pieces before it).  PF = 0: bf16 pieces (two carry 16 significant bits, three
// 24); PF = 1: IEEE fp16 pieces (two carry 22 bits; operands must lie inside fp16's exponent range), kept in the same 16-byte container.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <int PF>
__device__ __forceinline__ __bf16 piece16(float r, float& back) {
  if constexpr (PF == 1) {
    const _Float16 h = (_Float16)r;
    back = (float)h;
    return __builtin_bit_cast(__bf16, h);
  } else {
    const __bf16 h = (__bf16)r;
    back = (float)h;
    return h;
  }
}
template <int NP, int PF = 0>
__device__ __forceinline__ void splitN(const float (&v)[8], bf16x8_t (&pc)[NP]) {
  static_assert(PF == 0 || NP == 2, "fp16 pieces come in pairs");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = v[i];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      float back;
      pc[q][i] = piece16<PF>(r, back);
      if (q + 1 < NP) r -= back;
    }
  }
}

// acc += A . B over the kept piece products, smallest first (gemm_mfma.hip's order): NP = 2 -> a1 b0 + a0 b1 + a0 b0 (2^-16 with bf16
// pieces, ~2^-21.5 with fp16 pieces); NP = 3 -> a0 b2 + a2 b0 + a1 b1 + a0 b1 + a1 b0 + a0 b0 (2^-25)
template <int NP, int PF = 0>
__device__ __forceinline__ void mfma_pieces(f32x16_t& acc, const bf16x8_t (&pa)[NP], const bf16x8_t (&pb)[NP]) {
  constexpr int QA3[6] = {0, 2, 1, 0, 1, 0}, QB3[6] = {2, 0, 1, 1, 0, 0}, QA2[3] = {1, 0, 0}, QB2[3] = {0, 1, 0};
#pragma unroll
  for (int qi = 0; qi < (NP == 3 ? 6 : 3); ++qi) {
    const bf16x8_t x = pa[NP == 3 ? QA3[qi] : QA2[qi % 3]], y = pb[NP == 3 ? QB3[qi] : QB2[qi % 3]];
    if constexpr (PF == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, x), __builtin_bit_cast(f16x8_t, y), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
  }
}

}  // namespace paif
