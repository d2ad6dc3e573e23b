// Guided filter (radius 4, 1-channel guide, 32-channel NHWC target, both eps) -- inference form, fused (the coefficient maps
// A, b never touch HBM), with the HORIZONTAL 9-tap box sums on the matrix cores.
// Replaces the reference's two GuidedFilter(4, eps) calls of Cell_Decom.decomposition (core/model_fusion_auto.py:522-535;
// third-party guided_filter_pytorch.GuidedFilter, algorithm per oracle/shims/guided_filter_pytorch).
//
// Why (profiles/r02_pmc_fusion.txt): the all-VALU kernel (guided_filter.hip) was the largest single kernel
// of the headline step at 14.6 % of the HBM roof -- per eps 377 VALU instructions and 36 ds_read_b128 per thread-row, two
// barriers per row with two waves per SIMD to cover the LDS round trips, stage 1 computed twice (once per eps), matrix cores idle.
// Here:
//   * a lane owns 4 ADJACENT COLUMNS of ONE channel; a wave = 16 columns x 16 channels; a workgroup = 4 waves = a 64-column
//     strip of ONE 16-channel half, ONE WAVE PER SIMD with the whole 512-entry register file: the 9-row windows of all six
//     quantities (y, g*y, and A, b for both eps) stay in registers as rings of 3-row partial sums (4 adds per row and quantity,
//     direct sums of the nine terms, no running add/subtract); stage 1 is computed once for both eps;
//   * the horizontal 9-column sum is  S[m][c] = sum_k Band[m][k] * V[k][c]  on v_mfma_f32_16x16x32_f16: M = the wave's 16 output
//     columns, N = 16 channels, K = 16 columns x (hi, lo).  A lane's four vertical sums, each split into an f16 pair hi = rtz(v),
//     lo = rn(v - hi) (22 significant bits; products with the 0/1 band are exact, accumulation is fp32), ARE its B-operand
//     fragment -- no data movement.  The accumulator comes back with exactly the lane's own 4 columns (D row = 4*(lane>>4) + reg),
//     so stage 2 consumes it in place.  Only the 4 + 4 halo columns of a wave cross waves: one 16-byte fragment per edge lane and
//     quantity through LDS (before: every value, nine times);
//   * stage 1 of row r and stage 2 of row r-5 share ONE barrier per row;
//   * inputs stream through LDS by LDS-DMA (buffer_load_dword ... lds) with no VGPR cost; every vector-memory instruction of the
//     loop is issued unconditionally (masked lanes / warm-up rows store out of range of the buffer descriptor and are dropped by
//     the hardware), so the s_waitcnt vmcnt(N) that retires a row is a compile-time constant (N counts LOADS only); every LDS access of the loop is
//     inline asm (hipcc would drain vmcnt in front of each ds_read it can see while an LDS-DMA is in flight).
//   * the two 16-channel halves of a tile are blocks b and b + 8: the same XCD under round-robin dispatch, so the half-lines one of
//     them fetches are L2 hits for the other (speed only).
// f16 range: a 9-row sum beyond 65504 cannot be split.  The kernel tracks max|v| and raises *flag; the host wrapper then runs
// the all-VALU kernel (predicated on the flag) over the same output -- results are always those of an fp32 box filter.
#include <stdint.h>
#include <type_traits>

#include "paif_common.h"

namespace paif_gf_mfma {

constexpr int R = 4;
constexpr int SC = 64;            // input columns per workgroup
constexpr int SO = SC - 4 * R;    // 48 output columns
constexpr int PF = 6;             // the unroll factor = the partial-sum ring size (static ring slots)
#ifndef GF_PFY
#define GF_PFY 6
#endif
#ifndef GF_XPOSE
#define GF_XPOSE 0
#endif
constexpr int PFY = GF_PFY;       // input rows in flight per wave (LDS-DMA ring depth; slots are a running counter)
constexpr bool XPOSE = GF_XPOSE;  // stores: 1 = 16 bytes per lane through a wave-private LDS transposition (2 per iteration), 0 = 8 dword stores
constexpr int PD = PFY + 3;       // plane ring depth (see the slot-reuse argument at the DMA issue)
constexpr int NQ = 6;             // exchanged quantities: sum y, sum g*y, A0, b0, A1, b1
constexpr int ZSLOT = 8;          // halo slot that stays zero (outside neighbours of the strip's edge waves)
// vector-memory instructions per wave and iteration: plane DMAs (1 x 16 B per lane when W % 4 == 0, else 2 x 4 B), 1 input DMA
// (16 B per lane: the wave's 16 columns x 16 channels), 2 stores (16 B per lane, one per eps)
constexpr int NSTORE = XPOSE ? 2 : 8;
template <bool AL4> struct VmIter { static constexpr int planes = AL4 ? 1 : 2; };
constexpr unsigned RSRC_W3 = 0x00020000u;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS map (byte offsets inside the kernel's only __shared__ object, which therefore starts at LDS address 0)
constexpr int Y_OFF = 0;
constexpr int Y_BYTES = PFY * 4 * 1024;                   // [PFY][4 waves][16 cols][16 channels] float  49,152
constexpr int P_OFF = Y_OFF + Y_BYTES;
constexpr int P_SLOT = 12 * 256;                          // one iteration's plane set: 12 slots of 64 columns
constexpr int P_BYTES = PD * P_SLOT;                      // [PD][12][64 cols] float                      46,080
constexpr int H_OFF = P_OFF + P_BYTES;
constexpr int H_QTY = (ZSLOT + 1) * 256;                  // one quantity: 9 slots x 16 channels x 16 B
constexpr int H_BUF = NQ * H_QTY;                         // one buffer (iteration parity)
constexpr int H_BYTES = 2 * H_BUF;                        //                                              27,648
constexpr int T_OFF = H_OFF + H_BYTES;                    // per-wave output transposition: [4 waves][2 eps][16 cols][16 ch] float
constexpr int T_BYTES = 4 * 2048;
constexpr int LDS_BYTES = T_OFF + T_BYTES;

#ifndef GF_VMCAP
#define GF_VMCAP 63
#endif
#define GF_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) > GF_VMCAP ? GF_VMCAP : (n)) : "memory")
#define GF_RD32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define GF_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define GF_WR128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define GF_WR32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_off, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_off, 16, voff, soff, 0, 0);
}

// four fp32 values (the lane's 4 columns) -> B-operand fragment [hi(c0,c1), lo(c0,c1), hi(c2,c3), lo(c2,c3)]
__device__ __forceinline__ u32x4 split4(f32x4 v, float& vmax) {
  // max |v| seen (two v_max3_f32 with |.| source modifiers; a NaN operand is ignored -- it propagates to the output by itself)
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(vmax) : "v"(vmax), "v"(v[0]), "v"(v[1]));
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(vmax) : "v"(vmax), "v"(v[2]), "v"(v[3]));
  // hi = rtz_f16(v) (packed convert); lo = rn_f16(v - hi) in ONE instruction per value: v_fma_mix{lo,hi}_f16 evaluates
  // fma(hi as f16, -1.0, v) in fp32 (exact: v - hi has at most 13 significant bits) and rounds it into one half of the result
  u32x4 f;
  f[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
  f[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]));
  unsigned l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(f[0]), "v"(v[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(f[0]), "v"(v[1]));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(f[2]), "v"(v[2]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(f[2]), "v"(v[3]));
  f[1] = l01;
  f[3] = l23;
  return f;
}

__device__ __forceinline__ f32x4 band_mfma(u32x4 a_own, u32x4 b_own, u32x4 a_halo, u32x4 b_halo) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_own), __builtin_bit_cast(half8, b_own), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_halo), __builtin_bit_cast(half8, b_halo), acc, 0, 0, 0);
  return acc;
}

// band operand: lane (m = lane & 15, gk = lane >> 4) holds A[m][k = 8 gk + j], j = 4p + 2t + s  <->  column colbase + 2p + s
__device__ __forceinline__ u32x4 band_operand(int m, int colbase, bool live) {
  u32x4 d;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned v = 0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int col = colbase + 2 * p + s;
        const bool on = live && (m - col <= R) && (col - m <= R);
        v |= (on ? 0x3C00u : 0u) << (16 * s);      // f16 1.0
      }
      d[2 * p + t] = v;
    }
  return d;
}

// 9-row window of one quantity (4 columns) as three 3-row partial sums p(i) + p(i-3) + p(i-6), p(i) = x(i) + x(i-1) + x(i-2):
// six slots (p(i-1) .. p(i-6); p(i) replaces p(i-6) once that has been read) + the two previous rows -- direct sums of the nine
// terms, 4 adds per row; 8 register quadruples
struct Ring {
  f32x4 p[PF];
  f32x4 a1, a2;
};

template <int K>
__device__ __forceinline__ f32x4 ring_push(Ring& rg, f32x4 x) {
  const f32x4 p = (rg.a2 + rg.a1) + x;
  const f32x4 v = (p + rg.p[(K + 3) % PF]) + rg.p[K];
  rg.p[K] = p; rg.a2 = rg.a1; rg.a1 = x;
  return v;
}

// One (16-channel half, batch image, 64-column strip, row segment) per workgroup of 4 waves.
//   planes: [4][B*H*W] = mean_g, 1/(var_g + eps0), 1/(var_g + eps1), 1/n  (gf_guide_stats_kernel)
//   AL4: W % 4 == 0 -- a lane's 4 columns of a plane are one aligned 16-byte piece (one plane DMA per wave and iteration)
// BFO: the two low-frequency maps are written as bf16 (round to nearest even): the bf16 configuration's storage of the maps behind this
// block (the statistics, A, b and every sum stay fp32).  A lane holds ONE channel of 4 columns: lanes c and c ^ 1 exchange two columns
// (one DPP quad permute per value) so that each stores dwords = channel pairs (c & ~1, c | 1) -- 4 dword stores per row instead of 8;
// eight 2-byte stores per row made the kernel 11 % slower.
template <bool AL4, bool BFO>
__global__ __launch_bounds__(256, 1) void gf_mfma_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                         const float* __restrict__ planes, float* __restrict__ lf,
                                                         unsigned* __restrict__ flag, int B, int H, int W, int nstrip, int nseg,
                                                         int frows, int ntiles) {
  constexpr int VM_LOADS = VmIter<AL4>::planes + 1;   // LOADS per iteration (LDS-DMA): the only operations that retire in order with the awaited ones
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  asm volatile("" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem) : "memory");   // only asm touches it

  const int tid = threadIdx.x, l = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = 16-column group of the strip (wave-uniform on purpose)
  const int c = l & 15, g = l >> 4;
  // blocks b and b + 8 (same XCD under round-robin dispatch) are the two channel halves of one tile
  const int bid = blockIdx.x;
  const int chh = (bid >> 3) & 1;
  int t = (bid >> 4) * 8 + (bid & 7);
  if (t >= ntiles) return;
  const int strip = t % nstrip; t /= nstrip;
  const int seg = t % nseg;
  const int b = t / nseg;
  const int X0 = strip * SO - 2 * R;
  const int lc0 = 16 * q + 4 * g;                       // strip-local first column of this lane
  const int col0 = X0 + lc0;
  const int ybeg = seg * frows, yend = min(H, ybeg + frows);
  const size_t npix = (size_t)B * H * W;
  const size_t img = (size_t)b * H * W;

  // ---- per-lane constants ----
  bool cin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cin[i] = col0 + i >= 0 && col0 + i < W;
  // input DMA / output store geometry: lane -> (column l >> 2 of the wave's 16, channel quad l & 3): 16 bytes of one pixel
  const int pc = X0 + 16 * q + (l >> 2);
  const unsigned yoff16 = (unsigned)((min(max(pc, 0), W - 1) * 32 + 16 * chh + 4 * (l & 3)) * 4);
  const int plc = 16 * q + (l >> 2);                    // strip-local column of the stored pixel
  const unsigned soff16 = (plc >= 2 * R && plc < SC - 2 * R && pc >= 0 && pc < W) ? (unsigned)((pc * 32 + 16 * chh + 4 * (l & 3)) * 4) : 0xFFFFFFFFu;
  unsigned soff[4];                                     // direct-store path: (column 4g + i, channel c) or out of range
#pragma unroll
  for (int i = 0; i < 4; ++i)
    soff[i] = (lc0 >= 2 * R && lc0 < SC - 2 * R && cin[i]) ? (unsigned)(((col0 + i) * 32 + 16 * chh + c) * 4) : 0xFFFFFFFFu;
  const bool odd = (c & 1) != 0;                         // BFO: even lanes store columns 0, 1, odd lanes columns 2, 3 of the channel pair
  unsigned soff2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = odd ? 2 + j : j;
    soff2[j] = (lc0 >= 2 * R && lc0 < SC - 2 * R && (odd ? cin[2 + j] : cin[j])) ? (unsigned)(((col0 + i) * 32 + 16 * chh + (c & ~1)) * 2) : 0xFFFFFFFFu;
  }
  const u32x4 a_own = band_operand(l & 15, 4 * (l >> 4), true);
  const u32x4 a_halo = band_operand(l & 15, (l >> 4) == 0 ? -4 : 16, (l >> 4) < 2);
  // halo slots: publish own fragment as L (lanes g == 0) / R (g == 3); read left neighbour's R (g == 0), right neighbour's L (g == 1)
  const bool pub = g == 0 || g == 3;
  const int rd_slot = g == 0 ? (q > 0 ? (q - 1) * 2 + 1 : ZSLOT) : g == 1 ? (q < 3 ? (q + 1) * 2 : ZSLOT) : ZSLOT;
  const unsigned a_pub = H_OFF + (q * 2 + (g == 3 ? 1 : 0)) * 256 + c * 16;
  const unsigned a_rd = H_OFF + rd_slot * 256 + c * 16;
  const unsigned a_y = Y_OFF + q * 1024 + g * 256 + c * 4;       // y(column 4g + i, channel c) of a ring slot: + i * 64
  const unsigned a_pl = P_OFF + lc0 * 4;
  const unsigned a_tw = T_OFF + q * 2048 + g * 256 + c * 4;      // transposition: write (eps, column 4g + i, channel c): + eps * 1024 + i * 64
  const unsigned a_tr = T_OFF + q * 2048 + l * 16;               // read back 16 bytes = (column l >> 2, channels 4 (l & 3) ..): + eps * 1024
  const unsigned rowbytes_pl = (unsigned)W * 4u, rowbytes = (unsigned)W * 128u;
  const unsigned plane_bytes = (unsigned)(npix * 4);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, (int)((size_t)H * rowbytes_pl), RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + img), 0, (int)(3 * (size_t)plane_bytes + (size_t)H * rowbytes_pl), RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + img * 32), 0, (int)((size_t)H * rowbytes), RSRC_W3);
  static_assert(!(XPOSE && BFO), "bf16 outputs use the direct-store path");
  constexpr unsigned OES = BFO ? 2u : 4u;                 // bytes per stored output element
  const unsigned rowbytes_o = (unsigned)W * 32u * OES;
  char* const o0 = reinterpret_cast<char*>(lf) + img * 32 * OES, * const o1 = reinterpret_cast<char*>(lf) + (npix + img) * 32 * OES;
  const int out_bytes = (int)((size_t)H * rowbytes_o);
  // Plane staging: 12 slots of 64 columns per iteration --
  //   0..3 = {rn, mg, rden0, rden1}(r-4)   4 = g(r)   5 = g(r-9)   6, 7 = copies of 4, 5   8 = rn(r-9)   9..11 = copies / spare
  // AL4: ONE 16-byte DMA per wave fills four consecutive slots (lane group j = l >> 4 -> slot base + j, 4 columns per lane):
  //   wave 0 -> 0..3, wave 1 -> 4..7 (lane groups alternate between rows r and r-9), waves 2 and 3 -> 8..11 (the same data twice:
  //   every wave must issue the same number of vector-memory instructions);
  // else two 4-byte DMAs per wave, one column per lane: wave 0 -> 0, 1; wave 1 -> 2, 3; wave 2 -> 4, 5; wave 3 -> 8, 9.
  // Plane byte offsets inside the planes buffer: mg 0, rden0 1, rden1 2, rn 3 (x plane_bytes).
  const int j16 = l >> 4;
  const unsigned colb4 = (unsigned)min(max(X0 + 4 * (l & 15), 0), max(W - 4, 0)) * 4u;
  const unsigned colb1 = (unsigned)min(max(X0 + l, 0), W - 1) * 4u;
  const unsigned pl_plane4 = q == 0 ? (j16 == 0 ? 3u : j16 == 1 ? 0u : j16 == 2 ? 1u : 2u) : q == 1 ? 0u : 3u;
  const unsigned pl_voff4 = (q == 1 ? 0u : pl_plane4 * plane_bytes) + colb4;
  const bool pl_second_row = (j16 & 1) != 0;              // wave 1: odd lane groups fetch row r-9, even ones row r
  const unsigned pa_voff = (q == 0 ? 3u * plane_bytes : q == 1 ? 1u * plane_bytes : q == 2 ? 0u : 3u * plane_bytes) + colb1;
  const unsigned pb_voff = (q == 0 ? 0u : q == 1 ? 2u * plane_bytes : q == 2 ? 0u : 3u * plane_bytes) + colb1;
  const bool pl_guide = AL4 ? q == 1 : q == 2;
  const unsigned pl_dst = AL4 ? (q == 0 ? 0u : q == 1 ? 4u : 8u) * 256u : (q == 0 ? 0u : q == 1 ? 2u : q == 2 ? 4u : 8u) * 256u;

  // zero the permanent zero slot of both halo buffers
  if (tid < 2 * NQ * 16) *reinterpret_cast<uint4*>(smem + H_OFF + (tid >> 4) * H_QTY + ZSLOT * 256 + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  Ring ry, rgy, rA0, rB0, rA1, rB1;
#pragma unroll
  for (int k = 0; k < PF; ++k) { ry.p[k] = zero4; rgy.p[k] = zero4; rA0.p[k] = zero4; rB0.p[k] = zero4; rA1.p[k] = zero4; rB1.p[k] = zero4; }
  ry.a1 = ry.a2 = rgy.a1 = rgy.a2 = rA0.a1 = rA0.a2 = rB0.a1 = rB0.a2 = rA1.a1 = rA1.a2 = rB1.a1 = rB1.a2 = zero4;
  f32x4 wA0 = zero4, wB0 = zero4, wA1 = zero4, wB1 = zero4;           // vertical (A, b) window sums of the previous iteration
  float vmax = 0.f;

  const int r0 = ybeg - 2 * R;                           // first streamed input row
  const int n_it = ((yend - ybeg) + 4 * R + 1 + PF - 1) / PF * PF;

  auto issue_planes = [&](int it_target, int pslot) {    // planes of iteration it_target -> ring slot pslot = it_target % PD
    const unsigned base = P_OFF + pslot * P_SLOT + pl_dst;
    const int rr = r0 + it_target;
    const unsigned s_m4 = (unsigned)min(max(rr - R, 0), H - 1) * rowbytes_pl;            // rows r-4, r, r-9 (clamped)
    const unsigned s_0 = (unsigned)min(max(rr, 0), H - 1) * rowbytes_pl;
    const unsigned s_m9 = (unsigned)min(max(rr - (2 * R + 1), 0), H - 1) * rowbytes_pl;
    const __amdgpu_buffer_rsrc_t rs = pl_guide ? rs_g : rs_p;
    if (AL4) {
      const unsigned sa = q == 0 ? s_m4 : q == 1 ? s_0 : s_m9, sb = q == 0 ? s_m4 : s_m9;   // wave-uniform
      dma16(rs, pl_voff4 + (pl_second_row ? sb : sa), 0, base);
    } else {
      const unsigned sa = q <= 1 ? s_m4 : q == 2 ? s_0 : s_m9, sb = q <= 1 ? s_m4 : s_m9;
      dma4(rs, pa_voff, sa, base);
      dma4(rs, pb_voff, sb, base + 256);
    }
  };
  auto issue_y = [&](int it_target, int yslot) {         // input row of iteration it_target: 16 columns x 16 channels = 1 KB
    dma16(rs_y, yoff16, (unsigned)min(max(r0 + it_target, 0), H - 1) * rowbytes, Y_OFF + (yslot * 4 + q) * 1024);
  };
  auto dropped_stores = [&]() {                          // two stores through a zero-length descriptor: issued, counted, dropped
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(o0, 0, 0, RSRC_W3);
#pragma unroll
    for (int i = 0; i < NSTORE; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, rz, 0, 0, 0);
  };

  // ---- prologue: planes(0); then {planes(j+1), y(j), two dropped stores} for j = 0..PFY-1: the same vector-memory pattern as a
  // loop iteration, so that ONE wait constant is valid from the first iteration on ----
  issue_planes(0, 0);
#pragma unroll 1
  for (int j = 0; j < PFY; ++j) { issue_planes(j + 1, j + 1); issue_y(j, j); dropped_stores(); }
  GF_VMWAIT(VM_LOADS * PFY);                             // planes(0) landed
  lds_barrier();                                         // ... and are visible to every wave (so is the zero slot)
  int ps_rd = 0, ps_wr = PFY + 1, ys = 0;                // ring slots: planes read now / planes filled now / input row read and refilled now

  auto step = [&](auto ktag, int itb) {
    constexpr int k = decltype(ktag)::value;
    const int it = itb + k;
    const int r = r0 + it;
    // retire y(it) and planes(it+1): issued PFY iterations ago; YOUNGER LOADS in the queue: those of PFY-1 iterations.  Stores do not
    // retire in order with loads (conv_dma.hip: dropped stores retire at once), so they must not be counted as "allowed outstanding":
    // vmcnt <= (younger loads) is the bound that holds whatever the stores do.  Measured: no cost (0.518 vs 0.519 ms per launch)
    GF_VMWAIT(VM_LOADS * (PFY - 1));
    const unsigned a_pit = a_pl + (unsigned)ps_rd * (unsigned)P_SLOT;
    const unsigned a_yit = a_y + (unsigned)ys * 4096u;
    float y0, y1, y2, y3;
    f32x4 gv;
    GF_RD32(y0, a_yit, 0 * 64);
    GF_RD32(y1, a_yit, 1 * 64);
    GF_RD32(y2, a_yit, 2 * 64);
    GF_RD32(y3, a_yit, 3 * 64);
    GF_RD128(gv, a_pit, 4 * 256);                        // guide(r, 4 columns)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(gv)::"memory");   // the y slot is free again
    issue_planes(it + PFY + 1, ps_wr);
    issue_y(it + PFY, ys);
    ps_rd = ps_rd == PD - 1 ? 0 : ps_rd + 1;
    ps_wr = ps_wr == PD - 1 ? 0 : ps_wr + 1;
    ys = ys == PFY - 1 ? 0 : ys + 1;
    const bool rin = r >= 0 && r < H;
    f32x4 yy;
    yy[0] = (rin && cin[0]) ? y0 : 0.f;
    yy[1] = (rin && cin[1]) ? y1 : 0.f;
    yy[2] = (rin && cin[2]) ? y2 : 0.f;
    yy[3] = (rin && cin[3]) ? y3 : 0.f;
    const u32x4 f_y = split4(ring_push<k>(ry, yy), vmax);
    const u32x4 f_gy = split4(ring_push<k>(rgy, gv * yy), vmax);
    const u32x4 f_a0 = split4(wA0, vmax);
    const u32x4 f_b0 = split4(wB0, vmax);
    const u32x4 f_a1 = split4(wA1, vmax);
    const u32x4 f_b1 = split4(wB1, vmax);
    constexpr int HB = (k & 1) * H_BUF;                  // PF is even: the iteration's parity is static
    if (pub) {
      GF_WR128(a_pub, f_y, HB + 0 * H_QTY);
      GF_WR128(a_pub, f_gy, HB + 1 * H_QTY);
      GF_WR128(a_pub, f_a0, HB + 2 * H_QTY);
      GF_WR128(a_pub, f_b0, HB + 3 * H_QTY);
      GF_WR128(a_pub, f_a1, HB + 4 * H_QTY);
      GF_WR128(a_pub, f_b1, HB + 5 * H_QTY);
    }
    lds_barrier();
    // ---- stage 2: LF_e(r - 9) = (box(A_e) * g + box(b_e)) / n, transposed through a wave-private LDS tile so that a lane stores
    // 16 bytes (4 channels of one pixel).  Two store INSTRUCTIONS per iteration whatever the masks say: a lane without a valid
    // output (strip halo column, column outside the image) carries an out-of-range offset, a warm-up / tail row a zero-length
    // descriptor -- the hardware drops those stores ----
    {
      u32x4 h_a0, h_b0, h_a1, h_b1;
      f32x4 g2, rn2;
      GF_RD128(h_a0, a_rd, HB + 2 * H_QTY);
      GF_RD128(h_b0, a_rd, HB + 3 * H_QTY);
      GF_RD128(h_a1, a_rd, HB + 4 * H_QTY);
      GF_RD128(h_b1, a_rd, HB + 5 * H_QTY);
      GF_RD128(g2, a_pit, 5 * 256);                      // output row r - 9
      GF_RD128(rn2, a_pit, 8 * 256);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h_a0), "+v"(h_b0), "+v"(h_a1), "+v"(h_b1), "+v"(g2), "+v"(rn2)::"memory");
      const f32x4 s_a0 = band_mfma(a_own, f_a0, a_halo, h_a0);
      const f32x4 s_b0 = band_mfma(a_own, f_b0, a_halo, h_b0);
      const f32x4 s_a1 = band_mfma(a_own, f_a1, a_halo, h_a1);
      const f32x4 s_b1 = band_mfma(a_own, f_b1, a_halo, h_b1);
      const f32x4 out0 = (s_a0 * g2 + s_b0) * rn2;
      const f32x4 out1 = (s_a1 * g2 + s_b1) * rn2;
      if (XPOSE) {
        const float a0 = out0[0], a1 = out0[1], a2 = out0[2], a3 = out0[3], b0 = out1[0], b1 = out1[1], b2 = out1[2], b3 = out1[3];
        GF_WR32(a_tw, a0, 0 * 64); GF_WR32(a_tw, a1, 1 * 64); GF_WR32(a_tw, a2, 2 * 64); GF_WR32(a_tw, a3, 3 * 64);
        GF_WR32(a_tw, b0, 1024 + 0 * 64); GF_WR32(a_tw, b1, 1024 + 1 * 64); GF_WR32(a_tw, b2, 1024 + 2 * 64); GF_WR32(a_tw, b3, 1024 + 3 * 64);
      } else {
        const int ro = r - (2 * R + 1);
        const bool rowv = ro >= ybeg && ro < yend;
        const unsigned so = (unsigned)max(ro, 0) * rowbytes_o;
        const int nrec = rowv ? out_bytes : 0;           // zero-length descriptor: every store of this row is dropped
        const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(o0, 0, nrec, RSRC_W3);
        const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(o1, 0, nrec, RSRC_W3);
        if constexpr (BFO) {
          auto pair_store = [&](const f32x4& ov, const __amdgpu_buffer_rsrc_t& rs) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float a = ov[j], b = ov[2 + j];
              // partner lane (c ^ 1): quad permute [1, 0, 3, 2]
              const float pa = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(a), 0xB1, 0xF, 0xF, true));
              const float pb = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(b), 0xB1, 0xF, 0xF, true));
              const f32x2_t v = {odd ? pb : a, odd ? b : pa};     // (channel c & ~1, channel c | 1) of column j (even lanes) / 2 + j (odd)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)), rs, soff2[j], so, 2);
            }
          };
          pair_store(out0, ro0);
          pair_store(out1, ro1);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float o = out0[i]; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro0, soff[i], so, 2); }
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float o = out1[i]; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro1, soff[i], so, 2); }
        }
      }
    }
    // ---- stage 1: (A_e, b_e) of row r - 4, then their 9-row window sums for the next iteration ----
    {
      u32x4 h_y, h_gy, t0, t1;
      f32x4 rn1, mg1, rd0, rd1;
      if (XPOSE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is written (wave-private: no barrier)
        GF_RD128(t0, a_tr, 0);
        GF_RD128(t1, a_tr, 1024);
      } else {
        t0 = u32x4{0, 0, 0, 0}; t1 = t0;
      }
      GF_RD128(h_y, a_rd, HB + 0 * H_QTY);
      GF_RD128(h_gy, a_rd, HB + 1 * H_QTY);
      GF_RD128(rn1, a_pit, 0 * 256);                     // stage-1 row r - 4
      GF_RD128(mg1, a_pit, 1 * 256);
      GF_RD128(rd0, a_pit, 2 * 256);
      GF_RD128(rd1, a_pit, 3 * 256);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(h_y), "+v"(h_gy), "+v"(rn1), "+v"(mg1), "+v"(rd0), "+v"(rd1)::"memory");
      if (XPOSE) {
        const int ro = r - (2 * R + 1);
        const bool rowv = ro >= ybeg && ro < yend;
        const unsigned so = (unsigned)max(ro, 0) * rowbytes_o;
        const int nrec = rowv ? out_bytes : 0;           // zero-length descriptor: every store of this row is dropped
        __builtin_amdgcn_raw_buffer_store_b128(t0, __builtin_amdgcn_make_buffer_rsrc(o0, 0, nrec, RSRC_W3), soff16, so, 2);
        __builtin_amdgcn_raw_buffer_store_b128(t1, __builtin_amdgcn_make_buffer_rsrc(o1, 0, nrec, RSRC_W3), soff16, so, 2);
        // A 128-bit store reads its data registers quad by quad AFTER issue; hipcc (ROCm 7.2, gfx950) places no wait state
        // before a VALU write that reuses them -- observed: element 1 of the last lane quad of every 16 lanes stored the NEXT
        // value (W % 4 != 0 build only: the other build happened to schedule differently).  Keep the registers live across
        // two wait states.
        asm volatile("s_nop 2" : "+v"(t0), "+v"(t1));
      }
      const f32x4 s_y = band_mfma(a_own, f_y, a_halo, h_y);
      const f32x4 s_gy = band_mfma(a_own, f_gy, a_halo, h_gy);
      const int r1 = r - R;
      const bool in1 = r1 >= 0 && r1 < H;
      f32x4 rn;                                                   // outside the image the coefficients are zero padding
      rn[0] = (in1 && cin[0]) ? rn1[0] : 0.f;
      rn[1] = (in1 && cin[1]) ? rn1[1] : 0.f;
      rn[2] = (in1 && cin[2]) ? rn1[2] : 0.f;
      rn[3] = (in1 && cin[3]) ? rn1[3] : 0.f;
      const f32x4 my = s_y * rn;
      const f32x4 cov = s_gy * rn - mg1 * my;
      const f32x4 A0 = cov * rd0, A1 = cov * rd1;
      wA0 = ring_push<k>(rA0, A0);
      wB0 = ring_push<k>(rB0, my - A0 * mg1);
      wA1 = ring_push<k>(rA1, A1);
      wB1 = ring_push<k>(rB1, my - A1 * mg1);
    }
  };
  static_assert(PF == 6, "the unrolled body lists PF = 6 steps");
  for (int itb = 0; itb < n_it; itb += PF) {
    step(std::integral_constant<int, 0>{}, itb);
    step(std::integral_constant<int, 1>{}, itb);
    step(std::integral_constant<int, 2>{}, itb);
    step(std::integral_constant<int, 3>{}, itb);
    step(std::integral_constant<int, 4>{}, itb);
    step(std::integral_constant<int, 5>{}, itb);
  }

  GF_VMWAIT(0);                                          // no LDS-DMA may land after the workgroup's LDS is released
  // a 9-row sum beyond the f16 range cannot be split: tell the host wrapper's fallback launch
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
  if (l == 0 && !(vmax < 65000.f)) atomicOr(flag, 1u);
}

template __global__ void gf_mfma_kernel<true, false>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
template __global__ void gf_mfma_kernel<false, false>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
template __global__ void gf_mfma_kernel<true, true>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
template __global__ void gf_mfma_kernel<false, true>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);

}  // namespace paif_gf_mfma
