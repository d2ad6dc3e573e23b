// Guided filter (radius 4, 1-channel guide, 32-channel NHWC target, both eps), inference form -- round-4 engine.
// Replaces the reference's two GuidedFilter(4, eps) calls of Cell_Decom.decomposition (core/model_fusion_auto.py:522-535;
// third-party guided_filter_pytorch.GuidedFilter, algorithm per oracle/shims/guided_filter_pytorch).
//
// Same arithmetic as the round-3 engine (tools/parked/gf_mfma.hip: vertical 9-row windows as in-lane rings of 3-row partial sums, horizontal 9-column sums as f16
// hi/lo band-matrix products on the matrix cores, A / b never touch HBM, both eps in one pass) with a different mapping.  The
// round-3 kernel ran ONE wave per SIMD (a lane owned 4 columns: 192 registers of ring state, 320 VGPRs + 64 AGPRs) and was bound by
// that wave's own instruction issue and LDS round trips: 333 instructions per row at ~4 cycles each with nothing else to issue,
// instruction-active 64 % / parked 20 % / stalled 15 % of the wave cycles (profiles/r03_pmc_fusion_*.txt).  Here
//   * a lane owns 2 ADJACENT COLUMNS of one channel; a wave = 8 columns x 16 channels; a workgroup = 8 waves = the same 64-column
//     strip of one 16-channel half, TWO WAVES PER SIMD (96 registers of ring state per lane, <= 256 registers): the partner wave
//     issues while this one waits on LDS, the matrix pipe or memory;
//   * the horizontal sum is S[m][c] = sum_k Band[m][k] V[k][c] on v_mfma_f32_16x16x16_f16: K = 8 columns x (hi, lo); a lane's
//     two split values ARE its B fragment; D rows 4g, 4g + 1 come back as the lane's own two columns;
//   * EVERY lane publishes its fragment (8 bytes) and reads ONE neighbour fragment: the left / right 4-column halo of a wave is
//     the neighbouring wave's far half (lane ^ 32);
//   * inputs come straight from global memory into registers, three rows ahead (no LDS-DMA: the compiler keeps the waits, LDS
//     carries the halo fragments only); 1 / n comes from an 82-entry table in LDS indexed by n = ny * nx (IEEE division, once per
//     workgroup; entry 0 = 0 serves every out-of-image row and column), so the row loop has no branches;
//   * masks are buffer range checks: an out-of-image row loads through a zero-length descriptor, an out-of-image column through an
//     out-of-range offset (both return 0); invalid outputs are dropped the same way.
// f16 range: max |v| is tracked, *flag raised, the host wrapper's predicated VALU launch rewrites the output.
#include <stdint.h>
#include <type_traits>

#include "paif_common.h"

// This file is compiled twice: as itself (8 waves per workgroup, namespace paif_gf2, every output mode) and through gf_mfma2_w12.hip
// (GF2_NW = 12, namespace paif_gf2w12, the 16-bit high-frequency output modes 2 and 3 only: the modes whose 12-wave build keeps its
// row loop free of scratch accesses).
#ifndef GF2_NS
#define GF2_NS paif_gf2
#endif
namespace GF2_NS {

constexpr int R = 4;
#ifndef GF2_NW
#define GF2_NW 8
#endif
constexpr int NW = GF2_NW;        // waves per workgroup, 8 columns each (8: two waves per SIMD; 12: three, <= 168 registers)
constexpr int SC = 8 * NW;        // input columns per workgroup
constexpr int SO = SC - 4 * R;    // output columns per workgroup (48 of 64)
constexpr int PF = 6;             // unroll factor = partial-sum ring size
constexpr int NQ = 6;             // exchanged quantities: sum y, sum g*y, A0, b0, A1, b1
#ifndef GF2_DP
#define GF2_DP (GF2_NW >= 12 ? 2 : 3)
#endif
// Register diet of the 12-wave form (three waves per SIMD: <= 168 registers where the 8-wave form takes 207):
//   GF2_YD_LDS: y(r - 9) of the high-frequency output modes waits in a wave-private LDS ring (9 slots x 8 bytes per lane, one read + one
//               write per row at ONE address) instead of the 18-register in-lane delay line;
//   GF2_OWN_LDS: a lane's own fragments are read back from the halo buffer behind the barrier (3 more 16-byte LDS reads per row)
//               instead of staying live across it (12 registers).
#ifndef GF2_MINW
#define GF2_MINW (GF2_NW / 4)      // waves per SIMD the register allocation must allow
#endif
#ifndef GF2_YD_LDS
#define GF2_YD_LDS (GF2_NW >= 12)
#endif
#ifndef GF2_OWN_LDS
#define GF2_OWN_LDS (GF2_NW >= 12)
#endif
//   GF2_EARLY_AB: the (A, b) window sums are split and published at the END of the iteration that forms them, into the halo buffer of
//               the next iteration (every wave is past this iteration's barrier, so nobody still reads that buffer), instead of
//               waiting in 8 registers for the next iteration's publish;
//   GF2_DPL:    rows of per-pixel planes the two loader waves keep in flight (a register ring of 3 x GF2_DPL)
#ifndef GF2_EARLY_AB
#define GF2_EARLY_AB (GF2_NW >= 12)
#endif
#ifndef GF2_DPL
#define GF2_DPL (GF2_NW >= 12 ? 1 : GF2_DP)
#endif
constexpr int DP = GF2_DP;        // rows of input prefetch (register ring, static slots: PF % DP == 0)
constexpr int DPL = GF2_DPL;      // the same for the loader waves' plane ring
constexpr unsigned RSRC_W3 = 0x00020000u;
constexpr int MAXIT = 1032;        // iterations per workgroup (the host caps the rows per segment at 1000)

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef GF2_NOBARRIER   // diagnostic build (wrong results): how much of the row time is the workgroup barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// two fp32 values (the lane's 2 columns) -> B-operand fragment [hi(c0), hi(c1), lo(c0), lo(c1)]
__device__ __forceinline__ u32x2 split2(f32x2 v, float& vmax) {
#ifdef GF2_NOSPLIT     // diagnostic build: no hi/lo split arithmetic
  return __builtin_bit_cast(u32x2, v);
#endif
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(vmax) : "v"(vmax), "v"(v[0]), "v"(v[1]));
  u32x2 f;
  f[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
  unsigned lo;
  // lo = rn_f16(v - hi): v_fma_mix{lo,hi}_f16 evaluates fma(hi as f16, -1.0, v) in fp32 (exact) and rounds it into one half
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(f[0]), "v"(v[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(f[0]), "v"(v[1]));
  f[1] = lo;
  return f;
}

__device__ __forceinline__ f32x2 band_mfma(u32x2 a_own, u32x2 b_own, u32x2 a_halo, u32x2 b_halo) {
#ifdef GF2_NOMFMA      // diagnostic build: no matrix-core instructions
  return f32x2{__builtin_bit_cast(float, b_own[0] ^ a_own[0]) + __builtin_bit_cast(float, b_halo[0] ^ a_halo[0]), __builtin_bit_cast(float, b_own[1]) + __builtin_bit_cast(float, b_halo[1])};
#endif
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, a_own), __builtin_bit_cast(half4, b_own), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, a_halo), __builtin_bit_cast(half4, b_halo), acc, 0, 0, 0);
  return f32x2{acc[0], acc[1]};
}

// band operand: lane (m = lane & 15, gk = lane >> 4) holds A[m][k = 4 gk + j]; j = 0, 1: hi of columns colbase, colbase + 1;
// j = 2, 3: their lo halves.  D row m = 4 g + i carries output column 2 g + i for i < 2 (rows i = 2, 3 are zero).
__device__ __forceinline__ u32x2 band_operand(int m, int colbase) {
  u32x2 d = {0u, 0u};
  if ((m & 3) < 2) {
    const int oc = 2 * (m >> 2) + (m & 3);
    unsigned v = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int col = colbase + s;
      if (oc - col <= R && col - oc <= R) v |= 0x3C00u << (16 * s);      // f16 1.0
    }
    d[0] = v; d[1] = v;
  }
  return d;
}

// Round 6: TWO quantities per matrix instruction.  v_mfma_f32_16x16x32_f16: lane (n = l & 15, kg = l >> 4) holds B[8 kg + j][n], j = 0 .. 7 =
// quantity P's fragment (hi c0, hi c1, lo c0, lo c1) then quantity Q's; the band rows i = 0, 1 of every 4-row group select P's four k
// values, rows i = 2, 3 Q's -- D registers 0, 1 are P's two columns, 2, 3 are Q's (the 16x16x16 form left rows 2, 3 zero): 6 matrix
// instructions and 6 LDS operations per wave-row instead of 12 + 12.
#ifndef GF2_PAIRED
#define GF2_PAIRED 1
#endif
#if GF2_EARLY_AB && !GF2_PAIRED
#error "GF2_EARLY_AB is written for the paired form"
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x4 band_operand2(int m, int colbase) {
  const u32x2 d = band_operand((m & ~3) | (m & 1), colbase);       // the two-row pattern of output column 2 (m >> 2) + (m & 1)
  return (m & 2) ? u32x4{0u, 0u, d[0], d[1]} : u32x4{d[0], d[1], 0u, 0u};
}
__device__ __forceinline__ f32x4 band_mfma2(u32x4 a_own, u32x4 b_own, u32x4 a_halo, u32x4 b_halo) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_own), __builtin_bit_cast(half8, b_own), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a_halo), __builtin_bit_cast(half8, b_halo), acc, 0, 0, 0);
  return acc;
}

struct Ring {
  f32x2 p[PF];
  f32x2 a1, a2;
};

// 9-row window as three 3-row partial sums (direct sums of the nine terms, 4 adds per row)
template <int K>
__device__ __forceinline__ f32x2 ring_push(Ring& rg, f32x2 x) {
#ifdef GF2_NORING      // diagnostic build: no vertical window
  rg.a1 = rg.a1 + x; return rg.a1;
#endif
  const f32x2 p = (rg.a2 + rg.a1) + x;
  const f32x2 v = (p + rg.p[(K + 3) % PF]) + rg.p[K];
  rg.p[K] = p; rg.a2 = rg.a1; rg.a1 = x;
  return v;
}

// One run of rows of the strip list (see WORK DISTRIBUTION below) and one 16-channel half per workgroup of 8 waves.
//   planes: [3][B*H*W] = mean_g, 1/(var_g + eps0), 1/(var_g + eps1)   (gf_guide_stats_kernel; its fourth plane, 1/n, is not read)
// OM (output mode): 0 = the two low-frequency maps as fp32; 1 = as bf16 (round to nearest even): lanes c and c ^ 1 exchange one column
// by a DPP quad permute and store channel PAIRS as dwords (one store per eps and row); 2 (round 5, the fp16 configuration) = the two
// HIGH-frequency maps HF_e = y - LF_e as IEEE fp16, same store path: |HF| << |LF| ~ |y|, so the fp16 rounding of what the folded 1x1
// behind this block reads is ~8x smaller (core/model_fusion_auto.py:531-532 forms x - LF anyway).  y(r - 9) comes from an in-lane delay
// line (a 6-deep and a 3-deep ring, both statically indexed by the unrolled step: 18 registers, 4 moves per row).
// OM 3 (round 6): as OM 2, with y READ as IEEE fp16 -- the stem's 16-bit twin of its map (the only form of that map the fp16 forward then
// keeps: the stem writes no fp32 map at all).  HF = x16 - LF(x16): the same x16 the folded 1x1 behind the filter takes as its first source.
template <int OM>
__global__ __launch_bounds__(64 * NW, GF2_MINW) void gf2_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                     const float* __restrict__ planes, float* __restrict__ lf,
                                                     unsigned* __restrict__ flag, int B, int H, int W, int nstrip, int nslots,
                                                     int rows_per_slot, int total_rows) {
  // halo fragments: [parity][quantity][slot = wave + 1; slots 0 and NW + 1 stay zero][lane]
#if GF2_PAIRED
  __shared__ u32x4 hbuf[2][NQ / 2][NW + 2][64];                    // pairs (y, g y), (A0, b0), (A1, b1)
#else
  __shared__ u32x2 hbuf[2][NQ][NW + 2][64];
#endif
  __shared__ float rny_tab[MAXIT + 8];                             // 1 / ny of the row each iteration outputs; 0 outside the image
  __shared__ float pbuf[2][5][SC];                                  // per-pixel planes of one iteration: g(r), mean_g / rden0 / rden1 (r - 4), g(r - 9)
  __shared__ float tbuf[NW][2][8][16];                              // wave-private output transposition [eps][column][channel]
  constexpr bool YDL = GF2_YD_LDS && OM >= 2;
  __shared__ f32x2 ydl[YDL ? 9 : 1][YDL ? 64 * NW : 1];             // GF2_YD_LDS: the y delay line, [slot][thread]

  const int tid = threadIdx.x, l = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = 8-column group of the strip
  const int c = l & 15, g = l >> 4;
  // blocks b and b + 8 (same XCD under round-robin dispatch) are the two channel halves of one tile
  const int bid = blockIdx.x;
  const int chh = (bid >> 3) & 1;
  // WORK DISTRIBUTION (round 4): the B * nstrip full-height strips are laid end to end (total_rows = B * nstrip * H rows) and cut
  // into nslots equal runs of rows, one per workgroup PAIR (the two channel halves); a run that crosses a strip boundary is walked
  // as two pieces.  One round of workgroups, every CU busy to the end: at B = 8, 480 x 640 a CU streams 420 + 17..34 warm-up rows
  // (the round-3 tiling -- 7 rounds of 60-row segments, 78 iterations each -- streamed 546).
  const int slot_id = (bid >> 4) * 8 + (bid & 7);
  if (slot_id >= nslots) return;
  int run_lo = slot_id * rows_per_slot;
  const int run_hi = min(total_rows, run_lo + rows_per_slot);
  const size_t npix = (size_t)B * H * W;
  const int lc0 = 8 * q + 2 * g;                        // strip-local first column of this lane
  float vmax = 0.f;
  // zero the two permanent zero slots of both halo buffers
#if GF2_PAIRED
  for (int i = tid; i < 2 * (NQ / 2) * 2 * 64; i += 64 * NW) {
    const int ln = i & 63, sl = (i >> 6) & 1, qq = (i >> 7) % (NQ / 2), par = i / (128 * (NQ / 2));
    hbuf[par][qq][sl ? NW + 1 : 0][ln] = u32x4{0u, 0u, 0u, 0u};
  }
#else
  for (int i = tid; i < 2 * NQ * 2 * 64; i += 64 * NW) {
    const int ln = i & 63, sl = (i >> 6) & 1, qq = (i >> 7) % NQ, par = i / (128 * NQ);
    hbuf[par][qq][sl ? NW + 1 : 0][ln] = u32x2{0u, 0u};
  }
#endif
  while (run_lo < run_hi) {                              // workgroup-uniform
  const int sidx = run_lo / H;
  const int ybeg = run_lo - sidx * H, yend = min(H, ybeg + min(run_hi - run_lo, MAXIT - 32));
  run_lo += yend - ybeg;
  const int strip = sidx % nstrip, b = sidx / nstrip;
  const int X0 = strip * SO - 2 * R;
  const int col0 = X0 + lc0;
  const size_t img = (size_t)b * H * W;

  // ---- per-lane constants ----
  bool cin[2];
  float rnx[2];                                           // 1 / nx; 0 outside the image
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cc = col0 + i;
    cin[i] = cc >= 0 && cc < W;
    rnx[i] = cin[i] ? 1.0f / (float)(min(cc + R, W - 1) - max(cc - R, 0) + 1) : 0.f;
  }
  const bool outcol = lc0 >= 2 * R && lc0 < SC - 2 * R;          // the strip's 48 output columns
  constexpr bool BFO = OM != 0;                                    // 16-bit output (bf16 LF or fp16 HF)
  constexpr bool HFO = OM >= 2;                                    // the output is the high-frequency map y - LF as fp16
  constexpr bool Y16 = OM == 3;                                    // y is an fp16 map
  const bool odd = (c & 1) != 0;                                   // BFO: even lanes store column 0, odd lanes column 1 of the channel pair
#if GF2_PAIRED
  const u32x4 a_own = band_operand2(l & 15, 2 * (l >> 4));
  const u32x4 a_halo = band_operand2(l & 15, (l >> 4) < 2 ? -4 + 2 * (l >> 4) : 8 + 2 * ((l >> 4) - 2));
#else
  const u32x2 a_own = band_operand(l & 15, 2 * (l >> 4));
  const u32x2 a_halo = band_operand(l & 15, (l >> 4) < 2 ? -4 + 2 * (l >> 4) : 8 + 2 * ((l >> 4) - 2));
#endif
  const int rd_slot = (g < 2 ? q - 1 : q + 1) + 1;                 // left neighbour's right half / right neighbour's left half
  const int rd_lane = l ^ 32;
  const bool has_out = __builtin_amdgcn_readfirstlane((q != 0 && q != NW - 1) ? 1 : 0) != 0;

  // ---- addressing: NO per-row scalar arithmetic.  Every stream keeps a per-lane byte offset in a VGPR that advances by one row per
  // iteration (one v_add each); rows outside the image / the segment and lanes outside the image wrap to offsets >= num_records and
  // are range-checked away by the buffer hardware (loads return 0, stores are dropped).  The round-4 first cut computed clamped row
  // offsets and descriptor selects on the scalar unit -- 44 scalar instructions per wave and row, ONE scalar unit per CU for 8 waves:
  // 0.15 ms of the 0.49 ms launch (ablation builds).  Requires every per-image byte size < 2^31 - 2^20 (host check).
  constexpr unsigned YES = Y16 ? 2u : 4u;                          // bytes per y element
  const unsigned rowbytes_pl = (unsigned)W * 4u, rowbytes = (unsigned)W * 32u * YES;
  const int pl_bytes = (int)((size_t)H * rowbytes_pl);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, pl_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_mg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + img), 0, pl_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + npix + img), 0, pl_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + 2 * npix + img), 0, pl_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(y) + img * 32 * YES), 0, (int)((size_t)H * rowbytes), RSRC_W3);
  constexpr unsigned OES = BFO ? 2u : 4u;                 // bytes per stored output element
  const unsigned rowbytes_o = (unsigned)W * 32u * OES;
  // output descriptors cover the SEGMENT's rows only: a warm-up / tail row lies outside and its stores are dropped
  char* const o0 = reinterpret_cast<char*>(lf) + (img + (size_t)ybeg * W) * 32 * OES;
  char* const o1 = reinterpret_cast<char*>(lf) + (npix + img + (size_t)ybeg * W) * 32 * OES;
  const int seg_bytes = (int)((size_t)(yend - ybeg) * rowbytes_o);
  const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(o0, 0, seg_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(o1, 0, seg_bytes, RSRC_W3);
  // fp32 maps: one descriptor from the segment's first row of the eps0 map to its last row of the eps1 map (host check: < 2^31)
  const __amdgpu_buffer_rsrc_t ro01 = __builtin_amdgcn_make_buffer_rsrc(o0, 0, (int)(npix * 32 * OES) + seg_bytes, RSRC_W3);

  const int r0 = ybeg - 2 * R;                           // first streamed input row
  const int n_it = ((yend - ybeg) + 4 * R + 1 + PF - 1) / PF * PF;

  // 1 / ny per iteration: rny_tab[i] belongs to image row r0 - 9 + i (0 outside the image); 1 / n = (1 / nx) * (1 / ny)
  for (int i = tid; i < n_it + 8; i += 64 * NW) {
    const int row = r0 - (2 * R + 1) + i;
    rny_tab[i] = (row >= 0 && row < H) ? 1.0f / (float)(min(row + R, H - 1) - max(row - R, 0) + 1) : 0.f;
  }

  const f32x2 zero2 = {0.f, 0.f};
  Ring ry, rgy, rA0, rB0, rA1, rB1;
#pragma unroll
  for (int k = 0; k < PF; ++k) { ry.p[k] = zero2; rgy.p[k] = zero2; rA0.p[k] = zero2; rB0.p[k] = zero2; rA1.p[k] = zero2; rB1.p[k] = zero2; }
  ry.a1 = ry.a2 = rgy.a1 = rgy.a2 = rA0.a1 = rA0.a2 = rB0.a1 = rB0.a2 = rA1.a1 = rA1.a2 = rB1.a1 = rB1.a2 = zero2;
  f32x2 wA0 = zero2, wB0 = zero2, wA1 = zero2, wB1 = zero2;           // vertical (A, b) window sums of the previous iteration
  f32x2 yd6[(OM >= 2 && !YDL) ? PF : 1], yd3[(OM >= 2 && !YDL) ? 3 : 1];   // OM 2: y delayed by 6 and by 6 + 3 iterations (the output row's y)
#pragma unroll
  for (int k = 0; k < ((OM >= 2 && !YDL) ? PF : 1); ++k) yd6[k] = zero2;
#pragma unroll
  for (int k = 0; k < ((OM >= 2 && !YDL) ? 3 : 1); ++k) yd3[k] = zero2;
  int ys = 0;                                              // GF2_YD_LDS: the ring slot of this iteration (wave-uniform)
  if constexpr (YDL) {
#pragma unroll
    for (int k = 0; k < 9; ++k) ydl[k][tid] = zero2;      // wave-private: no barrier
  }

  // running offsets (wrapping 32-bit arithmetic on purpose)
  constexpr unsigned NEVER = 0x80000000u;                // + any row offset of the image stays out of range
  const unsigned lane_y = (unsigned)(((col0) * 32 + 16 * chh + c) * (int)YES);
  unsigned vy = cin[0] ? lane_y + (unsigned)r0 * rowbytes : NEVER + (unsigned)r0 * rowbytes;       // input row of the prefetch target
  const float m1 = cin[1] ? 1.f : 0.f;                   // a lane whose second column lies outside the image (odd W) reads the next row's first pixel there
  // ---- per-pixel planes: loaded ONCE per workgroup and row by the strip's two halo waves (which skip stage 2) and handed to the
  // other waves through LDS one iteration ahead.  A vector-memory instruction costs the CU's address unit ~16 cycles whatever it
  // moves: with every wave fetching its own five 8-byte plane pieces the unit was busy 640 of the ~1,760 cycles of a row (stamps).
  // wave 0: g(rr), g(rr - 9) -> planes 0, 4;   wave NW - 1: mean_g, 1/(var + eps0), 1/(var + eps1) at rr - 4 -> planes 1, 2, 3
  const bool loader = q == 0 || q == NW - 1;              // wave-uniform
  const bool ld_g = q == 0;
  // Strips wider than 64 columns (the 12-wave form: 96): each loader wave takes ALL five planes of one half of the strip's columns
  // (lanes 0 .. SC / 2 - 1) instead of two / three planes of all 64.
  constexpr bool PLH = SC > 64;
  const int pcol = X0 + l + ((PLH && !ld_g) ? SC / 2 : 0);        // the loader lane's column
  const unsigned lane_pl = (pcol >= 0 && pcol < W && (!PLH || l < SC / 2)) ? (unsigned)pcol * 4u : NEVER;
  const int pl_l = l + ((PLH && !ld_g) ? SC / 2 : 0);               // its column of pbuf (lanes >= SC / 2 of the half form: duplicates of in-range columns' slots are avoided by the mask below)
  const __amdgpu_buffer_rsrc_t rsA = (PLH || ld_g) ? rs_g : rs_mg, rsB = (PLH || ld_g) ? rs_g : rs_r0;
  unsigned vpa = lane_pl + (unsigned)((PLH || ld_g) ? r0 : r0 - R) * rowbytes_pl;
  unsigned vpb = lane_pl + (unsigned)((PLH || ld_g) ? r0 - (2 * R + 1) : r0 - R) * rowbytes_pl;
  unsigned vpc = lane_pl + (unsigned)(r0 - R) * rowbytes_pl;
  const int plA = (PLH || ld_g) ? 0 : 1, plB = (PLH || ld_g) ? 4 : 2;
  float pa[DPL], pb[DPL], pc[DPL];                        // the loader's ring: rows of iterations it + 1 .. it + DPL
  float pd[PLH ? DPL : 1], pe[PLH ? DPL : 1];
  auto plane_load = [&](int slot) {
    if (loader) {
      pa[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, vpa, 0, 0));
      pb[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, vpb, 0, 0));
      if constexpr (PLH) {
        pc[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_mg, vpc, 0, 0));
        pd[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r0, vpc, 0, 0));
        pe[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r1, vpc, 0, 0));
      } else {
        if (!ld_g) pc[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r1, vpc, 0, 0));
      }
      vpa += rowbytes_pl; vpb += rowbytes_pl; vpc += rowbytes_pl;
    }
  };
  auto plane_publish = [&](int slot, int par) {           // the loader's ring slot -> pbuf[par] (read by everyone after the next barrier)
    if constexpr (PLH) {
      if (loader && l < SC / 2) {
        pbuf[par][0][pl_l] = pa[slot];
        pbuf[par][4][pl_l] = pb[slot];
        pbuf[par][1][pl_l] = pc[slot];
        pbuf[par][2][pl_l] = pd[slot];
        pbuf[par][3][pl_l] = pe[slot];
      }
    } else if (loader) {
      pbuf[par][plA][l] = pa[slot];
      pbuf[par][plB][l] = pb[slot];
      if (!ld_g) pbuf[par][3][l] = pc[slot];
    }
  };
  // ---- output: fp32 maps leave as ONE 16-byte store per lane (4 channels of one pixel, lanes 0..31 eps0 / 32..63 eps1) after a
  // wave-private transposition through LDS (4 stores of 4 bytes per lane before: 24 of the workgroup's 80 vector-memory instructions
  // per row); bf16 maps as channel pairs exchanged by DPP (one dword store per eps)
  const unsigned rel0 = (unsigned)(r0 - (2 * R + 1) - ybeg) * rowbytes_o;                           // output row of iteration 0, relative to the segment
  const unsigned map_bytes = (unsigned)(npix * 32 * OES);                                           // eps1 map - eps0 map
  unsigned vo0;
  int ro = r0 - (2 * R + 1);                              // output row of the running iteration
  const int tcol = (l >> 2) & 7, tq = l & 3, te = l >> 5;  // transposed store: (eps, column of the wave's 8, channel quad)
  if (BFO) {
    vo0 = (outcol && (odd ? cin[1] : cin[0])) ? (unsigned)(((col0 + (odd ? 1 : 0)) * 32 + 16 * chh + (c & ~1)) * 2) + rel0 : NEVER + rel0;
  } else {
    const int sc = X0 + 8 * q + tcol;                     // the stored pixel's column
    vo0 = (has_out && sc >= 0 && sc < W) ? (unsigned)((sc * 32 + 16 * chh + 4 * tq) * 4) + (te ? map_bytes : 0u) + rel0 : NEVER + rel0;
  }

  // input prefetch ring: iteration `it` consumes slot it % DP and refills it with the row of iteration it + DP
  float py0[DP], py1[DP];
  auto prefetch = [&](int slot) {
#ifdef GF2_NOY         // diagnostic build: no input loads
    py0[slot] = __builtin_bit_cast(float, vy); py1[slot] = __builtin_bit_cast(float, vy ^ 77u);
#else
    if constexpr (Y16) {       // two 16-bit loads (the raw fp16 bits wait in the ring; converted where the row is consumed)
      py0[slot] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_y, vy, 0, 0));
      py1[slot] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs_y, vy + 64u, 0, 0));
    } else {
      py0[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, vy, 0, 0));
      py1[slot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, vy + 128u, 0, 0));
    }
#endif
    vy += rowbytes;
  };
#pragma unroll
  for (int j = 0; j < DP; ++j) prefetch(j);
  // planes of iteration 0 straight into buffer 0; iterations 1 .. DP into the loader's ring (slot j % DP)
  plane_load(0);
  plane_publish(0, 0);
#pragma unroll
  for (int j = 1; j <= DPL; ++j) plane_load(j % DPL);
#if GF2_EARLY_AB
  hbuf[0][1][q + 1][l] = u32x4{0u, 0u, 0u, 0u};            // iteration 0 reads the (A, b) fragments "of the iteration before": zeros
  hbuf[0][2][q + 1][l] = u32x4{0u, 0u, 0u, 0u};
#endif

  __syncthreads();                                       // the zero slots, the 1 / ny table and the first planes are visible

#ifdef GF2_STAMP   // diagnostic build: cycle stamps of one workgroup's waves into the (unused) fourth plane of the workspace
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(flag + 2);     // the 62 spare words behind the flag
#ifndef GF2_STAMP_WAVE
#define GF2_STAMP_WAVE 2
#endif
#define GF2_ST(j) do { if (bid == GF2_STAMP && q == GF2_STAMP_WAVE && l == 0 && it >= 30 && it < 35) stamps[(it - 30) * 6 + (j)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GF2_ST(j) do {} while (0)
#endif
  auto step = [&](auto ktag, int itb) {
    constexpr int k = decltype(ktag)::value;
    constexpr int slot = k % DP;
    constexpr int par = k & 1;                           // PF is even: the iteration's parity is static
    const int it = itb + k;
    GF2_ST(0);
    f32x2 yy;                                            // zero outside the image (range-checked loads)
    if constexpr (Y16) {
      const _Float16 h0 = __builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, py0[slot]));
      const _Float16 h1 = __builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, py1[slot]));
      yy = f32x2{(float)h0, (float)h1 * m1};
    } else {
      yy = f32x2{py0[slot], py1[slot] * m1};
    }
    f32x2 y9 = yy;
    if constexpr (HFO && YDL) {                           // y of the row stage 2 outputs in this iteration (r - 9): written 9 iterations ago
      y9 = ydl[ys][tid];
      ydl[ys][tid] = yy;                                  // same address, a wave's LDS operations execute in issue order
      ys = ys == 8 ? 0 : ys + 1;
    } else if constexpr (HFO) {
      const f32x2 y6 = yd6[k];
      yd6[k] = yy;
      y9 = yd3[k % 3];
      yd3[k % 3] = y6;
    }
    const f32x2 gv = *reinterpret_cast<const f32x2*>(&pbuf[par][0][lc0]), mg1 = *reinterpret_cast<const f32x2*>(&pbuf[par][1][lc0]);
    const f32x2 rd0 = *reinterpret_cast<const f32x2*>(&pbuf[par][2][lc0]), rd1 = *reinterpret_cast<const f32x2*>(&pbuf[par][3][lc0]);
    const f32x2 g2 = *reinterpret_cast<const f32x2*>(&pbuf[par][4][lc0]);
    prefetch(slot);
    // the loader waves: planes of the NEXT iteration -> the other buffer (visible after this iteration's barrier), ring refilled
    plane_publish((k + 1) % DPL, par ^ 1);
    plane_load((k + 1) % DPL);
    // 1 / n of the stage-2 row (r - 9) and of the stage-1 row (r - 4): (1 / nx) * (1 / ny); zero outside the image
    const float ry2 = rny_tab[it], ry1 = rny_tab[it + (R + 1)];
    const f32x2 rn2 = f32x2{rnx[0] * ry2, rnx[1] * ry2}, rn = f32x2{rnx[0] * ry1, rnx[1] * ry1};
    const u32x2 f_y = split2(ring_push<k>(ry, yy), vmax);
    const u32x2 f_gy = split2(ring_push<k>(rgy, gv * yy), vmax);
#if !GF2_EARLY_AB
    const u32x2 f_a0 = split2(wA0, vmax);
    const u32x2 f_b0 = split2(wB0, vmax);
    const u32x2 f_a1 = split2(wA1, vmax);
    const u32x2 f_b1 = split2(wB1, vmax);
#endif
    GF2_ST(1);
#ifndef GF2_NOLDS       // (diagnostic build without the halo exchange: -DGF2_NOLDS)
#if GF2_PAIRED
    const u32x4 p_yg = {f_y[0], f_y[1], f_gy[0], f_gy[1]};
    hbuf[par][0][q + 1][l] = p_yg;
#if !GF2_EARLY_AB
    const u32x4 p_0 = {f_a0[0], f_a0[1], f_b0[0], f_b0[1]}, p_1 = {f_a1[0], f_a1[1], f_b1[0], f_b1[1]};
    hbuf[par][1][q + 1][l] = p_0;
    hbuf[par][2][q + 1][l] = p_1;
#endif
#else
    hbuf[par][0][q + 1][l] = f_y;
    hbuf[par][1][q + 1][l] = f_gy;
    hbuf[par][2][q + 1][l] = f_a0;
    hbuf[par][3][q + 1][l] = f_b0;
    hbuf[par][4][q + 1][l] = f_a1;
    hbuf[par][5][q + 1][l] = f_b1;
#endif
    GF2_ST(2);
    lds_barrier();
#endif
    GF2_ST(3);
    // ---- stage 2: LF_e(r - 9) = (box(A_e) * g + box(b_e)) / n ----
    if (has_out) {                                       // the strip's first and last wave hold halo columns only
#ifdef GF2_NOLDS
      const u32x2 h_a0 = f_b1, h_b0 = f_a1, h_a1 = f_b0, h_b1 = f_a0;
#else
#if !GF2_PAIRED
      const u32x2 h_a0 = hbuf[par][2][rd_slot][rd_lane], h_b0 = hbuf[par][3][rd_slot][rd_lane];
      const u32x2 h_a1 = hbuf[par][4][rd_slot][rd_lane], h_b1 = hbuf[par][5][rd_slot][rd_lane];
#endif
#endif
#if GF2_PAIRED
      const u32x4 h_0 = hbuf[par][1][rd_slot][rd_lane], h_1 = hbuf[par][2][rd_slot][rd_lane];
#if GF2_OWN_LDS || GF2_EARLY_AB
      const u32x4 o_0 = hbuf[par][1][q + 1][l], o_1 = hbuf[par][2][q + 1][l];
#else
      const u32x4 o_0 = p_0, o_1 = p_1;
#endif
      const f32x4 s_0 = band_mfma2(a_own, o_0, a_halo, h_0), s_1 = band_mfma2(a_own, o_1, a_halo, h_1);
      const f32x2 s_a0 = {s_0[0], s_0[1]}, s_b0 = {s_0[2], s_0[3]}, s_a1 = {s_1[0], s_1[1]}, s_b1 = {s_1[2], s_1[3]};
#else
      const f32x2 s_a0 = band_mfma(a_own, f_a0, a_halo, h_a0);
      const f32x2 s_b0 = band_mfma(a_own, f_b0, a_halo, h_b0);
      const f32x2 s_a1 = band_mfma(a_own, f_a1, a_halo, h_a1);
      const f32x2 s_b1 = band_mfma(a_own, f_b1, a_halo, h_b1);
#endif
      const f32x2 out0 = (s_a0 * g2 + s_b0) * rn2;
      const f32x2 out1 = (s_a1 * g2 + s_b1) * rn2;
#ifndef GF2_NOSTORE    // (diagnostic build without the stores: -DGF2_NOSTORE)
      if constexpr (BFO) {
        auto pair_store = [&](const f32x2& lfv, const __amdgpu_buffer_rsrc_t& rs) {
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          const f32x2 ov = HFO ? y9 - lfv : lfv;         // OM 2 / 3: the high-frequency map
          // even lane keeps its column 0 and takes the partner's (channel c + 1) column 0; odd lane its column 1 and the partner's
          const float give = odd ? ov[0] : ov[1];        // what the partner needs from me: my value of ITS column
          const float got = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(give), 0xB1, 0xF, 0xF, true));
          const f32x2 v = {odd ? got : ov[0], odd ? ov[1] : got};     // (channel c & ~1, channel c | 1) of the lane's stored column
          unsigned bits;
          if constexpr (HFO) bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, paif::f16x2_t));
          else bits = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
          __builtin_amdgcn_raw_buffer_store_b32(bits, rs, vo0, 0, 2);
        };
        pair_store(out0, ro0);
        pair_store(out1, ro1);
      } else {
        tbuf[q][0][2 * g][c] = out0[0];
        tbuf[q][0][2 * g + 1][c] = out0[1];
        tbuf[q][1][2 * g][c] = out1[0];
        tbuf[q][1][2 * g + 1][c] = out1[1];
        // wave-private, and a wave's LDS operations execute in issue order: no barrier, no wait
        const f32x4 tv = *reinterpret_cast<const f32x4*>(&tbuf[q][te][tcol][4 * tq]);
        const bool rowv = ro >= ybeg && ro < yend;                   // wave-uniform: warm-up / tail rows are dropped
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv), ro01, rowv ? vo0 : NEVER, 0, 2);
      }
#endif
    }
    vo0 += rowbytes_o;
    ++ro;
    GF2_ST(4);
    // ---- stage 1: (A_e, b_e) of row r - 4, then their 9-row window sums for the next iteration ----
    {
#ifdef GF2_NOLDS
      const u32x2 h_y = f_gy, h_gy = f_y;
#else
#if !GF2_PAIRED
      const u32x2 h_y = hbuf[par][0][rd_slot][rd_lane], h_gy = hbuf[par][1][rd_slot][rd_lane];
#endif
#endif
#if GF2_PAIRED
      const u32x4 h_yg = hbuf[par][0][rd_slot][rd_lane];
#if GF2_OWN_LDS
      const u32x4 o_yg = hbuf[par][0][q + 1][l];
#else
      const u32x4 o_yg = p_yg;
#endif
      const f32x4 s_yg = band_mfma2(a_own, o_yg, a_halo, h_yg);
      const f32x2 s_y = {s_yg[0], s_yg[1]}, s_gy = {s_yg[2], s_yg[3]};
#else
      const f32x2 s_y = band_mfma(a_own, f_y, a_halo, h_y);
      const f32x2 s_gy = band_mfma(a_own, f_gy, a_halo, h_gy);
#endif
      const f32x2 my = s_y * rn;                         // rn = 0 outside the image: the coefficients there are zero padding
      const f32x2 cov = s_gy * rn - mg1 * my;
      const f32x2 A0 = cov * rd0, A1 = cov * rd1;
      wA0 = ring_push<k>(rA0, A0);
      wB0 = ring_push<k>(rB0, my - A0 * mg1);
      wA1 = ring_push<k>(rA1, A1);
      wB1 = ring_push<k>(rB1, my - A1 * mg1);
#if GF2_EARLY_AB
      const u32x2 f_a0 = split2(wA0, vmax), f_b0 = split2(wB0, vmax);
      hbuf[par ^ 1][1][q + 1][l] = u32x4{f_a0[0], f_a0[1], f_b0[0], f_b0[1]};
      const u32x2 f_a1 = split2(wA1, vmax), f_b1 = split2(wB1, vmax);
      hbuf[par ^ 1][2][q + 1][l] = u32x4{f_a1[0], f_a1[1], f_b1[0], f_b1[1]};
#endif
    }
    GF2_ST(5);
  };
  static_assert(PF == 6 && PF % DP == 0 && PF % DPL == 0, "the unrolled body lists PF = 6 steps");
  for (int itb = 0; itb < n_it; itb += PF) {
    step(std::integral_constant<int, 0>{}, itb);
    step(std::integral_constant<int, 1>{}, itb);
    step(std::integral_constant<int, 2>{}, itb);
    step(std::integral_constant<int, 3>{}, itb);
    step(std::integral_constant<int, 4>{}, itb);
    step(std::integral_constant<int, 5>{}, itb);
  }
  __syncthreads();                                       // the next piece rewrites the 1 / ny table and starts on halo buffer 0
  }   // pieces

  // a 9-row sum beyond the f16 range cannot be split: tell the host wrapper's fallback launch
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
  if (l == 0 && !(vmax < 65000.f)) atomicOr(flag, 1u);
}

// what the host launcher (guided_filter.hip) needs to know about the build
extern const int kStripOut = SO, kThreads = 64 * NW;

#if GF2_NW < 12
template __global__ void gf2_kernel<0>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
template __global__ void gf2_kernel<1>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
#endif
template __global__ void gf2_kernel<2>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);
template __global__ void gf2_kernel<3>(const float*, const float*, const float*, float*, unsigned*, int, int, int, int, int, int, int);

}  // namespace GF2_NS
