// Shared pieces of the streaming guided-filter kernels (gf_backward.hip: the reverse pass; gf_taped.hip: the taped forward).
// Geometry: a workgroup = 48 columns (40 outputs + 4 halo columns each side) x 16 channel PAIRS = 768 threads = 3 waves per SIMD
// (<= 168 VGPRs); a wave = 4 adjacent columns x 16 pairs, so a wave's 8-byte-per-lane load covers 4 pixels x 128 B contiguous and a
// DPP row (16 lanes) is one pixel.  The B x nstrip full-height strips are laid end to end and cut into equal runs of rows, one per
// workgroup (gf_mfma2.hip's work distribution); every stream keeps a per-lane byte offset that advances by one row per iteration, and
// rows / columns outside the image and rows outside the run are range-checked away by the buffer hardware (loads return 0, stores are
// dropped): the row loop has no branches but the wave-uniform "this wave holds output columns".
#pragma once
#include <stdint.h>

#include "paif_common.h"

namespace paif_gfs {

constexpr int R = 4, KB = 2 * R + 1;
constexpr int NC = 48;                 // columns per workgroup
constexpr int NO = NC - 2 * R;         // 40 output columns
constexpr int NT = NC * 16;            // threads: (column, channel pair)
constexpr int MAXIT = 1035;            // iterations per piece (a multiple of KB); the host caps nothing: long runs are walked in pieces
constexpr unsigned RSRC_W3 = 0x00020000u;
constexpr unsigned NEVER = 0xC0000000u;   // + any row offset of an image (< 2^30 bytes, host check) stays out of range

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x2 ld2(const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}
__device__ __forceinline__ float ld1(const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ void st2(f32x2 v, const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, off, 0, 0);
}

// sum over the 16 lanes of a DPP row (= the 16 channel pairs of one pixel), fixed order, every lane ends with the total
__device__ __forceinline__ float row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }

// LDS reads of the horizontal sums as ds_read_b64 with a 16-bit immediate offset from ONE address register (hipcc pairs the nine
// column reads into ds_read2_b64, whose 8-bit offsets need an address register per quantity and column group: 10 VGPRs of a budget of
// 168).  The compiler does not know these loads are in flight: lds_wait9 is the wait, and it carries the nine values as in / out
// operands so that no use can be scheduled in front of it.
template <int OFF>
__device__ __forceinline__ f32x2 lds_rd(unsigned addr) {
  f32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lds_wait9(f32x2 (&v)[9]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]));
}
template <int QOFF>
__device__ __forceinline__ f32x2 hsum9(unsigned addr) {       // addr: the thread's column - 4, its channel pair, the row's parity
  f32x2 v[9];
  v[0] = lds_rd<QOFF + 0 * 128>(addr); v[1] = lds_rd<QOFF + 1 * 128>(addr); v[2] = lds_rd<QOFF + 2 * 128>(addr);
  v[3] = lds_rd<QOFF + 3 * 128>(addr); v[4] = lds_rd<QOFF + 4 * 128>(addr); v[5] = lds_rd<QOFF + 5 * 128>(addr);
  v[6] = lds_rd<QOFF + 6 * 128>(addr); v[7] = lds_rd<QOFF + 7 * 128>(addr); v[8] = lds_rd<QOFF + 8 * 128>(addr);
  lds_wait9(v);
  return (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + v[8];
}

template <int NQ>
struct Rings {
  f32x2 r[NQ][KB];
};
// The 9-row window sum at unrolled step K (slot K holds the newest row): always oldest row first -- the result must not depend on
// where a run of rows starts (its phase in the ring), or a sample's values would depend on its position in the batch
// (tests/test_benchmarked_size_gpu.py holds a batch of eight copies bit-equal to the B = 1 run).  One accumulator per quantity (a tree
// costs four temporaries per quantity: 70 spilled registers in the six-quantity kernel).
template <int K, int NQ>
__device__ __forceinline__ f32x2 vsum9(const Rings<NQ>& rg, int qn) {
  f32x2 v = rg.r[qn][(K + 1) % KB];
  v = v + rg.r[qn][(K + 2) % KB]; v = v + rg.r[qn][(K + 3) % KB]; v = v + rg.r[qn][(K + 4) % KB]; v = v + rg.r[qn][(K + 5) % KB];
  v = v + rg.r[qn][(K + 6) % KB]; v = v + rg.r[qn][(K + 7) % KB]; v = v + rg.r[qn][(K + 8) % KB]; v = v + rg.r[qn][K];
  return v;
}


// Host side: the run length of a launch.  One 12-wave workgroup per CU, one round: runs of >= 48 rows; more rounds only where a run
// would exceed 1000 rows (the kernels' 1 / ny table holds one piece of a run; they cut longer runs into pieces themselves).
struct Plan {
  int nstrip, rows_per_slot, grid, total_rows;
  bool fits;        // wrapping 32-bit row offsets + the range check: an image's bytes and the flattened row count must stay small
};
inline Plan make_plan(int B, int H, int W) {
  Plan p;
  p.nstrip = (W + NO - 1) / NO;
  p.fits = (size_t)H * W * 128 < 0x40000000ull && (size_t)W * 128 * 16 < 0x10000000ull && (size_t)B * p.nstrip * H < 0x7FFFFFFFull;
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 1 ? n : 2;
  }();
  const long total_rows = (long)B * p.nstrip * H;
  long nslots = total_rows / 48;
  nslots = nslots < 1 ? 1 : (nslots > cus ? cus : nslots);
  if ((total_rows + nslots - 1) / nslots > 1000) nslots = (total_rows + 999) / 1000;
  p.rows_per_slot = (int)((total_rows + nslots - 1) / nslots);
  p.grid = (int)((total_rows + p.rows_per_slot - 1) / p.rows_per_slot);
  p.total_rows = (int)total_rows;
  return p;
}

}  // namespace paif_gfs
