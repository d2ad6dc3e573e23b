// ResidualDenseBlock (operations_m.py:435-449) of the 16-bit inference forward as ONE kernel:
//     x1 = P(c1(x));  x2 = P(c2([x, x1]));  out = P(c3([x, x1, x2])) * 0.333333 + x (+ up to two more residual maps)
// with k = 3, dilation 1, 32 channels, one shared PReLU slope.  Why: the three LDS-DMA convs of a block move 9 map
// passes (x read three times, x1 twice, x2 once, three outputs) at 4.2-5.6 TB/s -- the chip's practical copy rate -- i.e. they are
// bandwidth-bound with the matrix pipe a quarter busy; here x is read ONCE and out written ONCE (2 passes), x1 and x2 live in LDS,
// and the kernel is matrix-pipe bound: 1.22x the MFMA work (halo recompute) on a pipe that had 4x headroom.
//
// One persistent workgroup of 4 waves per CU (one wave per SIMD, 512 registers).  Output tile 8 rows x 28 columns, so that every stage
// works on 32-column row segments: x halo 14 x 34 (LDS pitch 36), x1 on 12 x 32, x2 on 10 x 32 (30 valid), out on 8 x 32 (28 valid).
// Waves = 2 cout halves x 2 row halves; products on v_mfma_f32_16x16x32_{f16,bf16} (M = 16 pixels, N = 16 couts, K = the 32 channels of
// one tap; the weights are the A operand and the pixel fragment the B operand, so a lane's 4 accumulators are 4 consecutive couts of one
// pixel): ALL weights of a wave's cout half stay in registers (54 taps x 4 VGPRs = 216); a pixel fragment read from LDS feeds the
// up-to-3 vertical taps that use its input row (252 ds_read_b128 for 504 MFMAs per wave and tile).  Stage outputs go back to LDS as
// [pixel][32 channels] 16-bit records (round-to-nearest-even, PReLU applied, zero outside the image = the next conv's padding) with the
// same 16-byte-chunk XOR swizzle as the tile (conflict-free fragment reads); the last stage parks fp32 values, and a store phase adds x
// (from the LDS tile: no second read of x), the other residual maps, rounds once, pools (optional ChannelPool) and stores 16 bytes per
// lane.  The NEXT tile's x halo is fetched into registers at the start of a tile (range-checked buffer loads: zero padding for free)
// and written to the other x buffer at its end: plain loads and LDS writes, the compiler keeps the waits.
#include <stdint.h>

#include <type_traits>

#include "paif_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TWO = 28, TH = 8;            // output tile
constexpr int PITCH = 36;                  // pixels per LDS row of every region (64 B each)
constexpr int ROWB = PITCH * 64;
constexpr int XR = TH + 6, X1R = TH + 4, X2R = TH + 2;
constexpr int X_BYTES = XR * ROWB, X1_BYTES = X1R * ROWB, X2_BYTES = X2R * ROWB;
constexpr int OUT_BYTES = TH * 32 * 128;   // [8 rows][32 px][32 ch] fp32
constexpr int X_OFF0 = 0, X_OFF1 = X_BYTES, X1_OFF = 2 * X_BYTES, X2_OFF = X1_OFF + X1_BYTES, OUT_OFF = X2_OFF + X2_BYTES;
constexpr int LDS_BYTES = OUT_OFF + OUT_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "regions exceed LDS");
constexpr int XCHUNKS = XR * PITCH * 4;    // 16-byte pieces of an x halo tile (incl. the two padding columns)
constexpr int NLD = (XCHUNKS + 255) / 256; // per thread
constexpr unsigned RSRC_W3 = 0x00020000u;
constexpr unsigned OOB = 0x80000000u;
#ifndef RDB_PD
#define RDB_PD 2      // fragment groups requested ahead of the MFMAs
#endif
#ifndef RDB_EXP
#define RDB_EXP 0     // timing experiments (wrong results): 1 one MFMA per group, 2 no fragment reads, 4 no stage-output conversion / LDS writes
#endif

struct RdbArgs {
  const void* x;          // NHWC-32 16-bit map
  const void* res[2];     // optional extra residual maps (16-bit), added to the output
  const void* wpk;        // [6 (stage, source)][9 taps][2 cout halves][64 lanes][8 x 16 bit]
  const float* prelu;     // the block's shared slope
  void* out;              // NHWC-32 16-bit map
  float* cpool;           // optional ChannelPool plane (pre-offset, 4 floats per pixel)
  float alpha;
  int nres, B, H, W, tilesX, tilesY, ntiles, reverse;
};

// 16-byte-chunk swizzle of the [pixel][4 chunks] records: physical chunk = logical ^ swz(column).  A fragment read takes, per lane
// (pixel = lane & 15, chunk = lane >> 4), 16 consecutive pixels x 4 chunks; ds_read_b128 is serviced in the lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with swz = 2 * ((column >> 2) & 1) every group touches 16 distinct
// 16-byte slots of the 256-byte bank window for EVERY column shift (brute-forced; the obvious (column >> 2) & 3 is 2-way: 8 cycles
// per read instead of 4, which made the LDS array, not the matrix pipe, the bound of the first version of this kernel)
__device__ __forceinline__ constexpr int swz(int col) { return ((col >> 2) & 1) * 2; }

template <int F> __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (F == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int F> __device__ __forceinline__ unsigned short to16(float v) {
  if constexpr (F == 2) return __builtin_bit_cast(unsigned short, (_Float16)v);
  else return __builtin_bit_cast(unsigned short, (__bf16)v);
}

// One stage: NS sources (LDS regions), output rows [R0, R0 + NR) of the stage's region, this wave's cout half.
//   source s: LDS base sbase[s], input row of (output row r, vertical tap dy) = r + dy + roff[s], column shift cshift[s] + dx.
// The (source, input row, horizontal tap) GROUPS -- two A fragments (the row's two 16-pixel segments), 2-6 MFMAs -- run as a software
// pipeline: the fragments of group g + PD are requested before the MFMAs of group g issue (one wave per SIMD: nothing else hides the
// ~100 cycles of an LDS read), order pinned with sched_barrier; the compiler's lgkmcnt waits are then partial (LDS returns in order).
template <int F, int NS, int NR, int WB>
__device__ __forceinline__ void stage(const unsigned char* lds, const int (&sbase)[3], const int (&roff)[3], const int (&cshift)[3], int R0,
                                      const unsigned (&a_off)[5][2], const u32x4 (&bw)[54], f32x4 (&acc)[6][2]) {
  constexpr int PD = RDB_PD, NB = PD + 1;
  constexpr int GPS = (NR + 2) * 3, G = NS * GPS;
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 A[NB][2];
  auto request = [&](int g) {
    const int sidx = g / GPS, rem = g - sidx * GPS, ir = rem / 3, dx = rem - ir * 3;
    const unsigned char* rowp = lds + sbase[sidx] + (R0 + roff[sidx] + ir) * ROWB;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      if constexpr (RDB_EXP & 2) A[g % NB][sg] = bw[(g + sg) % 54];
      else A[g % NB][sg] = *reinterpret_cast<const u32x4*>(rowp + a_off[cshift[sidx] + dx][sg]);
    }
  };
#pragma unroll
  for (int g = 0; g < PD; ++g) request(g);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + PD < G) request(g + PD);
    __builtin_amdgcn_sched_barrier(0);
    const int sidx = g / GPS, rem = g - sidx * GPS, ir = rem / 3, dx = rem - ir * 3;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int orow = ir - dy;
      if (orow >= 0 && orow < NR && !((RDB_EXP & 1) && dy > 0)) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
          if (!((RDB_EXP & 1) && sg > 0)) acc[orow][sg] = mma<F>(bw[WB + sidx * 9 + dy * 3 + dx], A[g % NB][sg], acc[orow][sg]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int F, bool CP>
__global__ __launch_bounds__(256, 1) void rdb_fused_kernel(RdbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = w & 1, rh = w >> 1;                  // cout half, row half
  const int n = l & 15, kq = l >> 4;
  const int H = a.H, W = a.W;

  // this workgroup's tiles: XCD x owns a contiguous range, its workgroups interleave over it (conv_dma.hip)
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  const int tpx = (a.ntiles + 7) >> 3;
  const int t_beg = xcd * tpx, t_end = min(a.ntiles, t_beg + tpx);
  const int cnt = t_beg + wg < t_end ? (t_end - t_beg - wg + nwg - 1) / nwg : 0;
  if (cnt == 0) return;
  auto tile_of = [&](int k, int& b, int& y0, int& x0) {
    const int pos = wg + min(k, cnt - 1) * nwg;
    int t = a.reverse ? t_end - 1 - pos : t_beg + pos;
    const int tx = t % a.tilesX;
    t /= a.tilesX;
    b = t / a.tilesY; y0 = (t % a.tilesY) * TH; x0 = tx * TWO;
  };

  const int map_bytes = a.B * H * W * 64;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.nres > 0 ? a.res[0] : a.x), 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.nres > 1 ? a.res[1] : a.x), 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_cp = __builtin_amdgcn_make_buffer_rsrc(CP ? (void*)a.cpool : a.out, 0, CP ? a.B * H * W * 16 - 8 : 0, RSRC_W3);

  // ---- weights of this wave's cout half: [s6][tap] -> 54 fragments, lane (n, kq) holds W[16 half + n][8 kq + j] ----
  u32x4 bw[54];
  {
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.wpk) + half * 64 + l;
#pragma unroll
    for (int i = 0; i < 54; ++i) bw[i] = wp[i * 128];
  }
  const float slope = *a.prelu;

  // ---- x halo tile fetch geometry of this thread (piece i = tid + 256 i of the 14 x 36 x 4 pieces) ----
  int f_rel[NLD];        // byte offset relative to the halo origin (y0 - 3, x0 - 3), or -1 for padding columns / past the end
  int f_rc[NLD];         // row | col << 8
  unsigned f_lds[NLD];   // byte offset inside an x buffer (swizzled chunk)
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int p = tid + 256 * i;
    const int pix = p >> 2, ch = p & 3;
    const int r = pix / PITCH, c = pix - r * PITCH;
    f_rc[i] = r | (c << 8);
    f_rel[i] = (p < XCHUNKS && c < TWO + 6) ? (r * W + c) * 64 + ch * 16 : -1;
    f_lds[i] = (unsigned)(pix * 64 + ((ch ^ swz(c)) * 16));
  }
  u32x4 xf[NLD];
  auto fetch_x = [&](int k) {
    int b, y0, x0;
    tile_of(k, b, y0, x0);
    const bool live = k < cnt;
    const int org = ((b * H + y0 - 3) * W + x0 - 3) * 64;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int gy = y0 - 3 + (f_rc[i] & 0xff), gx = x0 - 3 + (f_rc[i] >> 8);
      const bool in = live && f_rel[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      xf[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, in ? (unsigned)(org + f_rel[i]) : OOB, 0, 0);
    }
  };
  auto commit_x = [&](int buf) {
    unsigned char* xb = lds + (buf ? X_OFF1 : X_OFF0);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (tid + 256 * i < XCHUNKS) *reinterpret_cast<u32x4*>(xb + f_lds[i]) = xf[i];
  };

  // ---- A-fragment read offsets inside a region row: column = 16 sg + n + shift, the lane's channel chunk kq, swizzled ----
  unsigned a_off[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int c = 16 * sg + n + s;
      a_off[s][sg] = (unsigned)(c * 64 + ((kq ^ swz(c)) * 16));
    }
  // ---- stage outputs.  The products are formed TRANSPOSED (weights = the MFMA's A operand, rows = couts; the pixel fragment = its B
  //      operand, columns = pixels): a lane ends up with 4 CONSECUTIVE couts 16 half + 4 kq + i of ONE pixel (column 16 sg + n) --
  //      8 contiguous bytes of the [pixel][channel] record, one ds_write_b64 per (row, segment) instead of four 2-byte writes, the
  //      image mask per lane instead of per value.  chunk = (16 half + 4 kq) >> 3 = 2 half + (kq >> 1)
  const unsigned w_off = (unsigned)(n * 64 + (((2 * half + (kq >> 1)) ^ swz(n)) * 16) + 8 * (kq & 1));   // (swz(16 sg + n) = swz(n))
  const unsigned o_off = (unsigned)(n * 128 + (16 * half + 4 * kq) * 4);           // fp32 staging of the last stage: 16 bytes per lane
  auto prelu4 = [&](f32x4 v) -> f32x4 {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fmaxf(v[i], 0.f) + slope * fminf(v[i], 0.f);
    return r;
  };
  auto pack4 = [&](f32x4 v) -> uint2 { return paif::f32_to_h4<F>(make_float4(v[0], v[1], v[2], v[3])); };

  fetch_x(0);
  commit_x(0);
  int cur = 0;
  for (int k = 0; k < cnt; ++k) {
    int b, y0, x0;
    tile_of(k, b, y0, x0);
    __syncthreads();                       // x[cur] is complete; the previous tile's store phase is done with OUT and x[cur ^ 1]
    fetch_x(k + 1);                        // in flight during the three stages
    // residual maps of the store phase: 4 items per thread (row, pixel, chunk), fetched now
    u32x4 rr[2][4];
    if (a.nres > 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int item = tid + 256 * it;
        const int row = item / (TWO * 4), rem = item - row * (TWO * 4);
        const int px = rem >> 2, q = rem & 3;
        const bool ok = item < TH * TWO * 4 && y0 + row < H && x0 + px < W;
        const unsigned off = ok ? (unsigned)((((b * H + y0 + row) * W + x0 + px) * 64) + q * 16) : OOB;
        rr[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r0, off, 0, 0);
        rr[1][it] = a.nres > 1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_r1, off, 0, 0) : u32x4{0, 0, 0, 0};
      }
    }
    const int xoff = cur ? X_OFF1 : X_OFF0;
    f32x4 acc[6][2];
    // ---------------- stage 1: x1 = P(c1(x)) on rows y0 - 2 .. y0 + 9, columns x0 - 2 .. x0 + 29 ----------------
    {
      const int sbase[3] = {xoff, 0, 0}, roff[3] = {0, 0, 0}, cs[3] = {0, 0, 0};
      const int R0 = 6 * rh;
      stage<F, 1, 6, 0>(lds, sbase, roff, cs, R0, a_off, bw, acc);
      const bool colin[2] = {(unsigned)(x0 - 2 + n) < (unsigned)W, (unsigned)(x0 - 2 + 16 + n) < (unsigned)W};
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const bool rowin = (unsigned)(y0 - 2 + R0 + r) < (unsigned)H;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          if ((RDB_EXP & 4) && sg) continue;
          uint2 pk = pack4(prelu4(acc[r][sg]));
          if (!(rowin && colin[sg])) pk = make_uint2(0u, 0u);          // outside the image: the next conv's zero padding
          *reinterpret_cast<uint2*>(lds + X1_OFF + (R0 + r) * ROWB + sg * 1024 + w_off) = pk;
        }
      }
    }
    __syncthreads();
    // ---------------- stage 2: x2 = P(c2([x, x1])) on rows y0 - 1 .. y0 + 8, columns x0 - 1 .. x0 + 30 ----------------
    {
      const int sbase[3] = {xoff, X1_OFF, 0}, roff[3] = {1, 0, 0}, cs[3] = {1, 0, 0};
      const int R0 = 5 * rh;
      stage<F, 2, 5, 9>(lds, sbase, roff, cs, R0, a_off, bw, acc);
      const bool colin[2] = {(unsigned)(x0 - 1 + n) < (unsigned)W, (unsigned)(x0 - 1 + 16 + n) < (unsigned)W};
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const bool rowin = (unsigned)(y0 - 1 + R0 + r) < (unsigned)H;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          if ((RDB_EXP & 4) && sg) continue;
          uint2 pk = pack4(prelu4(acc[r][sg]));
          if (!(rowin && colin[sg])) pk = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(lds + X2_OFF + (R0 + r) * ROWB + sg * 1024 + w_off) = pk;
        }
      }
    }
    __syncthreads();
    // ---------------- stage 3: P(c3([x, x1, x2])) * alpha on rows y0 .. y0 + 7, columns x0 .. x0 + 31 -> fp32 staging ----------------
    {
      const int sbase[3] = {xoff, X1_OFF, X2_OFF}, roff[3] = {2, 1, 0}, cs[3] = {2, 1, 0};
      const int R0 = 4 * rh;
      stage<F, 3, 4, 27>(lds, sbase, roff, cs, R0, a_off, bw, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          if ((RDB_EXP & 4) && sg) continue;
          const f32x4 v = prelu4(acc[r][sg]) * a.alpha;
          *reinterpret_cast<f32x4*>(lds + OUT_OFF + ((R0 + r) * 32 + 16 * sg) * 128 + o_off) = v;
        }
    }
    __syncthreads();
    // ---------------- store phase: + x (LDS tile) + residual maps, one rounding, ChannelPool, 16-byte stores ----------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int item = tid + 256 * it;
      const int row = item / (TWO * 4), rem = item - row * (TWO * 4);
      const int px = rem >> 2, q = rem & 3;
      const bool live = item < TH * TWO * 4;
      const int rowc = live ? row : 0;
      const float4 v0 = *reinterpret_cast<const float4*>(lds + OUT_OFF + (rowc * 32 + px) * 128 + q * 32);
      const float4 v1 = *reinterpret_cast<const float4*>(lds + OUT_OFF + (rowc * 32 + px) * 128 + q * 32 + 16);
      const int xc = px + 3;
      const u32x4 xv = *reinterpret_cast<const u32x4*>(lds + xoff + (rowc + 3) * ROWB + xc * 64 + ((q ^ swz(xc)) * 16));
      float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      auto add16 = [&](u32x4 u) {
        const float4 lo = paif::h4_to_f32<F>(make_uint2(u.x, u.y)), hi = paif::h4_to_f32<F>(make_uint2(u.z, u.w));
        o[0] += lo.x; o[1] += lo.y; o[2] += lo.z; o[3] += lo.w; o[4] += hi.x; o[5] += hi.y; o[6] += hi.z; o[7] += hi.w;
      };
      add16(xv);
      if (a.nres > 0) add16(rr[0][it]);
      if (a.nres > 1) add16(rr[1][it]);
      const uint2 p0 = paif::f32_to_h4<F>(make_float4(o[0], o[1], o[2], o[3])), p1 = paif::f32_to_h4<F>(make_float4(o[4], o[5], o[6], o[7]));
      const bool ok = live && y0 + row < H && x0 + px < W;
      const unsigned goff = ok ? (unsigned)((((b * H + y0 + row) * W + x0 + px) * 64) + q * 16) : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{p0.x, p0.y, p1.x, p1.y}, rs_o, goff, 0, 2);
      if constexpr (CP) {        // the 4 lanes of a pixel are a DPP quad (item = ... + 4 px + q, 256 and 112 are multiples of 4)
        float mx = fmaxf(fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])), fmaxf(fmaxf(o[4], o[5]), fmaxf(o[6], o[7])));
        float sm = ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7]));
        mx = fmaxf(mx, paif::dpp_f<0xB1>(mx)); sm += paif::dpp_f<0xB1>(sm);
        mx = fmaxf(mx, paif::dpp_f<0x4E>(mx)); sm += paif::dpp_f<0x4E>(sm);
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v pv = {mx, sm * (1.0f / 32.0f)};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), pv), rs_cp,
                                              (ok && q == 0) ? (unsigned)(((b * H + y0 + row) * W + x0 + px) * 16) : OOB, 0, 0);
      }
    }
    commit_x(cur ^ 1);                     // the next tile's halo (nobody reads x[cur ^ 1] any more: the barrier at the loop top publishes it)
    cur ^= 1;
  }
}

// w1 [32][32][3][3], w2 [32][64][3][3], w3 [32][96][3][3] -> [s6][tap][half][64 lanes][8]: lane (n, kq), element j = W[16 half + n][32 src + 8 kq + j][tap]
template <int F>
__global__ void rdb_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                unsigned short* __restrict__ wpk) {
  const int total = 6 * 9 * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, half = (idx >> 9) & 1;
    int rest = idx >> 10;
    const int tap = rest % 9, s6 = rest / 9;
    const int n = lane & 15, kq = lane >> 4;
    const int co = 16 * half + n;
    const float* w = s6 == 0 ? w1 : (s6 < 3 ? w2 : w3);
    const int cin_tot = s6 == 0 ? 32 : (s6 < 3 ? 64 : 96);
    const int src = s6 == 0 ? 0 : (s6 < 3 ? s6 - 1 : s6 - 3);
    const float v = w[((size_t)co * cin_tot + 32 * src + 8 * kq + j) * 9 + tap];
    wpk[idx] = to16<F>(v);
  }
}

}  // namespace

extern "C" {

size_t paif_rdb_fused_wpk_floats(void) { return (size_t)6 * 9 * 2 * 64 * 8 / 2; }

int paif_rdb_fused_pack(const float* w1, const float* w2, const float* w3, float* wpk, int f16, paif_stream_t stream) {
  PAIF_REQUIRE(w1 && w2 && w3 && wpk, PAIF_EINVAL, "rdb_fused_pack: null pointer");
  hipStream_t st = paif::as_stream(stream);
  if (f16) hipLaunchKernelGGL(rdb_pack_kernel<2>, dim3(64), dim3(256), 0, st, w1, w2, w3, reinterpret_cast<unsigned short*>(wpk));
  else hipLaunchKernelGGL(rdb_pack_kernel<1>, dim3(64), dim3(256), 0, st, w1, w2, w3, reinterpret_cast<unsigned short*>(wpk));
  PAIF_LAUNCH_CHECK("rdb_fused_pack");
  return 0;
}

int paif_rdb_fused_fwd(const float* x, const float* wpk, const float* prelu, float alpha, const float* res0, const float* res1, float* out,
                       float* cpool, int f16, int reverse_tiles, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(x && wpk && prelu && out && B > 0 && H > 0 && W > 0, PAIF_EINVAL, "rdb_fused: bad arguments");
  PAIF_REQUIRE(!(res1 && !res0), PAIF_EINVAL, "rdb_fused: residual maps are packed from index 0");
  PAIF_REQUIRE((long long)B * H * W * 64 < (1ll << 31) - (1 << 24), PAIF_ENOSUP, "rdb_fused: maps beyond 2 GiB (32-bit byte offsets)");
  RdbArgs a{};
  a.x = x; a.res[0] = res0; a.res[1] = res1; a.wpk = wpk; a.prelu = prelu; a.out = out; a.cpool = cpool; a.alpha = alpha;
  a.nres = res1 ? 2 : (res0 ? 1 : 0); a.B = B; a.H = H; a.W = W;
  a.tilesX = (W + TWO - 1) / TWO; a.tilesY = (H + TH - 1) / TH; a.ntiles = B * a.tilesX * a.tilesY; a.reverse = reverse_tiles ? 1 : 0;
  hipStream_t st = paif::as_stream(stream);
  static bool raised = false;
  if (!raised) {
    const void* fns[4] = {reinterpret_cast<const void*>(&rdb_fused_kernel<1, false>), reinterpret_cast<const void*>(&rdb_fused_kernel<1, true>),
                          reinterpret_cast<const void*>(&rdb_fused_kernel<2, false>), reinterpret_cast<const void*>(&rdb_fused_kernel<2, true>)};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) {
        paif::set_error("rdb_fused: cannot raise dynamic LDS to %d: %s", LDS_BYTES, hipGetErrorString(e));
        return (int)e;
      }
    }
    raised = true;
  }
  const dim3 grid(256), blk(256);
  if (f16) {
    if (cpool) hipLaunchKernelGGL((rdb_fused_kernel<2, true>), grid, blk, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((rdb_fused_kernel<2, false>), grid, blk, LDS_BYTES, st, a);
  } else {
    if (cpool) hipLaunchKernelGGL((rdb_fused_kernel<1, true>), grid, blk, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((rdb_fused_kernel<1, false>), grid, blk, LDS_BYTES, st, a);
  }
  PAIF_LAUNCH_CHECK("rdb_fused");
  return 0;
}

}  // extern "C"
