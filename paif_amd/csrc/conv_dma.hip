// Dense 3x3 convolution (stride 1, same padding, 32 -> 32 channels per source, 1-3 sources, dilation 1) over bf16-STORED NHWC maps
// with plain bf16 weights (PAIF_CONV_BF16 + PAIF_ST_BF16: BASELINE configs[1] "bf16"), one bf16 MFMA per product, fp32 accumulate.
// Replaces, for that configuration, the 3x3 convs of operations_m.py ResidualDenseBlock (:435-449) and ResidualModule (:451-464).
//
// Why another kernel (profiles/r03_*): with bf16 maps the tile-per-workgroup and register-prefetch kernels of
// conv_mfma.hip hold the same NUMBER of halo tiles in flight as with fp32 maps but half the BYTES -- at ~3 us loaded latency that is
// 2.2-3 TB/s whatever the arithmetic does (they measure 0.29-0.52 of the HBM roof).  Here the bytes in flight are a property of LDS,
// not of registers:
//   * one persistent workgroup of 4 waves per CU, ONE WAVE PER SIMD with the whole 512-entry register file;
//   * (tile, source) STAGES stream HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no conversion -- the stored
//     bf16 value IS the MFMA operand) into a 5-slot ring, 4 stages (87 KB) ahead of the MFMAs; zero padding = out-of-range lanes of
//     the buffer descriptor; the 16-byte chunk a lane fetches is XOR-swizzled on the SOURCE side (the DMA writes LDS linearly in lane
//     order, so any permutation of what is fetched is free): the A-operand ds_read_b128 of 16 consecutive pixels hits 16 distinct
//     16-byte bank groups;
//   * the B operand (weights) of EVERY source is resident in registers (72 per source), loaded once per launch;
//   * a wave owns 2 output rows of the 8 x 32 tile = 4 input rows: one A fragment feeds the MFMAs of up to 3 vertical taps
//     (24 ds_read_b128 per 36 MFMAs -- at one MFMA per product the LDS pipe, 128 B/clk, would otherwise cap the kernel);
//   * everything a wave sends to the vector-memory pipe is issued unconditionally in a fixed pattern per stage (DMAs, residual
//     loads, stores; dead ones carry out-of-range offsets), so the s_waitcnt vmcnt(N) that retires a stage is a compile-time
//     constant (vmcnt is in order); residual maps are fetched into a register ring one to three tiles ahead -- early enough that
//     waiting for them never waits for a younger halo tile; every LDS access is inline asm (hipcc drains vmcnt in front of every
//     LDS access it can see while an LDS-DMA is in flight).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "conv_dma.h"
#include "paif_common.h"

namespace paif_conv_dma {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TW = 32, TH = 8;          // output tile
#ifndef CD_PF
#define CD_PF 4
#endif
#ifndef CD_ST_AUX
#define CD_ST_AUX 2   // cache policy of the output stores: 2 = streaming (nt); 0 = default (A/B: tools/build_variant1.sh ... -DCD_ST_AUX=0)
#endif
#ifndef CD_EXP
#define CD_EXP 0      // experiments (timing only, wrong results): 1 no stores, 2 no MFMAs, 4 no HBM reads (every DMA lane out of range)
#endif
constexpr int PARK_WAVE = 2 * 32 * 32 * 4;   // per-wave [64 px][32 ch] fp32 transposition buffer
// Geometry of the 3x3 kernel by dilation (round 6: DIL = 2 for the composed DilConv of operations_m.py:494-506).  DIL 1: 10 x 34 halo
// pixels = 21,760 B -> 6 DMAs per wave and stage, 5 ring slots (4 stages ahead).  DIL 2: 12 x 36 = 27,648 B -> 7 DMAs, 4 slots (3 ahead).
template <int DIL>
struct Geo {
  static constexpr int TWH = TW + 2 * DIL, THH = TH + 2 * DIL;
  static constexpr int NPIX = TWH * THH;                      // halo pixels x 64 B
  static constexpr int DPW = (NPIX * 4 + 255) / 256;          // DMA instructions per wave and stage (4 waves x 64 lanes x 16 B each)
  static constexpr int SLOT = 4 * DPW * 1024;
  static constexpr int PF = DIL == 1 ? CD_PF : 3;             // stages in flight ahead of the MFMAs
  static constexpr int NSLOT = PF + 1;
  static constexpr int ROWB = TWH * 64;                       // one halo row in LDS
  static constexpr int PARK_OFF = NSLOT * SLOT;               // the transposition buffers behind the ring
  static constexpr int LDS_BYTES = PARK_OFF + 4 * PARK_WAVE;
  static_assert(LDS_BYTES <= 160 * 1024, "ring + transposition buffers exceed LDS");
  static_assert(4 * DPW * 64 >= NPIX * 4, "DMA instructions do not cover the halo tile");
  // the wave's two output rows are `base` and `base + DIL`: their 3 vertical taps read input rows base + DIL * {0, 1, 2, 3} -- four
  // fragments feed six MFMAs at either dilation (rows 2 w, 2 w + 1 would need six fragments at dilation 2)
  static __device__ __forceinline__ int base_row(int w) { return DIL == 1 ? 2 * w : (w >> 1) * 4 + (w & 1); }
};
constexpr unsigned RSRC_W3 = 0x00020000u;
constexpr unsigned OOB = 0x80000000u;   // a byte offset no map reaches (checked at launch): the hardware returns 0 / drops the store
constexpr int NSTORE = 4;               // 16-byte stores per wave and tile (2 rows x 32 px x 64 B)
constexpr int VMCAP = 63;

#define CD_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) > VMCAP ? VMCAP : (n)) : "memory")
#define CD_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define CD_WR32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(uintptr_t)lds_off, 16, voff, 0, 0, 0);
}

// The vector-memory pattern of a stage (tile t, source s), in issue order:
//   (a) s == R_STAGE: 4 x NRES residual loads of tile t + D     (b) 6 DMAs of stage g + PF     (c) s == 0: 4 stores of tile t - 1
// (the epilogue of a tile runs in the MFMA shadow of the next tile's first stage).  Residual loads are issued just early enough
// that, by the time they are waited for (stage (t + 1, 0)), every DMA in front of them in the queue has been waited for anyway.
template <int NSRC, int NRES, int PF, int DPW>
struct Sched {
  static constexpr int RL = 4 * NRES;
  static constexpr int smod(int x) { return ((x % NSRC) + NSRC) % NSRC; }
  static constexpr int R_STAGE = smod(1 - PF);
  static constexpr int D = (R_STAGE + PF - 1 - NSRC) / NSRC;        // tiles the residual loads run ahead
  static constexpr int U = D + (R_STAGE == 0 ? 2 : 1);              // residual register sets = unroll factor of the tile loop
  static_assert(PF >= 2 && D >= 0 && U <= 4, "prefetch distance out of the range the tile loop lists");
  // The wait that retires stage g: its DMAs were issued PF stages ago; YOUNGER LOADS in the queue = the DMAs of the PF - 1 stages in
  // between (+ residual loads).  Loads retire in order among themselves; stores do not retire in order with loads (observed: the
  // dropped stores of the prologue retire at once, and a count that allowed "younger stores" as outstanding let a stage start before
  // its halo tile had landed).  vmcnt <= (number of younger loads) is sound whatever the stores do: at most that many LOADS are
  // then outstanding, and the oldest outstanding loads are the youngest issued.
  static constexpr int n1 = DPW * (PF - 1);
};

// F: 16-bit format of maps and weights (1 bf16, 2 fp16 -- round 5: same data path, v_mfma_f32_32x32x16_f16, fp16 conversions in the epilogue)
// CP: also write the ChannelPool of the output map (max_c, mean_c of the un-rounded fp32 values) to a.cpool: the 4 lanes of a pixel are a
// DPP quad -- two quad permutes finish the reduction of a lane's 8 channels; lane 0 of the quad stores 8 bytes
// DIL: dilation (1, 2).  IA: input activation on the A fragments as they arrive from LDS (0 none; 2 ReLU -- max(x, 0) of a stored 16-bit
// value is a 16-bit value: one v_pk_max_i16 per register, issued behind the MFMA in front of the fragment's first use): the composed
// DilConv (ReLU -> depthwise 3x3 dilation 2 -> 1x1 -> BN, operations_m.py:494-506, as ONE dense dilated conv) is <1, NRES, F, CP, 2, 2>.
template <int NSRC, int NRES, int F, bool CP = false, int DIL = 1, int IA = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_h16_dma(Args a, int ntiles, int tilesX, int tilesY) {
  typedef Geo<DIL> G;
  constexpr int TWH = G::TWH, NPIX = G::NPIX, DPW = G::DPW, SLOT = G::SLOT, PF = G::PF, NSLOT = G::NSLOT, ROWB = G::ROWB, PARK_OFF = G::PARK_OFF;
  typedef Sched<NSRC, NRES, PF, DPW> SC;
  constexpr int U = SC::U, D = SC::D;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  asm volatile("" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem) : "memory");   // only asm touches it

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = l & 31, hh = l >> 5;
  const int wb = G::base_row(w);        // this wave's output rows: wb and wb + DIL

  // this workgroup's tiles: XCD x owns the contiguous range [x * tpx, (x + 1) * tpx); its workgroups interleave over it
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int t_beg = xcd * tpx, t_end = min(ntiles, t_beg + tpx);
  const int cnt = t_beg + wg < t_end ? (t_end - t_beg - wg + nwg - 1) / nwg : 0;
  if (cnt == 0) return;
  const int H = a.H, W = a.W;
  // tile coordinates of this workgroup's k-th tile: computed once (lane i: tiles i and i + 64), fetched with v_readlane
  auto pack_tile = [&](int k) -> unsigned {
    const int pos = wg + min(k, cnt - 1) * nwg;
    int t = a.reverse ? t_end - 1 - pos : t_beg + pos;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    return ((unsigned)(t / tilesY) << 22) | ((unsigned)ty << 11) | (unsigned)tx;
  };
  const unsigned tab0 = pack_tile(l), tab1 = pack_tile(l + 64);
  auto tile_of = [&](int k, int& b, int& y0, int& x0) -> bool {   // false: before the first / past the last tile
    const bool ok = k >= 0 && k < cnt;
    const int kk = ok ? k : 0;
    const unsigned lo = __builtin_amdgcn_readlane(tab0, kk & 63), hi = __builtin_amdgcn_readlane(tab1, kk & 63);
    const unsigned pk = kk < 64 ? lo : hi;
    b = pk >> 22; y0 = ((pk >> 11) & 2047) * TH; x0 = (pk & 2047) * TW;
    return ok;
  };

  const int map_bytes = a.B * H * W * 64;
  __amdgpu_buffer_rsrc_t rs_src[NSRC];
#pragma unroll
  for (int s = 0; s < NSRC; ++s) rs_src[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[s]), 0, map_bytes, RSRC_W3);
  __amdgpu_buffer_rsrc_t rs_res[NRES > 0 ? NRES : 1];
#pragma unroll
  for (int r = 0; r < NRES; ++r) rs_res[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res[r]), 0, map_bytes, RSRC_W3);
  const int opitch = a.cout * 2;                         // bytes per output pixel: 32 channels, or 16 (stem_out.0; one source, no residual maps)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.B * H * W * opitch, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_cp = __builtin_amdgcn_make_buffer_rsrc(CP ? (void*)a.cpool : a.out, 0, CP ? a.B * H * W * 16 - 8 : 0, RSRC_W3);   // [B,H,W,4] fp32 plane, pre-offset

  // ---- B operand: [source][tap][k-step], lane (n = l & 31, k = 8 (l >> 5) + j) -- the hi halves of the split-bf16 pack ----
  u32x4 bw[NSRC][9][2];
  {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpk), 0, NSRC * 9 * 2 * 2 * 1024, RSRC_W3);
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          bw[s][tap][ks] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)l * 16u, (((s * 9 + tap) * 2 + ks) * 2) * 1024, 0);
  }
  // epilogue constants in the [pixel][8 channels] layout a lane stores: channels 8 (l & 3) + j.  act(z) * alpha = act(alpha z)
  // for alpha > 0 (checked at launch), and every activation is  max(z, 0) + slope * min(z, 0)  (none: slope 1, ReLU: slope 0)
  float esc[8], esh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * (l & 3) + j;
    esc[j] = ((a.scale && c < a.cout) ? a.scale[c] : 1.f) * a.alpha;
    esh[j] = ((a.shift && c < a.cout) ? a.shift[c] : 0.f) * a.alpha;
  }
  const float e_slope = a.act == 1 ? *a.prelu : (a.act == 2 ? 0.f : 1.f);

  // ---- DMA geometry of this lane: instruction i of wave w moves chunk n = (w * DPW + i) * 64 + l of the slot ----
  //   n -> halo pixel n >> 2 = (row r, column c), physical 16-byte chunk n & 3 = logical chunk ^ ((c >> 2) & 3)
  int d_rel[DPW];        // byte offset relative to the tile's halo origin (y0 - 1, x0 - 1)
  int d_rc[DPW];         // r | c << 8, or -1 for the slack chunks past the 340th pixel
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int n = (w * DPW + i) * 64 + l;
    const int pi = n >> 2;
    const int r = pi / TWH, c = pi - r * TWH;
    d_rel[i] = (r * W + c) * 64 + (((n & 3) ^ ((c >> 2) & 3)) * 16);
    d_rc[i] = pi < NPIX ? (r | (c << 8)) : -1;
  }
  unsigned d_voff[DPW];  // offsets of the tile the DMAs currently target
  auto target = [&](int k) {
    int b, y0, x0;
    const bool ok = tile_of(k, b, y0, x0);
    const int org = ((b * H + y0 - DIL) * W + x0 - DIL) * 64;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int gy = y0 - DIL + (d_rc[i] & 0xff), gx = x0 - DIL + (d_rc[i] >> 8);
      const bool in = !(CD_EXP & 4) && ok && d_rc[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      d_voff[i] = in ? (unsigned)(org + d_rel[i]) : OOB;
    }
  };
  // ---- A operand: input row r (of the wave's 4), horizontal tap dx, k-step ks: lane (pixel p, k half hh) ----
  unsigned a_rd[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = p + dx * DIL;
      a_rd[dx][ks] = (unsigned)(wb * ROWB + c * 64 + (((2 * ks + hh) ^ ((c >> 2) & 3)) * 16));
    }
  // ---- epilogue geometry ----
  const unsigned a_pw = PARK_OFF + w * PARK_WAVE + (4 * hh * 32 + p) * 4;          // + (sg * 32 + (r & 3) + 8 * (r >> 2)) * 128
  const unsigned a_pr = PARK_OFF + w * PARK_WAVE + (l >> 2) * 128 + (l & 3) * 32;   // + it * 2048 (+ 16): pixel 16 it + (l >> 2), 8 channels
  const unsigned e_lane = (unsigned)l * 16u;                                      // (pixel l >> 2, chunk l & 3) inside 16 pixels of a 32-channel map
  const unsigned o_lane = 8 * (l & 3) < a.cout ? (unsigned)((l >> 2) * opitch + (l & 3) * 16) : OOB;   // the same in the OUTPUT map

  // residual maps of a tile (this wave's 2 rows): 4 x 16 B per lane and map
  u32x4 rr[U][NRES > 0 ? NRES : 1][4];
  auto issue_res = [&](auto settag, int k) {
    constexpr int SET = decltype(settag)::value;
    int b, y0, x0;
    const bool ok = tile_of(k, b, y0, x0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int y = y0 + wb + (it >> 1) * DIL, x = x0 + (it & 1) * 16 + (l >> 2);
      const bool in = ok && y < H && x < W;
      const unsigned soff = (unsigned)(((b * H + y) * W + x0 + (it & 1) * 16) * 64);
#pragma unroll
      for (int r = 0; r < NRES; ++r) rr[SET][r][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res[r], in ? e_lane : OOB, soff, 0);
    }
  };

  auto relu_frag = [](u32x4& f) {      // max(x, 0) on 8 stored 16-bit values: negative floats are negative int16 (bf16 and fp16 alike)
    // (one whole-vector max: a per-dword loop over f[d] made hipcc 7.2 form ONE v_pk_max_i16 and splat it over the fragment)
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    f = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, f), z));
  };
  f32x16 acc[2];
  u32x4 t[4][2];          // a finished tile, transposed: [16-pixel group][channels 0-3 | 4-7 of the lane's 8] fp32 bits
#pragma unroll
  for (int it = 0; it < 4; ++it) t[it][0] = t[it][1] = u32x4{0, 0, 0, 0};
  float ev[8];
  float pmx = 0.f, psm = 0.f;   // CP: running (max, sum) of the lane's 8 channels of one pixel
  int cur = NSLOT - PF;    // ring slot of the current stage; the prologue's virtual stages -PF .. -1 bring it to 0

  // Epilogue micro-step m = 9 it + j of the tile in `t` (coordinates pb, py0, px0; residual set SETP): j < 8 one channel of the
  // lane's 8 -- scale / shift, activation, residual maps --, j == 8 the bf16 pack and the 16-byte store of pixel group `it`.
  // A handful of VALU instructions each: they run in the shadow of the MFMA they are placed behind.
  auto epi = [&](auto mtag, auto settag, bool pok, int pb, int py0, int px0) {
    constexpr int M = decltype(mtag)::value, SETP = decltype(settag)::value;
    constexpr int it = M / 9, j = M % 9;
    if constexpr (j < 8) {
      const unsigned raw = j < 4 ? t[it][0][j & 3] : t[it][1][j & 3];
      float v = __builtin_fmaf(__uint_as_float(raw), esc[j], esh[j]);
      v = __builtin_fmaf(e_slope, fminf(v, 0.f), fmaxf(v, 0.f));
#pragma unroll
      for (int r = 0; r < NRES; ++r) {
        const unsigned u = rr[SETP][r][it][j >> 1];
        if constexpr (F == 2) v += (float)__builtin_bit_cast(paif::f16x2_t, u)[j & 1];
        else v += __uint_as_float((j & 1) ? (u & 0xffff0000u) : (u << 16));
      }
      ev[j] = v;
      if constexpr (CP) {                  // the fused ChannelPool accumulates with the micro-steps (two instructions each)
        pmx = j == 0 ? v : fmaxf(pmx, v);
        psm = j == 0 ? v : psm + v;
      }
    } else {
      const uint2 o0 = paif::f32_to_h4<F>(make_float4(ev[0], ev[1], ev[2], ev[3]));
      const uint2 o1 = paif::f32_to_h4<F>(make_float4(ev[4], ev[5], ev[6], ev[7]));
      const int y = py0 + wb + (it >> 1) * DIL, x = px0 + (it & 1) * 16 + (l >> 2);
      const unsigned soff = (unsigned)(((pb * H + y) * W + px0 + (it & 1) * 16) * opitch);
      u32x4 od = {o0.x, o0.y, o1.x, o1.y};
      __builtin_amdgcn_raw_buffer_store_b128(od, rs_out, (!(CD_EXP & 1) && pok && y < H && x < W) ? o_lane : OOB, soff, CD_ST_AUX);
      asm volatile("s_nop 2" : "+v"(od));                 // 128-bit store data: WAR hazard hipcc does not pad
      if constexpr (CP) {
        // the lane's 8 channels are in (pmx, psm); the pixel's other three channel groups sit in the other lanes of the DPP quad
        float mx = pmx, sm = psm;
        mx = fmaxf(mx, paif::dpp_f<0xB1>(mx)); sm += paif::dpp_f<0xB1>(sm);       // quad_perm [1,0,3,2]
        mx = fmaxf(mx, paif::dpp_f<0x4E>(mx)); sm += paif::dpp_f<0x4E>(sm);       // quad_perm [2,3,0,1]
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v pv = {mx, sm * (1.0f / 32.0f)};
        const unsigned poff = (unsigned)(((pb * H + y) * W + px0 + (it & 1) * 16) * 16);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), pv), rs_cp,
                                              (pok && y < H && x < W && (l & 3) == 0) ? (unsigned)((l >> 2) * 16) : OOB, poff, 0);
      }
    }
  };

  // One stage.  VIRT: a stage in front of the first tile -- only its vector-memory pattern is issued (stores dropped).
  auto stage = [&](auto ktag, auto stag, auto virt, int k) {
    constexpr int K = decltype(ktag)::value, S = decltype(stag)::value;
    constexpr bool VIRT = decltype(virt)::value;
    constexpr int SETP = (K + U - 1) % U;                 // residual set of the previous tile
    const unsigned sb = (unsigned)cur * SLOT;
    const int fill = cur == 0 ? NSLOT - 1 : cur - 1;      // = (cur + PF) % NSLOT: the slot read one stage ago
    u32x4 A[2][4];
    unsigned ad[3][2];
    if constexpr (!VIRT) {
      CD_VMWAIT(SC::n1);                                   // this wave's share of stage g has landed ...
      asm volatile("s_barrier" ::: "memory");              // ... and so has everybody's; every wave is done reading slot `fill`
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) ad[dx][ks] = a_rd[dx][ks] + sb;
      if constexpr (S == 0) {                              // read the previous tile back from its transposition buffer, [pixel][channel]
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          CD_RD128(t[it][0], a_pr, it * 2048);
          CD_RD128(t[it][1], a_pr, it * 2048 + 16);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) CD_RD128(A[0][r], ad[0][0], r * DIL * ROWB);
    }
    // (a) residual maps, D tiles ahead
    if constexpr (NRES > 0 && S == SC::R_STAGE) issue_res(std::integral_constant<int, (K + D) % U>{}, k + D);
    asm volatile("" ::: "memory");                      // keep (a) in front of (b) in the vector-memory queue
    // (b) the DMAs of stage g + PF
    if constexpr ((S + PF) % NSRC == 0) target(k + (S + PF) / NSRC);
#pragma unroll
    for (int i = 0; i < DPW; ++i) dma16(rs_src[(S + PF) % NSRC], d_voff[i], (unsigned)(fill * SLOT + (w * DPW + i) * 1024));
    if constexpr (VIRT) {
      if constexpr (S == 0) {   // (c)
#pragma unroll
        for (int it = 0; it < NSTORE; ++it) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rs_null, 0, 0, 0);
      }
    } else {
      int pb = 0, py0 = 0, px0 = 0;
      bool pok = false;
      if constexpr (S == 0) pok = tile_of(k - 1, pb, py0, px0);
      // 6 groups (ks, dx): 4 input-row fragments -> 6 MFMAs (2 output rows x 3 vertical taps); reads one group ahead.
      // Stage 0 carries the previous tile's epilogue, one micro-step behind each MFMA ((c): its 4 stores).
      auto mma = [&](auto gitag, auto dytag, auto jtag) {
        constexpr int gi = decltype(gitag)::value, dy = decltype(dytag)::value, j = decltype(jtag)::value;
        constexpr int ks = gi / 3, dx = gi % 3;
        if constexpr (!((CD_EXP & 2) && (dy > 0 || j > 0))) {
          const u32x4 av = A[gi & 1][j + dy], bv = bw[S][dy * 3 + dx][ks];
          if constexpr (S == 0 && gi == 0 && dy == 0) {    // first product of a tile: C = 0
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[j] = paif::mfma16<F>(av, bv, z);
          } else {
            acc[j] = paif::mfma16<F>(av, bv, acc[j]);
          }
        }
        if constexpr (S == 0) epi(std::integral_constant<int, gi * 6 + dy * 2 + j>{}, std::integral_constant<int, SETP>{}, pok, pb, py0, px0);
        if constexpr (IA == 2) {      // the fragment the NEXT product reads first: (dy, j) = (0,0) (0,1) (1,0) (1,1) (2,0) (2,1) use fragments 0 1 1 2 2 3
          constexpr int call = dy * 2 + j;
          if constexpr (call == 0) relu_frag(A[gi & 1][1]);
          if constexpr (call == 2) relu_frag(A[gi & 1][2]);
          if constexpr (call == 4) relu_frag(A[gi & 1][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto group = [&](auto gitag) {
        constexpr int gi = decltype(gitag)::value;
        if constexpr (gi + 1 < 6) {
          constexpr int ks1 = (gi + 1) / 3, dx1 = (gi + 1) % 3;
#pragma unroll
          for (int r = 0; r < 4; ++r) CD_RD128(A[(gi + 1) & 1][r], ad[dx1][ks1], r * DIL * ROWB);
        }
        if constexpr (S == 0 && gi == 0) {   // LDS operations complete in order: the read-back issued in front of A[0] is in `t` too
          asm volatile("s_waitcnt lgkmcnt(4)"
                       : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]),
                         "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1])::"memory");
        } else if constexpr (gi + 1 < 6) {
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A[gi & 1][0]), "+v"(A[gi & 1][1]), "+v"(A[gi & 1][2]), "+v"(A[gi & 1][3])::"memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[gi & 1][0]), "+v"(A[gi & 1][1]), "+v"(A[gi & 1][2]), "+v"(A[gi & 1][3])::"memory");
        }
        if constexpr (IA == 2) relu_frag(A[gi & 1][0]);
        __builtin_amdgcn_sched_barrier(0);
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 2> I2;
        mma(gitag, I0{}, I0{}); mma(gitag, I0{}, I1{});
        mma(gitag, I1{}, I0{}); mma(gitag, I1{}, I1{});
        mma(gitag, I2{}, I0{}); mma(gitag, I2{}, I1{});
      };
      group(std::integral_constant<int, 0>{});
      group(std::integral_constant<int, 1>{});
      group(std::integral_constant<int, 2>{});
      group(std::integral_constant<int, 3>{});
      group(std::integral_constant<int, 4>{});
      group(std::integral_constant<int, 5>{});
      // the tile is complete: park the raw accumulators in the wave's transposition buffer; the next stage reads them back (a read
      // issued here would be waited for in another basic block, and hipcc then copies its not-yet-loaded destination registers)
      if constexpr (S == NSRC - 1) {
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA result -> LDS-store data: wait states inline asm does not get
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[j][r];
            CD_WR32(a_pw, v, (j * 32 + (r & 3) + 8 * (r >> 2)) * 128);
          }
      }
    }
    cur = cur == NSLOT - 1 ? 0 : cur + 1;
  };
  // the last tile's epilogue has no next stage to hide in
  auto drain = [&](auto ktag, int k) {
    constexpr int K = decltype(ktag)::value;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      CD_RD128(t[it][0], a_pr, it * 2048);
      CD_RD128(t[it][1], a_pr, it * 2048 + 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1])::"memory");
    int pb, py0, px0;
    const bool pok = tile_of(k, pb, py0, px0);
#define CD_M(m) epi(std::integral_constant<int, m>{}, std::integral_constant<int, K>{}, pok, pb, py0, px0);
    CD_M(0) CD_M(1) CD_M(2) CD_M(3) CD_M(4) CD_M(5) CD_M(6) CD_M(7) CD_M(8) CD_M(9) CD_M(10) CD_M(11)
    CD_M(12) CD_M(13) CD_M(14) CD_M(15) CD_M(16) CD_M(17) CD_M(18) CD_M(19) CD_M(20) CD_M(21) CD_M(22) CD_M(23)
    CD_M(24) CD_M(25) CD_M(26) CD_M(27) CD_M(28) CD_M(29) CD_M(30) CD_M(31) CD_M(32) CD_M(33) CD_M(34) CD_M(35)
#undef CD_M
  };

  // ---- prologue: the virtual stages -PF .. -1 (their DMAs are the real stages 0 .. PF-1) ----
  auto vstage = [&](auto jtag) {
    constexpr int J = decltype(jtag)::value;                       // -PF .. -1
    constexpr int T = (J - (NSRC - 1)) / NSRC;                     // floor(J / NSRC) for J < 0
    constexpr int S = J - T * NSRC;
    constexpr int K = ((T % U) + U) % U;
    stage(std::integral_constant<int, K>{}, std::integral_constant<int, S>{}, std::true_type{}, T);
  };
  static_assert(PF >= 1 && PF <= 4, "the prologue lists up to 4 virtual stages");
  if constexpr (PF >= 4) vstage(std::integral_constant<int, -4>{});
  if constexpr (PF >= 3) vstage(std::integral_constant<int, -3>{});
  if constexpr (PF >= 2) vstage(std::integral_constant<int, -2>{});
  vstage(std::integral_constant<int, -1>{});

  auto tile_step = [&](auto ktag, int k) {
    stage(ktag, std::integral_constant<int, 0>{}, std::false_type{}, k);
    if constexpr (NSRC > 1) stage(ktag, std::integral_constant<int, 1>{}, std::false_type{}, k);
    if constexpr (NSRC > 2) stage(ktag, std::integral_constant<int, 2>{}, std::false_type{}, k);
  };
  for (int k = 0;; k += U) {
    tile_step(std::integral_constant<int, 0>{}, k);
    if (k + 1 >= cnt) { drain(std::integral_constant<int, 0>{}, k); break; }
    if constexpr (U > 1) {
      tile_step(std::integral_constant<int, 1 % U>{}, k + 1);
      if (k + 2 >= cnt) { drain(std::integral_constant<int, 1 % U>{}, k + 1); break; }
    }
    if constexpr (U > 2) {
      tile_step(std::integral_constant<int, 2 % U>{}, k + 2);
      if (k + 3 >= cnt) { drain(std::integral_constant<int, 2 % U>{}, k + 2); break; }
    }
    if constexpr (U > 3) {
      tile_step(std::integral_constant<int, 3 % U>{}, k + 3);
      if (k + 4 >= cnt) { drain(std::integral_constant<int, 3 % U>{}, k + 3); break; }
    }
  }
  CD_VMWAIT(0);          // no LDS-DMA may land after the workgroup's LDS is released
}

// ---------------------------------------------------------------------------------------------------
// 7 x 7, one source, no residual maps (ResidualModule / ECA k = 7 of the searched cell).  The register-staged kernel is bound by its
// WEIGHT traffic: 98 KB of B operand per wave and tile from L2 (the 32 KB L1 cannot hold it): 14.7 MB per CU and launch at 64 B/clk
// = 110 us of the 268 it takes -- and by one A fragment per MFMA (LDS cycles = MFMA cycles).  Here, same architecture as the 3 x 3:
//   * the weights never move again: k-step 0 of all 49 taps in registers (196), k-step 1 in LDS (49 KB, loaded once per launch);
//   * a wave's 2 output rows read 8 input rows per (horizontal tap, k-step) and each fragment feeds up to 7 vertical taps:
//     8 (+7 B) ds_read_b128 per 14 MFMAs;
//   * a stage (= a tile: 196 MFMAs per wave, 3 us) is long enough for a 2-slot ring (one tile ahead).
// ---------------------------------------------------------------------------------------------------
namespace k7 {
constexpr int P = 3;
constexpr int TWH = TW + 2 * P, THH = TH + 2 * P;   // 38 x 14 halo
constexpr int NPIX = TWH * THH;                     // 532 pixels x 64 B
constexpr int DPW = 9;                              // 4 x 9 x 1 KB = 36 KB >= 34,048 B
constexpr int SLOT = 4 * DPW * 1024;
constexpr int NSLOT = 2;
constexpr int ROWB = TWH * 64;
constexpr int WB_OFF = NSLOT * SLOT;                // k-step 1 of the weights: [49 taps][64 lanes][16 B]
constexpr int WB_BYTES = 49 * 1024;
constexpr int PARK_OFF = WB_OFF + WB_BYTES;
constexpr int LDS_BYTES = PARK_OFF + 4 * PARK_WAVE;
static_assert(LDS_BYTES <= 160 * 1024, "7x7: ring + weights + transposition buffers exceed LDS");
static_assert(4 * DPW * 64 >= NPIX * 4, "7x7: DMA instructions do not cover the halo tile");
}  // namespace k7

#define CD_WR128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

template <int F>
__global__ __launch_bounds__(256, 1) void conv7x7_h16_dma(Args a, int ntiles, int tilesX, int tilesY) {
  constexpr int P = k7::P, TWH = k7::TWH, NPIX = k7::NPIX, DPW = k7::DPW, SLOT = k7::SLOT, ROWB = k7::ROWB, WB_OFF = k7::WB_OFF,
                PARK_OFF = k7::PARK_OFF;   // shadow the 3x3 kernel's file-scope constants
  __shared__ __attribute__((aligned(16))) unsigned char smem[k7::LDS_BYTES];
  asm volatile("" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem) : "memory");   // only asm touches it

  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = l & 31, hh = l >> 5;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int t_beg = xcd * tpx, t_end = min(ntiles, t_beg + tpx);
  const int cnt = t_beg + wg < t_end ? (t_end - t_beg - wg + nwg - 1) / nwg : 0;
  if (cnt == 0) return;
  const int H = a.H, W = a.W;
  auto pack_tile = [&](int k) -> unsigned {
    const int pos = wg + min(k, cnt - 1) * nwg;
    int t = a.reverse ? t_end - 1 - pos : t_beg + pos;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    return ((unsigned)(t / tilesY) << 22) | ((unsigned)ty << 11) | (unsigned)tx;
  };
  const unsigned tab0 = pack_tile(l), tab1 = pack_tile(l + 64);
  auto tile_of = [&](int k, int& b, int& y0, int& x0) -> bool {
    const bool ok = k >= 0 && k < cnt;
    const int kk = ok ? k : 0;
    const unsigned lo = __builtin_amdgcn_readlane(tab0, kk & 63), hi = __builtin_amdgcn_readlane(tab1, kk & 63);
    const unsigned pk = kk < 64 ? lo : hi;
    b = pk >> 22; y0 = ((pk >> 11) & 2047) * TH; x0 = (pk & 2047) * TW;
    return ok;
  };
  const int map_bytes = a.B * H * W * 64;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src[0]), 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, map_bytes, RSRC_W3);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wpk), 0, 49 * 2 * 2 * 1024, RSRC_W3);

  // ---- weights: k-step 0 -> registers, k-step 1 -> LDS (thread t moves 16-byte pieces t, t + 256, ... of the 49 KB) ----
  u32x4 bw0[49];
#pragma unroll
  for (int tap = 0; tap < 49; ++tap) bw0[tap] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)l * 16u, ((tap * 2 + 0) * 2) * 1024, 0);
#pragma unroll 1
  for (int i = tid; i < 49 * 64; i += 256) {
    const int tap = i >> 6, ln = i & 63;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)ln * 16u, ((tap * 2 + 1) * 2) * 1024, 0);
    const unsigned dst = WB_OFF + (unsigned)i * 16u;
    CD_WR128(dst, v, 0);
  }
  float esc[8], esh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * (l & 3) + j;
    esc[j] = (a.scale ? a.scale[c] : 1.f) * a.alpha;
    esh[j] = (a.shift ? a.shift[c] : 0.f) * a.alpha;
  }
  const float e_slope = a.act == 1 ? *a.prelu : (a.act == 2 ? 0.f : 1.f);

  int d_rel[DPW], d_rc[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int n = (w * DPW + i) * 64 + l;
    const int pi = n >> 2;
    const int r = pi / TWH, c = pi - r * TWH;
    d_rel[i] = (r * W + c) * 64 + (((n & 3) ^ ((c >> 2) & 3)) * 16);
    d_rc[i] = pi < NPIX ? (r | (c << 8)) : -1;
  }
  unsigned d_voff[DPW];
  auto target = [&](int k) {
    int b, y0, x0;
    const bool ok = tile_of(k, b, y0, x0);
    const int org = ((b * H + y0 - P) * W + x0 - P) * 64;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int gy = y0 - P + (d_rc[i] & 0xff), gx = x0 - P + (d_rc[i] >> 8);
      const bool in = ok && d_rc[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      d_voff[i] = in ? (unsigned)(org + d_rel[i]) : OOB;
    }
  };
  unsigned a_rd[7][2];
#pragma unroll
  for (int dx = 0; dx < 7; ++dx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = p + dx;
      a_rd[dx][ks] = (unsigned)(2 * w * ROWB + c * 64 + (((2 * ks + hh) ^ ((c >> 2) & 3)) * 16));
    }
  const unsigned a_wb = WB_OFF + (unsigned)l * 16u;                                  // + tap * 1024
  const unsigned a_pw = PARK_OFF + w * PARK_WAVE + (4 * hh * 32 + p) * 4;
  const unsigned a_pr = PARK_OFF + w * PARK_WAVE + (l >> 2) * 128 + (l & 3) * 32;
  const unsigned e_lane = (unsigned)l * 16u;

  f32x16 acc[2];
  u32x4 t[4][2];
#pragma unroll
  for (int it = 0; it < 4; ++it) t[it][0] = t[it][1] = u32x4{0, 0, 0, 0};
  float ev[8];
  auto epi = [&](auto mtag, bool pok, int pb, int py0, int px0) {
    constexpr int M = decltype(mtag)::value;
    constexpr int it = M / 9, j = M % 9;
    if constexpr (j < 8) {
      const unsigned raw = j < 4 ? t[it][0][j & 3] : t[it][1][j & 3];
      float v = __builtin_fmaf(__uint_as_float(raw), esc[j], esh[j]);
      ev[j] = __builtin_fmaf(e_slope, fminf(v, 0.f), fmaxf(v, 0.f));
    } else {
      const uint2 o0 = paif::f32_to_h4<F>(make_float4(ev[0], ev[1], ev[2], ev[3]));
      const uint2 o1 = paif::f32_to_h4<F>(make_float4(ev[4], ev[5], ev[6], ev[7]));
      const int y = py0 + 2 * w + (it >> 1), x = px0 + (it & 1) * 16 + (l >> 2);
      const unsigned soff = (unsigned)(((pb * H + y) * W + px0 + (it & 1) * 16) * 64);
      u32x4 od = {o0.x, o0.y, o1.x, o1.y};
      __builtin_amdgcn_raw_buffer_store_b128(od, rs_out, (pok && y < H && x < W) ? e_lane : OOB, soff, CD_ST_AUX);
      asm volatile("s_nop 2" : "+v"(od));
    }
  };

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this thread's share of the LDS-resident weights is written
  target(0);
#pragma unroll
  for (int i = 0; i < DPW; ++i) dma16(rs_src, d_voff[i], (unsigned)((w * DPW + i) * 1024));   // tile 0 -> slot 0
  int cur = 0;
  for (int k = 0; k < cnt; ++k) {
    const unsigned sb = (unsigned)cur * SLOT;
    CD_VMWAIT(0);                                        // tile k has landed (PF = 1: no younger load may be outstanding)
    asm volatile("s_barrier" ::: "memory");              // ... everybody's share too; slot cur ^ 1 is free; the LDS weights are in
    unsigned ad[7][2];
#pragma unroll
    for (int dx = 0; dx < 7; ++dx)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ad[dx][ks] = a_rd[dx][ks] + sb;
    u32x4 A[2][8], BL[7];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      CD_RD128(t[it][0], a_pr, it * 2048);
      CD_RD128(t[it][1], a_pr, it * 2048 + 16);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) CD_RD128(A[0][r], ad[0][0], r * ROWB);
    target(k + 1);
#pragma unroll
    for (int i = 0; i < DPW; ++i) dma16(rs_src, d_voff[i], (unsigned)((cur ^ 1) * SLOT + (w * DPW + i) * 1024));
    int pb, py0, px0;
    const bool pok = tile_of(k - 1, pb, py0, px0);
    auto mma = [&](auto gitag, auto dytag, auto jtag) {
      constexpr int gi = decltype(gitag)::value, dy = decltype(dytag)::value, j = decltype(jtag)::value;
      constexpr int dx = gi >> 1, ks = gi & 1;
      const u32x4 av = A[gi & 1][j + dy];
      const u32x4 bv = ks == 0 ? bw0[dy * 7 + dx] : BL[dy];
      if constexpr (gi == 0 && dy == 0) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[j] = paif::mfma16<F>(av, bv, z);
      } else {
        acc[j] = paif::mfma16<F>(av, bv, acc[j]);
      }
      constexpr int m = gi * 14 + dy * 2 + j;
      if constexpr (m < 36) epi(std::integral_constant<int, m>{}, pok, pb, py0, px0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto group = [&](auto gitag) {
      constexpr int gi = decltype(gitag)::value;
      if constexpr (gi + 1 < 14) {                         // the next group's operands: 8 A fragments (+ 7 B fragments of k-step 1)
        constexpr int dx1 = (gi + 1) >> 1, ks1 = (gi + 1) & 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) CD_RD128(A[(gi + 1) & 1][r], ad[dx1][ks1], r * ROWB);
        if constexpr (ks1 == 1) {
#pragma unroll
          for (int dy = 0; dy < 7; ++dy) CD_RD128(BL[dy], a_wb, (dy * 7 + dx1) * 1024);
        }
      }
      // wait for THIS group's operands: everything but what was just issued (LDS operations complete in order)
      if constexpr (gi == 0) {
        asm volatile("s_waitcnt lgkmcnt(15)"
                     : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[0][4]), "+v"(A[0][5]), "+v"(A[0][6]), "+v"(A[0][7]),
                       "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0])::"memory");
        asm volatile("" : "+v"(t[3][1])::"memory");        // (an asm statement takes 30 operands at most)
      } else if constexpr (gi + 1 >= 14) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2]), "+v"(A[1][3]), "+v"(A[1][4]), "+v"(A[1][5]), "+v"(A[1][6]), "+v"(A[1][7]),
                       "+v"(BL[0]), "+v"(BL[1]), "+v"(BL[2]), "+v"(BL[3]), "+v"(BL[4]), "+v"(BL[5]), "+v"(BL[6])::"memory");
      } else if constexpr ((gi & 1) == 1) {               // k-step 1: A[1] and BL, 8 reads (the next k-step-0 group) stay outstanding
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2]), "+v"(A[1][3]), "+v"(A[1][4]), "+v"(A[1][5]), "+v"(A[1][6]), "+v"(A[1][7]),
                       "+v"(BL[0]), "+v"(BL[1]), "+v"(BL[2]), "+v"(BL[3]), "+v"(BL[4]), "+v"(BL[5]), "+v"(BL[6])::"memory");
      } else {                                             // k-step 0: A[0]; the 15 reads of the next k-step-1 group stay outstanding
        asm volatile("s_waitcnt lgkmcnt(15)"
                     : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[0][4]), "+v"(A[0][5]), "+v"(A[0][6]), "+v"(A[0][7])::"memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      typedef std::integral_constant<int, 0> J0;
      typedef std::integral_constant<int, 1> J1;
#define CD_DY(d) mma(gitag, std::integral_constant<int, d>{}, J0{}); mma(gitag, std::integral_constant<int, d>{}, J1{});
      CD_DY(0) CD_DY(1) CD_DY(2) CD_DY(3) CD_DY(4) CD_DY(5) CD_DY(6)
#undef CD_DY
    };
#define CD_G(g) group(std::integral_constant<int, g>{});
    CD_G(0) CD_G(1) CD_G(2) CD_G(3) CD_G(4) CD_G(5) CD_G(6) CD_G(7) CD_G(8) CD_G(9) CD_G(10) CD_G(11) CD_G(12) CD_G(13)
#undef CD_G
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[j][r];
        CD_WR32(a_pw, v, (j * 32 + (r & 3) + 8 * (r >> 2)) * 128);
      }
    cur ^= 1;
  }
  // the last tile's epilogue
  {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      CD_RD128(t[it][0], a_pr, it * 2048);
      CD_RD128(t[it][1], a_pr, it * 2048 + 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1])::"memory");
    int pb, py0, px0;
    const bool pok = tile_of(cnt - 1, pb, py0, px0);
#define CD_M(m) epi(std::integral_constant<int, m>{}, pok, pb, py0, px0);
    CD_M(0) CD_M(1) CD_M(2) CD_M(3) CD_M(4) CD_M(5) CD_M(6) CD_M(7) CD_M(8) CD_M(9) CD_M(10) CD_M(11)
    CD_M(12) CD_M(13) CD_M(14) CD_M(15) CD_M(16) CD_M(17) CD_M(18) CD_M(19) CD_M(20) CD_M(21) CD_M(22) CD_M(23)
    CD_M(24) CD_M(25) CD_M(26) CD_M(27) CD_M(28) CD_M(29) CD_M(30) CD_M(31) CD_M(32) CD_M(33) CD_M(34) CD_M(35)
#undef CD_M
  }
  CD_VMWAIT(0);
}

int launch_7(const Args& a, hipStream_t st) {
  const int tilesX = (a.W + TW - 1) / TW, tilesY = (a.H + TH - 1) / TH;
  if (a.f16) hipLaunchKernelGGL(conv7x7_h16_dma<2>, dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  else hipLaunchKernelGGL(conv7x7_h16_dma<1>, dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    paif::set_error("conv2d(bf16 dma 7x7): launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

template <int NSRC, int NRES>
int launch_n(const Args& a, hipStream_t st) {
  const int tilesX = (a.W + TW - 1) / TW, tilesY = (a.H + TH - 1) / TH;
  if (a.cpool) {
    if constexpr (NSRC == 3 && (NRES == 1 || NRES == 3)) {      // can_cpool(): the last conv of a ResidualDenseBlock, inside a chain or closing it
      if (a.f16) hipLaunchKernelGGL((conv3x3_h16_dma<NSRC, NRES, 2, true>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
      else hipLaunchKernelGGL((conv3x3_h16_dma<NSRC, NRES, 1, true>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
    } else {
      paif::set_error("conv2d(dma): the fused ChannelPool is built for 3 sources with 1 or 3 residual maps");
      return PAIF_ENOSUP;
    }
  } else if (a.f16) hipLaunchKernelGGL((conv3x3_h16_dma<NSRC, NRES, 2>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  else hipLaunchKernelGGL((conv3x3_h16_dma<NSRC, NRES, 1>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    paif::set_error("conv2d(h16 dma): launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// dilation 2, one source, input ReLU: the composed DilConv (1 residual map = its own `+ x`; 3 = closing a chain: + the chain's input and
// the caller's residual, with the fused ChannelPool when asked)
template <int NRES, bool CP>
int launch_d2(const Args& a, hipStream_t st) {
  const int tilesX = (a.W + TW - 1) / TW, tilesY = (a.H + TH - 1) / TH;
  if (a.f16) hipLaunchKernelGGL((conv3x3_h16_dma<1, NRES, 2, CP, 2, 2>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  else hipLaunchKernelGGL((conv3x3_h16_dma<1, NRES, 1, CP, 2, 2>), dim3(256), dim3(256), 0, st, a, a.B * tilesX * tilesY, tilesX, tilesY);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    paif::set_error("conv2d(h16 dma, dilation 2): launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace

bool eligible(int nsrc, int nres, int B, int H, int W, float alpha) {
  static const bool on = [] {
    const char* e = getenv("PAIF_CONV_DMA");    // PAIF_CONV_DMA=0: the register-staged kernels of conv_mfma.hip everywhere (A/B runs)
    return !(e && e[0] == '0');
  }();
  if (!on || nsrc < 1 || nsrc > 3 || nres < 0 || nres > 3 || !(alpha > 0.f)) return false;
  if (nsrc == 1 && nres > 1) return false;   // 4 residual register sets x 2-3 maps do not fit the register file next to the pipeline   // alpha is folded through the activation
  const long long tiles = (long long)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  // tile table: 128 tiles per workgroup, 10 + 11 + 11 bits per entry
  return tiles >= 1024 && tiles <= 128 * 256 && B < 1024 && H < 2048 * TH && W < 2048 * TW && (long long)B * H * W * 64 < (1ll << 31);
}

// dilation 2 with an input ReLU: one source, 1 or 3 residual maps (the register file holds the 3 residual sets of PF = 3 next to the pipeline)
bool eligible_d2(int nsrc, int nres, int B, int H, int W, float alpha) {
  static const bool on = [] {
    const char* e = getenv("PAIF_CONV_DMA_D2");   // PAIF_CONV_DMA_D2=0: the register-staged persistent kernel (conv_mfma.hip) for A/B runs
    return !(e && e[0] == '0');
  }();
  return on && nsrc == 1 && (nres == 1 || nres == 3) && eligible(1, 1, B, H, W, alpha);
}

bool eligible16(int nsrc, int nres, int B, int H, int W, float alpha) { return nsrc == 1 && nres == 0 && eligible(1, 0, B, H, W, alpha); }
bool eligible7(int nsrc, int nres, int B, int H, int W, float alpha) { return nsrc == 1 && nres == 0 && eligible(1, 0, B, H, W, alpha); }

bool can_cpool(int nsrc, int nres, int kh, int cout, int dil) {
  return kh == 3 && cout == 32 && ((dil == 1 && nsrc == 3 && (nres == 1 || nres == 3)) || (dil == 2 && nsrc == 1 && (nres == 1 || nres == 3)));
}

int launch(const Args& a, hipStream_t st) {
  if (a.cpool && !can_cpool(a.nsrc, a.nres, a.kh, a.cout, a.dil)) {
    paif::set_error("conv2d(dma): no fused ChannelPool for this form");
    return PAIF_ENOSUP;
  }
  if (a.kh == 7) return launch_7(a, st);
  if (a.dil == 2) {
    if (a.nsrc == 1 && a.in_relu && a.nres == 1) return a.cpool ? launch_d2<1, true>(a, st) : launch_d2<1, false>(a, st);
    if (a.nsrc == 1 && a.in_relu && a.nres == 3) return a.cpool ? launch_d2<3, true>(a, st) : launch_d2<3, false>(a, st);
    paif::set_error("conv2d(h16 dma): dilation 2 is built for one source behind an input ReLU with 1 or 3 residual maps");
    return PAIF_ENOSUP;
  }
  if (a.in_relu) {
    paif::set_error("conv2d(h16 dma): the input ReLU is built for the dilation-2 form");
    return PAIF_ENOSUP;
  }
  switch (a.nsrc * 10 + a.nres) {
    case 10: return launch_n<1, 0>(a, st);
    case 11: return launch_n<1, 1>(a, st);
    case 20: return launch_n<2, 0>(a, st);
    case 21: return launch_n<2, 1>(a, st);
    case 22: return launch_n<2, 2>(a, st);
    case 23: return launch_n<2, 3>(a, st);
    case 30: return launch_n<3, 0>(a, st);
    case 31: return launch_n<3, 1>(a, st);
    case 32: return launch_n<3, 2>(a, st);
    case 33: return launch_n<3, 3>(a, st);
    default: break;
  }
  paif::set_error("conv2d(bf16 dma): %d sources / %d residual maps not built", a.nsrc, a.nres);
  return PAIF_ENOSUP;
}

}  // namespace paif_conv_dma
