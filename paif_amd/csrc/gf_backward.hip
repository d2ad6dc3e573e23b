// Guided-filter reverse pass, round-6 streaming form (replaces the round-1 gf_bwd1 x 2 + gf_bwd2 launches of fusion_backward.hip,
// which stay in the library as the PAIF_GF_BWD=v1 cross-check).  Reverse of Cell_Decom.decomposition's two GuidedFilter(4, eps) calls
// (core/model_fusion_auto.py:522-535 under autograd; third-party guided_filter_pytorch.GuidedFilter, algorithm per
// oracle/shims/guided_filter_pytorch).  The math is fusion_backward.hip's (same stage split, same workspaces t_my / t_mgy / t_g):
//   stage 1 (both eps in ONE launch):  dA'_e = box(dLF_e g / N), db'_e = box(dLF_e / N), boxA_e = box(A_e)
//       dA_e = dA'_e - db'_e mg;  dcov_e = dA_e / (var + eps_e);  dvar_e = -sum_c dcov_e A_e
//       t_my = sum_e (db'_e - dcov_e mg);  t_mgy = sum_e dcov_e
//       t_g = (sum_e[-sum_c(db'_e A_e + dcov_e my_e) - 2 mg dvar_e], sum_e dvar_e, sum_e sum_c (dLF_e / N) boxA_e)     [my_e = b_e + A_e mg]
//   stage 2:  dy = box(t_my / N) + g box(t_mgy / N);  dg = t_g.z + sum_c y box(t_mgy / N) + box(t_g.x / N) + 2 g box(t_g.y / N),
//       routed to the arg-max (+) and arg-min (-) channel of y (the guide is max_c y - min_c y), + `add`.
//
// What changed against the round-1 kernels (HBM-bound at ~4 TB/s on 14 + 6.5 map passes per stream):
//   * both eps in one launch: t_my / t_mgy / t_g are written once (no read-modify-write pass), A_e and dLF_e of the output pixel come
//     out of the register rings (the round-1 kernel re-read them);
//   * the streaming geometry of gf_stream.h: 48-column strips (40 outputs: halo x1.2, was 32 / 24: x1.33) x 16 channel PAIRS = 768
//     threads = 3 waves per SIMD, 640 columns = 16 strips exactly; full-height strips cut into equal runs of rows, one per CU, one
//     round (was 60-row segments + 8 warm-up rows: x1.13); per-lane byte offsets advancing one row per iteration with the buffer range
//     check as the only mask; loads one row ahead;
//   * the per-pixel guide statistics are the forward's planes (gf_guide_stats_kernel: mean_g, 1/(var+eps0), 1/(var+eps1));
//   * MC form (the default of the taped networks): the tape is gf_taped.hip's (mean_y, cov), A_e = cov / (var + eps_e) re-formed where
//     the ring is filled.
// Map passes per stream: stage 1 = 4 x 1.2 (dLF_e, A_e) + 2 (b_e) + 2 (t_my, t_mgy) = 8.8 (MC: 3 x 1.2 + 1 + 2 = 6.6), stage 2 =
// 2 x 1.2 + 3 (y, add, dy) = 5.4.  Measured at B=8 480x640: 2.31 ms (round 1) -> 1.02 ms (four-map tape) / 0.94 ms (MC) per stream,
// ~4 TB/s.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gf_stream.h"

// fusion_backward.hip: the round-1 form
extern "C" int paifi_gf_bwd_input_v1(const float* guide, const float* y, const float* ab, const float* dlf, float eps0, float eps1,
                                     const float* add, float* gstat, float* t_my, float* t_mgy, float* t_g, float* dy, int B, int H, int W,
                                     paif_stream_t stream);
// guided_filter.hip: the forward's per-pixel guide statistics (planes [4][B*H*W] + the flag line)
extern "C" int paifi_gf_guide_stats(const float* guide, float* workspace, float eps0, float eps1, int B, int H, int W, paif_stream_t stream);

namespace paif_gfb {

using namespace paif_gfs;

// ------------------------------------------------------------------------------------------------------------------------------
// stage 1
// ------------------------------------------------------------------------------------------------------------------------------
// MC: the tape is (mean_y, cov) [2][B,H,W,32] of gf_taped.hip -- A_e = cov / (var + eps_e) is re-formed at the source pixel from the
// per-pixel statistics (3 map streams with halo + 1 pointwise instead of 4 + 2); otherwise the round-1 tape (A_0, b_0, A_1, b_1).
template <bool MC>
__global__ __launch_bounds__(NT) void gf_bwd1_v2_kernel(const float* __restrict__ guide, const float* __restrict__ planes,
                                                        const float* __restrict__ ab, const float* __restrict__ dlf,
                                                        float* __restrict__ t_my, float* __restrict__ t_mgy, float* __restrict__ t_g,
                                                        int B, int H, int W, int nstrip, int rows_per_slot, int total_rows) {
  __shared__ f32x2 sbuf[2][6][NC][16];              // vertical sums of one row: [parity][quantity][column][channel pair]
  __shared__ float rny_tab[MAXIT + 8];              // 1 / ny of the row each iteration streams; 0 outside the image
  const int tid = threadIdx.x;
  const int cp = tid & 15, xi = tid >> 4;
  int run_lo = blockIdx.x * rows_per_slot;
  const int run_hi = min(total_rows, run_lo + rows_per_slot);
  const size_t npix = (size_t)B * H * W;
  const size_t plane = npix * 32;
  const unsigned rowbytes = (unsigned)W * 128u, rowbytes_g = (unsigned)W * 4u, rowbytes_t = (unsigned)W * 16u;
  const int img_bytes = (int)((size_t)H * rowbytes), img_bytes_g = (int)((size_t)H * rowbytes_g);
  unsigned par = 0;

  while (run_lo < run_hi) {                         // workgroup-uniform
    const int sidx = run_lo / H;
    const int ybeg = run_lo - sidx * H, yend = min(H, ybeg + min(run_hi - run_lo, MAXIT - 2 * R - KB));
    run_lo += yend - ybeg;
    const int strip = sidx % nstrip, b = sidx / nstrip;
    const int col = strip * NO - R + xi;
    const bool colin = col >= 0 && col < W;
    const bool outcol = xi >= R && xi < NC - R && col < W;
    const size_t img = (size_t)b * H * W;
    const float rnx = colin ? 1.0f / (float)(min(col + R, W - 1) - max(col - R, 0) + 1) : 0.f;

    // one descriptor per map: the SGPR offset of a buffer instruction takes part in gfx9's range check (records - offset), so the
    // planes of ab / dlf cannot share one
    const __amdgpu_buffer_rsrc_t rs_d0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dlf + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dlf + plane + img * 32), 0, img_bytes, RSRC_W3);
    // MC: rs_a0 = cov (streamed), rs_b0 = mean_y (pointwise); rs_a1 / rs_b1 unused
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ab + (MC ? plane : 0) + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_b0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ab + (MC ? 0 : plane) + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ab + (MC ? 0 : 2 * plane) + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_b1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ab + (MC ? 0 : 3 * plane) + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_mg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + npix + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + 2 * npix + img), 0, img_bytes_g, RSRC_W3);
    // output descriptors cover the piece's rows only: a warm-up / tail row lies outside and its stores are dropped
    const int seg_rows = yend - ybeg;
    const __amdgpu_buffer_rsrc_t ro_my = __builtin_amdgcn_make_buffer_rsrc(t_my + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);
    const __amdgpu_buffer_rsrc_t ro_mgy = __builtin_amdgcn_make_buffer_rsrc(t_mgy + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);
    const __amdgpu_buffer_rsrc_t ro_g = __builtin_amdgcn_make_buffer_rsrc(t_g + (img + (size_t)ybeg * W) * 4, 0, (int)((size_t)seg_rows * rowbytes_t), RSRC_W3);

    const int r0 = ybeg - R;                        // first streamed row
    const int n_it = (seg_rows + 2 * R + KB - 1) / KB * KB;
    __syncthreads();                                // the previous piece is done with the table
    for (int i = tid; i < n_it + 1; i += NT) {
      const int row = r0 + i;
      rny_tab[i] = (row >= 0 && row < H) ? 1.0f / (float)(min(row + R, H - 1) - max(row - R, 0) + 1) : 0.f;
    }
    __syncthreads();

    // running per-lane offsets (wrapping 32-bit arithmetic on purpose)
    const unsigned lane32 = (unsigned)(col * 128 + cp * 8), lane1 = (unsigned)(col * 4);
    unsigned vs = (colin ? lane32 : NEVER) + (unsigned)r0 * rowbytes;                       // streamed row (with halo columns)
    unsigned vg = (colin ? lane1 : NEVER) + (unsigned)r0 * rowbytes_g;
    unsigned vp = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R) * rowbytes;                // output row, pointwise maps
    unsigned vp1 = (outcol ? lane1 : NEVER) + (unsigned)(r0 - R) * rowbytes_g;
    unsigned vo = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R - ybeg) * rowbytes;         // output row relative to the piece
    unsigned vot = ((outcol && cp == 0) ? (unsigned)(col * 16) : NEVER) + (unsigned)(r0 - R - ybeg) * rowbytes_t;

    // LDS byte address of sbuf[0][0][xi - R][cp] (read by the output columns only)
    const unsigned rd_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&sbuf[0][0][0][0] + (unsigned)(((xi >= R ? xi - R : 0) * 16 + cp) * 8);
    Rings<6> rg;                                    // u1_0 = dLF_0 g / N, u2_0 = dLF_0 / N, A_0, u1_1, u2_1, A_1
#pragma unroll
    for (int qn = 0; qn < 6; ++qn)
#pragma unroll
      for (int k = 0; k < KB; ++k) rg.r[qn][k] = f32x2{0.f, 0.f};

    // one row ahead
    f32x2 nd0 = ld2(rs_d0, vs), nd1 = ld2(rs_d1, vs), na0 = ld2(rs_a0, vs), na1 = {0.f, 0.f};
    float ng = ld1(rs_g, vg), nr0 = 0.f, nr1 = 0.f;           // MC: 1 / (var + eps_e) of the streamed pixel
    if constexpr (MC) { nr0 = ld1(rs_r0, vg); nr1 = ld1(rs_r1, vg); } else na1 = ld2(rs_a1, vs);
    vs += rowbytes; vg += rowbytes_g;

    auto step = [&](auto ktag, int itb) {
      constexpr int k = decltype(ktag)::value;
      const int it = itb + k;
      const float rn = rnx * rny_tab[it];           // 1 / N of the streamed pixel; 0 outside the image
      const f32x2 u20 = nd0 * rn, u21 = nd1 * rn;
      rg.r[0][k] = u20 * ng; rg.r[1][k] = u20; rg.r[2][k] = MC ? na0 * nr0 : na0;
      rg.r[3][k] = u21 * ng; rg.r[4][k] = u21; rg.r[5][k] = MC ? na0 * nr1 : na1;
      // this output row's pointwise loads, then the next row's streamed loads (vmcnt retires in order: the wait for the pointwise
      // values leaves the five streamed loads in flight)
      const f32x2 b0 = ld2(rs_b0, vp);                  // MC: mean_y
      f32x2 b1 = {0.f, 0.f};
      if constexpr (!MC) b1 = ld2(rs_b1, vp);
      const float mg = ld1(rs_mg, vp1), rd0 = ld1(rs_r0, vp1), rd1 = ld1(rs_r1, vp1);
      vp += rowbytes; vp1 += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);
      nd0 = ld2(rs_d0, vs); nd1 = ld2(rs_d1, vs); na0 = ld2(rs_a0, vs);
      ng = ld1(rs_g, vg);
      if constexpr (MC) { nr0 = ld1(rs_r0, vg); nr1 = ld1(rs_r1, vg); } else na1 = ld2(rs_a1, vs);
      vs += rowbytes; vg += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);            // the loads stay HERE (hipcc sinks them to their first use: no prefetch at all)
      // vertical 9-row sums -> LDS
      f32x2* const sw = &sbuf[par][0][xi][cp];
#pragma unroll
      for (int qn = 0; qn < 6; ++qn) sw[qn * NC * 16] = vsum9<k>(rg, qn);
      lds_barrier();
      if (xi >= R && xi < NC - R) {                 // wave-uniform (a wave = 4 columns): the strip's first and last wave hold halo columns only
        // horizontal 9-column sums
        f32x2 hs[6];
        const unsigned ra = rd_base + par * (unsigned)(6 * NC * 16 * 8);
        hs[0] = hsum9<0 * NC * 128>(ra); hs[1] = hsum9<1 * NC * 128>(ra); hs[2] = hsum9<2 * NC * 128>(ra);
        hs[3] = hsum9<3 * NC * 128>(ra); hs[4] = hsum9<4 * NC * 128>(ra); hs[5] = hsum9<5 * NC * 128>(ra);
        // pointwise, output pixel (row r - 4): its A_e and dLF_e / N sit in the rings, 4 rows back
        constexpr int kc = (k + KB - R) % KB;
        const f32x2 A0 = rg.r[2][kc], A1 = rg.r[5][kc], q20 = rg.r[1][kc], q21 = rg.r[4][kc];
        const f32x2 dcov0 = (hs[0] - hs[1] * mg) * rd0, dcov1 = (hs[3] - hs[4] * mg) * rd1;
        const f32x2 my0 = MC ? b0 : b0 + A0 * mg, my1 = MC ? b0 : b1 + A1 * mg;
        const f32x2 xv = (hs[1] * A0 + dcov0 * my0) + (hs[4] * A1 + dcov1 * my1);
        const f32x2 vv = dcov0 * A0 + dcov1 * A1;
        const f32x2 gv = q20 * hs[2] + q21 * hs[5];
        const float X = row_sum(xv[0] + xv[1]), V = -row_sum(vv[0] + vv[1]), G = row_sum(gv[0] + gv[1]);
        const f32x2 o_my = (hs[1] - dcov0 * mg) + (hs[4] - dcov1 * mg);
        const f32x2 o_mgy = dcov0 + dcov1;
        st2(o_my, ro_my, vo);
        st2(o_mgy, ro_mgy, vo);
        const f32x4 og = {-X - 2.f * mg * V, V, G, 0.f};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, og), ro_g, vot, 0, 0);
      }
      // a use behind the branch: hipcc otherwise sinks the five pointwise loads INTO the branch (behind the barrier, in front of their
      // use) and their wait takes the next row's streamed loads with it
      asm volatile("" :: "v"(b0), "v"(b1), "v"(mg), "v"(rd0), "v"(rd1));
      par ^= 1u;
      vo += rowbytes; vot += rowbytes_t;
    };
    for (int itb = 0; itb < n_it; itb += KB) {
      step(std::integral_constant<int, 0>{}, itb);
      step(std::integral_constant<int, 1>{}, itb);
      step(std::integral_constant<int, 2>{}, itb);
      step(std::integral_constant<int, 3>{}, itb);
      step(std::integral_constant<int, 4>{}, itb);
      step(std::integral_constant<int, 5>{}, itb);
      step(std::integral_constant<int, 6>{}, itb);
      step(std::integral_constant<int, 7>{}, itb);
      step(std::integral_constant<int, 8>{}, itb);
    }
  }   // pieces
}

// ------------------------------------------------------------------------------------------------------------------------------
// stage 2
// ------------------------------------------------------------------------------------------------------------------------------
template <bool ADD>
__global__ __launch_bounds__(NT) void gf_bwd2_v2_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                        const float* __restrict__ t_my, const float* __restrict__ t_mgy,
                                                        const float* __restrict__ t_g, const float* __restrict__ add,
                                                        float* __restrict__ dy, int B, int H, int W, int nstrip, int rows_per_slot,
                                                        int total_rows) {
  __shared__ f32x2 sbuf[2][2][NC][16];              // [parity][t_my / N, t_mgy / N][column][channel pair]
  __shared__ f32x2 sg[2][NC];                       // [parity][column] = vertical sums of (t_g.x, t_g.y) / N
  __shared__ float rny_tab[MAXIT + 8];
  const int tid = threadIdx.x;
  const int cp = tid & 15, xi = tid >> 4;
  int run_lo = blockIdx.x * rows_per_slot;
  const int run_hi = min(total_rows, run_lo + rows_per_slot);
  const unsigned rowbytes = (unsigned)W * 128u, rowbytes_g = (unsigned)W * 4u, rowbytes_t = (unsigned)W * 16u;
  const int img_bytes = (int)((size_t)H * rowbytes), img_bytes_g = (int)((size_t)H * rowbytes_g), img_bytes_t = (int)((size_t)H * rowbytes_t);
  unsigned par = 0;

  while (run_lo < run_hi) {
    const int sidx = run_lo / H;
    const int ybeg = run_lo - sidx * H, yend = min(H, ybeg + min(run_hi - run_lo, MAXIT - 2 * R - KB));
    run_lo += yend - ybeg;
    const int strip = sidx % nstrip, b = sidx / nstrip;
    const int col = strip * NO - R + xi;
    const bool colin = col >= 0 && col < W;
    const bool outcol = xi >= R && xi < NC - R && col < W;
    const size_t img = (size_t)b * H * W;
    const float rnx = colin ? 1.0f / (float)(min(col + R, W - 1) - max(col - R, 0) + 1) : 0.f;

    const __amdgpu_buffer_rsrc_t rs_my = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t_my + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_mgy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t_mgy + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_tg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t_g + img * 4), 0, img_bytes_t, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((ADD ? add : y) + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, img_bytes_g, RSRC_W3);
    const int seg_rows = yend - ybeg;
    const __amdgpu_buffer_rsrc_t ro_dy = __builtin_amdgcn_make_buffer_rsrc(dy + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);

    const int r0 = ybeg - R;
    const int n_it = (seg_rows + 2 * R + KB - 1) / KB * KB;
    __syncthreads();
    for (int i = tid; i < n_it + 1; i += NT) {
      const int row = r0 + i;
      rny_tab[i] = (row >= 0 && row < H) ? 1.0f / (float)(min(row + R, H - 1) - max(row - R, 0) + 1) : 0.f;
    }
    __syncthreads();

    const unsigned lane32 = (unsigned)(col * 128 + cp * 8), lane1 = (unsigned)(col * 4), lane4 = (unsigned)(col * 16);
    unsigned vs = (colin ? lane32 : NEVER) + (unsigned)r0 * rowbytes;
    unsigned vt = (colin ? lane4 : NEVER) + (unsigned)r0 * rowbytes_t;
    unsigned vp = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R) * rowbytes;
    unsigned vp1 = (outcol ? lane1 : NEVER) + (unsigned)(r0 - R) * rowbytes_g;
    unsigned vpt = (outcol ? lane4 + 8u : NEVER) + (unsigned)(r0 - R) * rowbytes_t;          // t_g.z of the output pixel
    unsigned vo = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R - ybeg) * rowbytes;

    Rings<3> rg;                                    // t_my / N, t_mgy / N, (t_g.x, t_g.y) / N
#pragma unroll
    for (int qn = 0; qn < 3; ++qn)
#pragma unroll
      for (int k = 0; k < KB; ++k) rg.r[qn][k] = f32x2{0.f, 0.f};

    f32x2 n1 = ld2(rs_my, vs), n2 = ld2(rs_mgy, vs), n3 = ld2(rs_tg, vt);
    vs += rowbytes; vt += rowbytes_t;

    auto step = [&](auto ktag, int itb) {
      constexpr int k = decltype(ktag)::value;
      const int it = itb + k;
      const float rn = rnx * rny_tab[it];
      rg.r[0][k] = n1 * rn; rg.r[1][k] = n2 * rn; rg.r[2][k] = n3 * rn;
      const f32x2 yy = ld2(rs_y, vp);
      f32x2 ad = {0.f, 0.f};
      if constexpr (ADD) ad = ld2(rs_add, vp);
      const float g = ld1(rs_g, vp1), tgz = ld1(rs_tg, vpt);
      vp += rowbytes; vp1 += rowbytes_g; vpt += rowbytes_t;
      __builtin_amdgcn_sched_barrier(0);
      n1 = ld2(rs_my, vs); n2 = ld2(rs_mgy, vs); n3 = ld2(rs_tg, vt);
      vs += rowbytes; vt += rowbytes_t;
      __builtin_amdgcn_sched_barrier(0);            // pointwise loads first, the next row's streamed loads behind them, all of them HERE
      const f32x2 v1 = vsum9<k>(rg, 0), v2 = vsum9<k>(rg, 1), v3 = vsum9<k>(rg, 2);
      sbuf[par][0][xi][cp] = v1;
      sbuf[par][1][xi][cp] = v2;
      if (cp == 0) sg[par][xi] = v3;
      lds_barrier();
      f32x2 b1 = {0.f, 0.f}, b2 = b1, b3 = b1;
      if (xi >= R && xi < NC - R) {
        b1 = sbuf[par][0][xi - R][cp]; b2 = sbuf[par][1][xi - R][cp]; b3 = sg[par][xi - R];
#pragma unroll
        for (int j = 1; j < KB; ++j) {
          b1 = b1 + sbuf[par][0][xi - R + j][cp];
          b2 = b2 + sbuf[par][1][xi - R + j][cp];
          b3 = b3 + sg[par][xi - R + j];
        }
      }
      par ^= 1u;
      f32x2 o = b1 + b2 * g;
      const f32x2 yb = yy * b2;
      const float dgy = row_sum(yb[0] + yb[1]);
      // arg-max / arg-min channel of y over the pixel's 32 channels (first index on ties): butterfly over the DPP row
      float mx = yy[0]; int imx = 2 * cp;
      if (yy[1] > mx) { mx = yy[1]; imx = 2 * cp + 1; }
      float mn = yy[0]; int imn = 2 * cp;
      if (yy[1] < mn) { mn = yy[1]; imn = 2 * cp + 1; }
      auto fold = [&](auto ctl) {
        constexpr int C = decltype(ctl)::value;
        const float ox = dppf<C>(mx); const int oix = dppi<C>(imx);
        if (ox > mx || (ox == mx && oix < imx)) { mx = ox; imx = oix; }
        const float on = dppf<C>(mn); const int oin = dppi<C>(imn);
        if (on < mn || (on == mn && oin < imn)) { mn = on; imn = oin; }
      };
      fold(std::integral_constant<int, 0xB1>{});
      fold(std::integral_constant<int, 0x4E>{});
      fold(std::integral_constant<int, 0x141>{});
      fold(std::integral_constant<int, 0x140>{});
      const float dg = tgz + dgy + b3[0] + 2.f * g * b3[1];
      o[0] += (imx == 2 * cp ? dg : 0.f) - (imn == 2 * cp ? dg : 0.f);
      o[1] += (imx == 2 * cp + 1 ? dg : 0.f) - (imn == 2 * cp + 1 ? dg : 0.f);
      if constexpr (ADD) o = o + ad;
      st2(o, ro_dy, vo);
      vo += rowbytes;
    };
    for (int itb = 0; itb < n_it; itb += KB) {
      step(std::integral_constant<int, 0>{}, itb);
      step(std::integral_constant<int, 1>{}, itb);
      step(std::integral_constant<int, 2>{}, itb);
      step(std::integral_constant<int, 3>{}, itb);
      step(std::integral_constant<int, 4>{}, itb);
      step(std::integral_constant<int, 5>{}, itb);
      step(std::integral_constant<int, 6>{}, itb);
      step(std::integral_constant<int, 7>{}, itb);
      step(std::integral_constant<int, 8>{}, itb);
    }
  }
}

}  // namespace paif_gfb

static int gf_bwd_v2_launch(bool mc, const float* guide, const float* y, const float* tape, const float* stats, const float* dlf,
                            const float* add, float* t_my, float* t_mgy, float* t_g, float* dy, int B, int H, int W, paif_stream_t stream) {
  using namespace paif_gfb;
  const Plan p = make_plan(B, H, W);
  hipStream_t st = paif::as_stream(stream);
  if (mc)
    hipLaunchKernelGGL((gf_bwd1_v2_kernel<true>), dim3(p.grid), dim3(NT), 0, st, guide, stats, tape, dlf, t_my, t_mgy, t_g, B, H, W, p.nstrip,
                       p.rows_per_slot, p.total_rows);
  else
    hipLaunchKernelGGL((gf_bwd1_v2_kernel<false>), dim3(p.grid), dim3(NT), 0, st, guide, stats, tape, dlf, t_my, t_mgy, t_g, B, H, W, p.nstrip,
                       p.rows_per_slot, p.total_rows);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(1)");
  if (add)
    hipLaunchKernelGGL((gf_bwd2_v2_kernel<true>), dim3(p.grid), dim3(NT), 0, st, guide, y, t_my, t_mgy, t_g, add, dy, B, H, W, p.nstrip,
                       p.rows_per_slot, p.total_rows);
  else
    hipLaunchKernelGGL((gf_bwd2_v2_kernel<false>), dim3(p.grid), dim3(NT), 0, st, guide, y, t_my, t_mgy, t_g, add, dy, B, H, W, p.nstrip,
                       p.rows_per_slot, p.total_rows);
  PAIF_LAUNCH_CHECK("guided_filter_bwd(2)");
  return 0;
}

extern "C" int paif_guided_filter_bwd_input(const float* guide, const float* y, const float* ab, const float* dlf, float eps0, float eps1,
                                            const float* add, float* gstat, float* t_my, float* t_mgy, float* t_g, float* dy, int B, int H,
                                            int W, paif_stream_t stream) {
  using namespace paif_gfb;
  PAIF_REQUIRE(guide && y && ab && dlf && gstat && t_my && t_mgy && t_g && dy && B > 0, PAIF_EINVAL, "guided_filter_bwd: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter_bwd: H,W must exceed 9");
  const char* e = getenv("PAIF_GF_BWD");           // read per call: the tests run both forms in one process
  PAIF_REQUIRE(!e || !strcmp(e, "v1") || !strcmp(e, "v2"), PAIF_EINVAL, "PAIF_GF_BWD must be v1 or v2 (got '%s')", e);
  if ((e && !strcmp(e, "v1")) || !make_plan(B, H, W).fits)
    return paifi_gf_bwd_input_v1(guide, y, ab, dlf, eps0, eps1, add, gstat, t_my, t_mgy, t_g, dy, B, H, W, stream);
  int rc = paifi_gf_guide_stats(guide, gstat, eps0, eps1, B, H, W, stream);
  if (rc) return rc;
  return gf_bwd_v2_launch(false, guide, y, ab, gstat, dlf, add, t_my, t_mgy, t_g, dy, B, H, W, stream);
}

// The reverse pass over gf_taped.hip's tape: mc = (mean_y, cov) [2][B,H,W,32], stats = the workspace paif_guided_filter_taped_fwd filled.
extern "C" int paif_guided_filter_bwd_input_mc(const float* guide, const float* y, const float* mc, const float* stats, const float* dlf,
                                               const float* add, float* t_my, float* t_mgy, float* t_g, float* dy, int B, int H, int W,
                                               paif_stream_t stream) {
  using namespace paif_gfb;
  PAIF_REQUIRE(guide && y && mc && stats && dlf && t_my && t_mgy && t_g && dy && B > 0, PAIF_EINVAL, "guided_filter_bwd_mc: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter_bwd_mc: H,W must exceed 9");
  PAIF_REQUIRE(make_plan(B, H, W).fits, PAIF_ENOSUP, "guided_filter_bwd_mc: %dx%dx%d exceeds the streaming kernels' 32-bit row offsets", B, H, W);
  return gf_bwd_v2_launch(true, guide, y, mc, stats, dlf, add, t_my, t_mgy, t_g, dy, B, H, W, stream);
}
