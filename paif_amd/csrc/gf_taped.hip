// Guided filter, TAPED forward (the attack loops' and the training step's forward), round-6 streaming form: replaces gf_ab + gf_lf
// (guided_filter.hip, round 1: 472 + 583 us per stream at B=8 480x640) wherever the streaming geometry fits.  Forward of
// Cell_Decom.decomposition's two GuidedFilter(4, eps) calls (core/model_fusion_auto.py:522-535; third-party
// guided_filter_pytorch.GuidedFilter, algorithm per oracle/shims/guided_filter_pytorch).
//   stage A:  mean_y = box(y) / N,  cov = box(g y) / N - mean_g mean_y          -> the tape mc = (mean_y, cov), TWO maps
//   stage B:  A_e = cov / (var_g + eps_e),  b_e = mean_y - A_e mean_g  re-formed at every source pixel from the tape and the per-pixel
//             guide statistics;  LF_e = box(A_e) / N * g + box(b_e) / N   (both eps in one launch)
// The round-1 tape held (A_0, b_0, A_1, b_1): four maps written, four read back with halo by gf_lf and again by the reverse pass.  The
// (mean_y, cov) tape is half the bytes in all three places -- A_e = cov * rd_e costs one multiply where it is used (rd_e = 1 / (var_g +
// eps_e) comes from the statistics planes the fused inference kernel already uses: gf_guide_stats_kernel).
// Map passes per stream: stage A = 1.2 (y) + 2 = 3.2, stage B = 2 x 1.2 + 2 = 4.4 (round 1: 5 + 2 x (2 x 1.33 x 1.13 + 1) = 13).
// Geometry, addressing and the LDS exchange: gf_stream.h.
#include <stdlib.h>
#include <string.h>

#include "gf_stream.h"

extern "C" int paifi_gf_guide_stats(const float* guide, float* workspace, float eps0, float eps1, int B, int H, int W, paif_stream_t stream);

namespace paif_gft {

using namespace paif_gfs;

// ---- stage A: y -> (mean_y, cov) ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gf_mc_kernel(const float* __restrict__ guide, const float* __restrict__ y,
                                                   const float* __restrict__ planes, float* __restrict__ mc, int B, int H, int W,
                                                   int nstrip, int rows_per_slot, int total_rows) {
  __shared__ f32x2 sbuf[2][2][NC][16];              // vertical sums of one row: [parity][y, g y][column][channel pair]
  __shared__ float rny_tab[MAXIT + 8];              // 1 / ny of the row each iteration OUTPUTS; 0 outside the image
  const int tid = threadIdx.x;
  const int cp = tid & 15, xi = tid >> 4;
  int run_lo = blockIdx.x * rows_per_slot;
  const int run_hi = min(total_rows, run_lo + rows_per_slot);
  const size_t npix = (size_t)B * H * W;
  const size_t plane = npix * 32;
  const unsigned rowbytes = (unsigned)W * 128u, rowbytes_g = (unsigned)W * 4u;
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

    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_mg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + img), 0, img_bytes_g, RSRC_W3);
    const int seg_rows = yend - ybeg;
    const __amdgpu_buffer_rsrc_t ro_my = __builtin_amdgcn_make_buffer_rsrc(mc + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);
    const __amdgpu_buffer_rsrc_t ro_cov = __builtin_amdgcn_make_buffer_rsrc(mc + plane + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);

    const int r0 = ybeg - R;                        // first streamed row; iteration `it` streams row r0 + it and outputs row r0 + it - R
    const int n_it = (seg_rows + 2 * R + KB - 1) / KB * KB;
    __syncthreads();                                // the previous piece is done with the table
    for (int i = tid; i < n_it + 1; i += NT) {
      const int row = r0 - R + i;
      rny_tab[i] = (row >= 0 && row < H) ? 1.0f / (float)(min(row + R, H - 1) - max(row - R, 0) + 1) : 0.f;
    }
    __syncthreads();

    const unsigned lane32 = (unsigned)(col * 128 + cp * 8), lane1 = (unsigned)(col * 4);
    unsigned vs = (colin ? lane32 : NEVER) + (unsigned)r0 * rowbytes;
    unsigned vg = (colin ? lane1 : NEVER) + (unsigned)r0 * rowbytes_g;
    unsigned vp1 = (outcol ? lane1 : NEVER) + (unsigned)(r0 - R) * rowbytes_g;
    unsigned vo = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R - ybeg) * rowbytes;
    const unsigned rd_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&sbuf[0][0][0][0] +
                             (unsigned)(((xi >= R ? xi - R : 0) * 16 + cp) * 8);
    Rings<2> rg;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int k = 0; k < KB; ++k) rg.r[qn][k] = f32x2{0.f, 0.f};

    f32x2 ny = ld2(rs_y, vs);
    float ng = ld1(rs_g, vg);
    vs += rowbytes; vg += rowbytes_g;

    auto step = [&](auto ktag, int itb) {
      constexpr int k = decltype(ktag)::value;
      const int it = itb + k;
      rg.r[0][k] = ny; rg.r[1][k] = ny * ng;
      const float mg = ld1(rs_mg, vp1);
      vp1 += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);
      ny = ld2(rs_y, vs); ng = ld1(rs_g, vg);
      vs += rowbytes; vg += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);            // the loads stay HERE (hipcc sinks them to their first use: no prefetch at all)
      sbuf[par][0][xi][cp] = vsum9<k>(rg, 0);
      sbuf[par][1][xi][cp] = vsum9<k>(rg, 1);
      lds_barrier();
      if (xi >= R && xi < NC - R) {                 // wave-uniform
        const unsigned ra = rd_base + par * (unsigned)(2 * NC * 16 * 8);
        const f32x2 by = hsum9<0>(ra), bgy = hsum9<NC * 128>(ra);
        const float rn = rnx * rny_tab[it];         // 1 / N of the output pixel
        const f32x2 my = by * rn;
        const f32x2 cov = bgy * rn - my * mg;
        st2(my, ro_my, vo);
        st2(cov, ro_cov, vo);
      }
      asm volatile("" :: "v"(mg));                  // a use behind the branch (gf_backward.hip: the load must not sink into it)
      par ^= 1u;
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

// ---- stage B: (mean_y, cov) -> LF_0, LF_1 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gf_lf_mc_kernel(const float* __restrict__ guide, const float* __restrict__ planes,
                                                      const float* __restrict__ mc, float* __restrict__ lf, int B, int H, int W,
                                                      int nstrip, int rows_per_slot, int total_rows) {
  __shared__ f32x2 sbuf[2][4][NC][16];              // [parity][A_0, b_0, A_1, b_1][column][channel pair]
  __shared__ float rny_tab[MAXIT + 8];
  const int tid = threadIdx.x;
  const int cp = tid & 15, xi = tid >> 4;
  int run_lo = blockIdx.x * rows_per_slot;
  const int run_hi = min(total_rows, run_lo + rows_per_slot);
  const size_t npix = (size_t)B * H * W;
  const size_t plane = npix * 32;
  const unsigned rowbytes = (unsigned)W * 128u, rowbytes_g = (unsigned)W * 4u;
  const int img_bytes = (int)((size_t)H * rowbytes), img_bytes_g = (int)((size_t)H * rowbytes_g);
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

    const __amdgpu_buffer_rsrc_t rs_my = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mc + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_cov = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mc + plane + img * 32), 0, img_bytes, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(guide + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_mg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + npix + img), 0, img_bytes_g, RSRC_W3);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(planes + 2 * npix + img), 0, img_bytes_g, RSRC_W3);
    const int seg_rows = yend - ybeg;
    const __amdgpu_buffer_rsrc_t ro_l0 = __builtin_amdgcn_make_buffer_rsrc(lf + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);
    const __amdgpu_buffer_rsrc_t ro_l1 = __builtin_amdgcn_make_buffer_rsrc(lf + plane + (img + (size_t)ybeg * W) * 32, 0, (int)((size_t)seg_rows * rowbytes), RSRC_W3);

    const int r0 = ybeg - R;
    const int n_it = (seg_rows + 2 * R + KB - 1) / KB * KB;
    __syncthreads();
    for (int i = tid; i < n_it + 1; i += NT) {
      const int row = r0 - R + i;
      rny_tab[i] = (row >= 0 && row < H) ? 1.0f / (float)(min(row + R, H - 1) - max(row - R, 0) + 1) : 0.f;
    }
    __syncthreads();

    const unsigned lane32 = (unsigned)(col * 128 + cp * 8), lane1 = (unsigned)(col * 4);
    unsigned vs = (colin ? lane32 : NEVER) + (unsigned)r0 * rowbytes;
    unsigned vg = (colin ? lane1 : NEVER) + (unsigned)r0 * rowbytes_g;
    unsigned vp1 = (outcol ? lane1 : NEVER) + (unsigned)(r0 - R) * rowbytes_g;
    unsigned vo = (outcol ? lane32 : NEVER) + (unsigned)(r0 - R - ybeg) * rowbytes;
    const unsigned rd_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&sbuf[0][0][0][0] +
                             (unsigned)(((xi >= R ? xi - R : 0) * 16 + cp) * 8);
    Rings<4> rg;
#pragma unroll
    for (int qn = 0; qn < 4; ++qn)
#pragma unroll
      for (int k = 0; k < KB; ++k) rg.r[qn][k] = f32x2{0.f, 0.f};

    // one row ahead: the tape and the statistics of the streamed pixel (a pixel outside the image loads zeros: A = b = 0, the padding)
    f32x2 nmy = ld2(rs_my, vs), ncov = ld2(rs_cov, vs);
    float nmg = ld1(rs_mg, vg), nr0 = ld1(rs_r0, vg), nr1 = ld1(rs_r1, vg);
    vs += rowbytes; vg += rowbytes_g;

    auto step = [&](auto ktag, int itb) {
      constexpr int k = decltype(ktag)::value;
      const int it = itb + k;
      const f32x2 A0 = ncov * nr0, A1 = ncov * nr1;
      rg.r[0][k] = A0; rg.r[1][k] = nmy - A0 * nmg;
      rg.r[2][k] = A1; rg.r[3][k] = nmy - A1 * nmg;
      const float g = ld1(rs_g, vp1);
      vp1 += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);
      nmy = ld2(rs_my, vs); ncov = ld2(rs_cov, vs);
      nmg = ld1(rs_mg, vg); nr0 = ld1(rs_r0, vg); nr1 = ld1(rs_r1, vg);
      vs += rowbytes; vg += rowbytes_g;
      __builtin_amdgcn_sched_barrier(0);
      f32x2* const sw = &sbuf[par][0][xi][cp];
#pragma unroll
      for (int qn = 0; qn < 4; ++qn) sw[qn * NC * 16] = vsum9<k>(rg, qn);
      lds_barrier();
      if (xi >= R && xi < NC - R) {
        const unsigned ra = rd_base + par * (unsigned)(4 * NC * 16 * 8);
        const f32x2 ba0 = hsum9<0 * NC * 128>(ra), bb0 = hsum9<1 * NC * 128>(ra), ba1 = hsum9<2 * NC * 128>(ra), bb1 = hsum9<3 * NC * 128>(ra);
        const float rn = rnx * rny_tab[it];
        st2((ba0 * g + bb0) * rn, ro_l0, vo);
        st2((ba1 * g + bb1) * rn, ro_l1, vo);
      }
      asm volatile("" :: "v"(g));
      par ^= 1u;
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

}  // namespace paif_gft

// 1 if the streaming kernels (this file, gf_backward.hip) take the size; 0: the caller keeps the round-1 pair (paif_guided_filter_ab_fwd +
// paif_guided_filter_lf_fwd) and its four-map tape.
extern "C" int paif_guided_filter_taped_fits(int B, int H, int W) {
  return B > 0 && H > 2 * paif_gfs::R + 1 && W > 2 * paif_gfs::R + 1 && paif_gfs::make_plan(B, H, W).fits ? 1 : 0;
}

// guide [B,H,W], y [B,H,W,32] -> mc [2][B,H,W,32] = (mean_y, cov) (the tape of paif_guided_filter_bwd_input_mc) and lf [2][B,H,W,32]
// (eps0, eps1).  workspace: paif_guided_filter_fused_workspace_floats(B,H,W) floats; filled with the per-pixel guide statistics, which
// the reverse pass reads again (keep it with the tape).
extern "C" int paif_guided_filter_taped_fwd(const float* guide, const float* y, float* mc, float* lf, float eps0, float eps1, float* workspace,
                                            int B, int H, int W, paif_stream_t stream) {
  using namespace paif_gft;
  PAIF_REQUIRE(guide && y && mc && lf && workspace && B > 0, PAIF_EINVAL, "guided_filter_taped: bad arguments");
  PAIF_REQUIRE(H > 2 * R + 1 && W > 2 * R + 1, PAIF_EINVAL, "guided_filter: H,W must exceed 2r+1 = 9 (got %dx%d)", H, W);
  const Plan p = make_plan(B, H, W);
  PAIF_REQUIRE(p.fits, PAIF_ENOSUP, "guided_filter_taped: %dx%dx%d exceeds the streaming kernels' 32-bit row offsets", B, H, W);
  int rc = paifi_gf_guide_stats(guide, workspace, eps0, eps1, B, H, W, stream);
  if (rc) return rc;
  hipStream_t st = paif::as_stream(stream);
  hipLaunchKernelGGL(gf_mc_kernel, dim3(p.grid), dim3(NT), 0, st, guide, y, workspace, mc, B, H, W, p.nstrip, p.rows_per_slot, p.total_rows);
  PAIF_LAUNCH_CHECK("guided_filter_taped(mc)");
  hipLaunchKernelGGL(gf_lf_mc_kernel, dim3(p.grid), dim3(NT), 0, st, guide, workspace, mc, lf, B, H, W, p.nstrip, p.rows_per_slot, p.total_rows);
  PAIF_LAUNCH_CHECK("guided_filter_taped(lf)");
  return 0;
}
