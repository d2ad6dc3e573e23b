// Dense-conv weight gradient (first kernel of the training step, DESIGN.md "Plan for the training step" item 2):
//   dW[co][s*32+ci][ky][kx] = sum_{b,y,x} dAcc[b,y,x,co] * X_s[b, y + ky*d - P, x + kx*d - P, ci]        (zero outside the image)
//   dAcc = dOut * alpha * act'(z) * scale[co]      (z = saved pre-activation of the forward epilogue; act' of PReLU / ReLU)
// for the forward conv of conv_mfma.hip (`nn.Conv2d` inside BasicConv / RDB / ResidualModule ..., operations_m.py:435-464).
//
// v_mfma_f32_32x32x2_f32 consumes the NHWC layout directly: with K = 2 consecutive pixels of a row,
//   A[i = co][k] = dAcc[pixel k][co]   -> lane (co = l & 31, k = l >> 5) loads one dword; a wave-load is 2 px x 128 B contiguous
//   B[k][j = ci] = X[pixel k + tap][ci] -> the same, shifted by the tap
// so a wave streams pixel pairs (64-pixel chunks staged through a wave-private LDS region) and keeps one 32 x 32
// accumulator per horizontal tap kx; blockIdx.y selects (source, ky).
// A workgroup covers 8 image rows (4 waves x 2 rows); its 4 waves are reduced through LDS and written as one slab
// [block][source*KH + ky][kx][co][ci]; a second pass sums the slabs in block order (deterministic) into the PyTorch layout.
// Exact fp32 (the gradient-parity tolerance of the reference's own fp32 run is what this has to meet).
#include "paif_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 8;

struct WgradArgs {
  const float* src[3];
  const float* dout;
  const float* z;        // saved pre-activation (act 1/2) or NULL
  const float* scale;    // per-cout forward scale (folded BN) or NULL
  const float* prelu;    // slope (act 1)
  float* slabs;
  float alpha;
  int act, kh, dil, nsrc, B, H, W, nrb;   // nrb = row blocks per image
};

// Operands go through LDS, transposed to [channel][pixel]: the MFMA takes ONE dword per lane per operand, and fetching those
// dwords straight from global memory made every MFMA (64 cycles) wait on a 256-B wave load -- the kernel ran at the CU's
// L1 rate (64 B/clk): 21 TFLOP/s.  Staged with coalesced float4 loads (8 px x 128 B per wave instruction) into a wave-private
// LDS region, the same dwords cost one conflict-free ds_read_b32 each and the kernel is paced by the matrix pipe.
constexpr int CW = 64;              // pixels per staged chunk
constexpr int DSTR = CW + 1;        // row stride of the dAcc chunk [32][CW]   (stride % 32 == 1: lanes of one half-wave hit 32 banks)
constexpr int XSTR = CW + 12 + 1;   // row stride of the source chunk [32][CW + 2P], P <= 6

template <int KH>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ float wlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = lane >> 5, ch = lane & 31;
  const int s = blockIdx.y / KH, ky = blockIdx.y - s * KH;
  const int b = blockIdx.x / a.nrb, rb = blockIdx.x - b * a.nrb;
  const int P = a.dil * (KH - 1) / 2;
  const float* X = a.src[s];
  float* sD = wlds + wave * (32 * DSTR + 32 * XSTR);
  float* sX = sD + 32 * DSTR;
  const int q = lane & 7, pl = lane >> 3;                    // staging role: pixel lane, channel quad
  float4 sc4 = make_float4(a.alpha, a.alpha, a.alpha, a.alpha);
  if (a.scale) {
    const float4 t = *reinterpret_cast<const float4*>(a.scale + 4 * q);
    sc4 = make_float4(t.x * a.alpha, t.y * a.alpha, t.z * a.alpha, t.w * a.alpha);
  }
  const float slope = a.act == 1 ? *a.prelu : 0.f;
  f32x16 acc[KH];
#pragma unroll
  for (int t = 0; t < KH; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // The wave's chunks -- (output row rr = wave, wave + 4, ..., 64-pixel piece x0) with the source row inside the image -- form one
  // sequence, software-pipelined: the global loads of chunk i+1 are issued before the 32 * KH MFMAs of chunk i (the serial
  // load -> LDS -> MFMA form waited a memory latency per chunk).
  constexpr int NXV = (CW + 12 + 7) / 8;       // float4 slots of the source chunk per lane (P <= 6)
  float4 pd[CW / 8], pz[CW / 8], pxv[NXV];
  const int nx = CW + 2 * P;
  auto valid_row = [&](int rr) -> bool {       // wave-uniform
    const int y = rb * ROWS + rr;
    if (rr >= ROWS || y >= a.H) return false;
    const int ys = y + ky * a.dil - P;
    return ys >= 0 && ys < a.H;
  };
  auto advance = [&](int& rr, int& x0) {       // next chunk of this wave; rr >= ROWS when done
    x0 += CW;
    if (x0 < a.W) return;
    x0 = 0;
    rr += 4;
    while (rr < ROWS && rb * ROWS + rr < a.H && !valid_row(rr)) rr += 4;
    if (rr < ROWS && rb * ROWS + rr >= a.H) rr = ROWS;
  };
  auto gload = [&](int rr, int x0) {
    const int y = rb * ROWS + rr, ys = y + ky * a.dil - P;
    const size_t orow = ((size_t)b * a.H + y) * a.W, srow = ((size_t)b * a.H + ys) * a.W;
#pragma unroll
    for (int it = 0; it < CW / 8; ++it) {
      const size_t o = (orow + min(x0 + it * 8 + pl, a.W - 1)) * 32 + 4 * q;   // unconditional loads on clamped columns
      pd[it] = *reinterpret_cast<const float4*>(a.dout + o);
      if (a.act) pz[it] = *reinterpret_cast<const float4*>(a.z + o);
    }
#pragma unroll
    for (int it = 0; it < NXV; ++it) {
      const int xs = x0 - P + it * 8 + pl;
      pxv[it] = *reinterpret_cast<const float4*>(X + (srow + min(max(xs, 0), a.W - 1)) * 32 + 4 * q);
    }
  };
  int rr = wave, x0 = 0;
  while (rr < ROWS && rb * ROWS + rr < a.H && !valid_row(rr)) rr += 4;
  if (rr < ROWS && rb * ROWS + rr >= a.H) rr = ROWS;
  if (rr < ROWS) gload(rr, x0);
  while (rr < ROWS) {
    // ---- stage dAcc[x0 .. x0+CW) and X[x0-P .. x0+CW+P) of this row pair, transposed to [channel][pixel] ----
#pragma unroll
    for (int it = 0; it < CW / 8; ++it) {
      float4 d = pd[it];
      if (a.act) {
        const float4 zv = pz[it];
        if (a.act == 1) {
          d.x *= zv.x >= 0.f ? 1.f : slope; d.y *= zv.y >= 0.f ? 1.f : slope; d.z *= zv.z >= 0.f ? 1.f : slope; d.w *= zv.w >= 0.f ? 1.f : slope;
        } else {
          d.x *= zv.x > 0.f ? 1.f : 0.f; d.y *= zv.y > 0.f ? 1.f : 0.f; d.z *= zv.z > 0.f ? 1.f : 0.f; d.w *= zv.w > 0.f ? 1.f : 0.f;
        }
      }
      const bool in = x0 + it * 8 + pl < a.W;
      float* dst = sD + (4 * q) * DSTR + it * 8 + pl;
      dst[0] = in ? d.x * sc4.x : 0.f; dst[DSTR] = in ? d.y * sc4.y : 0.f;
      dst[2 * DSTR] = in ? d.z * sc4.z : 0.f; dst[3 * DSTR] = in ? d.w * sc4.w : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NXV; ++it) {
      const int xl = it * 8 + pl;                                       // local column 0 .. nx-1  <->  image column x0 - P + xl
      const int xs = x0 - P + xl;
      const float4 v = pxv[it];
      const bool in = xs >= 0 && xs < a.W && xl < nx;
      if (xl < nx) {
        float* dst = sX + (4 * q) * XSTR + xl;
        dst[0] = in ? v.x : 0.f; dst[XSTR] = in ? v.y : 0.f; dst[2 * XSTR] = in ? v.z : 0.f; dst[3 * XSTR] = in ? v.w : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();         // the region is wave-private and LDS operations of one wave complete in order:
    asm volatile("" ::: "memory");           // only the compiler has to be kept from reordering across the phases
    advance(rr, x0);
    if (rr < ROWS) gload(rr, x0);            // wave-uniform; in flight during the MFMAs below
    // ---- MFMA over the chunk's pixel pairs: A[co][k] = dAcc[pixel 2j+k][co], B[k][ci] = X[pixel 2j+k + tap][ci] ----
    const float* pdl = sD + ch * DSTR + k;
    const float* pxl = sX + ch * XSTR + k;
#pragma unroll 4
    for (int j = 0; j < CW / 2; ++j) {
      const float d = pdl[2 * j];
#pragma unroll
      for (int t = 0; t < KH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, pxl[2 * j + t * a.dil], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  // ---- reduce the 4 waves (fixed order) through the staging LDS and write the slab ----
  __syncthreads();
  float* red = wlds;                            // [KH][16][64] floats
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* slab = a.slabs + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * KH * 1024;
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * k;      // C/D layout of the 32x32 MFMA: row = co, column = lane & 31 = ci
        slab[(t * 32 + co) * 32 + ch] = acc[t][r];
      }
  }
}

// dw[co][s*32+ci][ky][kx] = sum over blocks of slab[blk][s*KH+ky][kx][co][ci]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int nblk, int nz,
                                                                int kh, int nsrc, int cout, int accumulate) {
  // one wave per weight: its 64 lanes stride over the workgroup slabs, then a fixed xor-shuffle tree (deterministic)
  const int total = cout * nsrc * 32 * kh * kh;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= total) return;      // wave-uniform
  int tmp = i;
  const int kx = tmp % kh; tmp /= kh;
  const int ky = tmp % kh; tmp /= kh;
  const int cin = tmp % (nsrc * 32);
  const int co = tmp / (nsrc * 32);
  const int s = cin >> 5, ci = cin & 31;
  const size_t off = ((size_t)(s * kh + ky) * kh + kx) * 1024 + (size_t)co * 32 + ci;
  float v = 0.f;
  for (int blk = lane; blk < nblk; blk += 64) v += slabs[((size_t)blk * nz) * kh * 1024 + off];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if (lane == 0) dw[i] = accumulate ? dw[i] + v : v;
}

}  // namespace

extern "C" size_t paif_conv2d_wgrad_workspace_floats(int nsrc, int kh, int B, int H) {
  return (size_t)B * ((H + ROWS - 1) / ROWS) * nsrc * kh * kh * 1024;
}

extern "C" int paif_conv2d_wgrad(const float* const* src, int nsrc, const float* dout, const float* z, const float* scale,
                                 const float* prelu, int act, float alpha, int kh, int dil, float* workspace, float* dw, int cout,
                                 int accumulate, int B, int H, int W, paif_stream_t stream) {
  PAIF_REQUIRE(src && dout && workspace && dw, PAIF_EINVAL, "conv2d_wgrad: null pointer");
  PAIF_REQUIRE(nsrc >= 1 && nsrc <= 3, PAIF_EINVAL, "conv2d_wgrad: nsrc=%d", nsrc);
  PAIF_REQUIRE(B > 0 && H > 0 && W > 0, PAIF_EINVAL, "conv2d_wgrad: empty shape");
  PAIF_REQUIRE(act >= 0 && act <= 2 && (!act || z) && (act != 1 || prelu), PAIF_EINVAL, "conv2d_wgrad: act=%d needs z (and the slope)", act);
  PAIF_REQUIRE(dil == 1 || dil == 2, PAIF_ENOSUP, "conv2d_wgrad: dil=%d", dil);
  PAIF_REQUIRE(cout >= 1 && cout <= 32, PAIF_EINVAL, "conv2d_wgrad: cout=%d (dout always has 32 channels; rows >= cout are dropped)", cout);
  WgradArgs a;
  for (int s = 0; s < 3; ++s) a.src[s] = s < nsrc ? src[s] : nullptr;
  for (int s = 0; s < nsrc; ++s) PAIF_REQUIRE(a.src[s], PAIF_EINVAL, "conv2d_wgrad: src[%d] null", s);
  a.dout = dout; a.z = z; a.scale = scale; a.prelu = prelu; a.slabs = workspace; a.alpha = alpha;
  a.act = act; a.kh = kh; a.dil = dil; a.nsrc = nsrc; a.B = B; a.H = H; a.W = W; a.nrb = (H + ROWS - 1) / ROWS;
  const dim3 grid(B * a.nrb, nsrc * kh);
  hipStream_t st = paif::as_stream(stream);
  constexpr size_t lds_bytes = (size_t)4 * (32 * DSTR + 32 * XSTR) * sizeof(float);      // 72.7 KB: two workgroups per CU
  static bool raised = false;
  if (!raised) {       // idempotent; > 64 KiB of dynamic LDS needs the attribute once per instantiation
    const void* fns[4] = {reinterpret_cast<const void*>(&conv_wgrad_kernel<1>), reinterpret_cast<const void*>(&conv_wgrad_kernel<3>),
                          reinterpret_cast<const void*>(&conv_wgrad_kernel<5>), reinterpret_cast<const void*>(&conv_wgrad_kernel<7>)};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) {
        paif::set_error("conv2d_wgrad: cannot raise dynamic LDS to %zu: %s", lds_bytes, hipGetErrorString(e));
        return (int)e;
      }
    }
    raised = true;
  }
  switch (kh) {
    case 1: hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), lds_bytes, st, a); break;
    case 3: hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), lds_bytes, st, a); break;
    case 5: hipLaunchKernelGGL(conv_wgrad_kernel<5>, grid, dim3(256), lds_bytes, st, a); break;
    case 7: hipLaunchKernelGGL(conv_wgrad_kernel<7>, grid, dim3(256), lds_bytes, st, a); break;
    default:
      paif::set_error("conv2d_wgrad: kernel size %d not built", kh);
      return PAIF_ENOSUP;
  }
  PAIF_LAUNCH_CHECK("conv2d_wgrad");
  const int total = cout * nsrc * 32 * kh * kh;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((total + 3) / 4), dim3(256), 0, st, workspace, dw, B * a.nrb, nsrc * kh, kh, nsrc,
                     cout, accumulate);
  PAIF_LAUNCH_CHECK("conv2d_wgrad_reduce");
  return 0;
}
