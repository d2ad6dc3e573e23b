// Phase timeline of the multi-source conv kernel (conv_bf16x3_ms): wave 0 of every workgroup records s_memtime at the
// phase boundaries; the host prints the mean cycles per phase.  Build (CPU container):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I paif_amd/csrc tools/microbench/conv_ms_trace.hip -o tools/microbench/conv_ms_trace
// Run on the GPU box: tools/microbench/conv_ms_trace [nsrc=3] [nres=0]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define NSTAMP 16
#define MAXWG 16384
__device__ unsigned long long g_stamps[MAXWG][4][NSTAMP];
__device__ unsigned g_where[MAXWG];   // XCC / SE / CU id of the workgroup
#define PAIF_MS_STAMP(i)                                                                         \
  do {                                                                                           \
    if (lane == 0 && blockIdx.x < MAXWG) {                                                       \
      g_stamps[blockIdx.x][wave][(i)] = __builtin_readcyclecounter();                            \
      if ((i) == 0) g_stamps[blockIdx.x][wave][13] = wall_clock64();                             \
      if ((i) == 15) g_stamps[blockIdx.x][wave][14] = wall_clock64();                            \
    }                                                                                            \
  } while (0)

// persistent kernels: per-phase cycle totals over all tiles of a workgroup (scalar registers)
#define PAIF_TRACE_DECL unsigned long long tr_last = __builtin_readcyclecounter(), tr_acc[5] = {0, 0, 0, 0, 0}; const unsigned long long tr_w0 = wall_clock64();
#define PAIF_TRACE(i)                                                    \
  do {                                                                   \
    const unsigned long long tr_now = __builtin_readcyclecounter();      \
    tr_acc[(i)] += tr_now - tr_last;                                     \
    tr_last = tr_now;                                                    \
  } while (0)
#define PAIF_TRACE_END                                                                           \
  if (lane == 0 && blockIdx.x < MAXWG) {                                                         \
    for (int tr_i = 0; tr_i < 5; ++tr_i) g_stamps[blockIdx.x][wave & 3][tr_i] = tr_acc[tr_i];    \
    g_stamps[blockIdx.x][wave & 3][5] = wall_clock64() - tr_w0;                                  \
  }

#include "../../paif_amd/csrc/conv_mfma.hip"

namespace paif {
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
}  // namespace paif

__global__ void fill(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((int)(x & 0xffff) - 32768) * (1.f / 32768.f);
  }
}

int main(int argc, char** argv) {
  const int nsrc = argc > 1 ? atoi(argv[1]) : 3, nres = argc > 2 ? atoi(argv[2]) : 0;
  const int B = 8, H = 480, W = 640;
  const size_t n = (size_t)B * H * W * 32;
  float *src[3], *res[3] = {nullptr, nullptr, nullptr}, *out, *w; void* wpk;
  for (int s = 0; s < nsrc; ++s) { hipMalloc(&src[s], n * 4); fill<<<2048, 256>>>(src[s], n, 17 + s); }
  for (int s = 0; s < nres; ++s) { hipMalloc(&res[s], n * 4); fill<<<2048, 256>>>(res[s], n, 99 + s); }
  hipMalloc(&out, n * 4);
  const size_t nw = (size_t)32 * 32 * nsrc * 9;
  hipMalloc(&w, nw * 4); fill<<<64, 256>>>(w, nw, 5);
  hipMalloc(&wpk, nw * 4);
  if (paif_pack_conv_weight_bf16x3(w, (float*)wpk, 32, nsrc, 3, nullptr)) return 1;
  paif_conv_desc d = {};
  for (int s = 0; s < nsrc; ++s) d.src[s] = src[s];
  for (int s = 0; s < nres; ++s) d.res[s] = res[s];
  d.nsrc = nsrc; d.wpk = (const float*)wpk; d.out = out; d.cin = 32; d.cout = 32; d.kh = 3; d.dil = 1; d.alpha = 1.f;
  d.precision = PAIF_CONV_BF16X3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (paif_conv2d_fwd(&d, B, H, W, nullptr)) return 1;
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) paif_conv2d_fwd(&d, B, H, W, nullptr);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("nsrc %d nres %d: %.1f us per launch\n", nsrc, nres, ms * 100.f);
  if (getenv("PAIF_TRACE_RES")) {
    std::vector<unsigned long long> st((size_t)MAXWG * 4 * NSTAMP);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* nm[5] = {"issue next halo tile", "MFMA phase", "epilogue", "convert -> LDS", "barrier"};
    for (int wv = 0; wv < 4; wv += 3) {
      double tot = 0, t[5] = {}, wall = 0;
      for (int wg = 0; wg < 256; ++wg) {
        for (int i = 0; i < 5; ++i) t[i] += (double)st[((size_t)wg * 4 + wv) * NSTAMP + i] / 256;
        wall += (double)st[((size_t)wg * 4 + wv) * NSTAMP + 5] / 256;
      }
      for (int i = 0; i < 5; ++i) tot += t[i];
      printf("wave %d: cycles per workgroup (mean over 256), %% of total\n", wv);
      for (int i = 0; i < 5; ++i) printf("  %-22s %10.0f  %5.1f %%\n", nm[i], t[i], 100 * t[i] / tot);
      printf("  total %.0f cycles in %.1f us  => %.0f MHz\n", tot, wall / 100, tot / (wall / 100));
    }
    return 0;
  }
  const int nwg = B * (H / 8) * (W / 32);
  std::vector<unsigned long long> st((size_t)MAXWG * 4 * NSTAMP);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
  auto S = [&](int wg, int wv, int i) { return st[((size_t)wg * 4 + wv) * NSTAMP + i]; };
  const char* names[16] = {"start", "A0 loaded+converted", "barrier", "mma s0", "barrier", "convert s1", "barrier", "mma s1",
                           "barrier", "convert s2", "barrier", "mma s2", "", "", "", "epilogue+end"};
  std::vector<int> idx = {0, 1, 2, 3};
  for (int s = 1; s < nsrc; ++s) for (int k = 0; k < 4; ++k) idx.push_back(4 * s + k);
  idx.push_back(15);
  for (int wv = 0; wv < 4; wv += 3) {
    printf("wave %d: mean cycles per phase over %d workgroups (s_memtime ticks)\n", wv, std::min(nwg, MAXWG));
    double tot = 0;
    for (size_t k = 1; k < idx.size(); ++k) {
      double sum = 0; int cnt = 0;
      for (int wg = 0; wg < std::min(nwg, MAXWG); ++wg) { sum += (double)(S(wg, wv, idx[k]) - S(wg, wv, idx[k - 1])); ++cnt; }
      printf("  %-22s %9.0f\n", names[idx[k]], sum / cnt); tot += sum / cnt;
    }
    printf("  %-22s %9.0f\n", "total", tot);
  }
  {
    double cyc = 0, real = 0; unsigned long long r0 = ~0ull, r1 = 0;
    for (int wg = 0; wg < std::min(nwg, MAXWG); ++wg) {
      cyc += (double)(S(wg, 0, 15) - S(wg, 0, 0)); real += (double)(S(wg, 0, 14) - S(wg, 0, 13));
      r0 = std::min(r0, S(wg, 0, 13)); r1 = std::max(r1, S(wg, 0, 14));
    }
    printf("cycle-counter ticks per 100 MHz wall tick: %.2f  (=> counter runs at %.0f MHz); kernel wall span %.1f us\n", cyc / real,
           cyc / real * 100.0, (r1 - r0) / 100.0);
  }
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int wg = 0; wg < std::min(nwg, MAXWG); ++wg) { t0 = std::min(t0, S(wg, 0, 0)); t1 = std::max(t1, S(wg, 0, 15)); }
  printf("kernel span %llu ticks\n", t1 - t0);
  for (int wg = 2000; wg < 2003; ++wg) {
    printf("wg %d:", wg);
    for (size_t k = 0; k < idx.size(); ++k) printf(" %llu", S(wg, 0, idx[k]) - t0);
    printf("\n");
  }
  return 0;
}
