// Issue rate of v_add_f32 / v_pk_add_f32 / v_pk_fma_f32 / ds_read_b128 on gfx950: one workgroup per CU, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 64
template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  __shared__ float4 lds[1024];
  lds[threadIdx.x & 1023] = make_float4(a0, a1, a2, a3);
  __syncthreads();
  float4 r0, r1, r2, r3;
  r0 = r1 = r2 = r3 = make_float4(0, 0, 0, 0);
  const float4* lp = lds + (threadIdx.x & 63);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < REP / 4; ++j) {
      if (MODE == 0) {
        asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7));
      } else if (MODE == 1) {
        asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4));
      } else if (MODE == 2) {
        asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4));
      } else if (MODE == 3) {
        asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7));
      } else {
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"((unsigned)(size_t)lp) : "memory");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0.x + r1.y + r2.z + r3.w;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
  float* out;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(out, 10);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * REP;
  const double cyc = ms * 1e-3 * 2.4e9;       // at the nominal 2.4 GHz
  printf("%-14s %d wave(s)/SIMD: %.2f ms  -> %.2f cycles (@2.4GHz) per instruction per wave, %.2f per SIMD slot\n", name, waves_per_simd, ms,
         cyc / instr_per_wave, cyc / instr_per_wave / waves_per_simd);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_add_f32", w);
    run<3>("v_fma_f32", w);
    run<1>("v_pk_add_f32", w);
    run<2>("v_pk_fma_f32", w);
    run<4>("ds_read_b128", w);
  }
  return 0;
}
