// Interface between the dense-conv dispatcher (conv_mfma.hip) and the LDS-DMA 3x3 kernel (conv_dma.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace paif_conv_dma {

struct Args {
  const void* src[3];   // NHWC-32 bf16 maps (the virtual concat)
  const void* res[3];   // NHWC-32 bf16 residual maps
  const void* wpk;      // split-bf16 weight pack [src][tap][ks][hi|lo][64 lanes][8 bf16]: the hi halves are used
  const float* scale;   // per-cout (NULL = 1)
  const float* shift;   // per-cout (NULL = 0)
  const float* prelu;   // 1 float (act == 1)
  void* out;            // NHWC-32 bf16
  float alpha;
  int nsrc, nres, act, kh;   // kh: 3 or 7
  int cout;                  // 32, or 16 (3x3, one source, no residual maps)
  int B, H, W, reverse;
  float* cpool;              // optional: fused ChannelPool of the output (paif_conv_desc.cpool); built for (3 sources, 1 or 3 residual maps)
  int f16;                   // 1: the maps and weights are IEEE fp16 (PAIF_ST_F16 / PAIF_CONV_F16; the fp16 hi pieces of the F16X2 pack), else bf16
  int dil;                   // 1, or 2 (3x3, one source, input ReLU: the composed DilConv)
  int in_relu;               // 1: ReLU on the source as it is read (dilation 2 only)
};

// true if the kernel is built for this source / residual count and the tensors fit its 32-bit addressing
bool eligible(int nsrc, int nres, int B, int H, int W, float alpha);
bool eligible16(int nsrc, int nres, int B, int H, int W, float alpha);  // 3x3 with 16 output channels: one source, no residual maps
bool eligible7(int nsrc, int nres, int B, int H, int W, float alpha);   // the 7x7 form: one source, no residual maps
bool eligible_d2(int nsrc, int nres, int B, int H, int W, float alpha);  // 3x3 dilation 2 behind an input ReLU: one source, 1 or 3 residual maps
bool can_cpool(int nsrc, int nres, int kh, int cout, int dil);   // the instantiations that write Args::cpool
int launch(const Args& a, hipStream_t st);

}  // namespace paif_conv_dma
