// The 12-wave build of the guided filter's matrix-core engine (gf_mfma2.hip): 96-column strips (80 outputs: halo x1.2 instead of x1.33,
// and 640 columns = 8 strips exactly where 48-column outputs leave a 16-column rest), THREE waves per SIMD (<= 168 registers: the
// register diet of gf_mfma2.hip's GF2_YD_LDS / GF2_OWN_LDS / GF2_EARLY_AB / GF2_DPL options), for the fp16 high-frequency output modes --
// what the fp16 configuration's forward (bench.py's `value`) runs.  Round 6: 328 -> ~265 us per B=8 480x640 launch.
#define GF2_NW 12
#define GF2_NS paif_gf2w12
#include "gf_mfma2.hip"
