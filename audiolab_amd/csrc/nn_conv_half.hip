// nn_half.hip's convolution kernel (nn_conv_hh_kernel) and its entry points as a unit of their own: the kernel sits at the 256-register limit
// of two waves per SIMD, and with the target feature packed-fp32-ops switched off (how every other unit is built, __graft_entry__.py) hipcc's
// register allocation tips over: 7 spilled dwords, every scratch access waited for with vmcnt(0), the prefetch gone -- 87 instead of 57 us
// per launch, MDX23C 1.14 instead of 0.93 s per 120 s.  This unit keeps the feature (the only float32 arithmetic in it is the epilogue's add).
#define ALSEP_NN_HALF_CONV_TU
#include "nn_half.hip"
