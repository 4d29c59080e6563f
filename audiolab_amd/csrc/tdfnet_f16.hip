// The TFC-TDF U-Net kernels compiled a second time with IEEE half (_Float16) as the 16-bit storage type: see alsep_common.h.
#define ALSEP_F16_TU 1
#include "tdfnet.hip"
