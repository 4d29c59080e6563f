// The STFT / iSTFT kernels compiled a second time with IEEE half (_Float16) spectrograms: see alsep_common.h.
#define ALSEP_F16_TU 1
#include "fft.hip"
