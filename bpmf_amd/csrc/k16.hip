// k16.hip -- the kernels and launch logic of num_latent = 16 (see launch.h).
#include "launch_impl.h"

BPMF_INSTANTIATE_K(16, false)
