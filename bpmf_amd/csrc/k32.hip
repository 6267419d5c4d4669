// k32.hip -- the kernels and launch logic of num_latent = 32 (see launch.h).
#include "launch_impl.h"

BPMF_INSTANTIATE_K(32, false)
