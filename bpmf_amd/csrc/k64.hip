// k64.hip -- the kernels and launch logic of num_latent = 64 (see launch.h).
#include "launch_impl.h"

BPMF_INSTANTIATE_K(64, false)
