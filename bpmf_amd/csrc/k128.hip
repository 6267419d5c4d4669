// k128.hip -- the kernels and launch logic of num_latent = 128 in the fp32 context (see launch.h).
#include "launch_impl.h"

BPMF_INSTANTIATE_K(128, true)
