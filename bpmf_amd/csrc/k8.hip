// k8.hip -- the kernels and launch logic of num_latent = 8 (see launch.h).
#include "launch_impl.h"

BPMF_INSTANTIATE_K(8, false)
