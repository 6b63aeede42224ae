// tv_fused_lazy.hip -- instantiations of the LAZY one-sweep Chambolle-Pock kernel (tv_fused.h, round 4), fp32
#include "tv_fused_launch.h"
TV_FUSED_INSTANTIATE_LAZY(float)
