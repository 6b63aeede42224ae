// tv_fused_admm.hip -- fp32 instantiations of the one-sweep ADMM z / u update + residual (tv_fused.h, ALG_ADMM; round 3).
#include "tv_fused_launch.h"
TV_FUSED_INSTANTIATE(float, ALG_ADMM)
