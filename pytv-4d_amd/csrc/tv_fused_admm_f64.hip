// tv_fused_admm_f64.hip -- fp64 instantiations of the one-sweep ADMM z / u update + residual (tv_fused.h, ALG_ADMM; round 3).
#include "tv_fused_launch.h"
TV_FUSED_INSTANTIATE(double, ALG_ADMM)
