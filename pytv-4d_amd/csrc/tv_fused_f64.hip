// tv_fused_f64.hip -- the fp64 instantiations of the one-sweep Chambolle-Pock iteration (tv_fused.h, round 3): a lane holds
// 2 doubles instead of 4 floats, everything else -- tile in lanes, LDS state, in-block column hand-off, fix-up classes -- is
// the same code.  A translation unit of its own so that it compiles next to tv_fused.hip.
#include "tv_fused_launch.h"
TV_FUSED_INSTANTIATE(double, ALG_CP)
