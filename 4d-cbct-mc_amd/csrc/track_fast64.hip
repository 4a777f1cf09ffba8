// track_fast64.hip -- the FAST personality with the reference's double-precision sub-steps (MCGPU_MODE_FAST_F64):
// rotate_double (K.cu:1103-1148), GRAa (K.cu:1181-1246) and GCOa's cdt1 / costh chain (K.cu:1329-1331) in double, as the
// reference computes them; scheduling, random-number streams and everything the reference does in float: track_fast.hip's.
#define MC_COMPAT 0
#define MC_FAST_F64 1
#include "track_pool.inc"
