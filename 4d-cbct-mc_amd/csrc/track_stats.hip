// track_stats.hip -- DIAGNOSTIC build of the FAST personality with scheduler statistics (never used for timing).
// Counters (TrackArgs::stats): 0 loop iterations per wave, 1 sum of flying lanes, 2/3 Compton rounds / lanes,
// 4/5 Rayleigh rounds / lanes, 6/7 tally+source rounds / lanes.
#define MC_COMPAT 0
#define MC_STATS 1
#include "track_pool.inc"
