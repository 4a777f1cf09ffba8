"""Diagnostic (GPU): where the COMPAT kernel's time goes on a bench workload.  Needs an engine library whose track_compat.o was
built with -DMC_COMPAT_STATS (tools/compat_stats.sh builds one under build/ab/ and runs this).
usage: MCGPU_AMD_LIB=build/ab/compat_stats.so MCGPU_COMPAT_STATS=1 compat_stats.py <workload dir> [histories]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import cases
eng = cases.pkg.engine
with eng.create(sys.argv[1] + "/input.in", device=0) as ctx:
    batches, hpt, total = ctx.reference_shape(int(float(sys.argv[2]) if len(sys.argv) > 2 else 2e7))
    ctx.run_projection(300, batches, mode="compat", seed=42, hpt=hpt)
    out = (C.c_ulonglong * 32)()
    ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, 32, 1)  # reset
    img, secs, done = ctx.run_projection(300, batches, mode="compat", seed=42, hpt=hpt)
    ctx.lib.mcgpu_scheduler_stats_ex(ctx.h, out, 32, 1)
    s = [int(v) for v in out]
    tot = max(s[17], 1)
    r = lambda a, b: round(a / max(b, 1), 2)
    print(json.dumps({
        "histories": done, "ms": round(secs * 1e3, 1), "loop_iterations_per_history": r(s[0], done) , "flying_lanes_per_iteration": r(s[1], s[0]),
        "compton": {"batches_per_history": round(s[3] / done, 4), "lanes_per_batch": r(s[4], s[3]), "share_of_wave_time": round(s[5] / tot, 3),
                    "angle_trials_share_of_wave_time": round(s[18] / tot, 3), "shell_momentum_trials_share_of_wave_time": round(s[19] / tot, 3), "batches_with_an_S0_pass": round(s[6] / max(s[3], 1), 4), "open_tests_per_trial": round(s[7] / max(s[4], 1), 5)},
        "rayleigh": {"batches_per_history": round(s[8] / done, 4), "lanes_per_batch": r(s[9], s[8]), "share_of_wave_time": round(s[10] / tot, 3)},
        "tally_source": {"batches_per_history": round(s[11] / done, 4), "lanes_per_batch": r(s[12], s[11]), "share_of_wave_time": round(s[13] / tot, 3)},
        "exchange": {"per_history": round(s[14] / done, 4), "lanes_each": r(s[15], s[14]), "share_of_wave_time": round(s[16] / tot, 3)},
        "flight_share_of_wave_time": round(s[2] / tot, 3),
        "rest_share_of_wave_time": round(1 - (s[2] + s[5] + s[10] + s[13] + s[16]) / tot, 3)}))
