# Dev check (GPU): every engine build under build/ab/*.so must reproduce the pinned FAST tallies (tests/golden/fast_pin.json) --
# a scheduling-only variant leaves every tally word unchanged -- then tools/ab.sh times them.
for lib in build/ab/*.so; do echo -n "$(basename $lib) pins: "; MCGPU_AMD_LIB=$PWD/$lib timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pinned and not workgroup" 2>&1 | tail -1; done
bash tools/ab.sh
