# Dev measurement (GPU): initialisation time of the drop-in executable for 1 and 4 device contexts (all on device 0 here): the
# input is parsed once, every device context is a clone built in its own thread (main.cpp)
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1
IN=/tmp/mcgpu_bench_catphan_512_894/input.in
sed 's/^894 /2 /' $IN > /tmp/init_probe.in   # two projections only
for devs in 0 0,0 0,0,0,0; do
  for text in 0 1; do
    if [ $text = 1 ]; then export MCGPU_IGNORE_VOXBIN=1; else unset MCGPU_IGNORE_VOXBIN; fi
    ./4d-cbct-mc_amd/MC-GPU_v1.3.x /tmp/init_probe.in --devices $devs --no-output 2>&1 | grep -E "INITIALIZATION finished|Execution time" | tr '\n' ' '; echo " devices=$devs text_parse=$text"
  done
done
