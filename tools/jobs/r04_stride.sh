# round 4: the timing stride test, then the early-regime lines with sampled no-fence events
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "timing_can_sample or chain_and_seven_launches_are" 2>&1 | tail -3
for rep in 1 2; do
  timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_stride_c3.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_stride_c3.json'));print('config3 value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'],d['roofline']['launches_timed'],d['roofline']['frac'])"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_stride_c3d.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_stride_c3d.json'));print('config3 driver flags value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'],d['roofline']['launches_timed'],d['roofline']['frac'])"
  timeout -k 10 200 python bench.py --rows 1024 --cols 2048 --seed 1002 --steps 5000 --warmup 500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_stride_c2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_stride_c2.json'));print('config2 value',round(d['value']),'ms',d['ms_per_step'])"
  timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --warmup 20000 --steps 3000 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_stride_c4.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_stride_c4.json'));print('config4 k~2200 value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'])"
  DZG_RL_GRID=512 timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --warmup 20000 --steps 3000 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_stride_c4g.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_stride_c4g.json'));print('config4 k~2200 grid 512 value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'])"
done
