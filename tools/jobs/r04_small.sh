# round 4: the fused small-k row pricing kernel -- parity tests, then A/B against DZG_PRICE_SMALL_K=0 on the same box
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain or seven or row_wise or oracle or degenerate or timing_can_sample or warm" > gpurun_out/t11.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t11.log
for rep in 1 2; do
for sk in 480 0; do
  export DZG_PRICE_SMALL_K=$sk
  timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_small_c3_$sk.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_small_c3_$sk.json'));print('SMALL_K=$sk config3 value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'],'frac',round(d['roofline']['frac'],3))"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_small_c3d_$sk.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_small_c3d_$sk.json'));print('SMALL_K=$sk config3 driver flags value',round(d['value']),'ms',d['ms_per_step'])"
  timeout -k 10 200 python bench.py --rows 1024 --cols 2048 --seed 1002 --steps 5000 --warmup 500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_small_c2_$sk.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_small_c2_$sk.json'));print('SMALL_K=$sk config2 value',round(d['value']),'ms',d['ms_per_step'])"
  timeout -k 10 300 python bench.py --rows 32768 --cols 65536 --seed 1005 --steps 300 --warmup 50 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_small_c5_$sk.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_small_c5_$sk.json'));print('SMALL_K=$sk config5 value',round(d['value']),'ms',d['ms_per_step'])"
done
done
