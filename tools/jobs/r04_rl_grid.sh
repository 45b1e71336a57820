mkdir -p gpurun_out
for rep in 1 2; do
for g in 2048 1024 512; do
  DZG_RL_GRID=$g timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --sparse-per-col 50 --seed 1004 --steps 3000 --warmup 1000 --late-pivots 100000 --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_rl_$g.json 2>/dev/null
  python3 -c "
import json;d=json.load(open('gpurun_out/r04_rl_$g.json'));print('GRID=$g value',round(d['value']),'late',round(d['late']['value']),d['late']['k_at_start'])"
done
done
