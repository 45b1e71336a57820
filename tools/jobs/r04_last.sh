# round 4: the whole GPU suite on the last code, smoke(), config 4's bench line
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 880 python -m pytest tests -m gpu -x -q > gpurun_out/t18_full.log 2>&1; echo "suite rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/t18_full.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for f in 1 0; do
  DZG_SP_FUSED=$f timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --sparse-per-col 50 --seed 1004 --steps 3000 --warmup 1000 --late-pivots 100000 --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_config4_final_fused$f.json 2>/dev/null
  python3 -c "
import json;d=json.load(open('gpurun_out/r04_config4_final_fused$f.json'));print('FUSED=$f value',d['value'],'k',d['config']['k_at_start'],d['config']['k_at_end'],'late',d['late']['value'],d['late']['k_at_start'],d['late'].get('kernel_us_per_pivot'),d['late']['roofline']['frac'])"
done
