# round 4: the artifacts of the final code -- kernel stats of the driver's command, config 4's line, the one-rank RCCL rehearsal
mkdir -p gpurun_out
bash tools/kernel_stats.sh r04_final_kernel_stats --steps 1500 --warmup 100 > gpurun_out/r04_final_kernel_stats.txt 2>&1; echo "kernel stats rc=$?"; head -8 gpurun_out/r04_final_kernel_stats.txt
for f in 1 0; do
  DZG_SP_FUSED=$f timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --sparse-per-col 50 --seed 1004 --steps 3000 --warmup 1000 --late-pivots 100000 --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_config4_final_fused$f.json 2>/dev/null
  python3 -c "
import json;d=json.load(open('gpurun_out/r04_config4_final_fused$f.json'));print('FUSED=$f value',d['value'],'k',d['config']['k_at_start'],d['config']['k_at_end'],'late',d['late']['value'],d['late']['k_at_start'],d['late'].get('kernel_us_per_pivot'),d['late']['roofline']['frac'])"
done
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-sharded > gpurun_out/r04_sharded_1rank.json 2> gpurun_out/r04_sharded_1rank.err; echo "sharded rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04_sharded_1rank.json'));print('sharded value',d['value'],'deep',d.get('deep',{}).get('value'),'deep basis replicated',d.get('deep',{}).get('basis_replicated',{}).get('value'));print(d.get('phases'))"
