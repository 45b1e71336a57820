mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-sharded --no-secondary --no-cpu-baseline > gpurun_out/r04_sharded_check.json 2> gpurun_out/r04_sharded_check.err; echo "sharded rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04_sharded_check.json'));print('sharded value',d['value'],d['roofline']['avg_launch_us'],d['roofline'].get('launches_timed'),d['roofline']['frac']);print(d['phases']['us_per_iteration_rank0'])"
