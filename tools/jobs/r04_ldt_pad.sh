# round 4: does a row stride of the row-major copy that is not a power of two help the row-wise pass?
mkdir -p gpurun_out
for pad in 0 32 528; do
  export DZG_LDT_PAD=$pad
  timeout -k 10 300 python bench.py --warmup 150000 --steps 1500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary --no-mfma > gpurun_out/r04_pad_deep_$pad.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_pad_deep_$pad.json'));print('PAD=$pad deep value',round(d['value']),'price us',round(d['roofline']['avg_launch_us'],2),'frac',round(d['roofline']['frac'],3),d['config']['k_at_start'])"
  timeout -k 10 300 python bench.py --warmup 20000 --steps 1500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary --no-mfma > gpurun_out/r04_pad_late_$pad.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_pad_late_$pad.json'));print('PAD=$pad late value',round(d['value']),'price us',round(d['roofline']['avg_launch_us'],2),'frac',round(d['roofline']['frac'],3),d['config']['k_at_start'])"
  timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary --no-mfma > gpurun_out/r04_pad_early_$pad.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_pad_early_$pad.json'));print('PAD=$pad early value',round(d['value']),'price us',round(d['roofline']['avg_launch_us'],2))"
done
