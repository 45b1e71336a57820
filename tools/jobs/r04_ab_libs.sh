# round 4: same-box A/B of two builds of the library (tools/jobs/_ab/lib_<name>.so): early regime of configs 3 and 2, config 4 (sparse) at k ~ 2 200
# usage: r04_ab_libs.sh nameA nameB
mkdir -p gpurun_out
cp dantzig_amd/libdantzig_amd.so /tmp/lib_keep.so
for rep in 1 2; do
for v in $1 $2; do
  cp tools/jobs/_ab/lib_$v.so dantzig_amd/libdantzig_amd.so
  timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_ab_${v}_c3.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_ab_${v}_c3.json'));print('$v config3 value',round(d['value']),'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'])"
  timeout -k 10 200 python bench.py --rows 1024 --cols 2048 --seed 1002 --steps 5000 --warmup 500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_ab_${v}_c2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_ab_${v}_c2.json'));print('$v config2 value',round(d['value']),'ms',d['ms_per_step'])"
  timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --warmup 20000 --steps 3000 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_ab_${v}_c4.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_ab_${v}_c4.json'));print('$v config4 value',round(d['value']),'ms',d['ms_per_step'])"
done
done
cp /tmp/lib_keep.so dantzig_amd/libdantzig_amd.so
