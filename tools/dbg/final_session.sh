python bench.py --config cfg3 > gpurun_out/r3k_bench_cfg3.json 2> gpurun_out/r3k_bench_cfg3.err
v=$(python -c "import json;print(int(json.loads(open('gpurun_out/r3k_bench_cfg3.json').read().strip().splitlines()[-1])['value']))")
echo "cfg3 $v"
if [ "$v" -lt 279500 ]; then echo "slow box: stopping"; exit 0; fi
for c in cfg2 cfg4; do python bench.py --config $c > gpurun_out/r3k_bench_$c.json 2> gpurun_out/r3k_bench_$c.err; echo $c done; done
python bench.py --config cfg5 --steps 2 --warmup 1 > gpurun_out/r3k_bench_cfg5.json 2> gpurun_out/r3k_bench_cfg5.err; echo cfg5 done
bash tools/prof.sh cfg3 > gpurun_out/r3k_prof.log 2>&1; tail -n 1 gpurun_out/r3k_prof.log
bash tools/prof.sh cfg2 > gpurun_out/r3k_prof2.log 2>&1; tail -n 1 gpurun_out/r3k_prof2.log
