mkdir -p gpurun_out/$1; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/$1/tests.log 2>&1; tail -3 gpurun_out/$1/tests.log; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$1/bench.out 2> gpurun_out/$1/bench.err; python - <<PY
import json
d=json.loads(open("gpurun_out/$1/bench.out").read().strip().splitlines()[-1])
print("cfg3", d["value"], d["roofline"]["avg_launch_ms"])
for k,v in d["workloads"].items():
    if isinstance(v,dict): print(k, v["value"], v.get("count_only_gibps"), v["roofline"]["avg_launch_ms"], v.get("build_s"), v.get("host_results"))
PY
