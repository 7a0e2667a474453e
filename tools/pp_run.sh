timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/t16.log 2>&1; tail -3 gpurun_out/t16.log | cut -c1-250
python bench.py --no-cpu-baseline > gpurun_out/bench_r02_i.json 2>/dev/null; python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/bench_r02_i.json") if l.startswith("{")][-1]
r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["traffic"])
for k,v in list(r["per_kernel_ms_per_step"].items())[:14]: print(k, v, r["per_kernel_tflops"].get(k,""))
print(r["per_class_ms_per_step"])
PY
