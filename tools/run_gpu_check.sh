#!/bin/bash
# standard GPU check: full gpu test suite, then bench (no cpu leg), compact table.  Usage: tools/run_gpu_check.sh [tag]
TAG="${1:-x}"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$TAG.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 4 gpurun_out/pytest_$TAG.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 280 python bench.py --steps 20 --no-cpu > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_$TAG.log") if l.startswith("{")][-1])
print("L1Box: %.1f GB/s  %.4f ms/step (wall)" % (d["roofline"]["achieved"], d["ms_per_step"]))
for k,v in d["other_operators"].items(): print("%-44s %9.4f ms %7.1f GB/s %.3f"%(k,v["ms"],v.get("gbs_algorithmic",v.get("gbs_over_pcie",0)),v.get("frac_of_peak",0)))
PY
