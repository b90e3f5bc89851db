#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SPX_NO_BUILD=1
for f in fuzz_misc_scenarios fuzz_topr_seeds fuzz_topr_views fuzz_obj_box fuzz_lhalf_scenarios fuzz_binf_scenarios fuzz_binf_ties fuzz_ragged_many fuzz_many_instances fuzz_lattice_separable fuzz_b2_scenarios fuzz_topr_big; do
  echo "== $f" >> gpurun_out/fuzzers.log
  timeout -k 10 240 python tools/$f.py > gpurun_out/fz_$f.log 2>&1; rc=$?
  echo "rc=$rc $(tail -n 2 gpurun_out/fz_$f.log | tr '\n' ' ' | cut -c1-300)" >> gpurun_out/fuzzers.log
done
cat gpurun_out/fuzzers.log
