#!/bin/bash
# bench.py's group lines with two libraries, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"   # (needs lib/libspx_r4a.so: the library of an older commit, built from `git archive <commit> shiftedproximaloperators.jl_amd/csrc include` with SPX_LIB_NAME=libspx_r4a.so)
mkdir -p gpurun_out/r4b
for rep in 1 2 3; do for l in libspx_r4a.so libspx.so; do
  SPX_LIB_NAME=$l SPX_NO_BUILD=1 timeout -k 10 200 python bench.py --no-cpu --steps 20 --warmup 3 > gpurun_out/r4b/ab_$l.$rep.json 2> gpurun_out/r4b/ab_err.txt || { echo "bench failed for $l"; tail -3 gpurun_out/r4b/ab_err.txt; exit 1; }
  python - "$l" "$rep" <<'P'
import json,sys
l,rep=sys.argv[1],sys.argv[2]
b=json.loads(open('gpurun_out/r4b/ab_%s.%s.json'%(l,rep)).read().strip().splitlines()[-1])
o=b['other_operators']
print("%-14s rep %s | L1Box %.4f | Binf 1e6x128 %.4f | plain 128 %.4f | Binf x8 %.4f | sparse %.4f | team Binf %.4f | top-r n/100 %.4f" % (l, rep, b['ms_per_step'], o['ShiftedGroupNormL2Binf_1000000x128']['ms'], o['ShiftedGroupNormL2_1000000x128']['ms'], o['ShiftedGroupNormL2Binf_12500000x8']['ms'], o['ShiftedGroupNormL2Binf_1000000x128_sparse_iterate']['ms'], o['ShiftedGroupNormL2Binf_1x100000000']['ms'], o['ShiftedIndBallL0BInf_r=n/100']['ms']), flush=True)
P
done; done
