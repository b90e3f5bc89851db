"""Do the `kernel` strings of a bench.py JSON line name kernels that rocprofv3 saw?  usage: check_bench_kernels.py <bench json> <kernel stats txt>
Every template instantiation named in the JSON (tokens like k_xxx<...> or k_xxx) must occur, with the namespace prefixes removed,
in the kernel-stats file (profiles/r04_all_ops_kernel_stats.txt)."""
import json, re, sys
line = [l for l in open(sys.argv[1]) if l.lstrip().startswith("{")][-1]
d = json.loads(line)
stats = open(sys.argv[2]).read().replace("(anonymous namespace)::", "").replace("void ", "")
names = set()
def walk(o):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "kernel" and isinstance(v, str):
                names.update(m.group(0) for m in re.finditer(r"k_[a-z0-9_]+(<(?:[^<>]|<[^<>]*>)*>)?", v))
            else:
                walk(v)
    elif isinstance(o, list):
        for v in o: walk(v)
walk(d)
missing = [n for n in sorted(names) if n not in stats]
print("%d kernel names in the bench line, %d not found in the profile" % (len(names), len(missing)))
for n in missing: print("  MISSING:", n)
sys.exit(1 if missing else 0)
