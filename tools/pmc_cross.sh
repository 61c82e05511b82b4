#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MfmaUtil of the Gram-column kernel in the A/B harness (exp/cross_bench), separate --pmc passes.
# usage: tools/pmc_cross.sh <tag> [f32]
set -o pipefail
TAG=${1:-r4_cross}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o c -- $R/exp/cross_bench 2000000 5000 1 $2 > $OUT/bench_$C.txt 2> $OUT/$C.err || { echo "pmc $C failed"; tail -5 $OUT/$C.err; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("cdk::", "")[:72]
for C in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
    g = glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True)
    if not g:
        print(f"{C}: no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == C: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "k_cross<" not in k: continue
        m = sum(v) / len(v)
        if C == "FETCH_SIZE": print(f"{C:12s} {k:72s} launches {len(v):5d}  {2 * 1024 * m / 1e6:12.2f} MB")
        elif C == "WRITE_SIZE": print(f"{C:12s} {k:72s} launches {len(v):5d}  {1024 * m / 1e6:12.2f} MB")
        else: print(f"{C:12s} {k:72s} launches {len(v):5d}  mean {m:8.3f}  max {max(v):8.3f}")
PY
