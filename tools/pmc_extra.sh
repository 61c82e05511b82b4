#!/bin/bash
# LDS bank conflicts and matrix-pipe utilisation of the dominant kernel (derived rocprofv3 counters,
# one --pmc pass each, never combined with trace domains).  usage: tools/pmc_extra.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in LDSBankConflict MfmaUtil LdsUtil; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o c -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sparse > $OUT/bench_$C.json 2> $OUT/$C.err || { echo "pmc $C failed"; tail -5 $OUT/$C.err; exit 1; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for C in ("LDSBankConflict", "MfmaUtil", "LdsUtil"):
    f = glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1]))[:4]:
        print(f"{C:16s} {k:60s} launches {len(v):4d}  mean {sum(v)/len(v):8.3f}  max {max(v):8.3f}")
PY
