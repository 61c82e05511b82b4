#!/bin/bash
# Kernel-trace statistics (rocprofv3 --kernel-trace --stats) of one bench.py configuration.
# usage: tools/kstats.sh <tag> <bench args...>     (CDHIP_SO selects the library build)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sparse > $OUT/bench.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
echo "== $TAG"
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-70s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
