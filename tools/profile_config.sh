#!/bin/bash
# rocprofv3 kernel-trace stats of one tools/bench_configs.py configuration (cfg3 | cfg4 | cfg5 | wls) under the environment
# given (CFG_CACHE, CFG_BLOCK, ...): top kernels by total time.  usage: tools/profile_config.sh <tag> <config>
set -o pipefail
TAG=$1; CFG=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/bench_configs.py $CFG > $OUT/line.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
cut -c1-400 $OUT/line.json
python3 - $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/trace/t_kernel_stats.csv")))
for r in rows[:14]:
    print("%-64s %6s %10.3f ms  avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
