#!/bin/bash
# rocprofv3 kernel trace of tools/ref_path_shape.py (benchmark/cd_bench.jl's shape: n = 3000, p = 5000, 60 lambdas down to 774
# non-zeros; three paths per cache mode): what runs on the device when supports outgrow the device loop's LDS block.
# usage: tools/profile_ref_path.sh <tag>
set -o pipefail
TAG=${1:-r4_ref_path}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/tools/ref_path_shape.py > $OUT/paths.txt 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("cdk::", "")[:64]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
print("# kernel trace of tools/ref_path_shape.py under rocprofv3 (n = 3000, p = 5000; 3 paths x 3 cache modes; the mode-0 paths are the streamed sweeps)")
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print("%-66s calls %7s avg %10.2f us total %9.2f ms" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("\n# the paths as timed under the profiler")
for l in open(out + "/paths.txt"):
    print(l.rstrip()[:400])
PY
