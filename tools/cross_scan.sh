#!/bin/bash
# k_cross tuning: kernel-trace total of the cfg3 path's Gram-column batches for each library build under exp_build/
# (hipcc -D-DCDH_CROSS_UH=.. -DCDH_CROSS_OCC=..).  usage: tools/cross_scan.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
for so in $R/exp_build/cross_*.so $R/coordinatedescent.jl_amd/csrc/libcdhip.so; do
  tag=$(basename $so .so)
  CDHIP_SO=$so CFG_BLOCK=32 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/cs_$tag -o trace -- python3 $R/tools/bench_configs.py cfg3 > $R/gpurun_out/prof/cs_$tag.json 2> $R/gpurun_out/prof/cs_$tag.err || { echo "$tag failed"; tail -3 $R/gpurun_out/prof/cs_$tag.err; continue; }
  python3 - $R/gpurun_out/prof/cs_$tag $tag <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_cross" in r["Name"]:
        print("%-16s k_cross calls %4s avg %8.1f us total %7.2f ms" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6), end="  ")
print("path %.3f s" % json.load(open(sys.argv[1] + ".json"))["seconds"])
PY
done
