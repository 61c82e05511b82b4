#!/bin/bash
# rocprofv3 evidence for the Gram-column kernel k_cross (grad_cache.hpp / small_solve.hpp use it) and for the cfg3 path that
# is built on it: kernel-trace stats, then FETCH_SIZE, WRITE_SIZE and the matrix-pipe counters in SEPARATE --pmc passes
# (never combined with trace domains).  usage: tools/profile_cross.sh <tag>   (needs exp/cross_bench: tools/build_cross_bench.sh)
set -o pipefail
TAG=${1:-r3_cross}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the library's kernel on cfg3's shape: one reference pass + 3 batches of 32 columns, as the cfg3 path fetches them
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/tools/bench_configs.py cfg3 > $OUT/cfg3_trace.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o c -- python3 $R/tools/bench_configs.py cfg3 > $OUT/cfg3_$C.json 2> $OUT/$C.err || { echo "pmc $C failed"; tail -5 $OUT/$C.err; }
done
# the A/B harness (round 2's kernel, k_cross2, the ring variants; loads-only / matrix-only / clock stamps)
$R/exp/cross_bench 2000000 5000 3 > $OUT/cross_bench_f64.txt 2>&1
$R/exp/cross_bench 2000000 5000 2 f32 > $OUT/cross_bench_f32.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def short(n): return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("cdk::", "")[:64]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
print("# kernel trace of tools/bench_configs.py cfg3 (n = 2e6, p = 5000, 100 lambdas)")
if f:
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print("%-66s calls %6s avg %10.2f us total %9.2f ms" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("\n# counters per launch (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KiB -> MB, FETCH_SIZE x2 on gfx950)")
for C in ("FETCH_SIZE", "WRITE_SIZE", "MfmaUtil"):
    g = glob.glob(f"{out}/pmc_{C}/**/*counter_collection.csv", recursive=True)
    if not g:
        print(f"{C}: no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == C: acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
        m = sum(v) / len(v)
        if C == "FETCH_SIZE": print(f"{C:12s} {k:64s} launches {len(v):5d}  {2 * 1024 * m / 1e6:12.2f} MB")
        elif C == "WRITE_SIZE": print(f"{C:12s} {k:64s} launches {len(v):5d}  {1024 * m / 1e6:12.2f} MB")
        else: print(f"{C:12s} {k:64s} launches {len(v):5d}  mean {m:8.3f}  max {max(v):8.3f}")
for j in ("cfg3_trace.json",):
    print("\n# " + j + "\n" + open(out + "/" + j).read().strip())
PY
