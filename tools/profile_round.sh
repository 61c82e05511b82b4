#!/bin/bash
# Profiles bench.py under rocprofv3 on the GPU box: kernel-trace stats, then HBM PMC
# counters in SEPARATE passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is
# never combined with other trace domains).  Output under gpurun_out/prof/<tag>/.
# usage: tools/profile_round.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sparse > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sparse > $OUT/bench_fetch.json 2> $OUT/fetch.err || { echo "pmc fetch failed"; tail -5 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sparse > $OUT/bench_write.json 2> $OUT/write.err || { echo "pmc write failed"; tail -5 $OUT/write.err; exit 1; }
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
