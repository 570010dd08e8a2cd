#!/bin/bash
# rocprofv3 evidence for the bench numbers (run on the GPU box from the repo root):
#   kernel-trace stats, then separate PMC passes for FETCH_SIZE and WRITE_SIZE
#   (they do not fit one pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).
# usage: tools/profile.sh <outdir> <bench args...>
set -u
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --launch eager --steps 50 --warmup 10 --no-cpu-baseline --no-eager-python $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.err"
# exact read bytes from the request-size counters (FETCH_SIZE tallies a 128-B request at 64 B)
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc_rdreq" -- $BENCH > /dev/null 2> "$OUT/rdreq.err"
find "$OUT" -name "*.csv" | head -20
