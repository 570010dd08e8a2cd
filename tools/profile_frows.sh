#!/bin/bash
# rocprofv3 --kernel-trace --stats of the SURVEY 8(f) kernels (tools/frow_step.py); usage: tools/profile_frows.sh <outdir>
set -u
OUT=$(realpath -m "$1")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/frow_step.py > "$OUT/run.log" 2> "$OUT/run.err"
find "$OUT" -name "*kernel_stats.csv" | head -3
