#!/bin/bash
# rocprofv3 PMC passes of the SQ (issue) counters for one bench configuration: how busy the SIMDs' issue ports are, by
# instruction kind.  usage: tools/pmc_sq.sh <outdir> <bench args...>     (run on the GPU box from the repo root)
set -u
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --launch eager --steps 30 --warmup 5 --no-cpu-baseline --no-eager-python $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_TRANS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- $BENCH > /dev/null 2> "$OUT/p$i.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in tot.items():
    if "ctc" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  (avg of %d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
