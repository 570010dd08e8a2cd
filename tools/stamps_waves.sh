#!/bin/bash
# phase stamps (tools/stamps.py) of several waves of workgroup 0, one run per wave, condensed to one line per wave
# usage: tools/stamps_waves.sh <variant> <wave> [<wave> ...]      e.g. tools/stamps_waves.sh binary 0 1 2 3 4 5 6 9 15
#        CTC_STAMPS_B=2048 tools/stamps_waves.sh noblank 0 2 1 15   (a batch beyond 2 x #CUs: the persistent form, LAST sample's stamps)
V=$1; shift
mkdir -p gpurun_out/stamps
OUT=gpurun_out/stamps/$V.log
: > $OUT
for wv in "$@"; do
  echo "== wave $wv" >> $OUT
  CTC_AMD_DEBUG_STOP=-$((wv+1)) python tools/stamps.py $V ${CTC_STAMPS_B:-256} 2>/dev/null | grep "slot " >> $OUT
done
grep -v "slot:" $OUT | sed "s/ *[0-9]* cyc//" | paste -sd" " | sed "s/== /\n== /g"; echo
