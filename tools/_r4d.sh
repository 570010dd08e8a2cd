mkdir -p gpurun_out/r4d
for w in 1 2 3 4; do CTC_AMD_DEBUG_STOP=-$w timeout -k 10 120 python tools/stamps.py > gpurun_out/r4d/stamps_w$w.txt 2>&1; done
for f in gpurun_out/r4d/stamps_w*.txt; do echo == $f; grep "slot [0-9]" $f | awk '{printf "%s %s | ", $2, $3}'; echo; done
