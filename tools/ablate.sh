#!/bin/bash
# phase ablation of the fused kernel: CTC_AMD_DEBUG_STOP=k leaves after phase k
for k in 1 2 3 4 0; do
  CTC_AMD_DEBUG_STOP=$k timeout -k 10 120 python bench.py --launch eager --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('stop=$k kernel_us avg %.2f (bracketed %.2f) ms/step %.4f' % (r['kernel_us_avg'], r['kernel_us_event_bracketed_avg'], d['ms_per_step']))"
done
