#!/bin/bash
# interleaved A/B of one library with / without an environment switch, ONE device
# usage: tools/abenv.sh VAR [bench args...]     (A: VAR unset, B: VAR=1)
V=$1; shift
for round in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $V=1; else unset $V; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$V=$on kernel_us %.2f ms/step %.4f frac %.3f parity %s' % (r['kernel_us_avg'], d['ms_per_step'], r['frac'], d.get('parity')))"
  done
done
