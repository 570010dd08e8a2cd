#!/bin/bash
# interleaved A/B of two builds of the library in ONE process group on ONE device
# usage: tools/ab.sh <libA.so> <libB.so> [bench args...]
A=$(realpath $1); B=$(realpath $2); shift 2
for round in 1 2 3; do
  for lib in $A $B; do
    CTC_AMD_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$(basename $lib) kernel_us %.2f ms/step %.4f frac %.3f' % (r['kernel_us_avg'], d['ms_per_step'], r['frac']))"
  done
done
