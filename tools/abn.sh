#!/bin/bash
# interleaved comparison of N library builds: tools/abn.sh lib1.so lib2.so ... -- [bench args]
LIBS=(); while [ "$1" != "--" ] && [ -n "$1" ]; do LIBS+=("$(realpath $1)"); shift; done; shift
for round in 1 2 3; do
  for lib in "${LIBS[@]}"; do
    CTC_AMD_LIB=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-eager-python --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$(basename $lib) kernel_us %.2f ms/step %.4f frac %.3f' % (r['kernel_us_avg'], d['ms_per_step'], r['frac']))"
  done
done
