#!/bin/bash
# kernel time over batch sizes, with and without an environment switch of the diagnostics build
# usage: tools/batch_sweep.sh <variant> <ENV_SWITCH> B1 B2 ...      e.g. tools/batch_sweep.sh binary CTC_AMD_BINARY_NOFLOW 64 128 256 512 1024 2048
V=$1; SW=$2; shift 2
export CTC_AMD_LIB=ctc_amd/lib/libctc_amd_diag.so
for B in "$@"; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $SW=1; else unset $SW; fi
    timeout -k 10 120 python bench.py --variant $V --batch $B --no-cpu-baseline --no-eager-python --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('B=%5d $SW=$on kernel_us %.2f frac %.3f' % ($B, r['kernel_us_avg'], r['frac']))"
  done
done
