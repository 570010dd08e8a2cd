#!/bin/bash
# per-SAMPLE stamps of the persistent no-blank launch (workgroup 0): slots 0..5 = when the wave leaves the barrier behind the
# lattice's initialisation for samples 0..5, slots 6..11 = when it is done with them; usage: tools/stamps_samples.sh B [waves...]
B=${1:-2048}; shift
for w in ${@:-0 2 1 8 15}; do
  echo "== wave $w B=$B"
  CTC_AMD_DEBUG_STOP=$((-400 - w)) python tools/stamps.py noblank $B 2>&1 | grep slot
done
