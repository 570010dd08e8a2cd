for w in -1 -3 -2 -8 -16; do echo "== wave stop=$w B=$1"; CTC_AMD_DEBUG_STOP=$w python tools/stamps.py noblank $1 2>&1 | tail -14; done
