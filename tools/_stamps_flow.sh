mkdir -p gpurun_out/flow
for wv in 1 2 3 4 5 6 7 10 16; do
  echo "== wave $((wv-1))" >> gpurun_out/flow/stamps.log
  CTC_AMD_DEBUG_STOP=-$wv python tools/stamps.py binary 2>/dev/null | grep slot >> gpurun_out/flow/stamps.log
done
cat gpurun_out/flow/stamps.log
