mkdir -p gpurun_out/r4e
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_status.py -x -q -m gpu -k "noblank or smooth or determin or shard or starved or status" > gpurun_out/r4e/pytest.txt 2>&1
tail -3 gpurun_out/r4e/pytest.txt
timeout -k 10 300 python bench.py --no-eager-python --no-cpu-baseline > gpurun_out/r4e/bench_noblank.json 2> gpurun_out/r4e/bench_noblank.err
timeout -k 10 300 python bench.py --no-eager-python --no-cpu-baseline --scaling strong --global-batch 2048 > gpurun_out/r4e/bench_2048.json 2> gpurun_out/r4e/bench_2048.err
for w in 1 2 3 9 5 16; do CTC_AMD_DEBUG_STOP=-$w timeout -k 10 120 python tools/stamps.py > gpurun_out/r4e/stamps_w$w.txt 2>&1; done
for f in gpurun_out/r4e/stamps_w*.txt; do echo == $f; grep "slot [0-9]" $f | awk '{printf "%s %s | ", $2, $3}'; echo; done
cat gpurun_out/r4e/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['roofline']['kernel_us_avg'], d['roofline']['frac'], d.get('parity'))"
