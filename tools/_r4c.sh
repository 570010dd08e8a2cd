set -x
mkdir -p gpurun_out/r4c
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_status.py -x -q -m gpu -k "noblank or smooth or determin or shard or starved or status" > gpurun_out/r4c/pytest.txt 2>&1
tail -15 gpurun_out/r4c/pytest.txt
timeout -k 10 300 python bench.py --no-eager-python --no-cpu-baseline > gpurun_out/r4c/bench_noblank.json 2> gpurun_out/r4c/bench_noblank.err
CTC_AMD_LIB=ctc_amd/lib/libctc_amd_diag.so CTC_AMD_NOKM=1 timeout -k 10 300 python bench.py --no-eager-python --no-cpu-baseline > gpurun_out/r4c/bench_noblank_r16.json 2> gpurun_out/r4c/bench_noblank_r16.err
timeout -k 10 300 python bench.py --no-eager-python --no-cpu-baseline --scaling strong --global-batch 2048 > gpurun_out/r4c/bench_2048.json 2> gpurun_out/r4c/bench_2048.err
for w in 1 2 3 4 5 6 16; do CTC_AMD_DEBUG_STOP=-$w timeout -k 10 120 python tools/stamps.py > gpurun_out/r4c/stamps_w$w.txt 2>&1; done
cat gpurun_out/r4c/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['roofline']['kernel_us_avg'], d['roofline']['frac'], d.get('parity'))"
