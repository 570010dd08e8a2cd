set -x
mkdir -p gpurun_out/r4b
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "noblank or smooth or determin or shard" > gpurun_out/r4b/pytest.txt 2>&1
tail -5 gpurun_out/r4b/pytest.txt
python bench.py --no-eager-python --no-cpu-baseline > gpurun_out/r4b/bench_noblank.json 2> gpurun_out/r4b/bench_noblank.err
python bench.py --no-eager-python --no-cpu-baseline --scaling strong --global-batch 2048 > gpurun_out/r4b/bench_2048.json 2> gpurun_out/r4b/bench_2048.err
for w in 1 2 3 9; do CTC_AMD_DEBUG_STOP=-$w python tools/stamps.py > gpurun_out/r4b/stamps_w$w.txt 2>&1; done
cat gpurun_out/r4b/bench_noblank.json gpurun_out/r4b/bench_2048.json | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][:40], d['ms_per_step'], d['roofline']['kernel_us_avg'], d['roofline']['frac'], d.get('parity'))"
