set -x
mkdir -p gpurun_out/r4a
python bench.py --no-eager-python > gpurun_out/r4a/bench_noblank.json 2> gpurun_out/r4a/bench_noblank.err
bash tools/pmc_sq.sh gpurun_out/r4a/sq_noblank > gpurun_out/r4a/sq_noblank.txt 2>&1
bash tools/pmc_sq.sh gpurun_out/r4a/sq_blank --variant blank > gpurun_out/r4a/sq_blank.txt 2>&1
bash tools/pmc_sq.sh gpurun_out/r4a/sq_noblank2048 --scaling strong --global-batch 2048 > gpurun_out/r4a/sq_noblank2048.txt 2>&1
for w in 1 2 3 4 8 16; do CTC_AMD_DEBUG_STOP=-$w python tools/stamps.py > gpurun_out/r4a/stamps_w$w.txt 2>&1; done
CTC_AMD_DEBUG_STOP=-50 python tools/stamps.py > gpurun_out/r4a/stamps_spread.txt 2>&1
./tools/micro/launch_cost > gpurun_out/r4a/launch_cost.txt 2>&1
rm -rf gpurun_out/r4a/sq_*/p*/
ls gpurun_out/r4a
