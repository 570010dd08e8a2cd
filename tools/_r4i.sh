mkdir -p gpurun_out/r4i
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_status.py -x -q -m gpu -k "noblank or smooth or determin or shard or starved or status or posteriors" > gpurun_out/r4i/pytest.txt 2>&1
tail -3 gpurun_out/r4i/pytest.txt
bash tools/abn.sh ctc_amd/lib/variants/step1.so ctc_amd/lib/variants/prev.so ctc_amd/lib/libctc_amd.so -- 2>&1 | tee gpurun_out/r4i/ab.txt
