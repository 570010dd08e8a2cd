mkdir -p gpurun_out/r4g
timeout -k 10 900 python -m pytest tests/test_producer_gpu.py -x -q -m gpu > gpurun_out/r4g/pytest.txt 2>&1
tail -25 gpurun_out/r4g/pytest.txt
