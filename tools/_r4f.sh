mkdir -p gpurun_out/r4f
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/pytest.txt 2>&1
tail -12 gpurun_out/r4f/pytest.txt
