mkdir -p gpurun_out/r4h
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "posteriors or noblank" > gpurun_out/r4h/pytest.txt 2>&1
tail -6 gpurun_out/r4h/pytest.txt
