python -m pytest tests/test_dist_gpu.py -q -m gpu -x -k phased 2>&1 | tail -40 > gpurun_out/gputest_c.txt
