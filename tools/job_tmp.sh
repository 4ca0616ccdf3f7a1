mkdir -p gpurun_out
for lib in "" ab/libmvlt_tnatomic.so "" ab/libmvlt_tnatomic.so; do echo "== $lib"; MVLT_HIP_LIB=$lib python tools/ubench_tn_vocab.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/tn_vocab.txt
python -m pytest tests/test_dist_gpu.py tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/gputest_b.txt
python -m cProfile -s tottime tools/host_time.py 2>&1 | grep -v amdgpu.ids | head -70 > gpurun_out/host_cprofile.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
