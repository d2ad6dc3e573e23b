python -m pytest tests/test_fusion_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err; tail -c 6000 gpurun_out/bench_r2d.json
