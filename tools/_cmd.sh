python -m pytest tests/test_fusion_gpu.py tests/test_attack_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-130
done
bash tools/prof_run.sh fusion_r2c --steps 20 --warmup 3 2>&1 | head -20 | cut -c1-110
