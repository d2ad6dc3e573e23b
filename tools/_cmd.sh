for i in 1 2; do
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-130
PAIF_LIB=$PWD/paif_amd/lib/libpaif_hip_nt.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-130
done
python -m pytest tests/test_fusion_gpu.py -m gpu -x -q 2>&1 | tail -2
