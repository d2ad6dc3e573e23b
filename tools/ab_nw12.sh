cd $GRAFT_REPO_ROOT
export PAIF_LIB=paif_amd/lib/libpaif_hip_nw12.so
python -m pytest tests/test_fusion_gpu.py tests/test_f16_storage_gpu.py tests/test_bf16_storage_gpu.py -x -q -m gpu -k "guided_filter or gf or decomposition" 2>&1 | tail -4
unset PAIF_LIB
for i in 1 2; do
for v in "" nw12; do
  if [ -z "$v" ]; then unset PAIF_LIB; else export PAIF_LIB=paif_amd/lib/libpaif_hip_$v.so; fi
  python tools/gf_time.py 2>&1 | grep -v amdgpu.ids
  python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=[$v]', round(r['value'],1), round(r['ms_per_step'],4))"
done; done
