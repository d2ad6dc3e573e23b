cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "" stdef; do
  if [ -z "$v" ]; then unset PAIF_LIB; else export PAIF_LIB=paif_amd/lib/libpaif_hip_$v.so; fi
  python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value'],1), round(r['ms_per_step'],4))"
done; done
