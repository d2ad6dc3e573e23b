cd $GRAFT_REPO_ROOT
for v in "" nomfma nold; do
  if [ -z "$v" ]; then unset PAIF_LIB; else export PAIF_LIB=paif_amd/lib/libpaif_hip_$v.so; fi
  echo "== $v"; python tools/rdb_time.py 2>&1 | grep "nres 0  one"
done
