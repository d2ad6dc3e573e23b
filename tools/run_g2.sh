cd $GRAFT_REPO_ROOT
python tools/gemm2_time.py bf16x3
for v in g2e1 g2e2 g2e4 g2e6 g2t4 g2t6; do PAIF_LIB=paif_amd/lib/libpaif_hip_$v.so python tools/gemm2_time.py bf16x3; done
