#!/bin/bash
# Usage (GPU box): tools/pmc_conv.sh <name> kh dil nsrc nres  -- SQ counters of one conv config (env PAIF_CONV_WS respected)
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$R/gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/conv_one.py "$@" 3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/b -- python3 $R/tools/conv_one.py "$@" 3 > $OUT/b.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f:
        print("no counters in", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv" not in k or "pack" in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        print(k[:70])
        for c, v in sorted(d.items()):
            print("   %-28s %.4g" % (c, v / 3))
PY
