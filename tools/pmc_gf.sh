#!/bin/bash
# PMC passes for the guided-filter timing script (separate runs per counter group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gf
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o gf -- python3 $R/tools/gf_time.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/tools/gf_time.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq2 -- python3 $R/tools/gf_time.py > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/fetch -- python3 $R/tools/gf_time.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/gf_time.py > $OUT/write.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:6]:
        print("%-60s %5s %10s %6s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"]))
for grp in ("sq","sq2","fetch","write"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % grp, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for ff in files:
        for row in csv.DictReader(open(ff)):
            k = row["Kernel_Name"][:40]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
    print("==", grp)
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:3]:
        print(k, {c: "%.4g" % (v / cnt[(k,c)]) for c, v in d.items()})
PY
