#!/bin/bash
# PMC passes for bench.py (separate runs per counter group; kernel-trace/stats not combined with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras ${@:2}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
python3 - <<PY
import csv, glob, collections
for grp in ("sq","fetch","write"):
    files = glob.glob("$OUT/%s/*/*counter_collection.csv" % grp)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])] += 1
    print("==", grp)
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
        print(k, {c: "%.4g" % (v / cnt[(k,c)]) for c, v in d.items()})
PY
