#!/bin/bash
# The four driver-shaped bench lines of the round-5 evidence set only (no profiler passes): refreshes gpurun_out/r05_bench_{fusion,fusion_seg,pgd,train}.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r05_bench_fusion.json 2> gpurun_out/r05_bench_fusion.err
python3 bench.py --workload fusion_seg --steps 10 --warmup 3 > gpurun_out/r05_bench_fusion_seg.json 2> gpurun_out/r05_bench_fusion_seg.err
python3 bench.py --workload pgd --steps 3 --warmup 1 --sustain-seconds 0 > gpurun_out/r05_bench_pgd.json 2> gpurun_out/r05_bench_pgd.err
python3 bench.py --workload train --steps 3 --warmup 1 --sustain-seconds 0 > gpurun_out/r05_bench_train.json 2> gpurun_out/r05_bench_train.err
python3 - <<'PY'
import json
for f in ["fusion", "fusion_seg", "pgd", "train"]:
    d = json.loads(open("gpurun_out/r05_bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"].get("traffic"), d["roofline"].get("traffic_note"),
          {k: round(v["value"], 1) for k, v in d.get("also", {}).items() if isinstance(v, dict) and "value" in v})
PY
