#!/bin/bash
# Round-6 evidence collection on the GPU box (one gpurun call; every step under its own `timeout`: a hung profiler run once cost 40 GPU-minutes): kernel stats of every workload, PMC passes, bench lines, the storage
# clause report.  Everything lands under gpurun_out/; the BUILD CONTAINER copies what is to be judged into profiles/ afterwards
# (tools/collect_r06_copy.sh) -- a copy made on the GPU box is lost with the box.
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
prof() { timeout 600 bash tools/prof_bench.sh "$@" > gpurun_out/prof_$1.txt 2>&1; }
prof fusion_f16 --storage f16 --steps 20 --warmup 5
prof fusion_f32 --storage f32 --steps 20 --warmup 5
prof fusion_bf16 --storage bf16 --steps 20 --warmup 5
prof fusion_seg --workload fusion_seg --steps 10 --warmup 3
prof pgd --workload pgd --steps 2 --warmup 1
prof train --workload train --steps 2 --warmup 1
timeout 900 bash tools/pmc_run.sh fusion_f16 --storage f16 > gpurun_out/pmc_fusion_f16.txt 2>&1
timeout 900 bash tools/pmc_run.sh fusion_f32 --storage f32 > gpurun_out/pmc_fusion_f32.txt 2>&1
timeout 900 bash tools/pmc_run.sh fusion_bf16 --storage bf16 > gpurun_out/pmc_fusion_bf16.txt 2>&1
timeout 900 bash tools/pmc_run.sh fusion_seg --workload fusion_seg > gpurun_out/pmc_fusion_seg.txt 2>&1
python3 tools/pmc_traffic.py fusion/f16 gpurun_out/pmc_fusion_f16 > gpurun_out/pmc_t1.json
python3 tools/pmc_traffic.py fusion/f32 gpurun_out/pmc_fusion_f32 gpurun_out/pmc_t1.json > gpurun_out/pmc_t2.json
python3 tools/pmc_traffic.py fusion/bf16 gpurun_out/pmc_fusion_bf16 gpurun_out/pmc_t2.json > gpurun_out/pmc_t3.json
python3 tools/pmc_traffic.py fusion_seg/f16 gpurun_out/pmc_fusion_seg gpurun_out/pmc_t3.json > gpurun_out/pmc_traffic.json
mkdir -p gpurun_out/profiles_tmp
cp profiles/pmc_traffic.json gpurun_out/profiles_tmp/pmc_traffic_before.json 2>/dev/null
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # (for the bench lines below, on this box only)
timeout 900 python3 -m pytest tests/test_f16_storage_gpu.py -q -k clause > gpurun_out/r06_clause_test.txt 2>&1
cp gpurun_out/f16_storage_report.json profiles/r06_f16_storage_report.json 2>/dev/null
PAIF_REQUIRE_RCCL_TEST=1 timeout 900 python3 -m pytest tests/test_00_rccl_gpu.py -q -m gpu > gpurun_out/r06_rccl_tests.txt 2>&1      # -> rccl_one_rank.json, bench_selflaunch.json
timeout 900 python3 bench.py > gpurun_out/r06_bench_fusion.json 2> gpurun_out/r06_bench_fusion.err
timeout 900 python3 bench.py --steps 20 --warmup 5 --storage f32 --no-also --no-cpu-baseline > gpurun_out/r06_bench_fusion_f32.json 2> gpurun_out/r06_bench_fusion_f32.err
timeout 900 python3 bench.py --workload fusion_seg --steps 10 --warmup 3 > gpurun_out/r06_bench_fusion_seg.json 2> gpurun_out/r06_bench_fusion_seg.err
timeout 900 python3 bench.py --workload pgd --steps 3 --warmup 1 --sustain-seconds 0 > gpurun_out/r06_bench_pgd.json 2> gpurun_out/r06_bench_pgd.err
timeout 900 python3 bench.py --workload train --steps 3 --warmup 1 --sustain-seconds 0 > gpurun_out/r06_bench_train.json 2> gpurun_out/r06_bench_train.err
timeout 900 python3 bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_fusion_graph.json 2> gpurun_out/r06_bench_fusion_graph.err
timeout 900 python3 bench.py --two-stream --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_fusion_two_stream_timed.json 2> gpurun_out/r06_bench_fusion_two_stream_timed.err
timeout 1500 python3 -m pytest tests/ -q -m gpu > gpurun_out/r06_gpu_suite.txt 2>&1; tail -3 gpurun_out/r06_gpu_suite.txt
python3 - <<'PY'
import json
for f in ["fusion", "fusion_f32", "fusion_seg", "pgd", "train", "fusion_graph", "fusion_two_stream_timed"]:
    try:
        d = json.loads(open("gpurun_out/r06_bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), round(d["ms_per_step"], 3), [(o["storage"], round(o["value"], 1)) for o in d.get("other_storage", [])],
              d.get("cpu_baseline", {}).get("value"), d["roofline"].get("frac"), d["roofline"].get("traffic"), d["roofline"].get("traffic_note"),
              "two_stream", d.get("two_stream", {}).get("value"), {k: round(v["value"], 1) for k, v in d.get("also", {}).items() if isinstance(v, dict) and "value" in v})
    except Exception as e:
        print(f, "ERR", e)
PY
