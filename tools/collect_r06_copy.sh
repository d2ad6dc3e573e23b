#!/bin/bash
# Run in the BUILD CONTAINER after tools/collect_r06.sh came back: copy the evidence that is to be judged from gpurun_out/ (scratch) into
# profiles/ (tracked).  tests/test_evidence.py then holds profiles/pmc_traffic.json to the kernel sources it was measured on.
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/pmc_traffic.json profiles/pmc_traffic.json
cp $G/f16_storage_report.json profiles/r06_f16_storage_report.json
for w in fusion_f16 fusion_f32 fusion_bf16 fusion_seg pgd train; do
  cp $G/prof_$w/${w}_kernel_stats.csv profiles/r06_${w}_kernel_stats.csv
done
for w in fusion_f16 fusion_f32 fusion_bf16 fusion_seg; do cp $G/pmc_$w.txt profiles/r06_pmc_$w.txt; done
for f in fusion fusion_f32 fusion_seg pgd train fusion_graph fusion_two_stream_timed; do cp $G/r06_bench_$f.json profiles/r06_bench_$f.json; done
ls -la profiles/ | grep r06
cp $G/rccl_one_rank.json profiles/r06_rccl_one_rank.json
cp $G/bench_selflaunch.json profiles/r06_bench_selflaunch.json
