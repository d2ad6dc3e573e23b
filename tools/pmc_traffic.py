#!/usr/bin/env python3
"""After `tools/pmc_run.sh <name> [bench args]` on the GPU box: turn the FETCH_SIZE / WRITE_SIZE passes into the record bench.py
reads (profiles/pmc_traffic.json), tagged with the hash of the kernel sources it was measured on.
    python3 tools/pmc_traffic.py <workload key> gpurun_out/pmc_<name> [existing json to merge into] > new.json
<workload key> = "fusion/bf16", "fusion/f32", "fusion_seg/bf16", "pgd", "train" (bench.py: workload + "/" + storage for the
inference workloads).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests at 64 B for wide
coalesced reads -> traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 bytes, averaged per launch of the kernel."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16  # noqa: E402


def clean(name):
    """rocprofv3 kernel name -> `kernel<template args>` without return type, namespaces and the argument list."""
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^paif_(gf2w12|gf2|gft|gfb|conv_dma)::", "", n)
    m = re.match(r"([A-Za-z_0-9]+(?:<.*?>)?)\(", n)
    return m.group(1) if m else n.split("(")[0]


wkey, out_dir = sys.argv[1], sys.argv[2]
allrec = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else {}
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for grp in ("fetch", "write"):
    for f in glob.glob(os.path.join(out_dir, grp, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                k = clean(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[(k, row["Counter_Name"])] += 1
allrec["_note"] = ("HBM bytes per launch from rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1 <workload args>` (tools/pmc_run.sh + "
                   "tools/pmc_traffic.py), gfx950 correction applied: traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, averaged over the launches of "
                   "the kernel; one section per workload key.  kernel_source_sha16 ties a section to the kernel sources it was measured on "
                   "(bench.py reports a stale one as such).")
sec = {"_kernel_source_sha16": kernel_source_sha16(), "_round": 6}
for k, d in tot.items():
    if not re.match(r"(conv_|conv3x3_|conv7x7_|gf_|gf2_|gemm_|sr_attention|attn_bwd|dwconv|layernorm|stem_|spa_|eca_|tail_|channel_|head_sum|im2col|col2im|upsample|resize)", k):
        continue
    fs = d["FETCH_SIZE"] / max(1, cnt[(k, "FETCH_SIZE")])
    ws = d["WRITE_SIZE"] / max(1, cnt[(k, "WRITE_SIZE")])
    sec[k] = {"fetch_size_kb": fs, "write_size_kb": ws, "traffic_bytes": int(2 * fs * 1024 + ws * 1024), "launches_fetch_pass": cnt[(k, "FETCH_SIZE")]}
allrec[wkey] = sec
print(json.dumps(allrec, indent=1, sort_keys=True))
