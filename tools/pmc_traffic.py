#!/usr/bin/env python3
"""After `tools/pmc_run.sh <name>` on the GPU box: turn the FETCH_SIZE / WRITE_SIZE passes into the record bench.py reads
(profiles/pmc_traffic.json), tagged with the hash of the kernel source it was measured on.
    python3 tools/pmc_traffic.py gpurun_out/pmc_<name> > gpurun_out/pmc_traffic.json
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads ->
traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 bytes, averaged per launch of the kernel."""
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

out_dir = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for grp in ("fetch", "write"):
    for f in glob.glob(os.path.join(out_dir, grp, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[(row["Kernel_Name"], row["Counter_Name"])] += 1
rec = {"_note": "HBM bytes per launch from rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1` (tools/pmc_run.sh + "
                "tools/pmc_traffic.py), gfx950 correction applied: traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, averaged over the "
                "launches of the kernel.  kernel_source_sha16 ties the record to the source it was measured on (bench.py refuses a stale one)."}
for k, d in tot.items():
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?((?:conv_|gf_)\w+(?:<[^>]*>)?)", k)
    if not m:
        continue
    fs = d["FETCH_SIZE"] / max(1, cnt[(k, "FETCH_SIZE")])
    ws = d["WRITE_SIZE"] / max(1, cnt[(k, "WRITE_SIZE")])
    rec[m.group(1)] = {"fetch_size_kb": fs, "write_size_kb": ws, "traffic_bytes": int(2 * fs * 1024 + ws * 1024),
                       "launches_fetch_pass": cnt[(k, "FETCH_SIZE")], "kernel_source_sha16": kernel_source_sha16(), "round": 2}
print(json.dumps(rec, indent=2))
