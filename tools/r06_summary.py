"""Prints the figures DESIGN section 6 quotes from the evidence set under profiles/ (after tools/collect_r06_copy.sh)."""
import csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(R, "profiles", f)
def line(f):
    return json.loads(open(P(f)).read().strip().splitlines()[-1])
for f in ("fusion", "fusion_f32", "fusion_seg", "pgd", "train", "fusion_graph", "fusion_two_stream_timed"):
    d = line("r06_bench_%s.json" % f)
    print(f, "value %.1f ms %.3f sustained %s two_stream %s" % (d["value"], d["ms_per_step"], d.get("sustained_value"), d.get("two_stream", {}).get("value")),
          "other", [(o["storage"], round(o["value"], 1)) for o in d.get("other_storage", [])],
          "also", {k: round(v["value"], 2) for k, v in d.get("also", {}).items() if isinstance(v, dict) and "value" in v},
          "roofline frac %s incl-res %s traffic %s" % (d["roofline"].get("frac"), d["roofline"].get("hbm_frac_counting_residual_reads"), d["roofline"].get("traffic")),
          "survey frac", d.get("whole_step_survey_hbm_frac"), "cpu", d.get("cpu_baseline", {}).get("by_batch"))
for w, steps in (("fusion_f16", 45), ("fusion_f32", 45), ("fusion_bf16", 45)):
    rows = list(csv.DictReader(open(P("r06_%s_kernel_stats.csv" % w))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", w, "GPU ms per step %.3f" % (tot / steps / 1e6))
    for r in rows[:19]:
        print("  %-92s x%-4s %8.1f us %5.2f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:92], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
for w in ("pgd", "train", "fusion_seg"):
    rows = list(csv.DictReader(open(P("r06_%s_kernel_stats.csv" % w))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", w, "GPU ms total %.1f" % (tot / 1e6))
    for r in rows[:14]:
        print("  %-92s x%-5s %8.1f us %5.2f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:92], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
    gf = [r for r in rows if "gf_" in r["Name"] or "gf2" in r["Name"]]
    print("  guided filter share %.2f %%" % (sum(float(r["TotalDurationNs"]) for r in gf) / tot * 100))
t = json.load(open(P("pmc_traffic.json")))["fusion/f16"]
print("PMC fusion/f16:", {k: round(v["traffic_bytes"] / 1e6, 1) for k, v in t.items() if not k.startswith("_") and v["traffic_bytes"] > 5e7})
