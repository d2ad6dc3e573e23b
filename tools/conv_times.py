"""Print per-launch conv / guided-filter kernel times of the last bench step from a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]][-n:]
out = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    out.append("%s:%.0f" % (name.split("(")[0][:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("  ".join(out))
print("span ms: %.3f" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
