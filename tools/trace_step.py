"""print one step of a rocprofv3 kernel trace: tools/trace_step.py <kernel_trace.csv> <marker kernel substring> [which]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
w = int(sys.argv[3]) if len(sys.argv) > 3 else -3
s, e = idx[w], idx[w + 1]
t0 = int(rows[s]["Start_Timestamp"])
busy = 0
for r in rows[s:e]:
    st = int(r["Start_Timestamp"]) - t0
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    print(f'{st/1e3:8.1f} {d/1e3:6.1f} q{r["Queue_Id"]} g{r["Grid_Size_X"]:>8s} {r["Kernel_Name"][:100]}')
print("launches", e - s, "busy us", busy / 1e3, "span us", (int(rows[e]["Start_Timestamp"]) - t0) / 1e3)
