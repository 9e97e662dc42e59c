"""Summarise a rocprofv3 kernel-trace CSV: per kernel duration and the gap to the previous kernel's end."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -90:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in tail:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    dur[name].append(en - st)
    if prev_end is not None: gap[name].append(st - prev_end)
    prev_end = en
print("span us", (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3, "kernels", len(tail))
for k in dur:
    g = gap.get(k, [0])
    print(f"{k:42s} n={len(dur[k]):3d} dur {sum(dur[k])/len(dur[k])/1e3:7.2f} us  gap-before {sum(g)/len(g)/1e3:7.2f} us")
