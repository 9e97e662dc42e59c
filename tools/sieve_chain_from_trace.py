"""The sieve step's dispatch chain from a rocprofv3 kernel trace: mean duration per position over the last STEPS steps.

    python tools/sieve_chain_from_trace.py gpurun_out/r3p/trace/t_kernel_trace.csv [steps=20]

A step is every `mir::` dispatch from one prep kernel (prep_queries16_kernel / prep_queries_i8_stats_kernel) to the next; the table in profiles/r03_bench_kernel_stats.md
is this script's output."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mir::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps_wanted = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps, cur = [], None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "prep_queries16_kernel" in name or "prep_queries_i8_stats_kernel" in name:
        cur = []
        steps.append(cur)
    if cur is not None:
        cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
shape = [n for n, _, _ in steps[-1]]
steps = [s for s in steps if [n for n, _, _ in s] == shape][-steps_wanted:]
print(f"{len(steps)} steps of {len(shape)} dispatches\n\n| dispatch | mean duration (us) |\n|---|---|")
for i, n in enumerate(shape):
    print(f"| `{n}` | {sum(s[i][2] - s[i][1] for s in steps) / len(steps) / 1e3:.1f} |")
filt = [i for i, n in enumerate(shape) if ("sieve_q16_kernel" in n or "sieve_i8_kernel" in n) and "true" not in n]
if len(filt) >= 2:
    a, b = filt[0], filt[-1]
    print(f"\nfirst filter launch start -> last filter launch end: {sum(s[b][2] - s[a][1] for s in steps) / len(steps) / 1e3:.1f} us")
print(f"step (first dispatch start -> last dispatch end): {sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3:.1f} us")
print(f"step to step (start to next start): {(steps[-1][0][1] - steps[0][0][1]) / (len(steps) - 1) / 1e3:.1f} us")
