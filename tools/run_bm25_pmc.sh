#!/bin/bash
# the BM25 fast passes at C3 per kernel: two rocprofv3 --pmc passes + a plain kernel trace over tools/bm25_pmc_driver.py
# (4096-query batches of the SURVEY 8(d) mix) -> a table per kernel on stdout (profiles/r04_bm25_pmc.md)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bm25p
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/bm25p/a -o p -- python3 $R/tools/bm25_pmc_driver.py > /tmp/bm25p_a.log 2>&1 || { tail -3 /tmp/bm25p_a.log; exit 1; }
timeout -k 5 300 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d /tmp/bm25p/b -o p -- python3 $R/tools/bm25_pmc_driver.py > /tmp/bm25p_b.log 2>&1 || { tail -3 /tmp/bm25p_b.log; exit 1; }
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bm25p/t -o p -- python3 $R/tools/bm25_pmc_driver.py > /tmp/bm25p_t.log 2>&1 || { tail -3 /tmp/bm25p_t.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"/tmp/bm25p/{d}/**/p_counter_collection.csv", recursive=True):
        per = collections.defaultdict(float); name = {}
        for r in csv.DictReader(open(f)):
            if "bm25_" in r["Kernel_Name"]:
                per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        for (did, c), v in per.items(): agg[name[did]][c].append(v)
    return agg
a, b = counters("a"), counters("b")
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/bm25p/t/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bm25_" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
def big(v):  # the 4096-query launches = the larger half
    v = sorted(v, reverse=True); v = v[: max(1, len(v) // 2)]; return sum(v) / len(v)
print("| kernel | launches | duration (trace) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | VALU instructions | LDS instructions | LDS array busy | bank-conflict share | FETCH_SIZE |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for k in sorted(dur):
    A, B = a.get(k, {}), b.get(k, {})
    g = lambda d, c: big(d[c]) if c in d and d[c] else 0.0
    wc = max(g(A, "SQ_WAVE_CYCLES"), 1.0); ga = max(g(B, "GRBM_GUI_ACTIVE"), 1.0)
    d_us = big(dur[k]) / 1e3; tot += d_us * (1 if len(dur[k]) else 0)
    idx = g(A, "SQ_LDS_IDX_ACTIVE")
    print(f"| `{k}` | {len(dur[k])} | {d_us:.0f} us | {100*g(A,'SQ_WAIT_ANY')/wc:.0f} % | {100*g(A,'SQ_ACTIVE_INST_ANY')/wc:.0f} % | {g(A,'SQ_INSTS_VALU')/1e6:.1f} M | {g(A,'SQ_INSTS_LDS')/1e6:.1f} M | "
          f"{100*idx/(256*ga/8):.0f} % | {100*g(A,'SQ_LDS_BANK_CONFLICT')/max(idx,1):.0f} % | {g(B,'FETCH_SIZE')/1024:.0f} MiB |")
print(f"\nsum of the per-kernel durations above: {tot:.0f} us")
PY
