"""profiles/traffic.json for the sieve: HBM bytes of one shard pass from two rocprofv3 PMC passes over tools/sieve_stats.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -o t -- python3 tools/sieve_stats.py 10000000 128
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -o t -- python3 tools/sieve_stats.py 10000000 128
    python tools/sieve_traffic.py OUT/fetch/t_counter_collection.csv OUT/write/t_counter_collection.csv 10000000 384 [queries] > profiles/traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads
(LDS-DMA included) -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.  The correction is calibrated for the
streaming filter launches; the scatter / select kernels between them (candidate records, list entries and the ~20 rows per query that are evaluated: a few MB) are listed
separately with the same factor, which over-states them at worst."""
import csv, json, sys
from collections import defaultdict

def per_kernel(path, counter):
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) if "Grid_Size" in r else 0, r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    out = defaultdict(list)
    for (name, grid, disp), v in by.items():
        out[name].append((int(disp), sum(v)))
    return out

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
n, d = int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 128  # queries per launch (what sieve_stats.py was run with)
def split_filter(by):  # the filter kernel's dispatches alternate first launch / second launch (the sample instance has another name)
    name = [k for k in by if ("sieve_q16_kernel" in k or "sieve_i8_kernel" in k) and "true" not in k.split("<")[1]][0]
    v = [x for _, x in sorted(by[name])][-32:]  # the timed steps
    return name, sum(v[0::2]) / len(v[0::2]), sum(v[1::2]) / len(v[1::2])
def mean_of(by, key):
    k = [x for x in by if key in x]
    v = [x for _, x in sorted(by[k[0]])][-32:] if k else [0.0]
    return sum(v) / len(v)
name, f1, f2 = split_filter(fetch)
_, w1, w2 = split_filter(write)
fv, fs = mean_of(fetch, "sieve_scatter"), mean_of(fetch, "sieve_select_kernel")
wv, ws = mean_of(write, "sieve_scatter"), mean_of(write, "sieve_select_kernel")
int8 = "sieve_i8_kernel" in name
reads = 2 * (f1 + f2) * 1024
between = 2 * (fv + fs) * 1024 + (wv + ws) * 1024  # one scatter + select pair sits inside the bracket (the mean is over both pairs)
hbm = reads + (w1 + w2) * 1024 + between
streamed = n * ((d + 127) // 128 * 128) * (1 if int8 else 2) + 4 * n + (n // 32 * 16 if int8 else 0) + B * d * 4 + B * 10 * 12
print(json.dumps({
    "sieve": True,
    "first_stage": "int8" if int8 else "bf16",
    "kernel": name + ": the two filter launches that stream one shard's " + ("int8 image" if int8 else "bf16 hi blocks") + " (first 1/16 of the tiles, then the rest), plus the scatter / select pair between them - what bench.py's HIP events bracket; the 32K-row sample launch before them is outside the bracket",
    "rows_per_launch": n, "dim": d, "queries_per_launch": B,
    "FETCH_SIZE_KiB_mean": {"first_launch": f1, "second_launch": f2, "scatter_kernel": fv, "select_kernel": fs},
    "WRITE_SIZE_KiB_mean": {"first_launch": w1, "second_launch": w2, "scatter_kernel": wv, "select_kernel": ws},
    "correction": "gfx950: FETCH_SIZE counts 1/2 of 16-B-per-lane streaming reads, LDS-DMA included (MI355X_MICROARCH.md, HBM) -> read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE exact",
    "hbm_bytes_per_launch": hbm,
    "streamed_bytes_per_launch_by_construction": streamed,
    "ratio": hbm / streamed,
    "survey_algorithmic_bytes_per_launch": n * d * 4 + 4 * n + B * d * 4 + B * 10 * 12,
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/sieve_stats.py %d %d (WRITE_SIZE in its own pass); tools/sieve_traffic.py" % (n, B),
}, indent=1))
