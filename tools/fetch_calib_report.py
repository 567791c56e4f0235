"""Per-kernel averages of the counter passes of tools/fetch_calib.sh against the probe's known bytes."""
import collections
import csv
import glob
import json

known = json.load(open("gpurun_out/fetch_calib_known.json"))


def collect(pattern, keep):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if keep(k):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # skip each kernel's first launch (cold caches / first touch)
    return {k: {c: sum(v[1:]) / max(1, len(v[1:])) for c, v in d.items()} for k, d in agg.items()}


out = {"probe": {}, "bench_lj": {}}
probe = collect("/tmp/fc_*/**/*counter_collection.csv", lambda k: k.startswith("calib_"))
for k, c in sorted(probe.items()):
    kb = known.get(k, {})
    row = dict(known=kb, counters=c)
    if "FETCH_SIZE" in c and "read_bytes" in kb:
        row["FETCH_SIZE_bytes"] = c["FETCH_SIZE"] * 1024.0
        row["known_over_FETCH_SIZE"] = kb["read_bytes"] / (c["FETCH_SIZE"] * 1024.0)
        for alt in ("read_bytes_128B_lines", "read_bytes_64B_lines"):
            if alt in kb:
                row[alt + "_over_FETCH_SIZE"] = kb[alt] / (c["FETCH_SIZE"] * 1024.0)
    if "WRITE_SIZE" in c and "write_bytes" in kb:
        row["WRITE_SIZE_bytes"] = c["WRITE_SIZE"] * 1024.0
        row["known_over_WRITE_SIZE"] = kb["write_bytes"] / (c["WRITE_SIZE"] * 1024.0)
    if "TCC_EA0_RDREQ_sum" in c and "read_bytes" in kb:
        row["read_bytes_per_EA_RDREQ"] = kb["read_bytes"] / c["TCC_EA0_RDREQ_sum"]
    if "TCC_EA0_WRREQ_sum" in c and "write_bytes" in kb:
        row["write_bytes_per_EA_WRREQ"] = kb["write_bytes"] / c["TCC_EA0_WRREQ_sum"]
    out["probe"][k] = row
out["bench_lj"] = collect("/tmp/fb_*/**/*counter_collection.csv", lambda k: "fused_forces" in k or "nve_step" in k)
out["_note"] = ("tools/fetch_calib.sh: rocprofv3 --pmc, one counter set per pass; averages over launches 2.. of each kernel. "
                "bench_lj: bench.py --no-cpu-baseline --no-mlp --no-fused --steps 20 --warmup 5 --equil 60 --settle 0 --windows 1")
print(json.dumps(out, indent=1, sort_keys=True))
