#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel (mean per launch) and record the HBM traffic
of the bench kernels for bench.py (`roofline.traffic`).

    python3 scripts/summarize_pmc.py <out_dir> [--record KEY=KERNEL_SUBSTRING ...] [--commit SHA]

<out_dir> holds one sub-directory per counter pass, named pmc_<COUNTER>[_<tag>] (separate passes, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (16 B per lane, global_load and buffer_load ... lds alike), so the read side
is doubled; WRITE_SIZE is exact for 16-byte streaming stores.  Both counters are in KiB.

--record writes profiles/hbm_traffic.json: {"sources_sha256", "commit", "records": {KEY: {"kernel",
"bytes_per_launch", "launches", "write_bytes_per_launch"}}}.  bench.py reports `traffic` only when
"sources_sha256" equals the hash of the kernel sources it runs on.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

args = sys.argv[1:]
out = args[0]
records, commit = [], "unknown"
i = 1
while i < len(args):
    if args[i] == "--record":
        i += 1
        while i < len(args) and not args[i].startswith("--"):
            records.append(args[i].split("=", 1))
            i += 1
    elif args[i] == "--commit":
        commit = args[i + 1]
        i += 2
    else:
        i += 1

counters = defaultdict(lambda: defaultdict(list))     # counter -> kernel -> values
for d in sorted(glob.glob(f"{out}/pmc_*")):
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                counters[row.get("Counter_Name")][row["Kernel_Name"].split("(")[0][:100]].append(float(row["Counter_Value"]))
names = sorted(counters)
kernels = sorted({k for c in counters.values() for k in c})
summary = {}
print("per-launch means; FETCH_SIZE / WRITE_SIZE in KiB (raw), read MB = 2 x FETCH_SIZE (gfx950 correction)")
for k in kernels:
    ent = {}
    for c in names:
        v = counters[c].get(k)
        if v:
            ent[c] = sum(v) / len(v)
            ent["launches"] = max(ent.get("launches", 0), len(v))
    if "FETCH_SIZE" in ent:
        ent["read_bytes_corrected"] = 2 * ent["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in ent:
        ent["write_bytes"] = ent["WRITE_SIZE"] * 1024
    summary[k] = ent
    cols = "  ".join(f"{c}={ent[c]:.1f}" for c in names if c in ent)
    extra = f"  read_MB={ent['read_bytes_corrected'] / 1e6:.2f}" if "read_bytes_corrected" in ent else ""
    print(f"{k:100s} n={ent.get('launches', 0):5d}  {cols}{extra}")
json.dump(summary, open(f"{out}/pmc_summary.json", "w"), indent=1)

if records:
    import bench
    rec = {"sources_sha256": bench.sources_digest(), "commit": commit, "records": {}}
    for key, sub in records:
        hits = [k for k in kernels if sub in k and "read_bytes_corrected" in summary[k]]
        if not hits:
            print(f"no FETCH_SIZE rows for a kernel matching {sub!r}", file=sys.stderr)
            continue
        k = max(hits, key=lambda x: summary[x]["read_bytes_corrected"] * summary[x]["launches"])
        name = sub
        rec["records"][key] = {"kernel": name, "kernel_full": k, "bytes_per_launch": summary[k]["read_bytes_corrected"],
                               "write_bytes_per_launch": summary[k].get("write_bytes"), "launches": summary[k]["launches"]}
    dst = os.environ.get("DEWI_TRAFFIC_OUT", f"{out}/hbm_traffic.json")
    json.dump(rec, open(dst, "w"), indent=1)
    print(f"wrote {dst}")
