#!/usr/bin/env python3
"""Regenerates the "numbers of the committed files" block of profiles/<tag>/README.md from the files themselves:
    python3 scripts/profiles_table.py r03
(bench_*.json: the line's own fields; kernel_stats_*.csv: mean / min duration of this package's kernels).  The prose around the
block quotes ranges over the round's calls; this block is the exact content of what is committed."""
import csv
import glob
import json
import sys
from pathlib import Path

args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args_[0] if args_ else "r03"
root = Path(__file__).resolve().parent.parent / "profiles" / tag
out = ["<!-- generated:begin (scripts/profiles_table.py) -->", "",
       "| bench line | value | ms / step | dominant kernel | mean kernel ms | frac of peak | traffic MB (PMC) | p50 ms (API) | CPU oracle |",
       "|---|---|---|---|---|---|---|---|---|"]
for f in sorted(glob.glob(str(root / "bench_*.json"))):
    d = json.loads(Path(f).read_text().strip().splitlines()[-1])
    r = d.get("roofline", {})
    cpu = d.get("cpu_baseline") or {}
    frac = f"{r.get('frac', 0) * 100:.1f} % HBM" if r else ""
    if "mfma" in r:
        frac += f" / {r['mfma']['frac'] * 100:.1f} % MFMA"
    traffic = f"{r['traffic'] / 1e6:.1f}" if r.get("traffic") else "—"
    unit = "q/s" if d["unit"].startswith("queries") else d["unit"]
    extra = d.get("opt_in_bf16_shadow")
    name = Path(f).name
    out.append(f"| `{name}` | {d['value']:,.0f} {unit} | {d['ms_per_step']:.4f} | `{r.get('kernel', '')}` | {r.get('mean_kernel_ms', 0):.4f} "
               f"({r.get('launches_timed', 0)} launches) | {frac} | {traffic} | {d.get('p50_latency_ms') or '—'} | "
               f"{(str(cpu.get('value')) + ' ' + cpu.get('unit', '') + ' @ ' + str(cpu.get('cores')) + ' threads') if cpu else '—'} |")
    if extra and "queries_per_s" in extra:
        out.append(f"| ↳ `opt_in_bf16_shadow` (outside `value`) | {extra['queries_per_s']:,.0f} q/s | {extra['ms_per_step']:.4f} | | | | | | "
                   f"bit-equal to the fp32 scan: {extra['answers_bit_equal_to_the_fp32_scan']} ({extra['queries_compared']} queries) |")
out += ["", "| kernel trace | kernel | launches | mean µs | min µs |", "|---|---|---|---|---|"]
for f in sorted(glob.glob(str(root / "kernel_stats_*.csv"))):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        nm = r["Name"]
        if "dewi::" not in nm or any(s in nm for s in ("normalize_rows", "payload_soa", "f32_to_bf16")):
            continue
        short = nm.split("(")[0].replace("void ", "").replace("dewi::", "").replace("fastfit::", "")
        out.append(f"| `{Path(f).name}` | `{short}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} |")
out += ["", "<!-- generated:end -->"]
readme = root / "README.md"
text = readme.read_text()
if "--check" in sys.argv:      # exit 1 when the README's block is not what the committed files say (tests/test_cpu_host_logic.py)
    have = text[text.index("<!-- generated:begin"): text.index("<!-- generated:end -->") + len("<!-- generated:end -->")] \
        if "<!-- generated:begin" in text else ""
    sys.exit(0 if have == "\n".join(out) else 1)
b, e = "<!-- generated:begin", "<!-- generated:end -->"
block = "\n".join(out)
if b in text:
    text = text[: text.index(b)] + block + text[text.index(e) + len(e):]
else:
    text = text.rstrip("\n") + "\n\n## Numbers of the committed files (generated)\n\n" + block + "\n"
readme.write_text(text)
print(block)
