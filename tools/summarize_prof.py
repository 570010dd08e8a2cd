#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<name>.md (+ traffic.json).

    python tools/summarize_prof.py gpurun_out/prof_r01_noblank profiles/r01_noblank_cfg2 noblank_B256 <x_bytes>

Traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come from separate --pmc passes, unit KiB; WRITE_SIZE is exact for streaming
stores; FETCH_SIZE under-counts wide reads on gfx950, so it is corrected by a factor
calibrated on THIS access pattern (the same kernel stopped after its row loads, which reads
exactly `x_bytes`), and cross-checked against the raw 32/64/128-byte request counters.
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(path, want):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def main():
    src, dst, key, x_bytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    want = sys.argv[5] if len(sys.argv) > 5 else "fused_kernel"
    lines = ["# rocprofv3 summary: %s" % key, "", "source: `%s` (tools/profile.sh on the MI355X box)" % src, ""]
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    lines += ["## kernel-trace --stats", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        lines.append("| `%s` | %s | %.2f | %.2f | %.2f | %s |" % (
            r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, r["Percentage"]))
    fetch = per_kernel(os.path.join(src, "pmc_fetch"), want)
    write = per_kernel(os.path.join(src, "pmc_write"), want)
    cal = per_kernel(os.path.join(src, "pmc_fetch_cal"), want)
    raw = per_kernel(os.path.join(src, "pmc_rdreq"), want)
    lines += ["", "## HBM traffic per launch (PMC, separate passes)", ""]
    out = {}
    for k in fetch:
        fs = fetch[k]["FETCH_SIZE"] * 1024.0
        ws = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        cs = cal.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
        factor = x_bytes / cs if cs else 2.0
        lines += ["kernel `%s`" % k[:90], "",
                  "* FETCH_SIZE raw = %.0f B; calibration run (reads x once = %.0f B) raw = %.0f B -> correction x%.3f"
                  % (fs, x_bytes, cs, factor),
                  "* WRITE_SIZE = %.0f B (exact for streaming stores)" % ws,
                  "* **traffic = FETCH_SIZE x %.3f + WRITE_SIZE = %.0f B per launch**" % (factor, fs * factor + ws)]
        if k in raw:
            r = raw[k]
            b = 32 * r.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * r.get("TCC_EA0_RDREQ_64B_sum", 0) + \
                128 * r.get("TCC_EA0_RDREQ_128B_sum", 0)
            lines.append("* cross-check, raw read requests: 32B x %.0f + 64B x %.0f + 128B x %.0f = %.0f B"
                         % (r.get("TCC_EA0_RDREQ_32B_sum", 0), r.get("TCC_EA0_RDREQ_64B_sum", 0),
                            r.get("TCC_EA0_RDREQ_128B_sum", 0), b))
            out["read_bytes_from_request_counters"] = b
        out.update({"hbm_bytes_per_launch": round(fs * factor + ws), "fetch_size_raw_bytes": round(fs),
                    "fetch_correction": round(factor, 4), "write_size_bytes": round(ws), "kernel": k[:120]})
        lines.append("")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst + ".md", "w").write("\n".join(lines) + "\n")
    tpath = os.path.join(os.path.dirname(dst), "traffic.json")
    allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
    allt[key] = out
    json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
