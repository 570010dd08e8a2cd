#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<name>.md (+ traffic.json).

    python tools/summarize_prof.py gpurun_out/prof_r01_noblank profiles/r01_noblank_cfg2 noblank_B256 <x_bytes>

Traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come from separate --pmc passes, unit KiB; WRITE_SIZE is exact for streaming
stores; on gfx950 FETCH_SIZE = (number of fabric read requests) x 64 B, i.e. it tallies a
128-byte request at 64 B.  The read side is therefore taken from the request-size counters
of a third pass (32B x n32 + 64B x n64 + 128B x n128, exact); FETCH_SIZE x 2 (the guide's
correction for wide streaming reads) is printed beside it.
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
    raw = per_kernel(os.path.join(src, "pmc_rdreq"), want)
    lines += ["", "## HBM traffic per launch (PMC, separate passes)", ""]
    out = {}
    for k in fetch:
        fs = fetch[k]["FETCH_SIZE"] * 1024.0
        ws = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        lines += ["kernel `%s`" % k[:90], "",
                  "* FETCH_SIZE raw = %.0f B (x2 = %.0f B)" % (fs, 2 * fs),
                  "* WRITE_SIZE = %.0f B (exact for streaming stores)" % ws]
        rd = 2 * fs
        if k in raw:
            r = raw[k]
            n32, n64, n128 = (r.get("TCC_EA0_RDREQ_%s_sum" % w, 0) for w in ("32B", "64B", "128B"))
            rd = 32 * n32 + 64 * n64 + 128 * n128
            lines.append("* read requests: 32B x %.0f + 64B x %.0f + 128B x %.0f = %.0f B  "
                         "(requests x 64 B = %.0f B = FETCH_SIZE)" % (n32, n64, n128, rd, 64 * (n32 + n64 + n128)))
        lines.append("* **traffic = reads %.0f B + writes %.0f B = %.0f B per launch** (logits once = %.0f B)"
                     % (rd, ws, rd + ws, x_bytes))
        out["hbm_bytes_per_launch"] = out.get("hbm_bytes_per_launch", 0) + round(rd + ws)   # summed over the
        out["read_bytes"] = out.get("read_bytes", 0) + round(rd)                              # variant's kernels
        out["write_bytes"] = out.get("write_bytes", 0) + round(ws)
        out["fetch_size_raw_bytes"] = out.get("fetch_size_raw_bytes", 0) + round(fs)
        out.setdefault("kernels", []).append(k[:120])
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
