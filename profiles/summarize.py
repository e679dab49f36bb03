"""Condense rocprofv3 output (gpurun_out/prof_*) into the small summaries committed under profiles/.

    tools/profile_bench.sh bf16            # on the GPU box: writes gpurun_out/prof_bf16/{stats,fetch,write}
    python profiles/summarize.py <tag> <stats_dir> <fetch_dir> <write_dir>

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, alvq kernels + top others),
profiles/<tag>_pmc.json (per-kernel average FETCH_SIZE / WRITE_SIZE per launch) and refreshes profiles/traffic.json
(HBM bytes per launch per kernel family, = (2*FETCH_SIZE + WRITE_SIZE) * 1024 following MI355X_MICROARCH.md "HBM":
FETCH_SIZE is in KiB and reports half of the bytes of a wide coalesced read on gfx950).
"""
import collections
import csv
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def family(name):
    m = re.search(r"alvq::(\w+)", name)
    return m.group(1) if m else None


def instantiation(name):
    """The name bench.py's KernelTimer gives the launch: function + template arguments as rocprofv3 prints them; the split-format
    convolutions keep their first two (OUT, KW) -- the tile variant behind them is the library's choice, not the caller's."""
    m = re.search(r"alvq::(\w+)<([^>]*)>", name)
    if not m:
        return family(name)
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) in ("conv1d_f16mx_kernel", "conv1d_bf16x3_kernel"):
        return "%s<%s, ...>" % (m.group(1), ", ".join(args[:2]))
    return "%s<%s>" % (m.group(1), ", ".join(args))


def pmc(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def _find(d, suffix):
    """the CSV rocprofv3 wrote under d (its -o prefix changes per round)"""
    hits = [f for f in sorted(os.listdir(d)) if f.endswith(suffix)]
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return os.path.join(d, hits[-1])


def main(tag, stats_dir, fetch_dir, write_dir):
    rows = list(csv.DictReader(open(_find(stats_dir, "_kernel_stats.csv"))))
    with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:25]:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    fetch = pmc(_find(fetch_dir, "_counter_collection.csv"))
    write = pmc(_find(write_dir, "_counter_collection.csv"))
    per_kernel, fam = {}, collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    inst = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for k in sorted(set(fetch) | set(write)):
        f, wr = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        if family(k) is None:
            continue
        per_kernel[k[:120]] = {"launches": f[0], "FETCH_SIZE_KiB_avg": f[1] / max(f[0], 1),
                               "WRITE_SIZE_KiB_avg": wr[1] / max(wr[0], 1)}
        a = fam[family(k)]
        a[0] += f[0]; a[1] += f[1]; a[2] += wr[0]; a[3] += wr[1]
        if instantiation(k) != family(k):
            a = inst[instantiation(k)]
            a[0] += f[0]; a[1] += f[1]; a[2] += wr[0]; a[3] += wr[1]
    json.dump(per_kernel, open(os.path.join(HERE, tag + "_pmc.json"), "w"), indent=1)
    traffic = {k: (2.0 * v[1] / max(v[0], 1) + v[3] / max(v[2], 1)) * 1024.0 for k, v in list(fam.items()) + list(inst.items())}
    tpath = os.path.join(HERE, "traffic.json")                      # merged: each dtype contributes its own families
    merged = json.load(open(tpath)) if os.path.exists(tpath) else {}
    merged.update(traffic)
    json.dump(merged, open(tpath, "w"), indent=1, sort_keys=True)
    durs = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        f = family(r["Name"])
        if f:
            durs[f][0] += int(r["Calls"]); durs[f][1] += float(r["TotalDurationNs"])
    for f, (n, t) in sorted(durs.items(), key=lambda kv: -kv[1][1]):
        print("%-28s launches %5d  avg %.4f ms  traffic/launch %.1f MB" % (f, n, t / n / 1e6, traffic.get(f, 0) / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
