"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (and grid size), average of each counter."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("alvq::", "").split("(")[0].replace("void ", "")[:34]
        key = (name, r["Grid_Size"])
        a = agg[key][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for key in sorted(agg):
    if not any(s in key[0] for s in ("conv1d", "wgrad")) or "reduce" in key[0]:
        continue
    c = {k: v[1] / v[0] for k, v in agg[key].items()}
    print("%-36s grid %-8s" % key)
    print("   " + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items())))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # MFMA_BUSY sums the matrix-pipe busy cycles of all 1024 SIMDs (= 16 per 16x16x32 bf16 MFMA);
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles, so /8 is the kernel's length in shader cycles
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print("   kernel length %.0f cycles; matrix-pipe utilisation = MFMA_BUSY / (cycles x 1024 SIMDs) = %.3f"
              % (cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)))
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if k in c:
                print("   %s / WAVE_CYCLES = %.3f" % (k, c[k] / c["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        print("   lds bank conflict share = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
