"""Developer tool: matrix-pipe utilisation per kernel symbol from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES,
GRBM_GUI_ACTIVE) of the bench child.  usage: pmc_mfma.py <pmc_dir> <out.csv>
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): the busy counter adds the cycles every
SIMD's matrix pipe is occupied (32 per v_mfma_f32_32x32x16_{bf16,f16}, MI355X_MICROARCH.md), GRBM_GUI_ACTIVE is the
dispatch's duration in cycles summed over the 8 XCDs."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
agg = {}
for r in csv.DictReader(open(f)):
    a = agg.setdefault(r["Kernel_Name"], {"n": {}, "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0})
    c = r["Counter_Name"]
    if c in a:
        a[c] += float(r["Counter_Value"])
        a["n"][c] = a["n"].get(c, 0) + 1
rows = []
for k, a in agg.items():
    n = max(a["n"].values()) if a["n"] else 0
    if n and a["GRBM_GUI_ACTIVE"] > 0:
        rows.append((a["GRBM_GUI_ACTIVE"], k, n, a["SQ_VALU_MFMA_BUSY_CYCLES"] / n, a["GRBM_GUI_ACTIVE"] / n,
                     a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)))
with open(sys.argv[2], "w") as out:
    out.write("kernel,dispatches,mean_SQ_VALU_MFMA_BUSY_CYCLES,mean_GRBM_GUI_ACTIVE,mfma_pipe_utilisation\n")
    for _, k, n, busy, act, util in sorted(rows, reverse=True)[:60]:
        out.write('"%s",%d,%.0f,%.0f,%.3f\n' % (k[:120].replace('"', "'"), n, busy, act, util))
for _, k, n, busy, act, util in sorted(rows, reverse=True)[:14]:
    print("%-70s %4d  util %.3f" % (k[:70], n, util))
