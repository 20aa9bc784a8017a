#!/usr/bin/env python3
"""Developer tool: summarise the LAST training step of a rocprofv3 kernel trace (csv) by kernel family."""
import collections, csv, glob, os, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
clusters = []
for i in adam:
    ts = int(rows[i]["Start_Timestamp"])
    if not clusters or ts - clusters[-1][-1][1] > 5e6:
        clusters.append([])
    clusters[-1].append((i, ts))
s0, s1 = clusters[-2][-1][0] + 1, clusters[-1][-1][0] + 1
seg = rows[s0:s1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
print("step wall ms %.1f  kernels %d  sum kernel ms %.1f" % ((t1 - t0) / 1e6, len(seg), sum(
    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
def cat(n):
    if os.environ.get("TRACE_NO_MERGE") == "1":  # every kernel symbol on its own line (no MINE:shell bucket)
        return n[:90]
    if "conv_mfma" in n or "wgrad_mfma" in n:
        return "MINE:mfma"
    if any(k in n for k in ("_kernel(float", "pack_weight", "wgrad_finish", "squeeze2d", "gauss", "affine_coupling",
                            "convlstm", "channel_stats", "actnorm_invconv", "conv_epilogue")):
        return "MINE:shell"
    return n[:90]
by_grid = len(sys.argv) > 3 and sys.argv[3] == "grid"  # keep launches of different grid sizes (= shapes) apart
by_template = len(sys.argv) > 3 and sys.argv[3] == "template"  # the instantiations of one kernel template on one line
if by_template:
    import re
    _cat = cat
    def cat(n):  # noqa: F811
        k = _cat(n)
        return k if k.startswith("MINE:") else re.sub(r"<.*", "", re.sub(r"^void ", "", k)).strip()[:90]
for r in seg:
    key = cat(r["Kernel_Name"])
    if by_grid and "Grid_Size_X" in r:
        key = "%s [grid %s,%s wg %s]" % (key[:70], r["Grid_Size_X"], r.get("Grid_Size_Y", "1"), r.get("Workgroup_Size_X", "?"))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-92s %6d %9.3f ms" % (k, v[0], v[1]))
