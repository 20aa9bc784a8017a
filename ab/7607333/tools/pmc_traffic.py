"""Developer tool: mean HBM bytes per launch of one kernel symbol from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
of the bench child.  usage: pmc_traffic.py <fetch_dir> <write_dir> <symbol substring> <out.json>
gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KB; FETCH_SIZE tallies 128-B requests at 64 B -> x2."""
import csv, glob, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def sha16(rel):
    return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
# the source file of each kernel whose traffic has been measured: bench.py drops the figure when the source changed
KERNEL_SOURCE = {"coupling_po_fwd_kernel": "recurrent-flows-msc_amd/csrc/coupling_po.hip",
                 "coupling_po_bwd_kernel": "recurrent-flows-msc_amd/csrc/coupling_po.hip",
                 "gemm_wgrad_b3_kernel": "recurrent-flows-msc_amd/csrc/wgrad_bf16x3.hip",
                 "gemm_wgrad_dma_kernel": "recurrent-flows-msc_amd/csrc/wgrad_bf16x3.hip",
                 "conv1x1_ws_kernel": "recurrent-flows-msc_amd/csrc/conv_bf16x3.hip"}
def matches(sym, name):
    """the bench labels the two instantiation families of coupling_po_fwd_kernel<..., BWD> as two kernels"""
    if sym == "coupling_po_fwd_kernel":
        return "coupling_po_fwd_kernel" in name and "false>" in name
    if sym == "coupling_po_bwd_kernel":
        return "coupling_po_fwd_kernel" in name and "true>" in name
    return sym in name
def mean_kb(d, counter, sym):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and matches(sym, r["Kernel_Name"])]
    return sum(vals) / len(vals), len(vals)
fetch, n1 = mean_kb(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write, n2 = mean_kb(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "dispatches": [n1, n2], "fetch_bytes_per_launch": 2.0 * fetch * 1024,
       "write_bytes_per_launch": write * 1024, "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024,
       "source": KERNEL_SOURCE.get(sys.argv[3]), "source_sha16": sha16(KERNEL_SOURCE[sys.argv[3]]) if sys.argv[3] in KERNEL_SOURCE else None,
       "corrections": "KB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B)",
       "command": "rocprofv3 --pmc <counter> -- python3 bench.py --child --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --secondary --no-parity"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
# per-kernel aggregate of both passes next to the summary (the raw counter_collection.csv files are hundreds of MB)
def by_kernel(d, counter):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg
fa, wa = by_kernel(sys.argv[1], "FETCH_SIZE"), by_kernel(sys.argv[2], "WRITE_SIZE")
with open(sys.argv[4].replace(".json", "_by_kernel.csv"), "w") as f:
    f.write("kernel,dispatches,mean_FETCH_SIZE_KB_raw,mean_WRITE_SIZE_KB_raw,mean_HBM_bytes_corrected\n")
    rows = []
    for k in set(fa) | set(wa):
        n = max(fa.get(k, [0, 0])[0], wa.get(k, [0, 0])[0])
        mf = fa[k][1] / fa[k][0] if k in fa else 0.0
        mw = wa[k][1] / wa[k][0] if k in wa else 0.0
        rows.append((n * (2 * mf + mw), k, n, mf, mw))
    for tot, k, n, mf, mw in sorted(rows, reverse=True)[:80]:
        f.write('"%s",%d,%.1f,%.1f,%.0f\n' % (k[:120].replace('"', "'"), n, mf, mw, (2 * mf + mw) * 1024))
print(json.dumps(out))
