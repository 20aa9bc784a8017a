"""developer probe: registers, spills and scratch of every kernel of one HIP source (hipcc -Rpass-analysis).
python tools/kernel_resources.py recurrent-flows-msc_amd/csrc/coupling_po.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for l in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +(.*?): (\S+) \[", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
keys = ("VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]")
for k, v in rows.items():
    if flt in k:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print("%-60s %s" % (name[:60], "  ".join("%s=%s" % (a.split(" [")[0], v.get(a, "?")) for a in keys)))
