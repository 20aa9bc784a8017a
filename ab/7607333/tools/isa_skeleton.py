"""Developer tool: control-flow / memory skeleton of one kernel in a hipcc -save-temps .s file.
usage: isa_skeleton.py file.s mangled_kernel_name_prefix"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r"^" + re.escape(sys.argv[2]) + r"[^\n]*:\s*(;[^\n]*)?\n", s, re.M)
i = m.end()
j = s.index("s_endpgm", i)
body = s[i:j].split("\n")
out = []
for n, l in enumerate(body):
    t = l.strip()
    if re.match(r"^\.LBB", t) or "s_cbranch" in t or "s_branch" in t or "s_barrier" in t or "vmcnt" in t:
        out.append((n, t[:80]))
    elif t.startswith(("global_load", "buffer_load")):
        out.append((n, "  LOAD " + t.split(" ")[0]))
    elif t.startswith(("global_store", "global_atomic")):
        out.append((n, "  STORE/ATOMIC"))
    elif t.startswith("v_mfma"):
        out.append((n, "  MFMA"))
    elif t.startswith("ds_write") or t.startswith("ds_store"):
        out.append((n, "  DS_WRITE"))
    elif t.startswith("scratch_"):
        out.append((n, "  SCRATCH " + t.split(" ")[0]))
prev, cnt, pn = None, 0, 0
for n, t in out:
    if t == prev:
        cnt += 1
        continue
    if prev is not None:
        print(pn, prev, ("x%d" % (cnt + 1)) if cnt else "")
    prev, cnt, pn = t, 0, n
print(pn, prev, ("x%d" % (cnt + 1)) if cnt else "")
