"""Developer tool: print gpurun_out/bench_kernel_table.json rows matching a substring."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_kernel_table.json"))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
print("---- kernels")
for r in d["rows"]:
    print("%-52s %4d %8.3f ms %7.1f TF %8.1f GB/s" % (r[0][:52], r[1], r[2], r[3], r[4]))
print("---- shapes")
for r in d["by_shape"]:
    if pat in r[0]:
        print("%-72s %4d %8.3f ms %7.1f TF %8.1f GB/s %6.0f us" % (r[0][:72], r[1], r[2], r[3], r[4], r[2] / r[1] * 1e3))
