"""developer probe: the backward / forward shell launches between two Glow steps at the five level shapes of the
canonical flow (N = 608 frames), timed as hipGraph replays of 20 launches (kernel time, no Python).
python tools/bench_shell.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
for p in (ROOT, os.path.join(ROOT, "recurrent-flows-msc_amd")):
    sys.path.insert(0, p)
import ctypes
import torch
from rfn_hip import lib as L

_i, _l = ctypes.c_int, ctypes.c_long


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(N, C, S):
    dev = "cuda"
    HW = S * S
    r = lambda *s: torch.randn(*s, device=dev)
    x, gz, o = r(N, C, S, S), r(N, C, S, S), r(N, C, S, S) * 0.3
    bias, logs, Wm = r(C) * 0.1, r(C) * 0.1, r(C, C) * 0.3
    gdl, l3 = r(N), r(C) * 0.1
    gW, gab, gal, gb3, gl3 = (torch.zeros(C, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                              torch.zeros(C, device=dev), torch.zeros(C, device=dev))
    gzp, gpre = torch.empty_like(x), torch.empty_like(x)

    def bwd():
        L.call("rfn_glow_shell_bwd_f32", L.dev(x), _l(C * HW), L.dev(bias), L.dev(logs), L.dev(Wm), L.dev(gz), _l(C * HW),
               L.dev(gW), L.dev(gab), L.dev(gal), L.dev(o), _l(C * HW), L.dev(gdl), None, None, L.dev(l3), L.dev(gzp),
               _l(C * HW), L.dev(gpre), _l(C * HW), None, None, L.dev(gb3), L.dev(gl3), _i(1), _i(1), _i(N), _i(C), _i(HW))

    ldf = int(L.load().rfn_glow_shell_fwd_ld_floats(N, C, S, S))
    ldp = torch.empty(ldf, device=dev)
    z, zn = r(N, C, S, S), torch.empty(N, C, S, S, device=dev)

    def fwd():
        L.call("rfn_glow_shell_fwd_f32", L.dev(z), _l(C * HW), None, L.dev(o), _l(C * HW), None, None, None, None, None,
               L.dev(ldp), _i(1), L.dev(bias), L.dev(logs), L.dev(Wm), L.dev(zn), _l(C * HW), _i(1), _i(N), _i(C), _i(S), _i(S))

    tb, tf = timed(bwd), timed(fwd)
    by = 4.0 * N * C * HW
    print("N%d C%d %dx%d  shell bwd %.1f us (%.0f GB/s of 5.5 tensors)   shell fwd %.1f us (%.0f GB/s of 3.5 tensors)" % (
        N, C, S, S, tb, 5.5 * by / tb / 1e3, tf, 3.5 * by / tf / 1e3), flush=True)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 608
    for C, S in ((4, 32), (8, 16), (16, 8), (32, 4), (64, 2)):
        run(N, C, S)
