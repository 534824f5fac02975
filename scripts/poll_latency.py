"""Launch one tiny library kernel that writes into pinned host memory, then (a) stream.synchronize
or (b) poll the pinned word from the host: which wait is cheaper, and is the write visible to a
polling host before the kernel's stream is synchronized?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ironcortex_amd import hip
dev = torch.device("cuda:0")
L, st = hip.lib(), hip.stream_handle()
pin = torch.zeros(64).pin_memory()
pn = pin.numpy()
W = torch.zeros(1, 4, device=dev)
x = torch.zeros(4, device=dev)
bias = [torch.full((1,), float(v), device=dev) for v in (1.0, 2.0)]
G = hip.CLASSES["ic_gemv_desc"]
descs = []
for b in bias:
    g = (G * 1)()
    g[0].W, g[0].b, g[0].x0, g[0].y = W.data_ptr(), b.data_ptr(), x.data_ptr(), pin.data_ptr() + 63 * 4
    g[0].n0, g[0].rows, g[0].act, g[0].nbatch = 4, 1, 0, 1
    descs.append(g)
sync = torch.cuda.current_stream().synchronize
def run(poll, n):
    t = []
    for i in range(n):
        want = float(1 + i % 2)
        t0 = time.perf_counter()
        rc = L.ic_gemv_chain(descs[i % 2], 1, st)
        assert rc == 0
        if poll:
            k = 0
            while pn[63] != want:
                k += 1
                if k > 2000000:
                    raise RuntimeError("flag never arrived")
        else:
            sync()
            assert pn[63] == want
        t.append(time.perf_counter() - t0)
        if poll:
            sync()
    t.sort()
    return 1e6 * t[len(t) // 2], 1e6 * t[len(t) // 10], 1e6 * t[-len(t) // 10]
for poll in (False, True, False, True):
    run(poll, 200)
    print("poll pinned word   " if poll else "stream.synchronize ", "median %.2f us  p10 %.2f  p90 %.2f" % run(poll, 2000))
