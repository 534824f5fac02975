"""ic_gate_select alone (rank by counting + greedy suppression walk) on a hex map: microseconds per launch for a few
(R, k_total), HIP events around 200 launches.  k_total = 2 leaves only the ranking (the walk stops at once)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import hip
dev = torch.device("cuda:0")
L, st = hip.lib(), hip.stream_handle()
for R in (64, 512, 1024):
    nb = ic.hex_neighbors(R)
    off = torch.tensor([0] + [sum(len(n) for n in nb[:i + 1]) for i in range(R)], dtype=torch.int32, device=dev)
    nbr = torch.tensor([j for n in nb for j in n], dtype=torch.int32, device=dev)
    S = 2
    scores = torch.rand(S, R, device=dev)
    mask = torch.zeros(S, R, dtype=torch.uint8, device=dev)
    for kt in (2, 34, 98, 2000):
        ga = hip.new("ic_gate_args", S=S, R=R, d=256, sensor_idx=0, motor_idx=R - 1, k_total=kt, force_io=1, w1=1.0, w2=0.3, w3=0.5,
                     w4=0.2, nbr_off=off, nbr=nbr, scores=scores, mask=mask)
        for _ in range(20):
            hip.check(L.ic_gate_select(ctypes.byref(ga), st), "sel")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            L.ic_gate_select(ctypes.byref(ga), st)
        e1.record(); torch.cuda.synchronize()
        print(f"R={R:5d} k_total={kt:3d}: {1e3 * e0.elapsed_time(e1) / 200:7.2f} us per launch, selected {int(mask[0].sum())}")
