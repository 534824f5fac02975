"""Host-visible cost of 'launch one tiny kernel, wait for it' under the HIP wait policies."""
import ctypes, sys, time, torch
flag = int(sys.argv[1]) if len(sys.argv) > 1 else -1
hip = ctypes.CDLL("libamdhip64.so")
if flag >= 0:
    print("hipSetDeviceFlags(%d) before init ->" % flag, hip.hipSetDeviceFlags(ctypes.c_uint(flag)))
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
st = torch.cuda.current_stream()
for name, fn in (("add_ + stream.synchronize", lambda: (x.add_(1), st.synchronize())),
                 ("add_ + tolist (D2H copy)", lambda: (x.add_(1), x.tolist()))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000):
        fn()
    dt = (time.perf_counter() - t0) / 3000
    print(f"flag {flag:2d} {name:34s} {dt*1e6:7.2f} us per round trip")
