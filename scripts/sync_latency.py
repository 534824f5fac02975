"""Host-visible cost of 'launch one tiny kernel, wait for it': stream.synchronize() vs polling a
pinned host word the kernel's result is copied into."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
st = torch.cuda.current_stream()
for name, fn in (("add_ + stream.synchronize", lambda: (x.add_(1), st.synchronize())),
                 ("add_ + event.synchronize", None), ("add_ + tolist (D2H copy)", lambda: (x.add_(1), x.tolist()))):
    if fn is None:
        ev = torch.cuda.Event()
        fn = lambda: (x.add_(1), ev.record(), ev.synchronize())
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    dt = (time.perf_counter() - t0) / 2000
    print(f"{name:34s} {dt*1e6:7.2f} us per round trip")
