"""Token mixer timing: sensor_vectors_batch forward + backward for N = 16 sequences at
T in {256, 1024}, both attention kinds (IronRoPE + anchors: default config; AFA: 10m config).  HIP events."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
import ironcortex_amd as ic  # noqa: E402


def main():
    dev = "cuda:0"
    for name in ("default", "10m"):
        for T in (256, 1024):
            cfg = bench.make_config(name) if hasattr(bench, "make_config") else None
            if cfg is None:
                raise SystemExit("bench.make_config missing")
            cfg.max_T = max(cfg.max_T, 2 * T)
            torch.manual_seed(0)
            model = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
            N = 16
            tokens = torch.randint(0, 255, (N, T), device=dev)
            focus = torch.rand(N, T, device=dev) < 0.15
            focus[: N // 2] = False
            probe = torch.randn(N, cfg.d, device=dev)

            def step():
                vec, en, ent = model.sensor_vectors_batch(tokens, focus)
                (vec * probe).sum().backward()

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 10
            e0.record()
            for _ in range(it):
                step()
            e1.record()
            torch.cuda.synchronize()
            row = {"config": name, "T": T, "N": N, "fwd_bwd_ms": e0.elapsed_time(e1) / it,
                   "peak_mem_MB": torch.cuda.max_memory_allocated() / 2**20}
            # the attention kernels alone (HIP events around each launch; a separate, untimed pass)
            from ironcortex_amd import hip
            hip.prof_enable(True)
            hip.prof_read()
            for _ in range(it):
                step()
            pr = hip.prof_read()
            hip.prof_enable(False)
            Ta = T + (0 if cfg.enable_afa_attention else 2 * sum(len(range(S // 2, T, S)) for S in (128, 32)) // 2 + 8)
            hd = cfg.d // 4
            dqk = hd if cfg.enable_afa_attention else hd + 64
            flop = {"attn_fwd": 2 * (2 * dqk + hd), "attn_bwd_kv": 2 * (2 * dqk + 2 * hd), "attn_bwd_q": 2 * (2 * dqk + hd)}
            for kname, (n, ms) in pr.items():
                if kname.startswith("attn"):
                    us = 1e3 * ms / n
                    row[kname + "_us"] = round(us, 1)
                    row[kname + "_TFLOPs"] = round(flop[kname] * N * 4 * Ta * Ta / (us * 1e-6) / 1e12, 1)
            print(json.dumps(row), flush=True)
            del model
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
