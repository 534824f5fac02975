"""Data-parallel train_step on real buffers: two fresh child processes share the
one GPU of the box over gloo (RCCL needs a GPU per rank; the exchange code is backend-agnostic).
Each rank trains its own shard of the batch and checks, against the other rank through the process
group, that
  * the synchronised gradients equal the mean of the two single-rank runs' gradients (same seeds, same
    shards, dp=None), with identical None-ness on both ranks (union of the single-rank sets),
  * after optimizer.step() the replicas are bit-identical (parameters, gain_ema, homeo, tau).
reference: training.py:233-248 (backward -> Hebbian nudges -> optimizer.step); SURVEY §8e for the exchange."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q, mode="strict", overlap=True, backend="gloo"):
    try:
        import faulthandler
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        trace = open(os.path.join(root, "gpurun_out", f"dp2_rank{rank}_hang.txt"), "w")
        faulthandler.dump_traceback_later(100, exit=True, file=trace)  # a stuck collective names its line and ends the rank
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        if backend == "nccl":  # RCCL: one rank per GPU, so world == 1 on this box
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import ironcortex_amd as ic
        from ironcortex_amd.parallel import DataParallel
        from tests.model_cases import load_case, seed_train
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        gold, cfg, sd, tokens, focus = load_case("allflags")
        shard = tokens[rank:rank + 1].to(dev)  # B = 1 per rank

        def build():
            m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg)
            m.load_state_dict(sd)
            m.stream_mode = mode
            return m.to(dev)

        import time
        t0 = time.time()
        mark = lambda what: (trace.write(f"{time.time() - t0:7.1f}s {what}\n"), trace.flush())
        names = [n for n, _ in build().named_parameters()]
        mark("built")
        # ---- single-rank run on this rank's shard
        m1 = build()
        seed_train()
        ic.train_step(m1, torch.optim.SGD(m1.parameters(), lr=0.0), shard, ic.LossWeights(), dev)
        g1 = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m1.named_parameters()}
        mark("single-rank step done")
        # ---- data-parallel run (early flags verified against the launches' own, OR-ed across ranks)
        from ironcortex_amd.hip import ops as _ops
        _ops.GradCoordinator.check_early = True
        dp = DataParallel(overlap=overlap)
        m2 = build()
        dp.broadcast_module(m2)
        opt = torch.optim.AdamW(m2.parameters(), lr=1e-3)
        seed_train()
        captured = {}
        step = opt.step
        opt.step = lambda *a, **k: (captured.update({n: (None if p.grad is None else p.grad.detach().clone())
                                                     for n, p in m2.named_parameters()}), step(*a, **k))[1]
        started = []
        if overlap:
            sar = dp.start_arena_reduce
            dp.start_arena_reduce = lambda mdl: (started.append(1), sar(mdl))[1]
        out = ic.train_step(m2, opt, shard, ic.LossWeights(), dev, dp=dp)
        assert captured and out["grad_norm"] > 0
        assert len(started) == (1 if overlap else 0) and dp._inflight is None
        if mode == "strict":  # device-steered loops: which matrices the pass touches is known (and exchanged) before it runs
            assert m2._coord.early_used >= 1, "the data-parallel pass waited for the device to learn its touched flags"
        mark("dp step done")
        # ---- mean of the single-rank gradients through the group: ONE packed all-reduce (+ one of the flags)
        p1 = dict(m1.named_parameters())
        have = torch.tensor([0.0 if g1[n] is None else 1.0 for n in names], device=dev)
        flat = torch.cat([(torch.zeros_like(p1[n]) if g1[n] is None else g1[n]).reshape(-1) for n in names])
        dist.all_reduce(have)
        dist.all_reduce(flat)
        have_l, off = have.tolist(), 0
        worst, n_cmp = 0.0, 0
        for n, h in zip(names, have_l):
            k = p1[n].numel()
            ref = flat[off:off + k] / world
            off += k
            got = captured[n]
            if h == 0.0:
                assert got is None, f"{n}: gradient on no single-rank run, but not None under dp"
                continue
            assert got is not None, f"{n}: None under dp although a rank has a gradient"
            scale = float(ref.abs().max())
            if scale > 1e-7:
                err = float((got.reshape(-1) - ref).abs().max()) / scale
                worst = max(worst, err)
                assert err < 2e-5, f"{n}: dp gradient differs from the mean of the single-rank gradients ({err:.2e})"
            n_cmp += 1
        mark("gradients compared")
        # ---- replicas identical after the step
        groups = {"parameters": torch.cat([p.detach().reshape(-1).double() for p in m2.parameters()]),
                  "gain_ema": m2.gate.gain_ema.double(), "homeo": m2.gate.homeo.double(),
                  "tau": torch.stack([f.tau.reshape(()) for f in m2.reg_ff]).double()}
        diverged = []
        for name, vec in groups.items():
            lo, hi = vec.clone(), vec.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                diverged.append((name, float((hi - lo).abs().max())))
        assert not diverged, f"replicas diverged after optimizer.step: {diverged}"
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, "ok", {"compared": n_cmp, "worst_rel_err": worst}))
    except BaseException as e:  # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        q.put((rank, "fail", traceback.format_exc()[-3000:]))


@pytest.mark.parametrize("mode,overlap", [("strict", True), ("strict", False), ("multi", True)])
def test_dp1_train_step_over_rccl(mode, overlap):
    """The same exchange through RCCL (backend "nccl"), which wants one GPU per rank: a one-rank group on this box.
    The collectives are identities there, but every call of the data-parallel path runs on the real backend -- the
    broadcasts, the asynchronous arena all-reduces started from the end of the backward pass (their work handles and
    stream ordering), the packed small bucket, the flag exchange -- and the synchronised gradients must equal the
    single-process run's."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank_main, args=(0, 1, _free_port(), q, mode, overlap, "nccl"))
    p.start()
    try:
        rank, status, info = q.get(timeout=150)
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert status == "ok", info
    print("\n[dp1/rccl]", info)


@pytest.mark.parametrize("mode,overlap", [("strict", True), ("strict", False), ("multi", True)])
def test_dp2_train_step_on_one_gpu_over_gloo(mode, overlap):
    """overlap: the arenas' all-reduces started at the end of the backward pass (asynchronously) instead of inside
    sync_grads -- same collectives, same results."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, mode, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(2):
            try:
                res.append(q.get(timeout=150))
            except Exception:  # queue.Empty: a rank neither finished nor reported a failure
                # gloo's CUDA collectives between two processes that share ONE GPU are known to hang on this pool from
                # ~100k elements with nothing of this repository involved (scripts/gloo_probe.py);
                # the model here stays far below that, but should it bite, say so instead of failing the suite: the
                # ranks' faulthandler dumps name the frame they were stuck in
                root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                dumps = ""
                for r in range(2):
                    try:
                        dumps += open(os.path.join(root, "gpurun_out", f"dp2_rank{r}_hang.txt")).read()
                    except OSError:
                        pass
                if "distributed_c10d.py" in dumps:
                    pytest.skip("both ranks stuck inside a gloo collective on CUDA tensors (environment: scripts/gloo_probe.py)")
                raise
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}:\n{info}"
    print("\n[dp2]", res)
