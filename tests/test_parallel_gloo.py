"""world_size-2 gloo tests (CPU) of the data-parallel exchange used by train_step(dp=...)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plain(x):
    """Tensors as numpy arrays for the trip through the queue: a torch tensor travels as a file descriptor that the
    parent fetches from the CHILD when it unpickles it -- from a child that has exited by then, that is an EOFError
    (seen once in ~12 runs of this file)."""
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    return x


def _tensors(x):
    import numpy as np
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    if isinstance(x, dict):
        return {k: _tensors(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_tensors(v) for v in x)
    return x


def _finish(q, ok):
    """End of a rank: meet the other rank before the group is torn down (a rank that closes its connections while the
    other still drains its last collective makes that one fail), and leave without the interpreter's teardown once
    the results are on their way (gloo's threads and atexit handlers raced there: a non-zero exit code once in ~12
    runs of this file, with every assertion already passed)."""
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001 -- the results are what the parent checks
        pass
    if ok:
        q.close()
        q.join_thread()
        os._exit(0)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = False
    try:
        from ironcortex_amd.flat import ArenaParameter, FlatParams
        from ironcortex_amd.parallel import DataParallel
        dp = DataParallel()
        torch.manual_seed(7)  # identical replicas
        lin = torch.nn.Linear(4, 3)
        a, b, c = (ArenaParameter(torch.randn(2, 2)) for _ in range(3))
        flat = FlatParams({"W": [a, b, c]})
        mod = torch.nn.Module()
        mod.lin, mod.a, mod.b, mod.c = lin, a, b, c
        # rank-dependent gradients; b has a gradient only on rank 1, c on no rank
        lin.weight.grad = torch.full((3, 4), float(rank + 1))
        lin.bias.grad = None if rank == 0 else torch.ones(3)
        flat.begin_pass()
        flat.stacked_grad("W")[0].fill_(rank + 1.0)
        flat.publish("W", [True, rank == 1, False])
        if rank == 1:
            flat.stacked_grad("W")[1].fill_(4.0)
        dp.sync_grads(mod)
        out = {
            "w": lin.weight.grad.clone(), "bias": lin.bias.grad.clone(), "a": a.grad.clone(), "b": b.grad.clone(),
            "c_none": c.grad is None, "mean": dp.mean(torch.tensor([float(rank)])),
            "any": dp.any(torch.tensor([rank == 1, False])), "gather": dp.gather(torch.full((2, 3), float(rank))),
            "many": dp.mean_many([torch.full((3,), float(rank)), torch.tensor(2.0 * rank), torch.full((2, 2), 4.0 - rank)]),
        }
        q.put((rank, _plain(out)))
        ok = True
    finally:
        _finish(q, ok)


def test_dp_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _tensors(dict(q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        o = res[r]
        assert torch.allclose(o["w"], torch.full((3, 4), 1.5))      # mean of 1 and 2
        assert torch.allclose(o["bias"], torch.full((3,), 0.5))     # None on rank 0 counts as zero
        assert torch.allclose(o["a"], torch.full((2, 2), 1.5))
        assert torch.allclose(o["b"], torch.full((2, 2), 2.0))      # (0 + 4) / 2, rank 0 had None
        assert o["c_none"]                                          # None everywhere stays None
        assert torch.allclose(o["mean"], torch.tensor([0.5]))
        assert o["any"].tolist() == [True, False]
        many = o["many"]  # one all-reduce for several tensors, shapes kept
        assert torch.equal(many[0], torch.full((3,), 0.5)) and many[1].shape == () and float(many[1]) == 1.0
        assert torch.equal(many[2], torch.full((2, 2), 3.5))
        assert torch.equal(o["gather"], torch.cat([torch.zeros(2, 3), torch.ones(2, 3)]))
    # replicas end identical
    for k in ("w", "bias", "a", "b"):
        assert torch.equal(res[0][k], res[1][k])


def _worker_arena(rank, world, port, q):
    """The train_step path: touched flags OR-ed across ranks inside GradCoordinator._end (one host read),
    gradient arenas all-reduced in place, the rest through one packed bucket."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = False
    try:
        from ironcortex_amd.flat import ArenaParameter, FlatParams
        from ironcortex_amd.hip.ops import GradCoordinator
        from ironcortex_amd.parallel import DataParallel
        dp = DataParallel()
        torch.manual_seed(7)
        lin = torch.nn.Linear(4, 3)
        a, b, c = (ArenaParameter(torch.randn(2, 2)) for _ in range(3))
        flat = FlatParams({"W": [a, b, c]})
        touched = torch.tensor([1, 1 if rank == 1 else 0, 0], dtype=torch.uint8)  # b only on rank 1, c nowhere

        def launch():  # stands for the deferred dW kernels: writes the arena, returns the device-side flags
            flat.stacked_grad("W")[0].fill_(rank + 1.0)
            if rank == 1:
                flat.stacked_grad("W")[1].fill_(4.0)
            return touched

        class Model:  # the two things DataParallel needs from a CortexReasoner
            _coord = GradCoordinator()

            @staticmethod
            def grad_layout():
                return [flat], [lin.weight, lin.bias]

        m = Model()
        m._coord.register_flagged(flat.begin_pass, launch, lambda fl: flat.publish("W", fl))
        lin.weight.grad = torch.full((3, 4), float(rank + 1))
        lin.bias.grad = None if rank == 0 else torch.ones(3)
        dp.attach(m)
        for f in m._coord._begin:
            f()
        m._coord._pending = True
        m._coord._end()  # what the autograd engine calls at the end of a backward pass
        assert dp._inflight is not None and len(dp._inflight) == 1  # the arena's all-reduce is already on its way (overlap)
        dp.detach(m)
        none_before = (b.grad is None, c.grad is None, lin.bias.grad is None)
        dp.sync_grads(m)
        q.put((rank, _plain({"w": lin.weight.grad.clone(), "bias": lin.bias.grad.clone(), "a": a.grad.clone(),
                             "b": b.grad.clone(), "c_none": c.grad is None, "none_before": none_before,
                             "a_is_view": a.grad.data_ptr() == flat.grad_arena().data_ptr()})))
        ok = True
    finally:
        _finish(q, ok)


def test_dp_arena_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_arena, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _tensors(dict(q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        o = res[r]
        assert o["none_before"] == (False, True, False)             # rank-consistent None-ness before the exchange
        assert o["a_is_view"]                                       # reduced in place: .grad still views the arena
        assert torch.allclose(o["w"], torch.full((3, 4), 1.5))
        assert torch.allclose(o["bias"], torch.full((3,), 0.5))
        assert torch.allclose(o["a"], torch.full((2, 2), 1.5))
        assert torch.allclose(o["b"], torch.full((2, 2), 2.0))
        assert o["c_none"]
    for k in ("w", "bias", "a", "b"):
        assert torch.equal(res[0][k], res[1][k])
