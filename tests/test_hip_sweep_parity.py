"""GPU parity of the kernel variants the roofline sweep times (BASELINE configs[2]; VERDICT r1
items 2 / weak 3-4): region phase and router at R in {64, 512}, d = 256, sparse top-k style masks,
S in {1, 8} streams, rows_fwd in {library default, 16, 32}, per-edge routing with and without
precision weighting -- forward AND backward, each against the CPU oracle (oracle/hotpath.py, pinned
to the reference by tests/test_oracle_vs_reference.py), never against another HIP launch.

reference: region.py:168-249 (cell step), model.py:271-294 (region loop), gate.py:251-335 (router).
Tolerances: every [.., d] row is compared relative to that row's own largest reference magnitude
(an elementwise rtol/atol pair misreads rows whose entries span e^-3..e^+3, as exp(k)-weighted
state rows do at d = 256): activations and new state 2e-5.  Gradients: relative to the largest
magnitude of the same [R,d] stream block / d x d matrix, 1e-4 (a row of dW = g_i * x inherits the
relative error of a g_i that is itself a cancelling sum, so single rows say nothing).  KWTA winners and the
set of parameters that receive a gradient must be identical.  Observed errors are logged
(tests/parity_log.py)."""
import pytest
import torch

from tests import parity_log as PL
from tests.golden import inputs as gi
from tests.test_hip_parity import build_bank

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 256


def row_err(a, b) -> float:
    """max over rows of max|a - b| / max|b| of that row (rows that are zero in both are skipped)."""
    a = a.detach().double().cpu().reshape(-1, a.shape[-1]) if a.dim() > 1 else a.detach().double().cpu().reshape(1, -1)
    b = b.detach().double().cpu().reshape(a.shape)
    scale = b.abs().amax(1)
    diff = (a - b).abs().amax(1)
    live = (scale > 0) | (diff > 0)
    if not bool(live.any()):
        return 0.0
    return float((diff[live] / scale[live].clamp_min(1e-30)).max())


_TAG = [""]


def block_err(a, b) -> float:
    """max over leading-index blocks of max|a - b| / max|b| of that block."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.dim() >= 3:
        a, b = a.reshape(a.shape[0], -1), b.reshape(a.shape[0], -1)
    else:
        a, b = a.reshape(1, -1), b.reshape(1, -1)
    return row_err(a, b)


def close(a, b, what="", tol=2e-5, err=None):
    e = (err or row_err)(a, b)
    PL.note(_TAG[0], what.split("[")[0], e, what)
    assert e <= tol, f"{_TAG[0]} {what}: row-relative error {e:.3e} > {tol:.0e}"


def gclose(a, b, what=""):
    if b is None:  # the oracle saw no use of this input (e.g. no sensor/motor region active)
        assert a is None or float(a.abs().max()) == 0.0, what
        return
    close(a, b, what, tol=1e-4, err=block_err)


class _PerRegion:
    """P["W"][r, j] for the oracle with one autograd leaf per ACTIVE region (indexing one [R,4,d,d]
    leaf would zero-fill a 0.5 GB gradient per use at R = 512)."""

    def __init__(self, W, active):
        self.leaf = {int(r): W[int(r)].clone().requires_grad_(True) for r in active}

    def __getitem__(self, idx):
        r, j = idx
        return self.leaf[int(r)][j]


def sparse_mask(seed, S, R, k):
    """Each stream: the first k entries of a seeded permutation of the regions."""
    g = torch.Generator().manual_seed(seed)
    mask = torch.zeros(S, R, dtype=torch.bool)
    for s in range(S):
        mask[s, torch.randperm(R, generator=g)[:k]] = True
    return mask


@pytest.mark.parametrize("R,S,k", [(64, 1, 8), (64, 8, 8), (512, 1, 32), (512, 8, 32)])
def test_region_phase_sweep_variants_vs_oracle(R, S, k, monkeypatch):
    from ironcortex_amd.hip import ops
    from oracle import hotpath as O
    seed = 7000 + R + S
    bank, cells, P = build_bank(R, D, False, False, "scalar", seed)
    st = {"num": gi.det(seed + 10, S, R, D, kind="rand"), "den": gi.det(seed + 11, S, R, D, kind="rand", shift=0.5),
          "var": torch.zeros(S, R, D), "pred": gi.det(seed + 13, S, R, D, scale=0.1), "radius": torch.zeros(S, R),
          "radius_var": torch.zeros(S, R), "surprise": torch.zeros(S, R),
          "dt": (gi.det(seed + 17, S, R, kind="rand") * 8).to(torch.int32)}
    M, sensor = gi.det(seed + 18, S, R, D), gi.det(seed + 19, S, D)
    mask = sparse_mask(seed + 20, S, R, k)
    nin = gi.det(seed + 21, D, scale=0.1, shift=1.0)
    gH, gnum, gden, gpred = (gi.det(seed + 40 + i, S, R, D, scale=sc) for i, sc in enumerate((1.0, 0.3, 0.3, 0.3)))
    active = mask.any(0).nonzero().flatten().tolist()
    # ---- oracle (CPU autograd)
    Wl = _PerRegion(P["W"], active)
    Pc = {k_: v.clone().requires_grad_(True) for k_, v in P.items() if k_ != "W"}
    Pc["W"], Pc["W_time"] = Wl, O.time_freqs(D)
    stc = {k_: (v.clone().requires_grad_(True) if k_ in ("num", "den", "pred") else v.clone()) for k_, v in st.items()}
    Mc, sc_, ninc = M.clone().requires_grad_(True), sensor.clone().requires_grad_(True), nin.clone().requires_grad_(True)
    Hc, newc = O.region_phase(Pc, stc, Mc, sc_, mask, 0.5, norm_in_w=ninc, sensor_idx=0, motor_idx=R - 1, kwta_k=D // 8)
    ((Hc * gH).sum() + (newc["num"] * gnum).sum() + (newc["den"] * gden).sum() + (newc["pred"] * gpred).sum()).backward()
    # ---- HIP, once per forward row-slice variant (0 = the library's own choice for this R)
    for rows in (0, 16, 32):
        monkeypatch.setattr(ops, "ROWS_FWD", rows)
        bank.reset_state(torch.device(DEV), S)
        leaves = {k_: st[k_].clone().to(DEV).requires_grad_(True) for k_ in ("num", "den", "pred")}
        for k_, v in leaves.items():
            setattr(bank, k_, v * 1.0)
        bank.dt = st["dt"].to(DEV)
        Mg, sg, ning = M.to(DEV).requires_grad_(True), sensor.to(DEV).requires_grad_(True), nin.to(DEV).requires_grad_(True)
        Hg = bank.phase(Mg, sg, ning, mask.to(torch.uint8).to(DEV), 0.5)
        ((Hg * gH.to(DEV)).sum() + (bank.num * gnum.to(DEV)).sum() + (bank.den * gden.to(DEV)).sum()
         + (bank.pred * gpred.to(DEV)).sum()).backward()
        tag = _TAG[0] = f"sweep-region/R{R}/S{S}/rows{rows}"
        assert torch.equal((Hg != 0).cpu(), Hc != 0), "KWTA winners differ"
        close(Hg, Hc, what="H")
        for k_ in ("num", "den", "pred"):
            close(getattr(bank, k_), newc[k_], what=k_)
        assert torch.equal(bank.dt.cpu(), newc["dt"])
        gclose(Mg.grad, Mc.grad, what="g.M"); gclose(sg.grad, sc_.grad, what="g.sensor"); gclose(ning.grad, ninc.grad, what="g.norm_in")
        for k_ in ("num", "den", "pred"):
            gclose(leaves[k_].grad, stc[k_].grad, what="g." + k_)
        on = set(active)
        for r, c in enumerate(cells):
            if r not in on:
                assert c.r_lin.weight.grad is None and c.norm.weight.grad is None  # untouched -> None (AdamW skip)
                continue
            gw = torch.stack([lin.weight.grad for lin in (c.r_lin, c.k_lin, c.v_lin, c.o_lin)])
            gclose(gw, Wl.leaf[r].grad, what=f"g.W[{r}]")
            gclose(c.norm.weight.grad, Pc["norm_w"].grad[r], what="g.norm_w")
            gclose(c.raw_decay.grad, Pc["raw_decay"].grad[r], what="g.raw_decay")
        for c in cells:
            for p in c.parameters():
                p.grad = None
        PL.flush(tag)


@pytest.mark.parametrize("d,afd,rt,noise", [(256, False, False, "scalar"), (256, True, True, "vector"),
                                            (128, True, False, "scalar"), (512, True, True, "scalar")])
def test_region_forward_many_streams_equals_stream_groups_bitwise(d, afd, rt, noise):
    """From 32 streams per launch the library takes the split forward (k_region_rkvG + k_region_state) instead of
    the fused k_region_rkv: same dot products and per-channel arithmetic, so 64 streams in ONE launch equal the
    same streams in eight launches of 8 (the fused kernel, itself pinned to the oracle above) bit for bit.
    IC_FWD_SPLIT=1 sends every non-chained launch of a process down the split path (run the suite under it)."""
    R, S, k, G = 48, 64, 6, 8
    seed = 7700 + d
    bank, cells, P = build_bank(R, d, afd, rt, noise, seed)
    st = {"num": gi.det(seed + 10, S, R, d, kind="rand"), "den": gi.det(seed + 11, S, R, d, kind="rand", shift=0.5),
          "var": gi.det(seed + 12, S, R, d, kind="rand", scale=0.2), "pred": gi.det(seed + 13, S, R, d, scale=0.1),
          "radius": gi.det(seed + 14, S, R, kind="rand", shift=0.5), "radius_var": gi.det(seed + 15, S, R, kind="rand", scale=0.2),
          "surprise": torch.zeros(S, R), "dt": (gi.det(seed + 17, S, R, kind="rand") * 8).to(torch.int32)}
    M, sensor = gi.det(seed + 18, S, R, d).to(DEV), gi.det(seed + 19, S, d).to(DEV)
    mask = sparse_mask(seed + 20, S, R, k).to(torch.uint8).to(DEV)
    mask[5] = 0  # a stream with nothing to do
    nin = gi.det(seed + 21, d, scale=0.1, shift=1.0).to(DEV)
    keys = ("num", "den", "var", "pred", "radius", "radius_var", "surprise", "dt")

    def run(lo, hi):
        bank.reset_state(torch.device(DEV), hi - lo)
        for k_ in keys:
            setattr(bank, k_, st[k_][lo:hi].clone().to(DEV))
        with torch.no_grad():
            H = bank.phase(M[lo:hi], sensor[lo:hi], nin, mask[lo:hi], 0.5)
        return [H.clone()] + [getattr(bank, k_).clone() for k_ in keys]

    whole = run(0, S)
    parts = [run(lo, lo + G) for lo in range(0, S, G)]
    assert float(whole[0].abs().sum()) > 0
    for i, name in enumerate(("H",) + keys):
        assert torch.equal(whole[i], torch.cat([p_[i] for p_ in parts])), name


@pytest.mark.parametrize("R,S,k", [(64, 1, 8), (64, 8, 8), (512, 1, 32), (512, 8, 32)])
@pytest.mark.parametrize("prec", [False, True], ids=["plain", "precision"])
def test_router_sweep_sizes_vs_oracle(R, S, k, prec):
    """per-edge routing at d = 256 (k_router_fwd_ms / k_router_edge_fwd + combine, and their
    backward kernels) against the per-stream oracle."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    seed = 7500 + R + S
    nb = hex_neighbors(R)
    coords = hex_axial_coords(R)
    router = Router(nb, D, R, enable_precision_routed_messages=prec).to(DEV)
    E = router.E
    mask = sparse_mask(seed, S, R, k)
    src = router.src_idx.cpu()
    e_act = mask[:, src].any(0).nonzero().flatten().tolist()          # edges with an active sender in some stream
    W_edge = gi.det(seed + 1, E, D, D, kind="xavier")
    key_W = gi.det(seed + 2, E, D, D, kind="xavier") if prec else None
    query_W = gi.det(seed + 3, R, D, D, kind="xavier") if prec else None
    raw_P = gi.det(seed + 4, E, D, scale=0.3, shift=-2.0)
    t = dict(W_reg=gi.det(seed + 5, 8, 2), beta_cos=gi.det(seed + 6, 8, scale=0.3), beta_sin=gi.det(seed + 7, 8, scale=0.3))
    with torch.no_grad():  # parameters are views of the router's arenas: fill the stacked views directly
        router._flat.stacked("W").copy_(W_edge)
        if prec:
            router._flat.stacked("key_W").copy_(key_W)
            router._flat.stacked("query_W").copy_(query_W)
            router._flat.stacked("raw_P").copy_(raw_P)
        router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
    H = gi.det(seed + 8, S, R, D) * mask.unsqueeze(-1)  # inactive regions carry zero states, as in the loop
    gM = gi.det(seed + 9, S, R, D)
    # ---- HIP
    Hg = H.to(DEV).requires_grad_(True)
    M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
    (M * gM.to(DEV)).sum().backward()
    # ---- oracle: one autograd leaf per touched matrix
    leaf = lambda x: x.clone().requires_grad_(True)
    on = set(e_act)
    P = {"W_edge": [leaf(W_edge[e]) if e in on else W_edge[e] for e in range(E)], "W_reg": t["W_reg"],
         "beta_cos": leaf(t["beta_cos"]), "beta_sin": leaf(t["beta_sin"])}
    if prec:
        P["key_W"] = [leaf(key_W[e]) if e in on else key_W[e] for e in range(E)]
        P["raw_P"] = [leaf(raw_P[e]) if e in on else raw_P[e] for e in range(E)]
        P["query_W"] = [leaf(query_W[r]) for r in range(R)]
    Hc = H.clone().requires_grad_(True)
    Mc = torch.stack([O.router_messages(P, Hc[s], mask[s], coords, nb, precision=prec)[0] for s in range(S)])
    (Mc * gM).sum().backward()
    tag = _TAG[0] = f"sweep-router/R{R}/S{S}/{'precision' if prec else 'plain'}"
    close(M, Mc, what="M")
    gclose(Hg.grad, Hc.grad, what="g.H")
    gclose(router.beta_cos.grad, P["beta_cos"].grad, what="g.beta_cos")
    gclose(router.beta_sin.grad, P["beta_sin"].grad, what="g.beta_sin")
    for e, key in enumerate(router.edge_keys):
        g = router.W_edge[key].weight.grad
        assert (g is not None) == (e in on), key  # inactive senders leave their edge weights without a gradient
        if g is not None:
            gclose(g, P["W_edge"][e].grad, what=f"g.W_edge[{key}]")
    if prec:
        # (an edge that dominates its destination's softmax has d loss / d log-weight ~ 0 by cancellation,
        # and raw_P's gradient is proportional to it: compared against the largest raw_P gradient of the call)
        keys = [router.edge_keys[e] for e in e_act]
        # and the key / query matrices' to d loss / d score, which cancels the same way)
        close(torch.stack([router.raw_P_edge[k_].grad for k_ in keys]).reshape(1, -1),
              torch.stack([P["raw_P"][e].grad for e in e_act]).reshape(1, -1), what="g.rawP", tol=1e-4)
        close(torch.stack([router.key_lin[k_].weight.grad for k_ in keys]).reshape(1, -1),
              torch.stack([P["key_W"][e].grad for e in e_act]).reshape(1, -1), what="g.key", tol=1e-4)
        zq = torch.zeros(D, D)
        close(torch.stack([router.query_lin[r].weight.grad.cpu() for r in range(R)]).reshape(1, -1),
              torch.stack([P["query_W"][r].grad if P["query_W"][r].grad is not None else zq for r in range(R)]).reshape(1, -1),
              what="g.query", tol=1e-4)
    PL.flush(tag)


@pytest.mark.parametrize("d", [256, 512, 48])
def test_router_forward_many_streams_equals_stream_groups_bitwise(d):
    """The per-edge router forward walks the streams 64 at a time, each edge's active streams from one ballot:
    64 + 6 streams in one launch equal the same streams in launches of 8 bit for bit (products and accumulation
    order do not depend on how many streams share a launch)."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    R, S, k, G = 90, 70, 12, 8
    seed = 7800 + d
    router = Router(hex_neighbors(R), d, R).to(DEV)
    coords = hex_axial_coords(R).to(DEV)
    with torch.no_grad():
        router._flat.stacked("W").copy_(gi.det(seed + 1, router.E, d, d, kind="xavier"))
    mask = sparse_mask(seed, S, R, k)
    mask[3] = False
    H = (gi.det(seed + 8, S, R, d) * mask.unsqueeze(-1)).to(DEV)
    mg = mask.to(torch.uint8).to(DEV)
    with torch.no_grad():
        whole = router.route(H, mg, coords).clone()
        parts = torch.cat([router.route(H[lo:lo + G], mg[lo:lo + G], coords).clone() for lo in range(0, S, G)])
    assert float(whole.abs().sum()) > 0
    assert torch.equal(whole, parts)


@pytest.mark.parametrize("gemm", [True, False], ids=["gemm", "per-edge-kernels"])
@pytest.mark.parametrize("R,S,d", [(36, 4, 256), (64, 8, 256), (10, 3, 48)])
def test_router_by_direction_mfma_vs_oracle(R, S, d, gemm, monkeypatch):
    """edge_transform_mode="by_direction" (gate.py:142-149,274-277): the forward is ONE fp32-MFMA GEMM
    [d, 6d] x [6d, S*R] (k_router_shared_fwd_mfma; at these sizes only when asked for, IC_RF_SHARED_GEMM -- by default
    the library serves fewer than 8192 columns with the per-edge kernels reading W[widx[e]], the second variant), the
    shared matrices' gradient one deterministic MFMA GEMM per direction (k_wgrad_shared_mfma, no float atomics) --
    against the per-stream oracle, and bit-identical from run to run."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.hip import ops
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    monkeypatch.setattr(ops, "SHARED_GEMM_ALWAYS", gemm)
    seed = 7900 + R + S
    nb = hex_neighbors(R)
    coords = hex_axial_coords(R)
    router = Router(nb, d, R, edge_transform_mode="by_direction").to(DEV)
    W_dir = gi.det(seed + 1, 6, d, d, kind="xavier")
    t = dict(W_reg=gi.det(seed + 5, 8, 2), beta_cos=gi.det(seed + 6, 8, scale=0.3), beta_sin=gi.det(seed + 7, 8, scale=0.3))
    with torch.no_grad():
        router._flat.stacked("W").copy_(W_dir)
        router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
    mask = gi.det(seed + 2, S, R, kind="rand") < 0.45
    mask[S - 1] = False  # a stream with no active sender: zeros
    H = gi.det(seed + 8, S, R, d) * mask.unsqueeze(-1)
    gM = gi.det(seed + 9, S, R, d)
    runs = []
    for _ in range(2):
        for p in router.parameters():
            p.grad = None
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
        assert router._shared_gemm  # the GEMM formulation is the path taken on a hex map
        (M * gM.to(DEV)).sum().backward()
        runs.append((M.detach().clone(), Hg.grad.clone(), torch.stack([l.weight.grad.clone() for l in router.W_dir])))
    for a_, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a_, b_), "not reproducible from run to run"
    M, gH, gW = runs[0]
    assert torch.all(M[S - 1] == 0)
    leaf = lambda x: x.clone().requires_grad_(True)
    P = {"W_dir": [leaf(W_dir[k]) for k in range(6)], "W_reg": t["W_reg"], "beta_cos": leaf(t["beta_cos"]), "beta_sin": leaf(t["beta_sin"])}
    Hc = H.clone().requires_grad_(True)
    Mc = torch.stack([O.router_messages(P, Hc[s], mask[s], coords, nb, mode="by_direction")[0] for s in range(S)])
    (Mc * gM).sum().backward()
    _TAG[0] = f"by_direction-{'mfma' if gemm else 'edge-kernels'}/R{R}/S{S}/d{d}"
    close(M, Mc, what="M")
    gclose(gH, Hc.grad, what="g.H")
    gclose(router.beta_cos.grad, P["beta_cos"].grad, what="g.beta_cos")
    gclose(router.beta_sin.grad, P["beta_sin"].grad, what="g.beta_sin")
    zeros = torch.zeros(d, d)
    gclose(gW, torch.stack([P["W_dir"][k].grad if P["W_dir"][k].grad is not None else zeros for k in range(6)]), what="g.W_dir")
    PL.flush(_TAG[0])


@pytest.mark.parametrize("gemm", [False, True, "sorted"], ids=["edge-kernels", "gemm", "gemm-sorted"])
@pytest.mark.parametrize("R,S,d", [(36, 4, 256), (10, 3, 48), (36, 9, 64)])
def test_router_by_direction_precision_vs_oracle(R, S, d, gemm, monkeypatch):
    """by_direction WITH precision routing (gate.py:274-277 with :292-318): messages and keys through the six shared
    matrices -- by the per-edge precision kernels reading W[widx[e]] / key_W[widx[e]] / raw_P[widx[e]] in place (the
    library's choice below 8192 (stream, edge) pairs), or as TWO shared-matrix GEMMs over the pairs (IC_RF_SHARED_GEMM
    here; with the pairs sorted by direction in the third variant) -- per-region queries, softmax-combined per
    destination: output and every gradient against the per-stream oracle, bit-identical from run to run."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.hip import ops
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    monkeypatch.setattr(ops, "SHARED_GEMM_ALWAYS", bool(gemm))
    monkeypatch.setattr(ops, "SHARED_SORT", gemm == "sorted")
    monkeypatch.setattr(ops, "SHARED_SORT_MIN", 64)
    seed = 8500 + R + S
    nb = hex_neighbors(R)
    coords = hex_axial_coords(R)
    router = Router(nb, d, R, edge_transform_mode="by_direction", enable_precision_routed_messages=True).to(DEV)
    W_dir, K_dir = gi.det(seed + 1, 6, d, d, kind="xavier"), gi.det(seed + 2, 6, d, d, kind="xavier")
    Q = gi.det(seed + 3, R, d, d, kind="xavier")
    rawP = gi.det(seed + 4, 6, d, scale=0.5)
    t = dict(W_reg=gi.det(seed + 5, 8, 2), beta_cos=gi.det(seed + 6, 8, scale=0.3), beta_sin=gi.det(seed + 7, 8, scale=0.3))
    with torch.no_grad():
        router._flat.stacked("W").copy_(W_dir); router._flat.stacked("key_W").copy_(K_dir)
        router._flat.stacked("query_W").copy_(Q); router._flat.stacked("raw_P").copy_(rawP)
        router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
    mask = gi.det(seed + 8, S, R, kind="rand") < 0.45
    mask[S - 1] = False
    H = gi.det(seed + 9, S, R, d) * mask.unsqueeze(-1)
    gM = gi.det(seed + 10, S, R, d)
    runs = []
    for _ in range(2):
        for p in router.parameters():
            p.grad = None
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
        (M * gM.to(DEV)).sum().backward()
        zero = lambda x: x.grad.clone() if x.grad is not None else torch.zeros_like(x)
        runs.append((M.detach().clone(), Hg.grad.clone(), torch.stack([zero(l.weight) for l in router.W_dir]),
                     torch.stack([zero(l.weight) for l in router.key_lin_dir]), torch.stack([zero(l.weight) for l in router.query_lin]),
                     torch.stack([zero(p_) for p_ in router.raw_P_dir]), router.beta_cos.grad.clone(), router.beta_sin.grad.clone()))
    for a_, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a_, b_), "not reproducible from run to run"
    M, gH, gW, gK, gQ, gP, gbc, gbs = runs[0]
    assert torch.all(M[S - 1] == 0) and float(M.abs().sum()) > 0
    from ironcortex_amd import hip  # which kernels served the forward: the variant asked for
    hip.prof_select("")
    hip.prof_enable(True)
    hip.prof_read()
    try:
        with torch.no_grad():
            router.route(H.to(DEV), mask.to(torch.uint8).to(DEV), coords.to(DEV))
        torch.cuda.synchronize()
        prof = hip.prof_read()
    finally:
        hip.prof_enable(False)
    assert ("router_mfma" in prof) == bool(gemm), sorted(prof)
    assert prof.get("router_mfma", (0, 0))[0] == (0 if not gemm else (6 if gemm == "sorted" else 2)), prof  # (2 GEMMs [+ 2 sorts of 2 launches])
    leaf = lambda x: x.clone().requires_grad_(True)
    P = {"W_dir": [leaf(W_dir[k]) for k in range(6)], "key_W": [leaf(K_dir[k]) for k in range(6)], "raw_P": [leaf(rawP[k]) for k in range(6)],
         "query_W": [leaf(Q[r]) for r in range(R)], "W_reg": t["W_reg"], "beta_cos": leaf(t["beta_cos"]), "beta_sin": leaf(t["beta_sin"])}
    Hc = H.clone().requires_grad_(True)
    Mc = torch.stack([O.router_messages(P, Hc[s], mask[s], coords, nb, mode="by_direction", precision=True)[0] for s in range(S)])
    (Mc * gM).sum().backward()
    z = lambda xs: torch.stack([x.grad if x.grad is not None else torch.zeros_like(x) for x in xs])
    _TAG[0] = f"by_direction-precision-{gemm}/R{R}/S{S}/d{d}"
    close(M, Mc, what="M")
    gclose(gH, Hc.grad, what="g.H")
    gclose(gW, z(P["W_dir"]), what="g.W_dir")
    # keys / queries / precisions: relative to the largest entry of the whole stack (a destination with ONE active
    # in-edge has a softmax over one logit: its query and key gradients are analytically zero, rounding noise on both sides)
    flat = lambda x: x.reshape(1, -1)
    close(flat(gK), flat(z(P["key_W"])), what="g.key_W", tol=1e-4)
    close(flat(gQ), flat(z(P["query_W"])), what="g.query_W", tol=1e-4)
    close(flat(gP), flat(z(P["raw_P"])), what="g.raw_P", tol=1e-4)
    gclose(gbc, P["beta_cos"].grad, what="g.beta_cos")
    gclose(gbs, P["beta_sin"].grad, what="g.beta_sin")
    PL.flush(_TAG[0])


@pytest.mark.parametrize("R,S,d,p_on", [(128, 32, 64, 0.3), (100, 41, 48, 0.06), (256, 16, 256, 0.0625)])
def test_router_by_direction_sorted_columns(R, S, d, p_on, monkeypatch):
    """With a workspace (ops: from 8192 columns on; here from 4096), the shared-matrix forward GEMM walks its columns sorted by activity pattern
    (k_shared_pattern / k_shared_scatter, ic_router_args.shared_ws), so that a 32-column tile multiplies only the
    direction matrices its columns use, and the data-gradient GEMM walks the active senders first.  Every column's
    result is its own: outputs and gradients must equal the unsorted kernels' bit for bit (twice in a row: the workspace header is cleared per call), and the oracle's on sampled
    streams (gate.py:142-149,274-277)."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.hip import ops
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    seed = 8100 + R + S
    monkeypatch.setattr(ops, "SHARED_GEMM_ALWAYS", True)  # (the library's own choice starts at 8192 columns)
    monkeypatch.setattr(ops, "SHARED_SORT_MIN", 4096)
    nb = hex_neighbors(R)
    coords = hex_axial_coords(R)
    router = Router(nb, d, R, edge_transform_mode="by_direction").to(DEV)
    W_dir = gi.det(seed + 1, 6, d, d, kind="xavier")
    t = dict(W_reg=gi.det(seed + 5, 8, 2), beta_cos=gi.det(seed + 6, 8, scale=0.3), beta_sin=gi.det(seed + 7, 8, scale=0.3))
    with torch.no_grad():
        router._flat.stacked("W").copy_(W_dir)
        router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
    mask = gi.det(seed + 2, S, R, kind="rand") < p_on
    mask[S - 1] = False
    H = gi.det(seed + 8, S, R, d) * mask.unsqueeze(-1)
    gM = gi.det(seed + 9, S, R, d).to(DEV)
    outs = {}
    for sort in (True, False, True):
        monkeypatch.setattr(ops, "SHARED_SORT", sort)
        for p in router.parameters():
            p.grad = None
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
        assert router._shared_gemm
        (M * gM).sum().backward()
        got = (M.detach().clone(), Hg.grad.clone(), router.beta_cos.grad.clone(), router.beta_sin.grad.clone(),
               torch.stack([l.weight.grad.clone() for l in router.W_dir]))
        if sort in outs:
            for x, y in zip(got, outs[sort]):
                assert torch.equal(x, y), "second sorted call differs from the first"
        outs[sort] = got
    assert any(k[0] == S * R for k in ops._SHARED_WS)
    assert float(outs[True][0].abs().sum()) > 0 and float(outs[True][1].abs().sum()) > 0
    for x, y, what in zip(outs[True], outs[False], ("M", "g.H", "g.beta_cos", "g.beta_sin", "g.W_dir")):
        assert torch.equal(x, y), f"{what}: sorted and unsorted columns differ"
    outs = {k: v[0] for k, v in outs.items()}
    P = {"W_dir": [W_dir[k] for k in range(6)], "W_reg": t["W_reg"], "beta_cos": t["beta_cos"], "beta_sin": t["beta_sin"]}
    _TAG[0] = f"by_direction-sorted/R{R}/S{S}/d{d}"
    for s in (0, S // 2, S - 1):
        close(outs[True][s], O.router_messages(P, H[s], mask[s], coords, nb, mode="by_direction")[0], what=f"M[{s}]")
    PL.flush(_TAG[0])


@pytest.mark.parametrize("R,S,d", [(36, 4, 256), (64, 8, 256), (10, 3, 48)])
def test_router_factorized_lib_vs_oracle(R, S, d, monkeypatch):
    """edge_transform_mode="factorized" (gate.py:150-159,278-281) on the library (IC_RF_FACTORIZED): V h per active
    sender and U z per destination as shared-matrix fp32-MFMA GEMMs, the per-edge diagonal in the mix kernels, dU / dV
    through the deterministic shared weight-gradient GEMM -- against the per-stream oracle (which applies U per EDGE,
    as the reference does), bit-identical from run to run; w_e of an inactive edge keeps grad None."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    from oracle import hotpath as O
    seed = 8100 + R + S
    nb = hex_neighbors(R)
    coords = hex_axial_coords(R)
    router = Router(nb, d, R, edge_transform_mode="factorized").to(DEV)
    assert router.fact_lib
    monkeypatch.setattr(router, "_route_factorized", None)  # the PyTorch formulation must not be the path taken
    E = router.E
    U, V = gi.det(seed + 1, d, d, kind="xavier"), gi.det(seed + 2, d, d, kind="xavier")
    w_e = 1.0 + gi.det(seed + 3, E, d, scale=0.3)
    t = dict(W_reg=gi.det(seed + 5, 8, 2), beta_cos=gi.det(seed + 6, 8, scale=0.3), beta_sin=gi.det(seed + 7, 8, scale=0.3))
    with torch.no_grad():
        router.U.weight.copy_(U); router.V.weight.copy_(V)
        for e, k in enumerate(router.edge_keys):
            router.w_edge[k].copy_(w_e[e])
        router.W_reg.copy_(t["W_reg"]); router.beta_cos.copy_(t["beta_cos"]); router.beta_sin.copy_(t["beta_sin"])
    mask = gi.det(seed + 4, S, R, kind="rand") < 0.45
    mask[S - 1] = False
    H = gi.det(seed + 8, S, R, d) * mask.unsqueeze(-1)
    gM = gi.det(seed + 9, S, R, d)
    runs = []
    for _ in range(2):
        for p in router.parameters():
            p.grad = None
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords.to(DEV))
        (M * gM.to(DEV)).sum().backward()
        gw = torch.stack([router.w_edge[k].grad.clone() if router.w_edge[k].grad is not None else torch.zeros(d, device=DEV)
                          for k in router.edge_keys])
        runs.append((M.detach().clone(), Hg.grad.clone(), router.U.weight.grad.clone(), router.V.weight.grad.clone(), gw))
    for a_, b_ in zip(runs[0], runs[1]):
        assert torch.equal(a_, b_), "not reproducible from run to run"
    M, gH, gU, gV, gw = runs[0]
    assert torch.all(M[S - 1] == 0)
    leaf = lambda x: x.clone().requires_grad_(True)
    P = {"U": leaf(U), "V": leaf(V), "w_edge": [leaf(w_e[e]) for e in range(E)], "W_reg": t["W_reg"],
         "beta_cos": leaf(t["beta_cos"]), "beta_sin": leaf(t["beta_sin"])}
    Hc = H.clone().requires_grad_(True)
    Mc = torch.stack([O.router_messages(P, Hc[s], mask[s], coords, nb, mode="factorized")[0] for s in range(S)])
    (Mc * gM).sum().backward()
    _TAG[0] = f"factorized-lib/R{R}/S{S}/d{d}"
    close(M, Mc, what="M")
    gclose(gH, Hc.grad, what="g.H")
    gclose(router.beta_cos.grad, P["beta_cos"].grad, what="g.beta_cos")
    gclose(router.beta_sin.grad, P["beta_sin"].grad, what="g.beta_sin")
    gclose(gU, P["U"].grad, what="g.U")
    gclose(gV, P["V"].grad, what="g.V")
    gclose(gw, torch.stack([P["w_edge"][e].grad if P["w_edge"][e].grad is not None else torch.zeros(d) for e in range(E)]), what="g.w_edge")
    act = mask[:, router.src_idx.cpu()].any(0)
    for e, k in enumerate(router.edge_keys):
        assert (router.w_edge[k].grad is not None) == bool(act[e]), k
    PL.flush(_TAG[0])


@pytest.mark.parametrize("R,S,d,p_on", [(128, 32, 64, 0.3), (256, 16, 256, 0.0625)])
def test_router_factorized_sorted_columns(R, S, d, p_on, monkeypatch):
    """The two row GEMMs of the factorized mode (t = V h over the active senders, M = U z over the destinations with
    an active in-edge; gate.py:150-159,278-281) and their input gradients take the column sort too
    (fact_rows passes ic_router_args.shared_ws on): outputs and gradients equal the unsorted run's bit for bit."""
    from ironcortex_amd.gate import Router
    from ironcortex_amd.hip import ops
    from ironcortex_amd.wiring import hex_axial_coords, hex_neighbors
    seed = 8300 + R + S
    monkeypatch.setattr(ops, "SHARED_SORT_MIN", 4096)
    router = Router(hex_neighbors(R), d, R, edge_transform_mode="factorized").to(DEV)
    assert router.fact_lib
    monkeypatch.setattr(router, "_route_factorized", None)
    coords = hex_axial_coords(R).to(DEV)
    with torch.no_grad():
        router.U.weight.copy_(gi.det(seed + 1, d, d, kind="xavier")); router.V.weight.copy_(gi.det(seed + 2, d, d, kind="xavier"))
        router.beta_cos.copy_(gi.det(seed + 6, 8, scale=0.3)); router.beta_sin.copy_(gi.det(seed + 7, 8, scale=0.3))
    mask = gi.det(seed + 4, S, R, kind="rand") < p_on
    mask[S - 1] = False
    H = gi.det(seed + 8, S, R, d) * mask.unsqueeze(-1)
    gM = gi.det(seed + 9, S, R, d).to(DEV)
    outs = {}
    for sort in (True, False):
        monkeypatch.setattr(ops, "SHARED_SORT", sort)
        for p in router.parameters():
            p.grad = None
        Hg = H.to(DEV).requires_grad_(True)
        M = router.route(Hg, mask.to(torch.uint8).to(DEV), coords)
        (M * gM).sum().backward()
        gw = torch.stack([router.w_edge[k].grad.clone() if router.w_edge[k].grad is not None else torch.zeros(d, device=DEV)
                          for k in router.edge_keys])
        outs[sort] = (M.detach().clone(), Hg.grad.clone(), router.U.weight.grad.clone(), router.V.weight.grad.clone(), gw,
                      router.beta_cos.grad.clone(), router.beta_sin.grad.clone())
    assert float(outs[True][0].abs().sum()) > 0 and float(outs[True][1].abs().sum()) > 0
    for x, y, what in zip(outs[True], outs[False], ("M", "g.H", "g.U", "g.V", "g.w_edge", "g.beta_cos", "g.beta_sin")):
        assert torch.equal(x, y), f"{what}: sorted and unsorted columns differ"
