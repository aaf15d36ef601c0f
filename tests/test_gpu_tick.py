"""fspann_tick_dev: encode of batch t+2, Route of batch t+1 and Refine of batch t as ONE kernel (tick.hip.h).

Each part must produce exactly what its stand-alone call produces — so the pipelined results are compared with the
oracle's QueryServiceImpl.search (QSI:101-352) batch by batch, bit-exact, for the dense boundary ([Q][B][d] rows handed
over) and the store boundary (rows read by id), with and without the hand-over buffer, for the fused kernel and for the
fall-back to stand-alone kernels; queries the bounded select hands over are finished by the refine role one tick later.
"""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu
K = 10


def _ctx(pkg, sc, jh=None):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"],
                                 max_global_candidates=p["hard_cap"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"], jh)
    return ctx


def _pipeline(pkg, ctx, sc, Qb, dense, handover, expect_fused=True):
    """Run the batches of Qb [nb][Q][d] through a 3-deep tick pipeline; returns per batch (ids, dist, count, sel, sel_count)."""
    import torch
    dev = torch.device("cuda", 0)
    p = sc["params"]
    nb, Q, d = Qb.shape
    B, TD, W = p["B"], p["T"] * p["D"], (p["m"] * p["lam"] + 63) // 64
    F32 = pkg._native.F32
    X = torch.from_numpy(sc["X"]).to(dev)
    qd = torch.from_numpy(Qb).to(dev)
    slots = 3
    codes = [torch.zeros((Q, TD, W), dtype=torch.int64, device=dev) for _ in range(slots)]
    bad = [torch.zeros(Q, dtype=torch.int32, device=dev) for _ in range(slots)]
    sel = [torch.full((Q, B), -1, dtype=torch.int32, device=dev) for _ in range(slots)]
    selc = [torch.zeros(Q, dtype=torch.int32, device=dev) for _ in range(slots)]
    hov = [torch.zeros(ctx.route_handover_bytes(Q), dtype=torch.uint8, device=dev) for _ in range(slots)] if handover else None
    cand = torch.zeros((Q, B, d), dtype=torch.float32, device=dev)
    oi = torch.zeros((Q, K), dtype=torch.int32, device=dev)
    od = torch.zeros((Q, K), dtype=torch.float64, device=dev)
    oc = torch.zeros(Q, dtype=torch.int32, device=dev)
    scn = torch.zeros(Q, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    out, fused_seen = [], []
    for t in range(nb + 2):
        enc = rt = rf = None
        if t < nb:
            s = t % slots
            enc = dict(nq=Q, q=qd[t].data_ptr(), codes=codes[s].data_ptr(), bad=bad[s].data_ptr())
        if 0 <= t - 1 < nb:
            s = (t - 1) % slots
            rt = dict(nq=Q, codes=codes[s].data_ptr(), limit=B, ids=sel[s].data_ptr(), count=selc[s].data_ptr(),
                      handover=hov[s].data_ptr() if handover else None)
        if 0 <= t - 2 < nb:
            s = (t - 2) % slots
            if dense:
                # the host's load + decrypt of F_q (here: a gather from plaintext).  With a hand-over buffer a PENDING query's
                # F_q does not exist yet: the host would decrypt it one tick later — the test packs those rows from the
                # stand-alone route instead, which is what the redo must reproduce
                ctx.sync()
                ids_h, cnt_h = sel[s].cpu().numpy(), selc[s].cpu().numpy()
                if (cnt_h == -2).any():
                    ref_rt = ctx.route(codes[s].cpu().numpy().view(np.uint64), limit=B, counters=False, allow_unmodelled=True)
                    ids_h = np.where((cnt_h == -2)[:, None], ref_rt["ids"][:, :B], ids_h)
                rows = np.zeros((Q, B, d), np.float32)
                rows[:] = sc["X"][np.clip(ids_h, 0, p["n"] - 1)]
                cand.copy_(torch.from_numpy(rows))
                torch.cuda.synchronize()
            rf = dict(nq=Q, q=qd[t - 2].data_ptr(), B=B, ids=sel[s].data_ptr(), count=selc[s].data_ptr(), k=K, out_ids=oi.data_ptr(),
                      out_dist=od.data_ptr(), out_count=oc.data_ptr(), scored=scn.data_ptr(), cand=cand.data_ptr() if dense else None,
                      codes=codes[s].data_ptr() if handover else None, handover=hov[s].data_ptr() if handover else None)
        ctx.tick_dev(enc, rt, rf)
        if rf is not None:
            fused_seen.append(ctx.last_tick_fused())    # (a tick without a Refine part may share a kernel even when B > 256)
        if rf is not None:
            ctx.sync()
            s = (t - 2) % slots
            out.append(dict(ids=oi.cpu().numpy().copy(), dist=od.cpu().numpy().copy(), count=oc.cpu().numpy().copy(),
                            scored=scn.cpu().numpy().copy(), sel=sel[s].cpu().numpy().copy(), sel_count=selc[s].cpu().numpy().copy(),
                            bad=bad[s].cpu().numpy().copy()))
    assert all(fused_seen) == expect_fused and any(fused_seen) == expect_fused, fused_seen
    del X
    return out


def _check(sc, Qb, out):
    o, B = sc["oracle"], sc["params"]["B"]
    for b, res in enumerate(out):
        ref = o.search(Qb[b].astype(np.float64), K)
        assert not ref["metrics"][:, 4].any()
        assert np.array_equal(res["sel_count"], ref["sel_count"]), b
        assert np.array_equal(np.where(np.arange(B)[None] < res["sel_count"][:, None], res["sel"], -1), ref["sel"][:, :B]), b
        wrong = np.where((res["ids"] != ref["ids"]).any(axis=1) | (res["dist"] != ref["dist"]).any(axis=1))[0]
        assert len(wrong) == 0, (b, wrong[:8].tolist(), [(res["ids"][q][:4].tolist(), ref["ids"][q][:4].tolist(), int(res["scored"][q]),
                                                           int(ref["metrics"][q, 2]), int(res["count"][q])) for q in wrong[:3]])
        assert np.array_equal(res["count"], ref["count"]) and np.array_equal(res["scored"], ref["metrics"][:, 2])
        assert not res["bad"].any()
    assert not o.unmodelled


@pytest.mark.parametrize("dense", [True, False], ids=["dense", "store"])
@pytest.mark.parametrize("handover", [True, False], ids=["handover", "second_launch"])
def test_tick_pipeline_matches_oracle(pkg, oracle, dense, handover):
    sc = make_scene(oracle, n=40000, d=64, T=8, D=1, m=12, lam=2, B=256, seed=5)
    Qb = sc["rng"].standard_normal((5, 200, 64)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])
        out = _pipeline(pkg, ctx, sc, Qb, dense, handover)
        assert ctx.unmodelled_queries() == 0
    _check(sc, Qb, out)


def test_tick_fallback_to_standalone_kernels(pkg, oracle, monkeypatch):
    """B > 256 (chunked scan + merge) does not fit the shared kernel: same API, stand-alone kernels, same results."""
    sc = make_scene(oracle, n=20000, d=32, T=6, D=1, m=12, lam=2, B=300, seed=6)
    Qb = sc["rng"].standard_normal((3, 64, 32)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])
        out = _pipeline(pkg, ctx, sc, Qb, dense=False, handover=True, expect_fused=False)
    _check(sc, Qb, out)
    monkeypatch.setenv("FSPANN_TICK_FUSE", "0")
    sc = make_scene(oracle, n=20000, d=32, T=6, D=1, m=12, lam=2, B=128, seed=7)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])
        out = _pipeline(pkg, ctx, sc, Qb, dense=True, handover=False, expect_fused=False)
    _check(sc, Qb, out)


@pytest.mark.parametrize("handover", [True, False], ids=["handover", "second_launch"])
def test_tick_finishes_handed_over_queries(pkg, oracle, handover, monkeypatch):
    """A tiny entry budget makes the bounded select hand most queries over (FSPANN_ROUTE_LAZY_CAP): with a hand-over buffer
    they stay PENDING until the workgroup that refines them runs the full select first; without one a second launch does."""
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "258")
    sc = make_scene(oracle, n=40000, d=16, T=10, D=1, m=12, lam=2, B=256, seed=23)
    Qb = sc["rng"].standard_normal((4, 96, 16)).astype(np.float32)
    for dense in (False, True):
        with _ctx(pkg, sc) as ctx:
            ctx.build_index(sc["X"])
            ctx.store_set(sc["X"])
            out = _pipeline(pkg, ctx, sc, Qb, dense, handover)
            info = ctx.last_route_info()
        _check(sc, Qb, out)
    assert info["lazy"]


@pytest.mark.parametrize("how", ["B300", "fuse_off"])
def test_tick_fallback_hands_over_on_consecutive_ticks(pkg, oracle, how, monkeypatch):
    """Regression (overflow-counter ping-pong): on the stand-alone fall-back path the tick's own Route plan and the
    fspann_route_dev it then calls both took the counters' turn, so consecutive fall-back ticks all counted into the counter
    nobody zeroes — stale overflow lists, then writes past the nq-sized list.  Several consecutive ticks whose bounded select
    hands queries over, on both ways into the fall-back (B > 256, FSPANN_TICK_FUSE=0)."""
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "258")
    if how == "fuse_off":
        monkeypatch.setenv("FSPANN_TICK_FUSE", "0")
    B = 300 if how == "B300" else 256
    sc = make_scene(oracle, n=40000, d=16, T=10, D=1, m=12, lam=2, B=B, seed=23)
    Qb = sc["rng"].standard_normal((6, 96, 16)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])
        out = _pipeline(pkg, ctx, sc, Qb, dense=False, handover=False, expect_fused=False)
        info = ctx.last_route_info()
        assert info["lazy"] and 0 < info["overflowed"] <= 96, info      # this call's list only, never an accumulated one
    _check(sc, Qb, out)


def test_tick_flags_treeified_queries_in_the_redo(pkg, oracle):
    """Degenerate hashCodes: every query is handed over AND its HashMap would treeify a bin -> the redo flags it (count -1),
    Refine returns nothing for it, the context counts it."""
    n = 30000
    sc = make_scene(oracle, n=n, d=16, T=8, D=1, m=12, lam=2, B=256, seed=24)
    o = sc["oracle"]
    jh = (np.arange(n) % 5).astype(np.int32)
    Qb = sc["rng"].standard_normal((2, 24, 16)).astype(np.float32)
    with _ctx(pkg, sc, jh) as ctx:
        for td in range(o.TD):
            ctx.set_index(td, **o.get_index(td))
        ctx.finalize()
        ctx.store_set(sc["X"])
        out = _pipeline(pkg, ctx, sc, Qb, dense=False, handover=True)
        assert ctx.unmodelled_queries() == 48
    for res in out:
        assert (res["sel_count"] == -1).all() and (res["count"] == 0).all()


def test_tick_partial_parts_and_errors(pkg, oracle):
    import torch
    sc = make_scene(oracle, n=5000, d=16, T=4, D=1, m=10, lam=2, B=64, seed=9)
    dev = torch.device("cuda", 0)
    Q = sc["rng"].standard_normal((32, 16)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        qd = torch.from_numpy(Q).to(dev)
        codes = torch.zeros((32, 4, 1), dtype=torch.int64, device=dev)
        with pytest.raises(pkg.FspannStateError, match="not finalized"):
            ctx.tick_dev(route=dict(nq=32, codes=codes.data_ptr(), limit=64, ids=codes.data_ptr(), count=codes.data_ptr()))
        ctx.build_index(sc["X"])
        ctx.tick_dev()                                         # nothing to do
        ctx.tick_dev(encode=dict(nq=32, q=qd.data_ptr(), codes=codes.data_ptr()))     # encode alone
        ctx.sync()
        assert np.array_equal(codes.cpu().numpy().view(np.uint64), sc["oracle"].encode(Q.astype(np.float64)))
        with pytest.raises(pkg.FspannStateError, match="store"):
            ctx.tick_dev(refine=dict(nq=32, q=qd.data_ptr(), B=64, ids=codes.data_ptr(), count=codes.data_ptr(), k=5, out_ids=codes.data_ptr(),
                                     out_dist=codes.data_ptr(), out_count=codes.data_ptr()))
        with pytest.raises(pkg.FspannArgumentError, match="go together"):
            ctx.tick_dev(refine=dict(nq=32, q=qd.data_ptr(), B=64, ids=codes.data_ptr(), count=codes.data_ptr(), k=5, out_ids=codes.data_ptr(),
                                     out_dist=codes.data_ptr(), out_count=codes.data_ptr(), cand=qd.data_ptr(), codes=codes.data_ptr()))


def test_route_after_a_tick_with_handed_over_queries(pkg, oracle, monkeypatch):
    """Regression: a tick's Route part hands queries over (they stay PENDING in the batch's hand-over buffer, the overflow list of
    that launch is never consumed) and the tick's Refine part prepares redo parameters.  A stand-alone fspann_route on the same
    context right after must start from an empty overflow list: it used to inherit the tick's list and ran the full select for
    another batch's query numbers over stale probe lists, overwriting correct results."""
    import torch
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "258")
    sc = make_scene(oracle, n=40000, d=16, T=10, D=1, m=12, lam=2, B=256, seed=23)
    o, p = sc["oracle"], sc["params"]
    B, TD = p["B"], p["T"]
    dev = torch.device("cuda", 0)
    F32 = pkg._native.F32
    Qs = sc["rng"].standard_normal((3, 96, 16)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])
        codes = [ctx.encode(q) for q in Qs]
        ctx.route(codes[2], limit=B, counters=False)                         # leaves batch C's probe lists in the workspace
        cd = [torch.from_numpy(c.view(np.int64)).to(dev) for c in codes]
        sel = torch.full((96, B), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(96, dtype=torch.int32, device=dev)
        hov = torch.zeros(ctx.route_handover_bytes(96), dtype=torch.uint8, device=dev)
        oi = torch.zeros((96, K), dtype=torch.int32, device=dev)
        od = torch.zeros((96, K), dtype=torch.float64, device=dev)
        oc = torch.zeros(96, dtype=torch.int32, device=dev)
        qd = torch.from_numpy(Qs[0]).to(dev)
        route = dict(nq=96, codes=cd[0].data_ptr(), limit=B, ids=sel.data_ptr(), count=cnt.data_ptr(), handover=hov.data_ptr())
        ctx.tick_dev(None, route, None)                                      # Route of batch A: some queries PENDING
        ctx.sync()
        assert (cnt.cpu().numpy() == -2).any()
        refine = dict(nq=96, q=qd.data_ptr(), B=B, ids=sel.data_ptr(), count=cnt.data_ptr(), k=K, out_ids=oi.data_ptr(), out_dist=od.data_ptr(),
                      out_count=oc.data_ptr(), codes=cd[0].data_ptr(), handover=hov.data_ptr())
        ctx.tick_dev(None, route, refine)                                    # Route again + Refine with the redo of A's PENDING queries
        ctx.sync()
        ref_a = o.search(Qs[0].astype(np.float64), K)
        assert np.array_equal(oi.cpu().numpy(), ref_a["ids"]) and np.array_equal(od.cpu().numpy(), ref_a["dist"])
        for b in (1, 2, 1):                                                  # stand-alone calls behind the ticks
            got = ctx.route(codes[b], limit=B, counters=False)
            ref = o.search(Qs[b].astype(np.float64), K)
            assert np.array_equal(got["count"], ref["sel_count"]), b
            assert np.array_equal(np.where(np.arange(B)[None] < got["count"][:, None], got["ids"][:, :B], -1), ref["sel"][:, :B]), b


@pytest.mark.parametrize("B", [200, 256, 400])
def test_front_launch_equals_encode_and_route(pkg, oracle, B):
    """fspann_tick_dev without a Refine part = front_kernel (encode of one batch + bounded select of another in ONE launch, small size
    classes allowed): the codes and the routed lists are exactly what fspann_encode / the oracle's Route give, with and without a
    hand-over buffer, also when queries overflow the class (second launch finishes them)."""
    import torch
    sc = make_scene(oracle, n=50000, d=32, T=16, D=1, m=14, lam=2, B=B, seed=41)
    o, p = sc["oracle"], sc["params"]
    TD, W = p["T"], 1
    dev = torch.device("cuda", 0)
    Qa = sc["rng"].standard_normal((300, 32)).astype(np.float32)
    Qb = sc["rng"].standard_normal((300, 32)).astype(np.float32)
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        codes_b_ref = o.encode(Qb.astype(np.float64))
        ref_b = o.search(Qb.astype(np.float64), K)
        qa = torch.from_numpy(Qa).to(dev)
        codes_a = torch.zeros((300, TD, W), dtype=torch.int64, device=dev)
        codes_b = torch.from_numpy(codes_b_ref.view(np.int64)).to(dev)
        bad = torch.zeros(300, dtype=torch.int32, device=dev)
        sel = torch.full((300, B), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(300, dtype=torch.int32, device=dev)
        ctx.tick_dev(encode=dict(nq=300, q=qa.data_ptr(), codes=codes_a.data_ptr(), bad=bad.data_ptr()),
                     route=dict(nq=300, codes=codes_b.data_ptr(), limit=B, ids=sel.data_ptr(), count=cnt.data_ptr()), refine=None)
        ctx.sync()
        assert ctx.last_tick_fused() and ctx.last_route_info()["lazy"]
        assert np.array_equal(codes_a.cpu().numpy().view(np.uint64), o.encode(Qa.astype(np.float64))) and not bad.cpu().numpy().any()
        c_h, s_h = cnt.cpu().numpy(), sel.cpu().numpy()
        assert np.array_equal(c_h, ref_b["sel_count"])
        assert np.array_equal(np.where(np.arange(B)[None] < c_h[:, None], s_h, -1), ref_b["sel"][:, :B])
