"""GPU: the HIP path against the committed golden fixtures (no oracle needed at run time),
both through the flat C ABI (FspannContext) and through the operator mirror
(QueryTokenFactory / PartitionedIndexService / QueryServiceImpl) incl. the adaptive retry."""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, load_scene_inputs
from tests_digest import index_digest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_quickcheck_codes(pkg):
    g = np.load(os.path.join(HERE, "golden", "quickcheck.npz"))
    cfg = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=24, lambda_=2, dim=128)
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(g["alpha"], g["r"], g["omega"])
        codes, hs = ctx.encode(g["v"][None], want_hashes=True)
    assert np.array_equal(hs[0, 0], g["H"])
    assert np.array_equal(codes[0, 0], g["code"])


@pytest.mark.parametrize("name", GOLDEN)
def test_flat_abi_matches_golden(pkg, name):
    g, X = load_scene_inputs(name)
    T, D, m, lam, d, B, K = (int(g[k]) for k in ("T", "D", "m", "lam", "d", "B", "K"))
    cfg = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B,
                                 max_global_candidates=int(g["hard_cap"]))
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(g["alpha"], g["r"], g["omega"])
        ctx.set_id_meta(int(g["n"]))
        ctx.build_index(X)
        for td in range(T * D):
            assert np.array_equal(index_digest(ctx.get_index(td)), g["index_digest"][td]), td
        codes, hs = ctx.encode(g["Q"], want_hashes=True)
        assert np.array_equal(codes, g["codes"]) and np.array_equal(hs, g["hashes"])
        for probes in (5, 10):
            nq = g[f"route{probes}_ids"].shape[0]
            res = ctx.route(codes[:nq], probe_override=probes)
            cnt = g[f"route{probes}_count"]
            assert np.array_equal(res["count"], cnt) and np.array_equal(res["raw_seen"], g[f"route{probes}_raw"])
            for i in range(nq):
                assert np.array_equal(res["ids"][i, :cnt[i]], g[f"route{probes}_ids"][i, :cnt[i]])
                assert np.array_equal(res["score"][i, :cnt[i]], g[f"route{probes}_score"][i, :cnt[i]])


@pytest.mark.parametrize("name", GOLDEN)
def test_operator_mirror_matches_golden(pkg, name):
    """insert -> finalizeForSearch -> createToken -> search, like ForwardSecureANNSystem.runQueries
    does per query (FSA:622-748), checked against the oracle's QSI.search restatement."""
    from fspann_amd import operators as ops
    g, X = load_scene_inputs(name)
    T, D, m, lam, d, B, K, n = (int(g[k]) for k in ("T", "D", "m", "lam", "d", "B", "K", "n"))
    cfg = ops.SystemConfig(m=m, lambda_=lam, divisions=D, tables=T, seed=int(g["seed"]), refinementLimit=B,
                           maxGlobalCandidates=int(g["hard_cap"]), kVariants=(K,))
    host = ops.InMemoryHost()
    ops.GFunctionRegistry.reset()
    if n < ops.MIN_SAMPLE_SIZE:
        # the reference's ITs pre-initialise the static registry when fewer than 1000 points are indexed
        # (it/src/test/java/com/fspann/it/BaseUnifiedIT.java:66-78)
        ops.GFunctionRegistry.install(g["alpha"], g["r"], g["omega"], d, m, lam, int(g["seed"]), T, D)
    index = ops.PartitionedIndexService(host, cfg, host, host)
    try:
        for i in range(n):
            index.insert(str(i), X[i])      # ids are decimal ordinals (FSA:515)
        index.finalizeForSearch()
        # natively generated GFunctions == the oracle's (same SplitMix64 stream, same libm, exact projections)
        assert np.array_equal(ops.GFunctionRegistry.alpha, g["alpha"])
        assert np.array_equal(ops.GFunctionRegistry.omega, g["omega"])
        assert np.array_equal(ops.GFunctionRegistry.r, g["r"])
        tf = ops.QueryTokenFactory(host, host, cfg)
        qs = ops.QueryServiceImpl(index, host, host, tf, cfg)
        for qi, q in enumerate(g["Q"]):
            tok = tf.create(q, K)
            assert np.array_equal(tok.getBitCodes().reshape(T * D, -1), g["codes"][qi])
            res = qs.search(tok)
            cnt = int(g["search_count"][qi])
            assert [int(r.id) for r in res] == list(g["search_ids"][qi, :cnt])
            assert [r.distance for r in res] == list(g["search_dist"][qi, :cnt])     # bit-exact fp64
            mt = g["search_metrics"][qi]
            assert (qs.getLastCandTotal(), qs.getLastCandKept(), qs.getLastCandDecrypted(), qs.getLastReturned()) == \
                tuple(int(x) for x in mt[:4])
        # the batched mirror (GpuQueryServiceImpl.searchBatch): ONE fspann_route + ONE fspann_refine for all tokens, the adaptive
        # retry (QSI:327-337) as a second batch of exactly the short queries — per token the same lists, bit for bit; the metric
        # getters describe the last token
        toks = tf.createBatch(list(g["Q"]), K)
        many = qs.searchBatch(toks)
        assert len(many) == len(g["Q"]) > 1
        for qi, res in enumerate(many):
            cnt = int(g["search_count"][qi])
            assert [int(r.id) for r in res] == list(g["search_ids"][qi, :cnt]), qi
            assert [r.distance for r in res] == list(g["search_dist"][qi, :cnt]), qi
        mt = g["search_metrics"][len(g["Q"]) - 1]
        assert (qs.getLastCandTotal(), qs.getLastCandKept(), qs.getLastCandDecrypted(), qs.getLastReturned()) == tuple(int(x) for x in mt[:4])
        assert qs.searchBatch([None, toks[0]])[0] == [] and [int(r.id) for r in qs.searchBatch([None, toks[0]])[1]] == list(g["search_ids"][0, :int(g["search_count"][0])])
    finally:
        if index.ctx is not None:
            index.ctx.close()
        ops.GFunctionRegistry.reset()
