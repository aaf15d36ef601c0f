#!/usr/bin/env python3
"""Compare a JVM dump (java/com/fspann/gpu/GoldenDumper.java) with the CPU oracle: the step that turns
"parity unpinned" into pinned.  Usage: python tests/golden/compare_jvm_dump.py jvm_golden.txt
Without an argument it prints the oracle's side (what the JVM is expected to print)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def hexd(x):
    return format(int(np.float64(x).view(np.uint64)), "x")


def oracle_lines():
    out = []
    for seed in (0, 13, 42, 12345):
        longs = O.splitmix_stream(seed, 4)
        dbls = O.splitmix_doubles(seed, 2)
        out.append("splitmix %d " % seed + " ".join(format(v, "x") for v in longs) + " " + " ".join(hexd(d) for d in dbls))
    v = np.arange(128) * 0.01
    alpha, r, w = O.build_random_g(128, 24, 1.0, 12345)
    out.append("gauss " + " ".join(hexd(a) for a in alpha[0, :8]))          # <= 1 ulp expected, compared loosely
    out.append("quick_r " + " ".join(hexd(x) for x in r))
    out.append("quick_H " + " ".join(str(int(h)) for h in O.H(v, alpha, r, w)))
    out.append("quick_C " + " ".join(str(int(np.int64(c))) for c in O.Ccode(v, alpha, r, w, 2).view(np.int64)))
    out.append("hashcode " + " ".join(str(O.string_hash(str(o))) for o in (0, 9, 10, 999, 1000, 123456, 999999, 1000000)))
    for cap in (4, 16, 64, 32768):
        keys, seen = [], {}
        for i in range(200):
            k = (i * 37) % 1009
            if k not in seen:
                seen[k] = len(keys)
                keys.append(k)
        order, _, unm = O.hashmap_order(cap, np.array(keys, np.int32), np.array([O.string_hash(str(k)) for k in keys], np.int32))
        out.append("hashmap %d " % cap + " ".join(str(int(k)) for k in order) + (" UNMODELLED" if unm else ""))
    for b in (5, 41):                                                       # tree bins: see GoldenDumper.java
        keys, crowded, i = [], 0, 0
        while crowded < 40:
            h = O.string_hash(str(i)) & 0xFFFFFFFF
            if ((h ^ (h >> 16)) & 63) == b:
                keys.append(i)
                crowded += 1
            if i % 97 == 0:
                keys.append(1000003 + i)
            i += 1
        order, _, tree, unm = O.hashmap_order_ex(64, np.array(keys, np.int32), np.array([O.string_hash(str(k)) for k in keys], np.int32), True)
        assert tree and not unm
        out.append("hashtree %d " % b + " ".join(str(int(k)) for k in order))
    for ops in ([5, -1, 3, 3, -1, -1], [5, -1, 3, 4, -1, 4, -1, -1], [5, -1, 3, 4, -1, 2, -1, -1], [7, 7, 7, -1, 7, -1, -1, -1]):
        out.append("pq " + " ".join(str(int(x)) for x in O.pq_trace(ops)))
    out.append("key %d %d %d" % (O.compute_key(np.array([0b1011], np.uint64)), O.compute_key(np.array([1 << 63, 1], np.uint64)),
                                 O.hamming(np.array([0xFF, 1], np.uint64), np.array([0x0F, 0], np.uint64))))
    o = O.Oracle(1, 1, 6, 1, 1)
    n = 150
    o.set_id_meta(n)
    codes = np.array([[(i * 73) & 0x3F] for i in range(n)], np.uint64).reshape(n, 1, 1)
    O.lib().orc_build_index(o._h, __import__("ctypes").c_int64(n), np.arange(n, dtype=np.int32).ctypes.data_as(__import__("ctypes").c_void_p),
                            codes.ctypes.data_as(__import__("ctypes").c_void_p))
    ix = o.get_index(0)
    for p in range(len(ix["min_key"])):
        ids = ix["ids"][ix["id_off"][p]:ix["id_off"][p + 1]]
        out.append("build %d %d %d %s" % (ix["min_key"][p], ix["max_key"][p], int(np.int64(ix["rep"][p, 0])), ",".join(str(int(i)) for i in ids)))
    out.extend(route_lines())
    out.append("cast " + " ".join(str(O.d2i(float(np.floor(x)))) for x in (1e300, -1e300, float("nan"), 2147483647.5, -2147483648.5, -0.5, 3.99)))
    return out


def route_scene():
    """The `route` case of GoldenDumper.java, number for number: three tables of 24 000 decimal ids with 16-bit codes, twelve ids of
    ONE bin of the 2 048-slot table placed beside query QA (six in table 0, six one bit away in table 1)."""
    import ctypes as C
    N, T = 24000, 3
    A, B = (7919, 104729, 1299709), (17, 4242, 31337)
    QA, QB = (0x1234, 0x0F0F, 0x5555), (0x8001, 0x7FFE, 0x00FF)
    codes = np.zeros((N, T, 1), np.uint64)
    i = np.arange(N, dtype=np.int64)
    for t in range(T):
        codes[:, t, 0] = ((i * A[t] + B[t]) % 65521) & 0xFFFF
    h = O.decimal_hashes(N).view(np.uint32)
    crowd = np.flatnonzero(((h ^ (h >> 16)) & 2047) == 312)[:12]
    assert len(crowd) == 12
    codes[crowd[:6], 0, 0] = QA[0]
    codes[crowd[6:], 1, 0] = QA[1] ^ 0x8000
    q = np.array([[[c] for c in QA], [[c] for c in QB]], np.uint64)
    return N, T, codes, q, crowd, C


def route_lines(with_flags=False):
    N, T, codes, q, crowd, C = route_scene()
    out, flags = [], []
    for hard_cap, probes in ((1500, 5), (100, 5), (1500, 10), (300, 5)):
        o = O.Oracle(T, 1, 8, 2, 1, max_global_candidates=hard_cap, refinement_limit=min(hard_cap, 64))
        o.set_id_meta(N)
        O.lib().orc_build_index(o._h, C.c_int64(N), np.arange(N, dtype=np.int32).ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p))
        ids, score, count, raw = o.route(q, probe_override=probes)
        tree = o.route_treeified(q, probe_override=probes)
        assert not o.unmodelled
        for qi, qn in enumerate(("QA", "QB")):
            n = int(count[qi])
            out.append("route %d %d %s %d %d " % (hard_cap, probes, qn, n, int(raw[qi])) +
                       " ".join("%d:%d" % (int(ids[qi, k]), int(score[qi, k])) for k in range(n)))
            flags.append((hard_cap, probes, qn, bool(tree[qi]), n))
    return (out, flags) if with_flags else out


def main():
    mine = oracle_lines()
    if len(sys.argv) < 2:
        print("\n".join(mine))
        return 0
    theirs = [l.rstrip("\n") for l in open(sys.argv[1]) if l.strip()]
    bad = 0
    for a, b in zip(mine, theirs):
        if a == b:
            continue
        if a.startswith("gauss") and b.startswith("gauss"):
            ua = [int(x, 16) for x in a.split()[1:]]
            ub = [int(x, 16) for x in b.split()[1:]]
            if all(abs(x - y) <= 1 for x, y in zip(ua, ub)):
                print("gauss: within 1 ulp (Math.log/cos are not bit-portable) - ok")
                continue
        bad += 1
        print("MISMATCH\n  oracle:", a[:200], "\n  jvm   :", b[:200])
    if len(mine) != len(theirs):
        bad += 1
        print("line count differs: oracle %d, jvm %d" % (len(mine), len(theirs)))
    print("PINNED: oracle == JVM" if bad == 0 else "%d mismatch(es)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
