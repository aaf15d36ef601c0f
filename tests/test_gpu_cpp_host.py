"""GPU: the C++ host mirror of the operator surface (tests/cpp/fspann_host.hpp) — compiled with g++ against
libfspann_hip.so and driven like ForwardSecureANNSystem drives the Java operators; results are compared with
the golden fixtures (oracle restatement of QSI.search, incl. the adaptive retry and the getLast* metrics)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN, load_scene_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fspann-query-system_amd")


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory, pkg):
    pkg._native.build()
    out = str(tmp_path_factory.mktemp("cpp") / "host_mirror_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", out, os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L" + PKG, "-lfspann_hip", "-Wl,-rpath," + PKG])
    return out


@pytest.mark.parametrize("name", GOLDEN)
def test_cpp_operator_mirror(host_binary, tmp_path, name):
    g, X = load_scene_inputs(name)
    n, d, T, D, m, lam, B, K = (int(g[k]) for k in ("n", "d", "T", "D", "m", "lam", "B", "K"))
    Q = np.ascontiguousarray(g["Q"], np.float64)
    pre = 1 if n < 1000 else 0    # the reference's ITs pre-initialise the registry below MIN_SAMPLE_SIZE
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<12q", n, d, T, D, m, lam, B, K, int(g["seed"]), len(Q), int(g["hard_cap"]), pre))
        f.write(np.ascontiguousarray(X, np.float64).tobytes())
        f.write(Q.tobytes())
        if pre:
            for k in ("alpha", "r", "omega"):
                f.write(np.ascontiguousarray(g[k], np.float64).tobytes())
    r = subprocess.run([host_binary, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(outp, "rb").read()
    rec = 4 + 4 * K + 8 * K + 16
    assert len(raw) == rec * len(Q)
    for qi in range(len(Q)):
        b = raw[qi * rec:(qi + 1) * rec]
        cnt = struct.unpack_from("<i", b, 0)[0]
        ids = np.frombuffer(b, np.int32, K, 4)
        dist = np.frombuffer(b, np.float64, K, 4 + 4 * K)
        met = np.frombuffer(b, np.int32, 4, 4 + 12 * K)
        assert cnt == int(g["search_count"][qi])
        assert np.array_equal(ids[:cnt], g["search_ids"][qi, :cnt])
        assert np.array_equal(dist[:cnt], g["search_dist"][qi, :cnt])      # bit-exact fp64
        assert np.array_equal(met, g["search_metrics"][qi, :4])
