"""CPU: the product's pure-host code under sanitizers (GPU AddressSanitizer / XNACK runs do not exist on the pool, so the host
code lives in headers a plain g++ can build):

  * host/pointstore.hpp — readers opening batches while one thread rotates + migrates every record and another re-seals records
    under the current version (tests/cpp/pointstore_stress.cpp), under ThreadSanitizer and under ASan + UBSan;
  * the oracle's C restatement under ASan + UBSan (oracle/Makefile: liboracle_asan.so) on the Java-semantics KATs and one search.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "pointstore_stress.cpp")


def _build(tmp, name, flags):
    out = str(tmp / name)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-Wall", "-Wextra", *flags, "-o", out, SRC, "-ldl"])
    return out


def _run(exe, env=None, args=("3000", "24", "3")):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    if r.returncode == 77:
        pytest.skip("libcrypto (OpenSSL 3) not present")
    return r


def test_pointstore_rotate_migrate_readers_plain(tmp_path):
    r = _run(_build(tmp_path, "ps_plain", ["-O2"]), args=("20000", "32", "4"))
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "failed 0 wrong 0" in r.stdout


def test_pointstore_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "ps_tsan", ["-fsanitize=thread"])
    # libcrypto is not instrumented: races INSIDE it would be false positives; everything in the product's header is instrumented
    r = _run(exe, env={"TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0 exitcode=66"})
    assert "failed 0 wrong 0" in r.stdout, r.stdout + r.stderr[-3000:]
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_pointstore_under_address_and_ub_sanitizers(tmp_path):
    exe = _build(tmp_path, "ps_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = _run(exe, env={"ASAN_OPTIONS": "detect_leaks=0"})
    assert r.returncode == 0 and "failed 0 wrong 0" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """liboracle_asan.so (oracle/Makefile) through the same Python binding, in a child process with the ASan runtime preloaded:
    HashMap KATs with tree bins, a small build + search."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from oracle import oracle as O
import ctypes as C
O._LIB = None
O._SO = {os.path.join(ROOT, "oracle", "liboracle_asan.so")!r}
L = O.lib()
rng = np.random.default_rng(2)
for it in range(8):
    n = int(rng.integers(100, 900))
    sp = (rng.integers(0, 3, n) + 64 * rng.integers(0, 1 << 14, n)).astype(np.uint32)
    h = (sp ^ (sp >> 16)).view(np.int32)
    keys = rng.permutation(1 << 20)[:n].astype(np.int32)
    out, cap, tree, unm = O.hashmap_order_ex(64, keys, h, True)
    assert sorted(out) == sorted(keys) and tree and not unm
X = rng.standard_normal((3000, 16))
a, r, w = O.registry_init(X[:1000], 8, 13, 3, 2)
o = O.Oracle(3, 2, 8, 2, 16, refinement_limit=64)
o.set_gfunctions(a, r, w); o.set_id_meta(3000); o.set_store(X); o.build_index(X)
res = o.search(rng.standard_normal((12, 16)), 5)
assert res["ids"].shape == (12, 5)
print("ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
