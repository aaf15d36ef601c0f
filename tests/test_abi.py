"""CPU: the C-ABI library loads and exports every symbol include/fspann.h declares; error
mapping and host-side argument checks that need no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "fspann.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fspann_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    pkg._native.build()
    L = pkg._native.lib()
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/fspann.h but not exported"
    # and the binding table covers the header (no silent drift)
    assert set(pkg._native.exported_symbols()) == set(syms)


def test_version_and_error_string(pkg):
    L = pkg._native.lib()
    assert b"gfx950" in L.fspann_version()
    assert isinstance(L.fspann_last_error(), bytes)


def test_null_and_argument_errors_without_gpu(pkg):
    N = pkg._native
    L = N.lib()
    h = C.c_void_p()
    assert L.fspann_ctx_create(0, None, C.byref(h)) == N.E_NULL
    bad = N.Cfg(0, 1, 8, 2, 16, 0, 0, -1, 0, 0, 0, 0)
    assert L.fspann_ctx_create(0, C.byref(bad), C.byref(h)) == N.E_ARG
    assert b"> 0" in L.fspann_last_error()
    bad = N.Cfg(2, 1, 8, 40, 16, 0, 0, -1, 0, 0, 0, 0)          # lambda > 32
    assert L.fspann_ctx_create(0, C.byref(bad), C.byref(h)) == N.E_ARG
    assert L.fspann_sync(None) == N.E_NULL
    assert L.fspann_encode(None, 1, None, 0, None, None) == N.E_NULL
    with pytest.raises(N.FspannNullError):
        N.check(N.E_NULL)
    with pytest.raises(ValueError):      # IllegalArgumentException is a ValueError
        N.check(N.E_ARG)


def test_no_gpu_means_device_error_not_fallback(pkg):
    """The product path must fail loudly without a GPU: there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.FspannDeviceError):
        pkg.FspannContext(pkg.PaperRuntimeConfig(tables=2, divisions=1, m=8, lambda_=2, dim=16), 0)


def test_product_never_imports_oracle():
    pkgdir = os.path.join(ROOT, "fspann-query-system_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
                assert "fspann_oracle" not in txt, f


def test_operator_surface_host_checks(pkg):
    from fspann_amd import operators as ops
    N = pkg._native
    host = ops.InMemoryHost()
    cfg = ops.SystemConfig(m=8, lambda_=2, divisions=1, tables=2)
    ops.GFunctionRegistry.reset()
    tf = ops.QueryTokenFactory(host, host, cfg)
    with pytest.raises(N.FspannNullError):           # Objects.requireNonNull(vec)
        tf.create(None, 5)
    with pytest.raises(N.FspannArgumentError):       # topK must be > 0
        tf.create(np.zeros(4), 0)
    with pytest.raises(N.FspannStateError):          # registry not initialised
        tf.create(np.zeros(4), 5)
    with pytest.raises(N.FspannNullError):
        ops.PartitionedIndexService(None, cfg, host, host)
    idx = ops.PartitionedIndexService(host, cfg, host, host)
    tok = ops.QueryToken(np.zeros((2, 1, 1), np.uint64), b"0" * 12, b"", 5, 2, 4, 1, 2, "dim_4_v1")
    with pytest.raises(N.FspannStateError):          # "Index not finalized" (PIS:594)
        idx.lookupCandidatesWithScores(tok)
    with pytest.raises(N.FspannStateError):          # cannot finalize with < 1000 samples and no registry
        idx.finalizeForSearch()
    with pytest.raises(N.FspannNullError):
        idx.insert(None, np.zeros(4))
    qs = ops.QueryServiceImpl(idx, host, host, tf, cfg)
    assert qs.search(None) == []                     # null token -> empty list (QSI:102)
    with pytest.raises(N.FspannArgumentError):
        tf.derive(tok, 0)
    t2 = tf.derive(tok, 7)
    assert t2.getTopK() == 7 and np.array_equal(t2.getBitCodes(), tok.getBitCodes())
    assert ops._java_string_hash("hello") == 99162322 and ops._java_string_hash("1000000") == 1958013297


def test_jni_shim_binds_every_header_symbol_and_type_checks():
    """jni/fspann_jni.cpp + FspannNative.java are generated from the header (tools/gen_jni.py): one native method per entry
    point.  No JDK exists here, so the shim is type-checked against tests/jni_stub/jni.h (syntax check only, never linked)."""
    import subprocess
    bound = open(os.path.join(ROOT, "jni", "bound_symbols.txt")).read().split()
    assert sorted(bound) == header_symbols()
    java = open(os.path.join(ROOT, "java", "com", "fspann", "gpu", "FspannNative.java")).read()
    assert java.count("public static native") == len(bound)
    # regenerating gives the committed files (the header and the shim cannot drift apart)
    before = {f: open(os.path.join(ROOT, f)).read() for f in ("jni/fspann_jni.cpp", "java/com/fspann/gpu/FspannNative.java")}
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_jni.py")], stdout=subprocess.DEVNULL)
    for f, txt in before.items():
        assert open(os.path.join(ROOT, f)).read() == txt, f + " is stale: run tools/gen_jni.py"
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "fspann_jni.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


def test_no_kernel_uses_scratch_memory(pkg, tmp_path):
    """Any scratch use (a spilled register, a by-value struct copied to the stack, a real device-function call) makes the
    runtime manage scratch per dispatch — measured ~2x slower for every kernel of the queue.  The kernel
    metadata of the built code object must therefore say private_segment_fixed_size = 0 for every kernel, and the kernels
    that share a CU with others must keep the register budget their launch bounds promise."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not in this image")
    so = str(tmp_path / "libfspann_hip.so")
    shutil.copy(pkg._native._SO, so)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=str(tmp_path))
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert len(objs) == 1, objs
    notes = subprocess.run([readelf, "--notes", str(tmp_path / objs[0])], check=True, capture_output=True, text=True).stdout
    kernels = {}
    name = None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m and m.group(1).startswith("_Z"):
            name = m.group(1)
            kernels[name] = {}
        m = re.search(r"\.(private_segment_fixed_size|vgpr_count|group_segment_fixed_size):\s+(\d+)", line)
        if m and name:
            kernels[name][m.group(1)] = int(m.group(2))
    assert len(kernels) >= 40, len(kernels)
    spilled = {k: v["private_segment_fixed_size"] for k, v in kernels.items() if v.get("private_segment_fixed_size", 0) != 0}
    assert not spilled, spilled
    # four workgroups of 256 threads per CU = 128 registers per lane
    for frag in ("route_select_lazy_kernel", "tick_kernel", "20refine_stream_kernelIffLi32ELb0E"):
        hit = [k for k in kernels if frag in k]
        assert hit, frag
        for k in hit:
            assert kernels[k]["vgpr_count"] <= 128, (k, kernels[k])
