#!/usr/bin/env python3
"""Dev tool (no GPU needed): registers, scratch and static LDS of every kernel in the built libfspann_hip.so.
usage: python tools/kregs.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "fspann-query-system_amd", "libfspann_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so=SO):
    with tempfile.TemporaryDirectory() as td:
        import shutil
        shutil.copy(so, os.path.join(td, "lib.so"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", os.path.join(td, "lib.so")], check=True, capture_output=True, cwd=td)
        obj = [f for f in os.listdir(td) if "gfx950" in f][0]
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, obj)], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
            out.setdefault(cur, {})
        m = re.search(r"\.(private_segment_fixed_size|vgpr_count|sgpr_count|agpr_count|group_segment_fixed_size):\s+(\d+)", line)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
    names = [k for k in out if k.startswith("_Z")]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return {d: out[n] for n, d in zip(names, dem)}


if __name__ == "__main__":
    pats = sys.argv[1:]
    for name, v in sorted(kernels().items()):
        if pats and not any(p in name for p in pats):
            continue
        short = re.sub(r"\(.*", "", name)[:96]
        print(short.ljust(98), "vgpr", v.get("vgpr_count"), "agpr", v.get("agpr_count", 0), "sgpr", v.get("sgpr_count"),
              "scratch", v.get("private_segment_fixed_size"), "lds", v.get("group_segment_fixed_size"))
