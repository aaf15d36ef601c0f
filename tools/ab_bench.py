"""Dev tool: bench.py against a VARIANT build of the library (AB_LIB=<path to .so>), for same-box A/B runs — boxes differ by
+-0.5 us for the same binary, so variants are compared inside ONE gpurun call.  Never used by tests or the driver."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
if os.environ.get("AB_LIB"):
    pkg._native._SO = os.path.abspath(os.environ["AB_LIB"])
import bench  # noqa: E402

bench.main()
