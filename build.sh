#!/bin/bash
# Build libfspann_hip.so (gfx950) and the CPU oracle. Usage: ./build.sh
cd "$(dirname "$0")" && python -c "import __graft_entry__ as g; g.build()"
