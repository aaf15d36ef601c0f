#!/bin/bash
# dev helper: rebuild the in-tree .so files (they travel with the snapshot), then run a command on a GPU box.
# usage: tools/gr.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/fspann_build.log 2>&1 || { tail -30 /tmp/fspann_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun "$@"
