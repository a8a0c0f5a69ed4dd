#!/bin/bash
# Refuses to start a GPU visit with a stale library: rebuilds first and stops on a compile error.  usage: tools/gpu_visit.sh TIMEOUT 'command'
set -e
cd "$(dirname "$0")/.."
python -c "import sys; sys.path.insert(0,'.'); from anyedit_amd import build; build.build_library(verbose=False)"
git rev-parse --short HEAD > .commit_id 2>/dev/null || true   # the box has no .git: tools/traffic.sh stamps its output with this
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
