#!/bin/bash
# Quick check on the GPU box after a kernel change: the GPU tests, smoke(), the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5 || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 bench.py 2>/dev/null | tail -1 | cut -c1-400
