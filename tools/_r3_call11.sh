#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_apc_snapshots.py tests/test_original_chips.py -m gpu -x -q 2>&1 | tail -25
