#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python -m pytest tests/test_apc_snapshots.py -m gpu -q > /tmp/t.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error|assert" /tmp/t.txt | tail -12
