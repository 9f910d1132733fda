#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_jit.py tests/test_sharding.py -m gpu -x -q > /tmp/t.txt 2>&1; echo "rc=$?"; grep -E "passed|failed|error" /tmp/t.txt | tail -5
