#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py -x -q -k "one_launch or step_n" 2>&1 | tail -8
