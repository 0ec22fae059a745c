#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py tests/test_env_surface_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tail -4
