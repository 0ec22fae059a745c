#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; nproc; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" ; timeout 600 python scripts/probe_cpu_scaling.py; MVO_PIN=0 timeout 300 python scripts/probe_cpu_scaling.py 64 256
