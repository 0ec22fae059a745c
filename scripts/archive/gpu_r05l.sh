#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --steps 30000 --warmup 100"
for a in "" "--scenario ObstaclesHard" "--agents 4 --envs-per-gpu 512" "--scenario Empty"; do echo "$a: $($B $a 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), d['checksum'])"; tail -1 /tmp/err.txt | cut -c1-200)"; done
