#!/bin/bash
# r10a: the host cost of a policy-in-the-loop half-step, piece by piece (scripts/probe_host_cost.py), 512 and 1024 envs
tag=${1:-r10a}; out=gpurun_out/$tag; mkdir -p $out
python scripts/probe_host_cost.py 512 2000 > $out/host_cost_512.txt 2> $out/host_cost_512.err
python scripts/probe_host_cost.py 1024 2000 > $out/host_cost_1024.txt 2> $out/host_cost_1024.err
cat $out/host_cost_512.txt $out/host_cost_1024.txt; tail -3 $out/host_cost_512.err
