"""DESIGN.md section 10's table from the bench lines of one measurement pass: python scripts/measured_table.py profiles/r08z_"""
import glob
import json
import os
import sys

ROWS = [("tower_bench.json", "**TowerBuilding 1024 envs x 1, 128x128 (BASELINE configs[1])**"),
        ("tower_bench_driver_style.json", "same, the driver's form (`--gpus 1 --steps 20 --warmup 5`)"),
        ("tower_no_overlap_bench.json", "same, passes not overlapped (`--pass-overlap off`: round 6's headline until `r10zb`)"),
        ("tower_8_ticks_per_call_bench.json", "same, 8 ticks per call (`--batch 8`)"),
        ("tower_no_multitick_bench.json", "same, k step launches + k raster launches per call (`MV_STEP_TICKS=0 --batch 8`)"),
        ("tower_512x4_bench.json", "TowerBuilding 512 envs x 4 agents (configs[3])"),
        ("tower_512_bench.json", "TowerBuilding 512 envs x 1 (overlapped passes)"), ("tower_512_no_overlap_bench.json", "same, passes not overlapped"),
        ("tower_4096_bench.json", "TowerBuilding 4096 envs x 1"),
        ("obstacles_hard_512_bench.json", "ObstaclesHard 512 envs x 1 (one GPU's share of configs[2]), overlapped passes"),
        ("obstacles_hard_512_no_overlap_bench.json", "same, passes not overlapped (`--pass-overlap off`)"),
        ("obstacles_hard_1024_bench.json", "ObstaclesHard 1024 x 1 (overlapped passes)"),
        ("obstacles_hard_1024_2_cores_bench.json", "same on two host cores"), ("obstacles_hard_512_2_cores_bench.json", "ObstaclesHard 512 x 1 on two host cores"),
        ("Collect_bench.json", "Collect 1024 x 1 (overlapped passes)"), ("Collect_no_overlap_bench.json", "same, passes not overlapped"),
        ("Collect_device_generator_bench.json", "same, episodes drawn on the device (`MV_COLLECT_DEVICE_GEN=1`)"),
        ("Collect_2_cores_bench.json", "same on two host cores (`taskset -c 0,1`: the rule draws on the device there)"),
        ("Collect_2_cores_host_2_threads_bench.json", "same on two host cores as before round 6's last changes (host feeder, two threads)"),
        ("Rearrange_bench.json", "Rearrange 1024 x 1 (overlapped passes)"), ("Rearrange_no_overlap_bench.json", "same, passes not overlapped"),
        ("Sokoban_bench.json", "Sokoban 1024 x 1 (synthetic Boxoban-format levels; overlapped passes)"),
        ("HexMemory_bench.json", "HexMemory 1024 x 1 (overlapped passes)"), ("HexMemory_no_overlap_bench.json", "same, passes not overlapped"),
        ("HexExplore_bench.json", "HexExplore 1024 x 1 (overlapped passes)"), ("HexExplore_no_overlap_bench.json", "same, passes not overlapped"),
        ("Empty_bench.json", "Empty 1024 x 1"), ("Empty_800_steps_bench.json", "Empty 1024 x 1, 800 steps"),
        ("mixed_64_bench.json", "**Mixed: all eight `megaverse8` scenarios round-robin, 1024 x 1, 64x64 (one GPU's share of configs[4])**"),
        ("mixed4_64_bench.json", "Mixed4: TowerBuilding, ObstaclesEasy, ObstaclesHard, Collect round-robin, 1024 x 1, 64x64 (BASELINE.md 3 row 5)"),
        ("mixed_64_2_cores_bench.json", "Mixed 64x64 on two host cores"), ("mixed4_64_2_cores_bench.json", "Mixed4 64x64 on two host cores"),
        ("mixed_128_bench.json", "Mixed, 1024 envs, 128x128"),
        ("tower_128x72_bench.json", "TowerBuilding 1024 x 1, 128x72 (the reference's own obs size)"),
        ("tower_64x64_bench.json", "TowerBuilding 1024 x 1, 64x64"),
        ("Collect_128x72_bench.json", "Collect 1024 x 1, 128x72"),
        ("tower_exact_pixels_bench.json", "same as the headline, exact pixels (`--pixels exact`)"),
        ("tower_single_bit_bench.json", "same, the reference benchmark's single-bit policy (`--policy single-bit`)"),
        ("tower_planar_off_bench.json", "same, tiles not classified (`MV_PLANAR=0`)"),
        ("tower_step_pipe_bench.json", "same, the two-wave step kernels (`MV_STEP_PIPE=1`; the rule takes one wave at 1024 envs)"),
        ("tower_256_bench.json", "TowerBuilding 256 envs x 1 (two-wave step kernels by the rule)"),
        ("tower_512_one_wave_step_bench.json", "TowerBuilding 512 envs x 1, one-wave step kernels (`MV_STEP_PIPE=0`)"),
        ("obstacles_hard_512_one_wave_step_bench.json", "ObstaclesHard 512 envs x 1, one-wave step kernels (`MV_STEP_PIPE=0`)"),
        ("Empty_one_wave_step_bench.json", "Empty 1024 x 1, one-wave step kernels (`MV_STEP_PIPE=0`)"),
        ("tower_normal_priority_bench.json", "same, simulation stream at default priority (`MV_SIM_PRIORITY=normal`)"),
        ("obstacles_hard_512_normal_priority_bench.json", "ObstaclesHard 512, overlapped passes, `MV_SIM_PRIORITY=normal`")]


def main():
    prefix = sys.argv[1]
    print("| config | obs/s | ms/step | ticks/call | raster | step | single | serial | loop | 2 halves |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, label in ROWS:
        p = prefix + name
        if not os.path.exists(p):
            continue
        try:
            d = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            continue
        leg = lambda k: ("%.2f" % (d[k] / 1e6)) if isinstance(d.get(k), (int, float)) else ""
        print("| %s | %.2f M | %.4f | %s | %.3f | %.3f | %s | %s | %s | %s |" % (
            label, d["value"] / 1e6, d["ms_per_step"], d["config"].get("ticks_per_call", ""), d["roofline"]["avg_launch_ms"], d["roofline_physics"]["avg_launch_ms"],
            leg("value_single_step"), leg("value_unpipelined"), leg("value_closed_loop"), leg("value_closed_loop_double_buffered")))


if __name__ == "__main__":
    main()
