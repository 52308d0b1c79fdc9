#!/bin/bash
# Round 4: the observation noise drawn behind the integration (SCG_OBS_NOISE_LATE=1: 79 instead of 89 VGPRs in the disturbance tier,
# 6 instead of 5 waves per SIMD) against drawing it with the other disturbances at the top (=0), and 7 waves demanded of the tier
# (SCG_DIST_WAVES=7: 72 VGPRs + 36 B of scratch); arms are lib/variants builds, interleaved on one box (tools/tune.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04/obsnoise; mkdir -p $out
T="timeout -k 10 500 python tools/tune.py --rounds 2 --arms ${ARMS:-obs_early,obs_late,dist7}"
run() { name="$1"; shift; echo "== $name"; $T "$@" 2>&1 | tee "$out/tune_$name.log" | grep -A4 "^----" | tail -3; }
run obs_noise --steps 600 -- --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}"
run act_noise --steps 600 -- --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}]}"
run all_three --steps 600 -- --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}], dynamics: [{disturbance_func: uniform, low: [-0.01, -0.01, -0.01], high: [0.01, 0.01, 0.01]}], observation: [{disturbance_func: white_noise, std: 0.01}]}"
run c3_obs_noise --steps 1500 -- --workload c3 --envs-per-gpu 262144 --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}"
run headline --steps 1000
