#!/bin/bash
# One GPU-box session that produces everything committed under profiles/<tag>/ (run via gpurun).
# usage: tools/round_measure.sh <tag>
set -u
tag="${1:-r01}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd "$root"
echo "== tests"; tools/run_gpu_tests.sh > "$out/gpu_tests.log" 2>&1; tail -4 "$out/gpu_tests.log"
echo "== smoke"; python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench c4"; python bench.py --python-loop --fused > "$out/bench_c4.json" 2> "$out/bench_c4.err"; cat "$out/bench_c4.json"
echo "== bench c4 without episode-return tracking (bare 125-byte step)"; python bench.py --no-episode-stats --no-cpu-baseline > "$out/bench_c4_bare.json" 2>/dev/null; cat "$out/bench_c4_bare.json"
echo "== bench c3"; python bench.py --workload c3 --envs-per-gpu 262144 --steps 5000 --fused > "$out/bench_c3.json" 2>/dev/null; cat "$out/bench_c3.json"
echo "== bench c2"; python bench.py --workload c2 --envs-per-gpu 65536 --steps 10000 --fused > "$out/bench_c2.json" 2>/dev/null; cat "$out/bench_c2.json"
echo "== N sweep"; for e in 10 12 14 16 18 20 22 23; do python bench.py --envs-per-gpu $((1<<e)) --steps 2000 --no-cpu-baseline --no-episode-stats 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('N=2^$e  %.3e env-steps/s  %.2f us/step  frac %.3f' % (j['value'], j['roofline']['kernel_us_per_launch'], j['roofline']['frac']))"; done | tee "$out/n_sweep.txt"
echo "== precise mode"; python bench.py --no-cpu-baseline --no-episode-stats --steps 1500 --env-kw compensated_state=true > "$out/bench_c4_precise.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_c4_precise.json')); print('precise mode %.3e env-steps/s %.2f us' % (j['value'], j['roofline']['kernel_us_per_launch']))"
echo "== disturbed step (white action + uniform force + white obs noise, drawn in the kernel)"; python bench.py --no-cpu-baseline --steps 1000 --env-kw 'disturbances={action: [{disturbance_func: white_noise, std: 0.002}], dynamics: [{disturbance_func: uniform, low: [-0.05, -0.05, -0.05], high: [0.05, 0.05, 0.05]}], observation: [{disturbance_func: white_noise, std: 0.01}]}' --env-kw info_level=full > "$out/bench_c4_disturbed.json" 2>/dev/null; python -c "import json; j=json.load(open('$out/bench_c4_disturbed.json')); print('disturbed %.3e env-steps/s %.2f us' % (j['value'], j['roofline']['kernel_us_per_launch']))"
echo "== linearize"; python tools/bench_linearize.py 3 2>/dev/null | tail -1 | tee "$out/bench_linearize.json"; python tools/bench_linearize.py 2 2>/dev/null | tail -1 | tee -a "$out/bench_linearize.json"
echo "== normalize"; python tools/bench_normalize.py 12 2>/dev/null | tail -1 | tee "$out/bench_normalize.json"
echo "== gae"; python tools/bench_gae.py 2>/dev/null | tail -1 | tee "$out/bench_gae.json"
echo "== 2 ranks sharing this GPU over gloo (rehearsal of the N>1 path)"
SCG_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 600 --warmup 50 --envs-per-gpu 524288 > "$out/bench_2rank_rehearsal.json" 2> "$out/bench_2rank.err"; tail -1 "$out/bench_2rank_rehearsal.json"
echo "== rocprof kernel trace"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$root/bench.py" --steps 500 --warmup 50 --no-cpu-baseline --no-bare > "$out/prof.log" 2>&1 )
find "$out/prof" -name "*kernel_stats.csv" -exec cp {} "$out/c4_kernel_stats.csv" \;
head -3 "$out/c4_kernel_stats.csv" | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof2" -- python3 "$root/tools/bench_linearize.py" 3 > "$out/prof2.log" 2>&1 )
find "$out/prof2" -name "*kernel_stats.csv" -exec cp {} "$out/linearize_kernel_stats.csv" \;
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof3" -- python3 "$root/tools/bench_normalize.py" 12 > "$out/prof3.log" 2>&1 )
find "$out/prof3" -name "*kernel_stats.csv" -exec cp {} "$out/normalize_kernel_stats.csv" \;
rm -rf "$out/prof2" "$out/prof3"
echo "== pmc (bare C4 step, all counter groups)"; profiles/pmc.sh "$out/pmc" --no-episode-stats > "$out/pmc_summary.txt" 2>&1; cat "$out/pmc_summary.txt"
echo "== pmc (default workload: + episode-return tracking; traffic counters only)"; PMC_TRAFFIC_ONLY=1 profiles/pmc.sh "$out/pmc_c5" > "$out/pmc_summary_c5shard.txt" 2>&1; cat "$out/pmc_summary_c5shard.txt"
rm -rf "$out"/pmc_c5/pass*
rm -rf "$out/prof" "$out"/pmc/pass*
