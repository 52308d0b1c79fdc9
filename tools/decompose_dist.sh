run() { python bench.py --workload $1 --steps 1000 --no-cpu-baseline --no-sub-records "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-4s %-90s %.2f us/step' % ('$1', ' '.join(sys.argv[1:])[:90], j['ms_per_step']*1e3))" "${@:2}"; }
for w in c2 c4; do
run $w
run $w --env-kw info_level=full
run $w --env-kw "disturbances={action: [{disturbance_func: white_noise, std: 0.001}]}"
run $w --env-kw "disturbances={action: [{disturbance_func: uniform, low: [-0.001], high: [0.001]}]}"
run $w --env-kw "disturbances={action: [{disturbance_func: step, magnitude: 0.001, step_offset: 2}]}"
run $w --env-kw "disturbances={observation: [{disturbance_func: white_noise, std: 0.01}]}"
run $w --env-kw "disturbances={observation: [{disturbance_func: step, magnitude: 0.001, step_offset: 2}]}"
done
