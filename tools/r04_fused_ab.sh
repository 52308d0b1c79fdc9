#!/bin/bash
# A/B of a fused-rollout kernel change through bench.py's sub-records: $1 = variant library name under lib/variants
cd "$(dirname "$0")/.." || exit 1
root=$(pwd); out="$root/gpurun_out/r04/fused_ab"; mkdir -p "$out"
for round in 1 2; do
  for arm in variant shipped; do
    if [ "$arm" = variant ]; then export SCG_ROLLOUT_LIB="$root/safe_control_gym_amd/lib/variants/$1.so"; else unset SCG_ROLLOUT_LIB; fi
    timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > "$out/bench_${arm}_$round.json" 2> "$out/bench_${arm}_$round.err" || { tail -5 "$out/bench_${arm}_$round.err"; exit 1; }
    python - "$out/bench_${arm}_$round.json" "$arm $round" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=lambda d,k: '%.2f'%d[k] if k in d else '-'
print(sys.argv[2], 'headline %.2f'%(r['ms_per_step']*1e3), '| c3 fused %s +outputs %s | c2 fused %s +outputs %s |'%(f(r['c3'],'open_loop_fused_us_per_step'),f(r['c3'],'fused_with_outputs_us_per_step'),f(r['c2'],'open_loop_fused_us_per_step'),f(r['c2'],'fused_with_outputs_us_per_step')),
      ' '.join('2^%d: %s/%s'%(p['envs'].bit_length()-1,f(p,'open_loop_fused_us_per_step'),f(p,'fused_with_outputs_us_per_step')) for p in r['n_sweep']['points']))
PY
  done
done 2>&1 | tee "$out/table_$1.txt"
