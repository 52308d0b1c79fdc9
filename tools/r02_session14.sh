#!/bin/bash
# round 2, GPU session 14: attitude precise mode with angle-addition stage trigonometry.
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02n"
mkdir -p "$out"
cd "$root"
echo "== precise-mode tests"; timeout -k 10 600 python -m pytest tests -m gpu -q -rA -s -k "precise or attitude or compensated" > "$out/precise_tests.log" 2>&1; tail -2 "$out/precise_tests.log"
grep -E "^(FAILED|ERROR)|drift" "$out/precise_tests.log" | head -30
echo "== bench with sub-records"
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > "$out/bench_c4.json" 2> "$out/bench_c4.err" || tail -3 "$out/bench_c4.err"
python - <<PY
import json
j = json.load(open('$out/bench_c4.json'))
print('main %.2f us' % (j['ms_per_step'] * 1e3))
for k, v in j.get('sub_records', {}).items():
    print('%-28s %.2f us  frac %.3f' % (k, v['us_per_step'], v['frac']))
PY
