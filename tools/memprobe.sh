#!/bin/bash
# Memory-side probes with (almost) no arithmetic: explicit Euler, no terminations (tuning only).  plane_pad = the plane-stride
# skew in units of 64 lanes (make(..., plane_pad=k)).
run() { local label="$1"; shift
  python bench.py --steps 1500 --no-cpu-baseline --no-sub-records --env-kw physics=euler --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-40s %.2f us  %.0f GB/s' % ('$label', j['roofline']['kernel_us_per_launch'], j['roofline']['achieved']))"
}
for pad in 0 1 2 3 4 8 16 17 33 64; do run "plane_pad=$pad" --env-kw plane_pad=$pad; done
run "n=1000000" --envs-per-gpu 1000000
run "n=2^19" --envs-per-gpu 524288
run "n=2^21" --envs-per-gpu 2097152
run "n=2^22" --envs-per-gpu 4194304
