#!/bin/bash
# Memory-side probes with (almost) no arithmetic: explicit Euler, no terminations (tuning only).
run() { local label="$1"; shift
  python bench.py --steps 1500 --no-cpu-baseline --env-kw physics=euler --env-kw done_on_out_of_bound=false --env-kw episode_len_sec=600 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-40s %.2f us  %.0f GB/s' % ('$label', j['roofline']['kernel_us_per_launch'], j['roofline']['achieved']))"
}
export SCG_PACKED=0
for pad in 0 1 2 3 4 8 16 17 33 64; do SCG_LD_PAD=$pad run "unpacked ld_pad=$pad"; done
run "unpacked n=1000000" --envs-per-gpu 1000000
run "unpacked n=2^19" --envs-per-gpu 524288
run "unpacked n=2^21" --envs-per-gpu 2097152
run "unpacked n=2^22" --envs-per-gpu 4194304
export SCG_PACKED=1
for pad in 0 1 3 17; do SCG_LD_PAD=$pad run "packed ld_pad=$pad"; done
SCG_PACKED_BLOCKS_PER_CU=100000 run "packed nocap pad0"
SCG_LD_PAD=17 SCG_PACKED_BLOCKS_PER_CU=100000 run "packed nocap pad17"
