#!/bin/bash
# Collect PMC counters for the step kernel, one rocprofv3 pass per counter group (TCC slots:
# FETCH_SIZE and WRITE_SIZE do not fit one pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# No trace domains besides the implicit kernel dispatch records (gpurun refuses --pmc + sys-trace).
# usage: profiles/pmc.sh <out_dir> [bench args...]
set -u
out="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
if [ "${PMC_TRAFFIC_ONLY:-0}" = "1" ]; then
  groups=("FETCH_SIZE" "WRITE_SIZE")
elif [ "${PMC_GROUPS_ONLY:-}" = "SQ" ]; then
  groups=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA")
else
  groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
          "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
          "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
fi
for group in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- python3 "$root/bench.py" --steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-sub-records "$@" > "$out/pass$i.log" 2>&1 || echo "pass $i ($group) failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'scg_step' not in row.get('Kernel_Name', ''):
            continue
        k = row['Counter_Name']
        acc[k][0] += float(row['Counter_Value']); acc[k][1] += 1
with open(out + '/summary.txt', 'w') as fo:
    for k, (s, c) in sorted(acc.items()):
        line = '%-28s per-launch avg %.6g   (launches %d)' % (k, s / max(c, 1), c)
        print(line); fo.write(line + '\n')
PY
