#!/bin/bash
# round 2, GPU session 27: PMC counters of the matrix-exponential kernel (what bounds it).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/r02y"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- python3 "$root/tools/bench_expm.py" > "$out/pass$i.log" 2>&1 || echo "pass $i ($group) failed: $(tail -2 $out/pass$i.log)"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'expm' not in row.get('Kernel_Name', ''):
            continue
        key = (row['Kernel_Name'][:60], row.get('Grid_Size', '?'))
        a = acc[key][row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
with open(out + '/pmc_summary_expm.txt', 'w') as fo:
    for key in sorted(acc):
        fo.write('%s grid %s\n' % key)
        for k, (s, c) in sorted(acc[key].items()):
            fo.write('  %-30s per-launch avg %.6g   (launches %d)\n' % (k, s / max(c, 1), c))
print(open(out + '/pmc_summary_expm.txt').read())
PY
rm -rf "$out"/pass*/
