#!/bin/bash
# Static ISA statistics of one step-kernel instantiation (default: the headline C4 kernel).
# usage: tools/dev_isa.sh [out.s] [extra hipcc flags / -DSCG_DEV_ARGS=...]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${1:-/tmp/scg_dev_kernel.s}"; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --offload-device-only -S "$root/tools/dev_kernel.hip" -o "$out" "$@" 2>&1 | grep -v "hip-link" || true
python3 "$root/tools/isa_stats.py" "$out" scg_step_kernel --dump "${out%.s}.body.s" | head -${ISA_HEAD:-14}
grep -E "^\s+\.(sgpr|vgpr)_(count|spill_count):|\.vgpr_count|\.sgpr_count|NumVgprs|NumSgprs|ScratchSize|Occupancy" "$out" | head -8
