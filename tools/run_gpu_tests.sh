#!/bin/bash
# Full GPU suite, then the lean-tier parity tests again on the opt-in packed kernel.
set -u
python -m pytest tests -m gpu -q "$@" 2>&1 | tail -5
SCG_PACKED=1 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "lean_tier" 2>&1 | tail -3
