#!/bin/bash
# Full GPU suite (one process).
set -u
python -m pytest tests -m gpu -q "$@" 2>&1 | tail -5
