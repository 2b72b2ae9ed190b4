#!/bin/bash
# The side measurements quoted in DESIGN.md / profiles/README.md, one JSON line each (tools/side_measurements.sh > profiles/rNN_side_measurements.jsonl)
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1; }
run                                   # the metric's configuration (batch 256, frame mode, pipelined)
run --serial                          # one launch per hop, no added latency
run --conv bf16                       # BASELINE config 3
run --stream --graph --batch 1024     # BASELINE config 5 under hipGraph replay
run --batch 1024
run --batch 4096
run --batch 8192
run --preset R1
run --preset R1 --batch 1024
run --preset R2
run --pcie
