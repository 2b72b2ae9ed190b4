#!/bin/bash
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/r03_d"
mkdir -p "$out"
cd "$root"
t() { python tools/pipe_time.py "$@" 2>/dev/null | tail -1; }
(
t 256 1 1 32; t 256 1 1 0; t 256 1 1 32 300 0; t 256 1 1 0 300 0
t 256 1 2 32; t 256 1 2 0
t 256 4 0 32; t 256 4 0 16; t 256 4 0 0
t 256 3 0 32; t 256 3 0 0
t 1024 1 2 32; t 1024 1 2 0; t 1024 1 1 32; t 1024 1 1 0
) | tee "$out/times.txt"
