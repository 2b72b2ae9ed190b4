#!/bin/bash
# round 3: deep pipe (dn_pipe_set_depth): parity on the GPU, then throughput per depth and batch
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/r03_c"
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deep_pipe or wavefront_per_stream or head_start_is_bit or captured_streaming or pipelined" > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1; }
for b in 256 512; do
  for dp in 1 2 3 4; do
    echo "{\"depth\": $dp, \"batch\": $b, \"line\": $(DN_PIPE_DEPTH=$dp run --batch $b)}" >> "$out/depth.jsonl"
  done
done
python - <<PY
import json
for l in open("$out/depth.jsonl"):
    d = json.loads(l); x = d["line"]
    print(d["depth"], d["batch"], x["value"], x["ms_per_step"])
PY
