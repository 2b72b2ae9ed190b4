#!/bin/bash
# round 3: the wavefront-per-stream Griffin-Lim: parity on the GPU, then throughput per schedule and batch
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/r03_b"
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wavefront_per_stream or head_start_is_bit or odd_batch or captured_streaming" > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -3 "$out/tests.log"
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1; }
for b in 256 512 768 1024 2048 4096 8192; do
  for s in 1 2; do
    echo "{\"sched\": $s, \"batch\": $b, \"line\": $(DN_GL_SCHEDULE=$s run --batch $b)}" >> "$out/sched.jsonl"
  done
  echo batch $b
done
DN_GL_SCHEDULE=2 run --stream --graph --batch 1024 > "$out/stream_graph_1024.json"
python - <<PY
import json
for l in open("$out/sched.jsonl"):
    d = json.loads(l); x = d["line"]
    print(d["sched"], d["batch"], x["value"], x["ms_per_step"])
PY
