#!/bin/bash
# instruction-cache counters of the hop launch: does the front half's straight-line code push the chain loop out of the cache two CUs share?
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/icache_pmc"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for cfg in "256 4 0 32" "256 1 1 32" "256 1 1 0" "8192 1 2 32"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH -d "$out/$tag" -o p --output-format csv -- python3 "$root/tools/pipe_time.py" $cfg 30 > "$out/$tag.log" 2>&1
  echo "== $cfg"; cd "$root"; python tools/pmc_summary.py "$out/$tag/*counter_collection.csv"; cd /tmp
done
