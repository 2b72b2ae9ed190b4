#!/bin/bash
# Where do the bf16 conv tiles (BASELINE config 3) lose to the fp32 ones?  Instruction mix and matrix-pipe occupancy of the stand-alone cell kernel,
# both precisions:   tools/cell_counters.sh [outdir]  -> <outdir>/cell_counters.txt
root="$(cd "$(dirname "$0")/.." && pwd)"
out=${1:-$root/gpurun_out/cell_counters}
mkdir -p "$out"
out="$(cd "$out" && pwd)"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
  i=$((i + 1))
  rocprofv3 --pmc $set -d "$out/pmc$i" -o p --output-format csv -- python3 "$root/tools/cell_both.py" 10 > "$out/pmc$i.log" 2>&1 || { echo "pass $i failed: $out/pmc$i.log" >&2; tail -5 "$out/pmc$i.log" >&2; }
done
cd "$root"
python tools/pmc_summary.py "$out/pmc*/*counter_collection.csv" > "$out/cell_counters.txt" || exit 1
python tools/cell_time.py >> "$out/cell_counters.txt" 2>/dev/null
cat "$out/cell_counters.txt"
