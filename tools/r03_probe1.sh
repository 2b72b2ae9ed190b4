#!/bin/bash
# round 3, first GPU call: wave placement, baseline numbers, SQ counters at 1,024 and 8,192 streams
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/r03_a"
mkdir -p "$out"
cd "$root"
tools/build/wave_place > "$out/wave_place.txt" 2>&1
echo placed
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1; }
( run; run --batch 1024; run --batch 8192; run --stream --graph --batch 1024 ) > "$out/base.jsonl"
echo based
cd /tmp && export TMPDIR=/tmp
for b in 1024 8192; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_SQ_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ_$b.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES -d "$out/pmc_SQ2_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ2_$b.log" 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM -d "$out/pmc_SQ3_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ3_$b.log" 2>&1
  echo pmc $b
done
cd "$root"
for b in 1024 8192; do python tools/pmc_summary.py "$out/pmc_SQ*_$b/*counter_collection.csv" > "$out/pmc_$b.txt" 2>&1; done
echo done
