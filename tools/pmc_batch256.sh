#!/bin/bash
# PMC passes only (batch 256), into gpurun_out/r04_v3
root="$(cd "$(dirname "$0")/.." && pwd)"; out=$root/gpurun_out/r04_v3; mkdir -p $out/logs
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 40 > "$out/logs/pmc_$c.log" 2>&1 || exit 1; done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_SQ_256" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 28 256 > "$out/logs/pmc_SQ.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES -d "$out/pmc_SQ2_256" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 28 256 > "$out/logs/pmc_SQ2.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM -d "$out/pmc_SQ3_256" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 28 256 > "$out/logs/pmc_SQ3.log" 2>&1 || exit 1
# the deep pipe of round 3 on the same box, for the per-hop comparison
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_D4" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 40 256 depth 4 > "$out/logs/pmc_D4.log" 2>&1 || exit 1
cd $root
python tools/pmc_summary.py --traffic-json "$out/pmc_traffic.json" --kernel group_kernel --frames-per-launch 1024 --key group_kernel_hbm_bytes_per_launch --tag r04_v3 "$out/pmc_FETCH_SIZE/*counter_collection.csv" "$out/pmc_WRITE_SIZE/*counter_collection.csv" > "$out/r04_v3_pmc_hbm.txt" || exit 1
python tools/pmc_summary.py "$out/pmc_SQ*_256/*counter_collection.csv" > "$out/r04_v3_pmc_sq_256.txt" || exit 1
python tools/pmc_summary.py "$out/pmc_D4/*counter_collection.csv" > "$out/r04_v3_pmc_depth4.txt" || exit 1
cat "$out/r04_v3_pmc_hbm.txt" "$out/r04_v3_pmc_sq_256.txt" "$out/r04_v3_pmc_depth4.txt"
