#!/bin/bash
# SQ instruction counts of the front half alone: the pipelined hop with n_iter = 0 (the Griffin-Lim workgroups then only run one istft), batch 256
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/front_pmc"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$out/a" -o p --output-format csv -- python3 "$root/tools/pipe_time.py" 256 1 1 0 50 > "$out/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$out/b" -o p --output-format csv -- python3 "$root/tools/pipe_time.py" 256 1 1 32 50 > "$out/b.log" 2>&1
cd "$root"
echo "n_iter 0"; python tools/pmc_summary.py "$out/a/*counter_collection.csv"
echo "n_iter 32"; python tools/pmc_summary.py "$out/b/*counter_collection.csv"
