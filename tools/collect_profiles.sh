#!/bin/bash
# Collect the evidence files of one version of the kernels into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   tools/collect_profiles.sh r03_v1
# bench line, rocprofv3 kernel statistics of the same command, PMC passes (HBM bytes; SQ instruction mix and busy counters at batch 256 and in
# the saturated regime) on tools/prof_step.py, the side measurements, the stamped timelines of the diagnostic build, the head-start sweep.
tag=${1:-run}
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd "$root"
python bench.py > "$out/${tag}_bench.json" 2> "$out/bench.log" || exit 1
echo bench
tools/side_measurements.sh > "$out/${tag}_side_measurements.jsonl" || exit 1
echo side
( for a in "256 4" "256 1" "1024 1" ; do python tools/glw_probe.py $a 2>/dev/null | tail -6; done ) > "$out/${tag}_glw_timelines.txt"
( python tools/hop_wg_probe.py 2>/dev/null | tail -3; DN_PRESET=R1 python tools/hop_wg_probe.py 2>/dev/null | tail -3 ) > "$out/${tag}_hop_workgroup_stamps.txt"
DN_LIB_PATH=$root/audio-denoising_amd/lib/libdn_probe.so python tools/gl_probe.py hop 256 > "$out/${tag}_gl_iteration_stamps.txt" 2>/dev/null
( for m in "frames 4" "frames 1" "stream 4"; do python tools/pairing_probe.py $m 2>/dev/null | tail -12; done ) > "$out/${tag}_workgroup_census.txt"
tools/head_start_sweep.sh > "$out/${tag}_head_start_sweep.txt" 2>/dev/null
echo stamps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 "$root/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$out/kt.log" 2>&1 || exit 1
find "$out/kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
echo trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 20 > "$out/pmc_$c.log" 2>&1 || exit 1
done
for b in 256 1024 8192; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_SQ_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ_$b.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES -d "$out/pmc_SQ2_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ2_$b.log" 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM -d "$out/pmc_SQ3_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 10 $b > "$out/pmc_SQ3_$b.log" 2>&1
  echo pmc $b
done
cd "$root"
python tools/pmc_summary.py "$out/pmc_FETCH_SIZE/*counter_collection.csv" "$out/pmc_WRITE_SIZE/*counter_collection.csv" > "$out/${tag}_pmc_hbm.txt" 2>&1
for b in 256 1024 8192; do python tools/pmc_summary.py "$out/pmc_SQ*_$b/*counter_collection.csv" > "$out/${tag}_pmc_sq_$b.txt" 2>&1; done
echo done
