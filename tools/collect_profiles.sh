#!/bin/bash
# Collect the evidence files of one version of the kernels into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   tools/collect_profiles.sh r02_v3
# bench line, rocprofv3 kernel statistics of the same command, PMC passes (HBM bytes; SQ instruction mix) on tools/prof_step.py,
# the side measurements, the stamped breakdowns of the diagnostic build and the head-start sweep.
tag=${1:-run}
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd "$root"
python bench.py > "$out/${tag}_bench.json" 2> "$out/bench.log" || exit 1
tools/side_measurements.sh > "$out/${tag}_side_measurements.jsonl" || exit 1
python tools/cell_probe.py > "$out/${tag}_cell_phase_stamps.txt" 2>/dev/null
python tools/hop_wg_probe.py 2>/dev/null | tail -3 > "$out/${tag}_hop_workgroup_stamps.txt"
DN_LIB_PATH=$root/audio-denoising_amd/lib/libdn_probe.so python tools/gl_probe.py hop 256 > "$out/${tag}_gl_iteration_stamps.txt" 2>/dev/null
tools/head_start_sweep.sh > "$out/${tag}_head_start_sweep.txt" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 "$root/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$out/kt.log" 2>&1 || exit 1
cp "$out"/kt/*kernel_stats.csv "$out/${tag}_kernel_stats.csv" 2>/dev/null || find "$out/kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 20 > "$out/pmc_$c.log" 2>&1 || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_SQ" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 20 > "$out/pmc_SQ.log" 2>&1
cd "$root"
python tools/pmc_summary.py "$out/pmc_*/*counter_collection.csv" > "$out/${tag}_pmc.txt" 2>&1
echo done
