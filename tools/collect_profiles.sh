#!/bin/bash
# Collect the evidence files of one version of the kernels into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   make -C audio-denoising_amd/csrc probe      (HERE, before gpurun: the stamped build travels with the snapshot)
#   tools/collect_profiles.sh r04_v1 [quick]
# bench line, rocprofv3 kernel statistics of the same command, PMC passes (HBM bytes -> pmc_traffic.json in the same pass; SQ instruction mix and
# busy counters at batch 256 and in the saturated regime) on tools/prof_step.py, the side measurements, the stamped timelines of the diagnostic
# build.  Every tool's stderr goes to <out>/logs/; a probe that prints nothing FAILS the collection (round 3 committed empty files).
tag=${1:-run}
quick=${2:-}
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$tag"
mkdir -p "$out/logs"
cd "$root"
probe="$root/audio-denoising_amd/lib/libdn_probe.so"
die() { echo "collect_profiles: $*" >&2; exit 1; }
[ -f "$probe" ] || die "$probe is missing: run make -C audio-denoising_amd/csrc probe before gpurun"
for f in audio-denoising_amd/csrc/*.hip audio-denoising_amd/csrc/*.hpp include/dn_denoise.h; do
  [ "$probe" -nt "$f" ] || die "$probe is older than $f: rebuild it (make -C audio-denoising_amd/csrc probe)"
done
# run NAME OUTFILE cmd...: stdout -> OUTFILE, stderr -> logs/NAME.log; fails on a non-zero status or an empty OUTFILE
run() {
  local name=$1 dst=$2; shift 2
  "$@" >> "$dst" 2>> "$out/logs/$name.log" || die "$name failed (rc $?): see $out/logs/$name.log"
  [ -s "$dst" ] || die "$name printed nothing: see $out/logs/$name.log"
}
python bench.py > "$out/${tag}_bench.json" 2> "$out/logs/bench.log" || die "bench.py failed"
echo bench
: > "$out/${tag}_group_timelines.txt"
run group_probe "$out/${tag}_group_timelines.txt" python tools/group_probe.py 256 4
run group_probe "$out/${tag}_group_timelines.txt" python tools/group_probe.py 256 2
: > "$out/${tag}_glw_timelines.txt"
for a in "256 4" "256 1" "1024 1"; do run glw_probe "$out/${tag}_glw_timelines.txt" python tools/glw_probe.py $a; done
: > "$out/${tag}_hop_workgroup_stamps.txt"
run hop_wg_probe "$out/${tag}_hop_workgroup_stamps.txt" python tools/hop_wg_probe.py
DN_PRESET=R1 run hop_wg_probe_r1 "$out/${tag}_hop_workgroup_stamps.txt" python tools/hop_wg_probe.py
: > "$out/${tag}_gl_iteration_stamps.txt"
DN_LIB_PATH=$probe run gl_probe "$out/${tag}_gl_iteration_stamps.txt" python tools/gl_probe.py hop 256
: > "$out/${tag}_workgroup_census.txt"
for m in "frames 4" "frames 1" "stream 4"; do run pairing_probe "$out/${tag}_workgroup_census.txt" python tools/pairing_probe.py $m; done
echo stamps
if [ -z "$quick" ]; then
  tools/side_measurements.sh > "$out/${tag}_side_measurements.jsonl" 2> "$out/logs/side.log" || die "side measurements failed"
  [ -s "$out/${tag}_side_measurements.jsonl" ] || die "side measurements printed nothing"
  echo side
fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 "$root/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$out/logs/kt.log" 2>&1 || die "kernel trace failed"
find "$out/kt" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
[ -s "$out/${tag}_kernel_stats.csv" ] || die "no kernel statistics"
echo trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- python3 "$root/tools/prof_step.py" 40 > "$out/logs/pmc_$c.log" 2>&1 || die "pmc $c failed"
done
batches="256"
[ -z "$quick" ] && batches="256 1024 8192"
for b in $batches; do
  sq_steps=12; [ "$b" = 256 ] && sq_steps=28          # (batch 256: groups of four hops per launch -- enough full launches for the median)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$out/pmc_SQ_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" $sq_steps $b > "$out/logs/pmc_SQ_$b.log" 2>&1 || die "pmc SQ $b failed"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES -d "$out/pmc_SQ2_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" $sq_steps $b > "$out/logs/pmc_SQ2_$b.log" 2>&1 || die "pmc SQ2 $b failed"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM -d "$out/pmc_SQ3_$b" -o p --output-format csv -- python3 "$root/tools/prof_step.py" $sq_steps $b > "$out/logs/pmc_SQ3_$b.log" 2>&1 || die "pmc SQ3 $b failed"
  echo pmc $b
done
cd "$root"
python tools/pmc_summary.py --traffic-json "$out/pmc_traffic.json" --kernel group_kernel --frames-per-launch 1024 --key group_kernel_hbm_bytes_per_launch --tag "$tag" \
  "$out/pmc_FETCH_SIZE/*counter_collection.csv" "$out/pmc_WRITE_SIZE/*counter_collection.csv" > "$out/${tag}_pmc_hbm.txt" 2> "$out/logs/pmc_summary.log" || die "pmc summary failed"
for b in $batches; do python tools/pmc_summary.py "$out/pmc_SQ*_$b/*counter_collection.csv" > "$out/${tag}_pmc_sq_$b.txt" 2>> "$out/logs/pmc_summary.log" || die "pmc sq summary $b failed"; done
for f in "$out"/${tag}_*; do [ -s "$f" ] || die "$f is empty"; done
echo done
