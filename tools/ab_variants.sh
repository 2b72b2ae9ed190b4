#!/bin/bash
# A/B of experiment builds (make variant VARIANT=..): tools/ab_variants.sh name1 name2 ..  -> the bench's ms_per_step for the shipped library and each variant
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-10s %.4f ms  %.0f %s' % ('$1', d['ms_per_step'], d['value'], d['unit']))"; }
for rep in 1 2; do
  run base
  for v in "$@"; do DN_LIB_PATH=$PWD/audio-denoising_amd/lib/libdn_$v.so run $v; done
done
