#!/bin/bash
# Wave priority of the head-start iterations (variants built with make variant VARIANT=hsprioN EXTRA=-DDN_HS_PRIO=N) x head-start length.
for v in 1 0 2 3; do
  if [ $v = 1 ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=$PWD/audio-denoising_amd/lib/libdn_hsprio$v.so; fi
  for s in 3 5 7; do
    DN_GL_HEAD_START=$s python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('head-start prio $v, iterations $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
  done
done
