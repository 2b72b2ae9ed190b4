#!/bin/bash
cd /root/repo
for v in base gl0hs0 gl3hs3 gl1hs0 gl3hs2; do
  if [ $v = base ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=$PWD/audio-denoising_amd/lib/libdn_$v.so; fi
  for s in 8 10 12; do
    DN_GL_HEAD_START=$s python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v head start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
  done
done
