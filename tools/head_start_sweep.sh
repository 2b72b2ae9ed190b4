#!/bin/bash
# Sweep the Griffin-Lim head start of the one-hop pipe (DN_GL_HEAD_START = iterations the front workgroup runs of its own frame's chain) on the bench workloads.
cd "$(dirname "$0")/.."
line() { python bench.py --no-cpu-baseline --no-extras --depth 1 "$@" 2>/dev/null | tail -1; }
for s in 0 4 6 7 8 9 10; do
  DN_GL_HEAD_START=$s line --steps 400 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('S b256 depth 1 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, launch', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'], 'unpipelined', d['serial_ms_per_step'])"
done
for s in 0 8 10 11 12 13 14; do
  DN_GL_HEAD_START=$s line --preset R1 --steps 400 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('R1 b256 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
for s in 0 5 8; do
  DN_GL_HEAD_START=$s line --preset R2 --steps 400 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('R2 b256 depth 1 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
