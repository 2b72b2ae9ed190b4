#!/bin/bash
# Sweep the Griffin-Lim head start (DN_GL_HEAD_START = iterations the front workgroup runs of its own frame's chain) on the bench workloads.
for s in 0 4 6 7 8 9 10; do
  DN_GL_HEAD_START=$s python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('S b256 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, launch', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'], 'unpipelined', d['serial_ms_per_step'])"
done
for s in 0 7 8 9 10 11 12; do
  DN_GL_HEAD_START=$s python bench.py --no-cpu-baseline --preset R1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('R1 b256 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
for s in 0 5 8; do
  DN_GL_HEAD_START=$s python bench.py --no-cpu-baseline --preset R2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('R2 b256 head_start $s :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done
