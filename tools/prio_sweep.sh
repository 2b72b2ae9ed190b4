for v in 3 0 1 2; do
  if [ $v = 3 ]; then unset DN_LIB_PATH; else export DN_LIB_PATH=$PWD/audio-denoising_amd/lib/libdn_prio$v.so; fi
  python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('GL prio $v :', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, launch', d['roofline']['launch_ms'])"
done
