cd /root/repo
for round in 1 2 3; do
for v in 0 1; do
  ILM_LIGHTS_IN_PLACE=$v python bench.py --no-cpu-baseline --no-cfg4 --no-next-rows --steps 20 --light-frames 8 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in place $v', {k:(v['roofline']['launch_ms'], v['ms_per_frame']) for k,v in d['lighting'].items()})"
done
done
