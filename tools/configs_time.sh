# usage (GPU box, repo root): bash tools/configs_time.sh > profiles/rNN_configs.txt
# the canonical workload (bench.py, M1) at the picture formats / QPs of BASELINE.json's configs, one MI355X; one line per run
echo "bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-real-mix at the picture formats / QPs BASELINE.json names (one MI355X; pictures/s of the hot path, M1)"
for cfg in "1920 1080 32" "3840 2160 22" "3840 2160 27" "3840 2160 32" "3840 2160 37" "7680 4320 32"; do
  set -- $cfg
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-real-mix --width $1 --height $2 --qp $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print('%dx%d QP %d: %8.1f pictures/s  %.4f ms per picture  (with input stream %.1f)  dominant %s %.4f ms  stages %s' % (c['width'], c['height'], c['qp'], d['value'], c['ms_per_picture'], d['input_stream']['value'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['stage_ms']))"
done
