# fused aggregation pairs on (1, the default) against off (0) at a ladder of frame sizes
for sz in "320 240 64" "450 375 64" "640 480 64" "640 480 128" "960 540 128" "960 540 256" "1280 720 64" "1280 720 128" "1920 1080 64" "1920 1080 128"; do
  set -- $sz
  for f in 0 1; do
    ADC_DEBUG_ENV=1 ADC_AGG_FUSE=$f timeout -k 10 120 python bench.py --quick --steps 40 --warmup 5 --width $1 --height $2 --disp $3 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1x$2x$3 fuse=$f', round(j['ms_per_frame'],3), 'aggr', j['stage_ms']['aggr'])" || exit 1
  done
done
