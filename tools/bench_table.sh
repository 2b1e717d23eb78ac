#!/bin/bash
# the (B, N) table of DESIGN.md section 6: one bench.py line per configuration (no CPU baseline, no secondary numbers)
for cfg in "1 12" "256 25" "256 50" "1024 6" "1024 12" "1024 20" "1024 25" "1024 32" "1024 40" "1024 50" "4096 50" "1024 64" "1024 70" "1024 77" "1024 100" "1024 150"; do
  set -- $cfg
  steps=200; [ $2 -gt 77 ] && steps=10
  python bench.py --batch $1 --feat $2 --steps $steps --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('B=%5d N=%3d  %8.4f ms/step  %10.0f steps/s  frac %.4f' % ($1, $2, j['ms_per_step'], j['value'], j['roofline']['frac']))"
done
