#!/bin/bash
# default heuristics vs a few overrides across grid sizes
for n in 128 512 1024 2048 4096; do
  steps=$([ $n -le 512 ] && echo 1000 || echo 100)
  for cfg in "" "VOF2D_TB=1" "VOF2D_TB=2"; do
    env $cfg python bench.py --nx $n --steps $steps --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n=$n [$cfg] us/step', round(1e3*d['ms_per_step'],1), 'Gcells/s', round(d['value']/1e9,2))"
  done
done
env python bench.py --nx 2048 --dtype f32 -ic 2 --steps 100 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2048 f32 bubble us/step', round(1e3*d['ms_per_step'],1), 'Gcells/s', round(d['value']/1e9,2))"
env VOF2D_TB=1 python bench.py --nx 2048 --dtype f32 -ic 2 --steps 100 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2048 f32 bubble TB=1 us/step', round(1e3*d['ms_per_step'],1), 'Gcells/s', round(d['value']/1e9,2))"
