#!/bin/bash
for tb in 1 2 5; do for r in 0 4 8 16 32; do
  [ $tb = 1 ] && [ $r != 0 ] && continue
  env VOF2D_TB=$tb VOF2D_TB_ROWS=$r python bench.py --nx 2048 --dtype f32 -ic 2 --steps 100 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 2048 tb=$tb rows=$r us/step', round(1e3*d['ms_per_step'],1))"
done; done
for cfg in "" "VOF2D_MOM_ROWS=8" "VOF2D_MOM_ROWS=16" "VOF2D_MOM_ROWS=32" "VOF2D_FCTX_ROWS=8" "VOF2D_FCTX_ROWS=32" "VOF2D_ROWS=8" "VOF2D_ROWS=32"; do
  env VOF2D_TB=1 $cfg python bench.py --nx 2048 --dtype f32 -ic 2 --steps 100 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 2048 tb=1 [$cfg] us/step', round(1e3*d['ms_per_step'],1), d['kernels_us_dispatch_start_to_stop'])"
done
