cd $GRAFT_REPO_ROOT
VOF2D_MOM_PF=2 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused_step or random_conf or strip_decomp" 2>&1 | tail -2
S="python tools/strip_shape.py --n 1 --rank 0 --nx 4096 --ny 4096 --skip 5 --kernel k_momentum"
for R in 14 21 28 50; do echo "MOM_ROWS=$R"; VOF2D_MOM_ROWS=$R $S --sweep momentum_prefetch=0,1,2,3,4,6; done
