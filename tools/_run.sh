cd $GRAFT_REPO_ROOT
B=taichi-2d-vof_amd/csrc/build/variants/libvof2d_base.so
for rep in 1 2 3; do
for L in "--lib $B" ""; do echo -n "${L:-new}: "; python tools/strip_shape.py $L --n 1 --rank 0 --nx 4096 --ny 4096 --skip 5 --steps 40 | grep -E "momentum"; done
echo -n "new virt=0: "; VOF2D_VIRTUAL_GHOSTS=0 python tools/strip_shape.py --n 1 --rank 0 --nx 4096 --ny 4096 --skip 5 --steps 40 | grep -E "momentum"
done
