V=taichi-2d-vof_amd/csrc/build/variants/libvof2d_tmold.so
for rep in 1 2 3; do
  timeout 200 python3 tools/probes/halves_sweep.py "overlap_halves=0,fuse_tm=1" --rounds 3 --skip 640 --engines 2 | sed 's/^/new  late  /'
  timeout 200 python3 tools/probes/halves_sweep.py "overlap_halves=0,fuse_tm=1" --rounds 3 --skip 640 --engines 2 --lib $V | sed 's/^/old  late  /'
done
for rep in 1 2; do
  timeout 200 python3 tools/probes/halves_sweep.py "overlap_halves=0,fuse_tm=1" --rounds 3 --engines 2 | sed 's/^/new  front /'
  timeout 200 python3 tools/probes/halves_sweep.py "overlap_halves=0,fuse_tm=1" --rounds 3 --engines 2 --lib $V | sed 's/^/old  front /'
done
