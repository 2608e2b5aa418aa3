cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "b1:overlap_halves=0:jacobi_tb_adapt=1" "b0:overlap_halves=0:jacobi_tb_adapt=0" "h1:overlap_halves=1:jacobi_tb_adapt=1" "h0:overlap_halves=1:jacobi_tb_adapt=0"; do
  IFS=: read tag p1 p2 <<< "$cfg"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $R/tools/bound_run.py --steps 1000 --param $p1 --param $p2 > /tmp/tr_$tag.log 2>&1
  python3 $R/tools/probes/../summarize_overlap_trace.py tmp_$tag $tag=/tmp/tr_$tag --windows 701-1000
done
