# Collects what profiles/<tag>_* is made of (run on the GPU box through gpurun; summaries: tools/summarize_profiles.py).
# usage: bash tools/collect_profiles.sh <tag>     e.g. r03a
# rocprofv3 gets the program itself after `--` (python3 ...), never a wrapper; --pmc passes carry no other trace domain.
T=${1:-r03a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scaling-reference --no-extras"
F32="python3 bench.py --nx 2048 -ic 2 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --jacobi-sweeps-timed 20"
F64S="python3 bench.py --nx 2048 -ic 2 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --jacobi-sweeps-timed 20"
for d in stats fetch write long f32 sq_f32 sq_f64; do rm -rf gpurun_out/${T}_$d; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- $B > gpurun_out/${T}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_fetch -- $B > gpurun_out/${T}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_write -- $B > gpurun_out/${T}_write.log 2>&1
# 1000 steps from set_init_F (the tiny-value front crosses the grid in steps ~65-600): per-kernel averages of a long run
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_long -- python3 tools/adapt_ab.py jacobi_tb_adapt 1 4096 1000 > gpurun_out/${T}_long.log 2>&1
# BASELINE configs[4]: 2048^2 rising bubble fp32; and the issue counters of the same workload in fp32 and fp64
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_f32 -- $F32 > gpurun_out/${T}_f32.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${T}_sq_f32 -- $F32 > gpurun_out/${T}_sq_f32.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${T}_sq_f64 -- $F64S > gpurun_out/${T}_sq_f64.log 2>&1
python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python3 bench.py --nx 2048 -ic 2 --dtype f32 --no-cpu-baseline > gpurun_out/${T}_bench_2048_bubble_f32.json 2>> gpurun_out/${T}_bench.err
find gpurun_out/${T}_stats gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_long gpurun_out/${T}_f32 gpurun_out/${T}_sq_f32 gpurun_out/${T}_sq_f64 -name "*.csv" | head -40
cut -c1-300 gpurun_out/${T}_bench.json
