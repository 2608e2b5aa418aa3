# Collects what profiles/<tag>_* is made of (run on the GPU box through gpurun) and summarises it there, so that the
# bench lines at the end are printed with the PMC hash of THIS session's sources (profiles/jacobi_pmc.json).
# usage: bash tools/collect_profiles.sh <tag>     e.g. r04a     -> gpurun_out/<tag>_profiles/ (copy into profiles/)
# rocprofv3 gets the program itself after `--` (python3 ...), never a wrapper.  The --pmc passes carry --kernel-trace
# (needed for per-dispatch rows) and no other trace domain.
T=${1:-r04a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out   # (every command below under its own timeout: a hung one must not eat the box's time)
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scaling-reference --no-extras --profile-steps 14"
F32="python3 bench.py --nx 2048 -ic 2 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --jacobi-sweeps-timed 20 --profile-steps 14"
F64S="python3 bench.py --nx 2048 -ic 2 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --jacobi-sweeps-timed 20 --profile-steps 14"
for d in stats fetch write long long2 long3 tmfetch tmwrite f32 sq_f32 sq_f64; do rm -rf gpurun_out/${T}_$d; done
# Per-kernel durations and counters want one kernel at a time on the whole grid: the one-chain schedule (the default
# on large grids runs every kernel as two overlapping launches; rocprofv3 --pmc serialises dispatches anyway).
export VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=0
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats -- $B > gpurun_out/${T}_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_fetch -- $B > gpurun_out/${T}_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_write -- $B > gpurun_out/${T}_write.log 2>&1
# 1000 steps from set_init_F (the tiny-value front crosses the grid in steps ~65-600): per-kernel averages of a long run
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_long -- python3 tools/bound_run.py --steps 1000 > gpurun_out/${T}_long.log 2>&1
# BASELINE configs[4]: 2048^2 rising bubble fp32; and the issue counters of the same workload in fp32 and fp64
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_f32 -- $F32 > gpurun_out/${T}_f32.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${T}_sq_f32 -- $F32 > gpurun_out/${T}_sq_f32.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${T}_sq_f64 -- $F64S > gpurun_out/${T}_sq_f64.log 2>&1
# the chain form (the default without k_tm) and the k_tm form against the one-chain trace above: span per step vs the kernels' own durations
unset VOF2D_OVERLAP_HALVES
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_long2 -- python3 tools/bound_run.py --steps 1000 > gpurun_out/${T}_long2.log 2>&1
export VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_long3 -- python3 tools/bound_run.py --steps 1000 > gpurun_out/${T}_long3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${T}_tmfetch -- python3 tools/bound_run.py --steps 60 > gpurun_out/${T}_tmfetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${T}_tmwrite -- python3 tools/bound_run.py --steps 60 > gpurun_out/${T}_tmwrite.log 2>&1
unset VOF2D_OVERLAP_HALVES VOF2D_FUSE_TM
python3 tools/summarize_overlap_trace.py $T one-chain=gpurun_out/${T}_long chains=gpurun_out/${T}_long2 k_tm=gpurun_out/${T}_long3 --cmd "rocprofv3 --kernel-trace -- python3 tools/bound_run.py --steps 1000 (VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=0 for the one-chain run, VOF2D_FUSE_TM=0 for chains, VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=1 for k_tm)"
python3 tools/summarize_profiles.py ${T}_tm "$(find gpurun_out/${T}_long3 -name '*kernel_stats.csv' | head -1)" "$(find gpurun_out/${T}_tmfetch -name '*counter_collection.csv' | head -1)" "$(find gpurun_out/${T}_tmwrite -name '*counter_collection.csv' | head -1)" --tm-json --cmd "VOF2D_OVERLAP_HALVES=0 VOF2D_FUSE_TM=1 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bound_run.py --steps 1000"
# summaries, here: profiles/<tag>_*.md and profiles/jacobi_pmc.json (with the hash of the sources just profiled)
f() { find gpurun_out/${T}_$1 -name "*$2" | head -1; }
python3 tools/summarize_profiles.py $T "$(f stats kernel_stats.csv)" "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" --cmd "VOF2D_OVERLAP_HALVES=0 rocprofv3 --kernel-trace --stats --output-format csv -- $B"
python3 tools/summarize_profiles.py ${T}_long "$(f long kernel_stats.csv)" --cmd "VOF2D_OVERLAP_HALVES=0 rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bound_run.py --steps 1000"
python3 tools/summarize_profiles.py ${T}_f32 "$(f f32 kernel_stats.csv)" --nx 2048 --ny 2048 --dtype f32 --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- $F32"
python3 tools/summarize_sq.py $T f32="$(f sq_f32 counter_collection.csv)" f64="$(f sq_f64 counter_collection.csv)" --nx 2048 --ny 2048 --cmd "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -- python3 bench.py --nx 2048 -ic 2 --dtype f32|f64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --jacobi-sweeps-timed 20" > /dev/null
# the bench command as the driver runs it (default environment: the form the handle keeps), under the profiler
rm -rf gpurun_out/${T}_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_default -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-reference --no-extras > gpurun_out/${T}_default.log 2>&1
python3 tools/summarize_profiles.py ${T}_default "$(find gpurun_out/${T}_default -name '*kernel_stats.csv' | head -1)" --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-reference --no-extras"
# the bench lines LAST: roofline.traffic is quoted from the jacobi_pmc.json written a moment ago
timeout 1200 python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 900 python3 bench.py --nx 2048 -ic 2 --dtype f32 --no-cpu-baseline > gpurun_out/${T}_bench_2048_bubble_f32.json 2>> gpurun_out/${T}_bench.err
timeout 900 python3 bench.py --nx 4096 --dtype f32 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_4096_f32.json 2>> gpurun_out/${T}_bench.err
for n in 128 1024 2048 8192; do timeout 600 python3 bench.py --nx $n --no-cpu-baseline --no-extras --profile-steps 40 $([ $n = 8192 ] && echo "--steps 60 --warmup 10") 2>> gpurun_out/${T}_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%s: %.4f ms/step, %.2f G cell-updates/s' % (d['config']['workload'], d['ms_per_step'], d['value']/1e9))"; done > gpurun_out/${T}_sizes.txt 2>&1
mkdir -p gpurun_out/${T}_profiles
cp profiles/${T}_default_kernel_stats.md profiles/${T}_overlap_trace.md profiles/${T}_tm_kernel_stats.md profiles/${T}_tm_hbm_pmc.md profiles/${T}_kernel_stats.md profiles/${T}_hbm_pmc.md profiles/${T}_long_kernel_stats.md profiles/${T}_f32_kernel_stats.md profiles/${T}_sq_counters.md profiles/jacobi_pmc.json profiles/tm_pmc.json gpurun_out/${T}_profiles/ 2>/dev/null
cp gpurun_out/${T}_bench.json gpurun_out/${T}_bench_2048_bubble_f32.json gpurun_out/${T}_bench_4096_f32.json gpurun_out/${T}_sizes.txt gpurun_out/${T}_profiles/
find gpurun_out/${T}_stats gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_long gpurun_out/${T}_long2 gpurun_out/${T}_default gpurun_out/${T}_long3 gpurun_out/${T}_tmfetch gpurun_out/${T}_tmwrite gpurun_out/${T}_f32 gpurun_out/${T}_sq_f32 gpurun_out/${T}_sq_f64 -type f -size +6M -delete      # (gpurun brings back 64 MiB at most; the raw per-kernel CSVs stay)
ls gpurun_out/${T}_profiles; cut -c1-400 gpurun_out/${T}_bench.json; cat gpurun_out/${T}_sizes.txt
