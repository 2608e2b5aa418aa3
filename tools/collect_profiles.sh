# Collects what profiles/r01g_* is made of (run on the GPU box through gpurun; summaries: tools/summarize_profiles.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scaling-reference"
rm -rf gpurun_out/r01g_stats gpurun_out/r01g_fetch gpurun_out/r01g_write gpurun_out/r01g_p2ptrace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01g_stats -- $B > gpurun_out/r01g_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01g_fetch -- $B > gpurun_out/r01g_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01g_write -- $B > gpurun_out/r01g_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r01g_p2ptrace -- python3 tools/p2p_overhead.py --modes native-fused --steps 30 --rounds 1 > gpurun_out/r01g_p2ptrace.log 2>&1
python3 tools/p2p_overhead.py --steps 40 --rounds 2 > gpurun_out/r01g_p2p.log 2>&1
python3 bench.py > gpurun_out/r01g_bench.json 2> gpurun_out/r01g_bench.err
find gpurun_out/r01g_stats gpurun_out/r01g_fetch gpurun_out/r01g_write gpurun_out/r01g_p2ptrace -name "*.csv" | head -20
cat gpurun_out/r01g_bench.json | cut -c1-300
