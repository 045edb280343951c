cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/kt_joint; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python bench.py --workload joint --serial --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/bench.json 2> $O/log
DB=$(find $O/kt -name "*.db" | head -1)
python tools/prof_summary.py "$DB" 1 > $O/kernel_stats.txt
rm -rf $O/kt
