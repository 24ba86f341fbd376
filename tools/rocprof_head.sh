cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -- python $GRAFT_REPO_ROOT/tools/time_head.py 8 D 20 > /tmp/ph.log 2>&1
grep "ia::" /tmp/ph/*/*_kernel_stats.csv | cut -d, -f1-4
