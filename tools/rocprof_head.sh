# rocprofv3 kernel stats of the head kernels (tools/time_head.py): batch, input kind (A / D), iterations
cd /tmp; export TMPDIR=/tmp
B=${1:-8}; KIND=${2:-D}; IT=${3:-20}
rm -rf /tmp/ph
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -- python $GRAFT_REPO_ROOT/tools/time_head.py $B $KIND $IT > /tmp/ph.log 2>&1
tail -5 /tmp/ph.log
python $GRAFT_REPO_ROOT/tools/kstats.py "/tmp/ph/*/*_kernel_stats.csv" ia::
