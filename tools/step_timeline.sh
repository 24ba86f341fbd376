cd /tmp; export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train --no-other-configs > /tmp/warm.log 2>&1
rm -rf /tmp/pb
rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-other-configs > /tmp/pb.log 2>&1
python $ROOT/tools/trace_timeline.py /tmp/pb/*/*_kernel_trace.csv k_affine_relu_maxpool k_finalize\( | cut -c1-150 > $ROOT/gpurun_out/step_timeline.txt
python $ROOT/tools/summarize_trace.py /tmp/pb/*/*_kernel_trace.csv --steps 3 | tail -12
