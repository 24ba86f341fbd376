#!/bin/bash
# All randomised campaigns (tools/fuzz_*.py) with fresh seeds in one gpurun call; writes
# gpurun_out/fuzz/report.txt (copy to profiles/rNN_fuzz_report.txt).  Arguments: seed base (default 600000).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BASE=${1:-600000}
OUT=$ROOT/gpurun_out/fuzz; mkdir -p $OUT
REP=$OUT/report.txt
echo "Randomised campaigns on an MI355X (tools/fuzz_*.py), seeds from $BASE:" > $REP
run() {
  echo >> $REP; echo "== python tools/$*" >> $REP
  local t0=$(date +%s)
  timeout 1500 python $ROOT/tools/$* > $OUT/last.log 2>&1; local rc=$?
  local t1=$(date +%s)
  local n_ok=$(grep -c "^ok" $OUT/last.log); local n_bad=$(grep -ci "^FAIL\|Traceback\|mismatch" $OUT/last.log)
  echo "   rc $rc, $n_ok ok lines, $n_bad failure lines, $((t1 - t0)) s" >> $REP
  echo "   first / last lines:" >> $REP
  grep -v "amdgpu.ids\|UserWarning\|Consider using\|assert rel" $OUT/last.log | head -3 | cut -c1-220 | sed 's/^/     /' >> $REP
  tail -2 $OUT/last.log | cut -c1-220 | sed 's/^/     /' >> $REP
  if [ $rc -ne 0 ] || [ $n_bad -ne 0 ]; then cp $OUT/last.log $OUT/failed_$1.log; fi
}
run fuzz_get_bboxes.py ${N_GB:-400} $((BASE + 0))
run fuzz_nms_ops.py ${N_NMS:-500} $((BASE + 1000))
run fuzz_fused_model.py ${N_FM:-80} $((BASE + 2000))
run fuzz_detector.py ${N_DET:-80} $((BASE + 3000))
run fuzz_losses.py ${N_LOSS:-200} $((BASE + 4000))
run fuzz_preproc_soft.py ${N_PRE:-300} $((BASE + 5000))
run fuzz_targets.py ${N_TGT:-300} $((BASE + 6000))
run fuzz_train.py ${N_TR:-30} $((BASE + 7000))
cat $REP
