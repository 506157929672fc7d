#!/bin/bash
# what do the detection tail, the staging copies and the NCHW -> NHWC copy cost the PIPELINE?  bench.py three times each
# way on one box (LSFA_LAB_SKIP drops the launches; results are garbage then, only frames/s is read).  -> gpurun_out/tail_ablation.txt
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=${OUT:-gpurun_out/tail_ablation.txt}
: > $O
one() {
  python bench.py --steps 60 --warmup 5 --no-parity --no-cpu-baseline --no-frame-by-frame --no-spread 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   %8.1f frames/s  (%.3f ms per interval)' % (d['value'], d['ms_per_step']))"
}
for rep in 1 2 3; do
  for skip in ${SKIPS:-none tail copy nhwc tail,copy,nhwc}; do      # (none: nothing dropped, the baseline of the same bench flags)
    echo "rep $rep  LSFA_LAB_SKIP='$skip'" >> $O
    LSFA_LAB_SKIP=$skip one >> $O 2>&1
  done
done
cat $O
