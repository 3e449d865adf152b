#!/bin/bash
out=$RR_OUT
python tools/strong_scaling_probe.py > $out/scaling.txt 2>&1; cat $out/scaling.txt
cd /tmp && export TMPDIR=/tmp
for n in 8 1; do
  rocprofv3 --kernel-trace --output-format csv -d $out/tl$n -- python3 $RR_CODE_ROOT/tools/timeline.py run $n > $out/tl$n.log 2>&1 || { echo "trace failed"; tail -3 $out/tl$n.log; }
  python3 $RR_CODE_ROOT/tools/timeline.py show $out/tl$n > $out/timeline_$n.txt 2>&1; cat $out/timeline_$n.txt
done
