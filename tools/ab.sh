#!/bin/bash
# A/B developer bench on ONE box: ab.sh "<bench args>" lib1 lib2 ...   (boxes differ by several %, never compare across calls)
args=$1; shift
for lib in "$@"; do
  echo "== $lib $args"
  RUSTRAY_HIP_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $args 2> /tmp/ab_err.txt > /tmp/ab_out.txt || { echo "bench failed:"; tail -n 3 /tmp/ab_err.txt; continue; }
  python -c "
import json; r=json.loads(open('/tmp/ab_out.txt').read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['kernel_ms_per_frame'].items()}, r['frame_checksum'])"
done
