#!/bin/bash
# A/B developer bench: runs bench.py against several builds of the library (RUSTRAY_HIP_LIB) on one box.
for lib in "$@"; do
  echo "== $lib"
  RUSTRAY_HIP_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['kernel_ms_per_frame'].items()}, r['frame_checksum'])"
done
