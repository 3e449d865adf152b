#!/bin/bash
# developer bench: same library, several bench.py argument sets (one per line of arguments, separated by ';')
IFS=';' read -ra SETS <<< "$1"
for a in "${SETS[@]}"; do
  echo "== $a"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['kernel_ms_per_frame'].items()}, r['frame_checksum'], r['rays_per_frame'])"
done
