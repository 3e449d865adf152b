#!/bin/bash
# developer bench: same library and args, several environment settings: ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...
args=$1; shift
for e in "$@"; do
  echo "== [$e] $args"
  env $e python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), round(r['ms_per_step'],2), {k: round(v,2) for k,v in r['kernel_ms_per_frame'].items()}, r['frame_checksum'])"
done
