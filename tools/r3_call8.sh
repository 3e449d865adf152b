#!/bin/bash
out=$RR_OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_gpu.txt
python bench.py --one-process --gpus 2 --same-device --steps 2 > $out/one_process.json 2> $out/one_process.err || tail -5 $out/one_process.err
cat $out/one_process.json | cut -c1-900
for n in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --dist-backend gloo --no-cpu-baseline > $out/gloo_$n.json 2> $out/gloo_$n.err || { echo "gloo $n failed"; tail -8 $out/gloo_$n.err; }
  python - $out/gloo_$n.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith('{')]
if l:
    r=json.loads(l[-1]); keys=['value','ms_per_step','n_gpus','dist_backend','world_size','render_only_ms','gather_ms','gather_bytes_per_rank','frame_checksum','frame_checksum_single_gpu','frame_checksum_matches_single_gpu']
    print({k:r.get(k) for k in keys}); print('ranks',[(x['rank'],x['device'],x['region_pixels']) for x in r.get('ranks',[])]); print('one_process',{k:v for k,v in (r.get('one_process') or {}).items() if k!='config'})
PY
done
