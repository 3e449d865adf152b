#!/bin/bash
# Round-end evidence on the FINAL build (run through tools/gpu.sh; then tools/summarize_profiles.py rNN rNN and commit profiles/): full GPU suite, smoke(), profile round (bench line + kernel stats + PMC passes), multi-rank rehearsals
out=$RR_OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.txt
tools/profile_round.sh ${ROUND:-r03} > $out/profile.log 2>&1; tail -2 $out/profile.log | cut -c1-300
./build/valu_issue > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/valu_issue.txt 2>&1 || true
python tools/strong_scaling_probe.py > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/scaling.txt 2>&1
for n in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 3 --warmup 1 --dist-backend gloo --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/gloo_$n.json 2> $out/gloo_$n.err || { echo "gloo $n failed"; tail -5 $out/gloo_$n.err; }
done
python bench.py --one-process --gpus 2 --same-device --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/${ROUND:-r03}/one_process.json 2> $out/one_process.err || tail -3 $out/one_process.err
echo final done
