#!/bin/bash
# Round-end evidence on the FINAL build (run through tools/gpu.sh; then tools/summarize_round.sh rNN and commit profiles/): full GPU suite, smoke(),
# the profile round of the contract workload (bench line + kernel stats + every PMC pass), SQ + kernel stats of the other BASELINE configs (their
# `roofline` in other_configs), multi-rank rehearsals on the one GPU.
out=$RR_OUT; R=${ROUND:-r04}; G=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.txt
tools/profile_round.sh ${R} full > $out/profile.log 2>&1; tail -1 $out/profile.log | cut -c1-200
tools/profile_round.sh ${R}_helmet_syn full --scene helmet_syn --spp 64 --no-extras --no-cpu-baseline > $out/profile_helmet.log 2>&1; tail -1 $out/profile_helmet.log | cut -c1-120
tools/profile_round.sh ${R}_lotus_syn quick --scene lotus_syn --spp 512 > $out/profile_lotus.log 2>&1; tail -1 $out/profile_lotus.log | cut -c1-120
tools/profile_round.sh ${R}_monkey quick --scene monkey --width 800 --height 600 --spp 16 > $out/profile_monkey.log 2>&1; tail -1 $out/profile_monkey.log | cut -c1-120
tools/profile_round.sh ${R}_spheres quick --scene spheres --width 256 --height 256 --spp 1 --monte-carlo 0 > $out/profile_spheres.log 2>&1; tail -1 $out/profile_spheres.log | cut -c1-120
[ -x ./build/valu_issue ] && ./build/valu_issue > $G/${R}/valu_issue.txt 2>&1 || true
python tools/strong_scaling_probe.py > $G/${R}/scaling.txt 2>&1
python tools/update_probe.py > $G/${R}/update_probe.txt 2>&1; python tools/update_probe.py helmet_syn >> $G/${R}/update_probe.txt 2>&1
for n in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 3 --warmup 1 --dist-backend gloo --no-cpu-baseline > $G/${R}/gloo_$n.json 2> $out/gloo_$n.err || { echo "gloo $n failed"; tail -5 $out/gloo_$n.err; }
done
python bench.py --one-process --gpus 2 --same-device --steps 2 > $G/${R}/one_process.json 2> $out/one_process.err || tail -3 $out/one_process.err
echo final done
