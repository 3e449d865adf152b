#!/bin/bash
out=$RR_OUT
tools/ab.sh "" build/lib_nopkw.so build/lib_pkc.so build/lib_nopkw.so build/lib_pkc.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene monkey --width 800 --height 600 --spp 64" build/lib_nopkw.so build/lib_pkc.so > $out/ab_monkey.txt 2>&1
tools/ab.sh "--scene room_kbert --spp 64" build/lib_nopkw.so build/lib_pkc.so > $out/ab_kbert.txt 2>&1
cat $out/ab_sponza.txt $out/ab_monkey.txt $out/ab_kbert.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_gpu.txt
python tools/strong_scaling_probe.py > $out/scaling.txt 2>&1; cat $out/scaling.txt
