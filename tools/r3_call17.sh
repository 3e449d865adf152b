#!/bin/bash
out=$RR_OUT
tools/ab.sh "" build/lib_nopkw.so build/lib_pkw.so build/lib_nopkw.so build/lib_pkw.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene lotus_syn --spp 128" build/lib_nopkw.so build/lib_pkw.so > $out/ab_lotus.txt 2>&1
tools/ab.sh "--scene helmet_syn --spp 64" build/lib_nopkw.so build/lib_pkw.so > $out/ab_helmet.txt 2>&1
cat $out/ab_sponza.txt $out/ab_lotus.txt $out/ab_helmet.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -6 $out/pytest_gpu.txt
