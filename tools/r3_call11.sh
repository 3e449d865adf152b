#!/bin/bash
out=$RR_OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_gpu.txt
RUSTRAY_HIP_LIB=$RR_CODE_ROOT/build/lib_base.so tools/profile_round.sh r03a_base quick > $out/prof_base.log 2>&1; tail -2 $out/prof_base.log | cut -c1-300
RUSTRAY_HIP_LIB=$RR_CODE_ROOT/build/lib_w3.so tools/profile_round.sh r03a_w3 quick > $out/prof_w3.log 2>&1; tail -2 $out/prof_w3.log | cut -c1-300
