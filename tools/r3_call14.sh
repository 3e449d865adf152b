#!/bin/bash
out=$RR_OUT
tools/ab.sh "" build/lib_base.so build/lib_d16.so build/lib_w5.so build/lib_s5.so build/lib_c5.so build/lib_w6.so build/lib_base.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene helmet_syn --spp 64" build/lib_base.so build/lib_d16.so build/lib_w5.so build/lib_w6.so > $out/ab_helmet.txt 2>&1
tools/ab.sh "--scene monkey --width 800 --height 600 --spp 64" build/lib_base.so build/lib_d16.so build/lib_w5.so > $out/ab_monkey.txt 2>&1
tools/ab.sh "--scene lotus_syn --spp 128" build/lib_base.so build/lib_w5.so > $out/ab_lotus.txt 2>&1
cat $out/ab_sponza.txt $out/ab_helmet.txt $out/ab_monkey.txt $out/ab_lotus.txt
