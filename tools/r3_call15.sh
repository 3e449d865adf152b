#!/bin/bash
out=$RR_OUT
tools/ab.sh "" build/lib_base.so build/lib_texd.so build/lib_base.so build/lib_texd.so > $out/ab_sponza.txt 2>&1
tools/ab.sh "--scene helmet_syn --spp 64" build/lib_base.so build/lib_texd.so build/lib_base.so build/lib_texd.so > $out/ab_helmet.txt 2>&1
cat $out/ab_sponza.txt $out/ab_helmet.txt
