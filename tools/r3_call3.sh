#!/bin/bash
out=$RR_OUT
tools/ab.sh "" build/lib_base.so build/lib_abl_nojit.so build/lib_abl_nopow.so build/lib_abl_noacc.so build/lib_abl_noaux.so build/lib_abl_nosq.so build/lib_abl_notex.so build/lib_abl_nochild.so build/lib_abl_nolight.so build/lib_base.so > $out/abl_sponza.txt 2>&1
cat $out/abl_sponza.txt
