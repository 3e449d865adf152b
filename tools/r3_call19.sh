#!/bin/bash
out=$RR_OUT
for sc in "--scene monkey --width 800 --height 600 --spp 64" "--scene helmet_syn --spp 64" "--scene room_kbert --spp 16" "--scene spheres_room --spp 64" "--scene monkey_room --spp 64"; do
tools/ab.sh "$sc" build/lib_cur.so build/lib_bm1.so build/lib_cur.so build/lib_bm1.so >> $out/ab.txt 2>&1
done
cat $out/ab.txt
