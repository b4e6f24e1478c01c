export SQAIR_TOOL_LIB=tools/bin/libsqair_hip_knobs.so
SH="640,362,1152,0 640,312,768,0 640,400,256,1 640,256,256,1 640,256,109,0 1280,362,1152,0 1280,256,256,1"
echo "== default dispatch"; python tools/time_linear.py $SH 2>&1 | grep "M="
for s in 1,2 1,3 1,4 2,2; do echo "== big $s from 600 rows"; SQAIR_MT_ROWS=600 SQAIR_BIG_SHAPE=$s python tools/time_linear.py $SH 2>&1 | grep "M="; done
