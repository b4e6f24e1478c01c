#!/bin/bash
# Tile shapes of the LDS-tiled dense kernel (k_linear_big<TMW, TNW>) on the large-row layer shapes of the pass, back to back
# (knob build): bash tools/big_shapes.sh   -> one line per shape and tile, us per launch
export SQAIR_TOOL_LIB=tools/bin/libsqair_hip_knobs.so
SHAPES="1920,362,1152,0 1920,312,768,0 2560,362,1152,0 2560,312,768,0 2560,256,256,1 2560,256,400,0 5120,362,1152,0 5120,312,768,0 5120,256,256,1 5120,256,400,0 6400,256,256,1 6400,256,400,0 4800,256,256,1 51200,256,256,1"
for t in 4,2 3,3 3,2 2,2 2,3 2,4 3,4 4,3 4,4; do
  echo "tile $t"
  SQAIR_BIG_SHAPE=$t timeout -s KILL 200 python tools/time_linear.py $SHAPES 2>&1 | grep "^M=" | awk '{print "  "$1,$2,$3,$6,$7}'
done
