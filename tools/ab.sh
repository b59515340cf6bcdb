#!/bin/bash
# A/B of the built library against ab/libbase.so on one box:  bash tools/ab.sh <tag> "<cmd>" [passes]
TAG=$1; CMD=$2; P=${3:-2}
O=gpurun_out; mkdir -p $O
for r in $(seq 1 $P); do
  echo "== base (pass $r)"; DHZ_LIB_PATH=$(pwd)/ab/libbase.so $CMD
  echo "== new (pass $r)"; $CMD
done > $O/${TAG}.txt 2>&1
