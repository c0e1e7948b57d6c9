#!/bin/bash
cd "$(dirname "$0")/.."
for items in 256 600 1152 1728 2304; do
  echo "== 4x4 144->288 @128 target items $items"; FDGAN_DEBUG_WGRAD_ITEMS=$items python tools/wgrad_one.py 128 144 288 4 2>&1 | tail -1
done
for shape in "128 160 128 3" "32 640 512 3" "32 1024 256 3" "64 512 128 3" "128 72 144 3" "128 36 72 3" "128 128 32 3" "256 128 32 3"; do
  for items in 256 1024 2048; do
    echo "== $shape items $items"; FDGAN_DEBUG_WGRAD_ITEMS=$items python tools/wgrad_one.py $shape 2>&1 | tail -1
  done
done
