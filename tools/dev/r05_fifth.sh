#!/bin/bash
# round 5, fifth dev call: config 5 as two / three groups started out of phase
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05g_split_streams_offset.txt
: > $O
for off in 100 150 200 300 450; do
  python tools/dev/split_streams.py --groups 2 --offset-us $off >> $O 2>&1 || exit 1
done
python tools/dev/split_streams.py --groups 3 --batch 1020 --offset-us 100 >> $O 2>&1
python tools/dev/split_streams.py --groups 4 --offset-us 80 >> $O 2>&1
grep -v amdgpu.ids $O
