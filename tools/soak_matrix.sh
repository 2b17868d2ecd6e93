#!/bin/bash
# r5: the long soak with the window kinds isolated on the closed loop path (SyntheticRoom(path="loop")).
# The default arc leaves the room at frame 194: that, not a window kind, was r4's drift.
set -e
N=${1:-300}
ROOM="{'path': 'loop'}"
run() { tag=$1; shift; python tools/soak_slam.py $N "$@" > gpurun_out/soak_$tag.log 2>&1; echo "$tag: $(tail -1 gpurun_out/soak_$tag.log) | max $(grep ' err ' gpurun_out/soak_$tag.log | awk '{if ($3>m) {m=$3; f=$1}} END {print m" cm at frame "f}') | tb $(grep -c ' tb 1' gpurun_out/soak_$tag.log) | kf $(grep ' err ' gpurun_out/soak_$tag.log | tail -1 | awk '{print $12}')"; }
run base 1 "{}" "$ROOM"
run noextra 1 "{'mapping': {'extra_rays': False}}" "$ROOM"
run nojoint 1 "{'mapping': {'joint_opt': False}}" "$ROOM"
run eager 0 "{}" "$ROOM"
run tex4 1 "{}" "{'path': 'loop', 'tex_freq': 4.0}"
