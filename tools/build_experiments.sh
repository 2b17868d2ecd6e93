#!/bin/bash
# tools/build_experiments.sh [extra flags]: build libunislam_hip.so WITH the measured-slower variants (include/unislam_hip_experiments.h:
# packed 8-byte records, the one-launch encode + decode kernel) so that tests/test_gpu_experiments.py and the timings of DESIGN.md 5d / 5e
# can be repeated.  `make -C uni-slam_amd/csrc` (or __graft_entry__.build()) afterwards restores the shipped library.
cd "$(dirname "$0")/.." || exit 1
touch uni-slam_amd/csrc/*.hip
make -s -j4 -C uni-slam_amd/csrc EXTRA="-DUS_EXPERIMENTS $*"
