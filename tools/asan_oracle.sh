#!/bin/bash
# tools/asan_oracle.sh -- the C restatement of the hash grid (oracle/hashgrid_ref.c) under AddressSanitizer + UBSan, CPU only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fopenmp $R/oracle/hashgrid_ref.c $R/tools/asan_oracle_main.c -lm -o /tmp/asan_oracle
ASAN_OPTIONS=detect_leaks=1 /tmp/asan_oracle
