#!/bin/bash
# GPU script: BZip2_3 on 256 MiB of the benchmark stream under the knob settings given as lines of "NAME=value ..." on stdin (default: a built-in list),
# one line of tests/gpu_bz2_perf.py per setting, the defaults first and last.  usage: bash tests/bz_knob_sweep.sh < settings
run() { echo "== $*"; env "$@" BZ_MIBS=${BZ_MIBS:-256} python tests/gpu_bz2_perf.py 2>&1 | grep -v amdgpu | grep "MiB:" ; }
run X=1
if [ -t 0 ]; then set -- ; else while read -r line; do [ -n "$line" ] && run $line; done; fi
run X=1
