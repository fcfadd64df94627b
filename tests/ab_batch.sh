#!/bin/bash
# GPU script: one A/B round on ONE box -- the product's library, then every variant named (zip-ada_amd/ab/<name>.so, built with
# `make -C zip-ada_amd/csrc variant NAME=<name> DEFS=...` and copied there: variants/ does not travel), then the product again; a line of
# tests/gpu_phase.py per run.  usage: bash tests/ab_batch.sh [MiB] name...
mib=1024
case "$1" in ''|*[!0-9]*) ;; *) mib=$1; shift;; esac
run() { python tests/gpu_phase.py $mib 2>&1 | grep -v amdgpu | tail -1 | cut -c1-360; }
run
for v in "$@"; do ZADA_LIB=zip-ada_amd/ab/$v.so run; done
run
