#!/bin/bash
# A/B of experimental library builds (build/exp/libpcdhip_<tag>.so) against the in-tree one: tools/ab_libs.sh "<perf_msm specs>" tag...
specs="$1"; shift
echo "== in-tree"; PERF_NOCHECK=${PERF_NOCHECK:-0} python tools/perf_msm.py $specs 2>&1 | grep "^msm"
for t in "$@"; do echo "== $t"; PCDHIP_LIB=$PWD/build/exp/libpcdhip_$t.so python tools/perf_msm.py $specs 2>&1 | grep "^msm"; done
