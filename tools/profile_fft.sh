#!/bin/bash
# tools/profile_fft.sh <tag>: rocprofv3 on the transform passes of the CURRENT sources (tools/fft_bench.py, coset FFT at 2^20 over the 298- and
# the 753-bit scalar field): a kernel trace (per-dispatch durations) and three counter runs of their own -- FETCH_SIZE, WRITE_SIZE, the SQ
# instruction mix -- as the MI355X guide prescribes (no trace domain beside --pmc; the program itself right after `--`).
# tools/fft_profile_summary.py then writes gpurun_out/<tag>_fft_pass.csv (per kernel and PASS: duration, HBM bytes, GB/s, VALU mix) and
# gpurun_out/traffic_fft_pass.json (stamped with the hash of fft.hip.h / fp.hip.h / inst_field.hip): copy both into profiles/.
tag=${1:-fft}
export TMPDIR=/tmp
export FFT_CASES=${FFT_CASES:-1:20,3:20} FFT_REPS=${FFT_REPS:-20}
mkdir -p gpurun_out
rm -rf /tmp/prof_${tag}_*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_trace -- python3 tools/fft_bench.py > gpurun_out/${tag}_fft_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_${tag}_fetch -- python3 tools/fft_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_${tag}_write -- python3 tools/fft_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS --output-format csv -d /tmp/prof_${tag}_sq -- python3 tools/fft_bench.py > /dev/null 2>&1
python3 tools/fft_profile_summary.py /tmp/prof_${tag}_trace /tmp/prof_${tag}_fetch /tmp/prof_${tag}_write /tmp/prof_${tag}_sq ${tag}
cat gpurun_out/${tag}_fft_bench.log | grep "fft field"
