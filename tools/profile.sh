#!/bin/bash
# tools/profile.sh <tag> <python script and args...>: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as
# the MI355X guide prescribes) of one command; summaries land in gpurun_out/<tag>_{stats,fetch,write}.csv -- copy into profiles/.
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_stats -- python3 "$@" > gpurun_out/${tag}_run.log 2>&1
python3 tools/pmc_summary.py /tmp/prof_${tag}_stats gpurun_out/${tag}_stats.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/prof_${tag}_$c -- python3 "$@" > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/prof_${tag}_$c gpurun_out/${tag}_$(echo $c | tr A-Z a-z | cut -d_ -f1).csv > /dev/null
done
if [ -n "$PROFILE_SQ" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/prof_${tag}_sq -- python3 "$@" > /dev/null 2>&1
  python3 tools/pmc_summary.py /tmp/prof_${tag}_sq gpurun_out/${tag}_sq.csv > /dev/null
fi
head -12 gpurun_out/${tag}_stats.csv; grep -i "fft_pass\|accumulate" gpurun_out/${tag}_fetch.csv gpurun_out/${tag}_write.csv
