mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_witness_like.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/t3_witness.txt
AB_WITNESS=1 AB_SCHEDULES=0,2:8,2:0,0 timeout 600 python tools/ab_step.py 0 20 > gpurun_out/r05/ab_c0_witness.txt 2>&1
AB_WITNESS=1 AB_SCHEDULES=0,2:8,0 timeout 600 python tools/ab_step.py 1 16 > gpurun_out/r05/ab_c1_witness.txt 2>&1
AB_WITNESS=1 timeout 900 python tools/acc_probe.py 0 20 > gpurun_out/r05/acc_probe_c0_witness.txt 2>&1
cd /tmp && rm -rf /tmp/pt
PT_WITNESS=1 PT_SCHED=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/tools/proof_timeline.py run > $GRAFT_REPO_ROOT/gpurun_out/r05/pt_run_witness.txt 2>&1
cd $GRAFT_REPO_ROOT && PT_MIN_US=60 python3 tools/proof_timeline.py report /tmp/pt > gpurun_out/r05/pt_witness.txt 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/t4_all.txt
cat gpurun_out/r05/t3_witness.txt gpurun_out/r05/ab_c0_witness.txt gpurun_out/r05/ab_c1_witness.txt gpurun_out/r05/acc_probe_c0_witness.txt gpurun_out/r05/t4_all.txt
