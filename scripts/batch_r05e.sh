mkdir -p gpurun_out/r05e
timeout 600 python -m pytest tests/test_instantiations.py -q -k "stream" 2>&1 | tail -30 > gpurun_out/r05e/stream_tests.log
timeout 600 python -m pytest tests/test_hip_parity.py -q -k "k520 or k264 or k600 or fewer_workgroups or busy_stream" 2>&1 | tail -30 > gpurun_out/r05e/parity.log
for pf in 1 0; do
  KH_STREAM_PF=$pf timeout 300 python scripts/perf_sweeps.py 1024 64 1001 1 distinct 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r05e/perf.log
  KH_STREAM_PF=$pf timeout 300 python scripts/perf_sweeps.py 2048 64 501 1 distinct 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r05e/perf.log
  KH_STREAM_PF=$pf timeout 300 python scripts/perf_sweeps.py 768 64 1001 1 distinct 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r05e/perf.log
  KH_ENS=0 KH_STREAM_PF=$pf timeout 300 python scripts/perf_sweeps.py 1024 64 1001 1 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf shared-drift KH_ENS=0 /" >> gpurun_out/r05e/perf.log
done
cat gpurun_out/r05e/perf.log; tail -n 4 gpurun_out/r05e/stream_tests.log gpurun_out/r05e/parity.log
