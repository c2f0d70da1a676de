mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_hip_parity.py -q -k "generic or kernel_families or edge_cases or nonuniform or several_controls or more_controls or sparse or second_order_update or sweeps_match" 2>&1 | tail -15 > gpurun_out/r05g/tests.log
for a in "256 64 501 6" "256 64 501 8" "256 64 501 5" "256 48 501 8"; do
  timeout 600 python scripts/perf_sweeps.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05g/cliffs.txt
done
KH_KERNEL=generic timeout 600 python scripts/perf_sweeps.py 256 64 501 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05g/cliffs.txt
KH_KERNEL=generic timeout 600 python scripts/perf_sweeps.py 256 96 201 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05g/cliffs.txt
cat gpurun_out/r05g/cliffs.txt; tail -n 5 gpurun_out/r05g/tests.log
