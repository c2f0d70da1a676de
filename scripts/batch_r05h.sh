mkdir -p gpurun_out/r05h
for a in "256 160 101 1" "64 256 101 1" "256 128 101 2"; do
  timeout 600 python scripts/perf_sweeps.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05h/cliffs.txt
done
KH_KERNEL=generic timeout 600 python scripts/perf_sweeps.py 256 128 101 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05h/cliffs.txt
cat gpurun_out/r05h/cliffs.txt
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r05h/tests_full.log
tail -n 6 gpurun_out/r05h/tests_full.log
