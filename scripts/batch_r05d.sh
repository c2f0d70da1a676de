mkdir -p gpurun_out/r05d
python -m pytest tests/test_hip_parity.py -q -k "bench_self_launches or bench_gpus_8" 2>&1 | tail -40 > gpurun_out/r05d/bench_tests.log
python -m pytest tests/test_instantiations.py -q -k "ell_" 2>&1 | tail -30 > gpurun_out/r05d/ell.log
python -m pytest tests/test_hip_parity.py -q -k "sparse or three_states or lindblad" 2>&1 | tail -30 > gpurun_out/r05d/sparse.log
hipcc --offload-arch=gfx950 -O3 scripts/ubench_gather.hip -o /tmp/ubench_gather 2>/dev/null; /tmp/ubench_gather > gpurun_out/r05d/ubench_gather.txt 2>&1
python scripts/perf_sparse.py 40 201 3 > gpurun_out/r05d/perf_sparse_d40.log 2>&1
KH_KERNEL=generic python scripts/perf_sparse.py 40 201 3 csr > gpurun_out/r05d/perf_sparse_d40_generic.log 2>&1
python scripts/perf_sparse.py 33 201 3 csr > gpurun_out/r05d/perf_sparse_d33.log 2>&1
python scripts/perf_sparse.py 25 501 16 csr > gpurun_out/r05d/perf_sparse_d25.log 2>&1
python scripts/perf_sweeps.py 1024 64 1001 1 distinct > gpurun_out/r05d/k1024_distinct.log 2>&1
python scripts/perf_sweeps.py 2048 64 501 1 distinct > gpurun_out/r05d/k2048_distinct.log 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKH_TIMING -Iinclude krotov_amd/csrc/krotov_hip.hip -o gpurun_out/libkrotov_hip_timing.so 2>/dev/null
python scripts/timing_stream.py 1024 1 --distinct > gpurun_out/r05d/timing_stream_distinct.log 2>&1
KH_ENS=0 python scripts/timing_stream.py 1024 1 > gpurun_out/r05d/timing_stream_shared.log 2>&1
python scripts/timing_coop.py > gpurun_out/r05d/config4_timing.txt 2>&1
rm -f gpurun_out/libkrotov_hip_timing.so
tail -n 3 gpurun_out/r05d/*.log
