mkdir -p gpurun_out/r05f
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKH_TIMING -Iinclude krotov_amd/csrc/krotov_hip.hip -o gpurun_out/libkrotov_hip_timing.so 2>/dev/null
for pf in 1 0; do
KH_STREAM_PF=$pf python scripts/timing_stream.py 1024 1 --distinct 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r05f/timing.log
KH_STREAM_PF=$pf python scripts/timing_stream.py 768 1 --distinct 2>&1 | grep -v amdgpu.ids | sed "s/^/PF=$pf /" >> gpurun_out/r05f/timing.log
done
rm -f gpurun_out/libkrotov_hip_timing.so
cat gpurun_out/r05f/timing.log
