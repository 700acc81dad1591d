mkdir -p gpurun_out/r5a
export TRON_TUNING=1
timeout 900 python -m pytest tests/test_gpu_arc.py -x -q -m gpu > gpurun_out/r5a/arc_tests.log 2>&1
cp tron_amd/lib/libtronhip.so /tmp/orig.so
for a in "1 128" "2 128" "8 128"; do WARM=20 python tools/gridbench.py $a fast 5 2>&1 | tail -1; done > gpurun_out/r5a/gridbench.log
cp tron_amd/lib/libtronhip_aprof.so tron_amd/lib/libtronhip.so
for a in "1 128" "2 128"; do WARM=20 python tools/arcprof.py $a 2>&1 | tail -14; done > gpurun_out/r5a/phase.log
cp /tmp/orig.so tron_amd/lib/libtronhip.so
tail -3 gpurun_out/r5a/arc_tests.log; cat gpurun_out/r5a/gridbench.log gpurun_out/r5a/phase.log
