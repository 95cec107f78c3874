set -u
mkdir -p gpurun_out/r5trace
COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so timeout 300 python3 tools/trace_fill.py 10000 > gpurun_out/r5trace/trace_parts.txt 2>&1
COATI_HIP_CK_SPLIT=0 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so timeout 300 python3 tools/trace_fill.py 10000 > gpurun_out/r5trace/trace_noparts.txt 2>&1
COATI_HIP_CK_DEBUG=2 timeout 300 python3 tools/fill_loop.py 10000 3 > gpurun_out/r5trace/stats.txt 2>&1
COATI_HIP_CK_DEBUG=1 COATI_HIP_LIB=coati_amd/_build/libcoati_hip_trace.so timeout 300 python3 tools/trace_fill.py 10000 > gpurun_out/r5trace/trace_parts_fillonly.txt 2>&1
