python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3
python bench.py --no-pmc --no-cpu-baseline --model-queries 0 --steps 24 > gpurun_out/s4_dp1.json 2> gpurun_out/s4_dp1.err; grep "PCIe-inclusive\|kernel ms" gpurun_out/s4_dp1.err
python bench.py --config C5 --no-pmc --no-cpu-baseline --model-queries 0 --steps 12 > gpurun_out/s4_dp1c5.json 2> gpurun_out/s4_dp1c5.err; grep "PCIe-inclusive\|kernel ms" gpurun_out/s4_dp1c5.err
