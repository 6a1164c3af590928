set -e
timeout -k 10 900 python -m pytest tests/test_gpu_mesh_box.py tests/test_gpu_api.py tests/test_operator_study.py -x -q 2>&1 | tail -15
timeout -k 10 300 python tools/api_case.py 128 2>&1 | tail -3
API_WARM=1 timeout -k 10 300 python tools/api_case.py 128 2>&1 | tail -3
timeout -k 10 300 python tools/prof_api.py 128 > gpurun_out/prof_api2.log 2>&1
