set -x
mkdir -p gpurun_out/s16
timeout -k 10 600 python -m pytest tests/test_hip_integrators.py -m gpu -q -x > gpurun_out/s16/tests.log 2>&1; tail -3 gpurun_out/s16/tests.log
for v in 1 2; do timeout -k 10 200 python profiles/bench_integrators.py --steps 4 --no-cpu 2>/dev/null | grep '"bdpt"' | cut -c1-110; done
timeout -k 10 200 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2>/dev/null | grep '"bdpt"' | cut -c1-110
