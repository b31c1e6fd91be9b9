set -x
mkdir -p gpurun_out/s11
timeout -k 10 900 python -m pytest tests/test_hip_integrators.py -m gpu -q -s -x > gpurun_out/s11/tests.log 2>&1
tail -25 gpurun_out/s11/tests.log
timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu 2>/dev/null | grep "^{" > gpurun_out/s11/integrators.jsonl; cat gpurun_out/s11/integrators.jsonl | cut -c1-200
TUTU_BDPT_UNIT_KERNEL=1 timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu 2>/dev/null | grep "^{" > gpurun_out/s11/integrators_unit.jsonl; cat gpurun_out/s11/integrators_unit.jsonl | cut -c1-200
timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2>/dev/null | grep "^{" > gpurun_out/s11/integrators_veach.jsonl; cat gpurun_out/s11/integrators_veach.jsonl | cut -c1-200
