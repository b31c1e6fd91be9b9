set -x
mkdir -p gpurun_out/s15
make -C tuturenderer_amd/csrc > /dev/null 2>&1
for v in dense sparse dense sparse; do
  if [ $v = sparse ]; then export TUTU_BDPT_SPARSE_EVENTS=1; else unset TUTU_BDPT_SPARSE_EVENTS; fi
  timeout -k 10 200 python profiles/bench_integrators.py --steps 4 --no-cpu 2>/dev/null | grep '"bdpt"' | cut -c1-110 | sed "s/^/$v /"
done
unset TUTU_BDPT_SPARSE_EVENTS
timeout -k 10 200 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2>/dev/null | grep '"bdpt"' | cut -c1-110
TUTU_BDPT_UNIT_KERNEL=1 timeout -k 10 200 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2>/dev/null | grep '"bdpt"' | cut -c1-110 | sed "s/^/unit /"
