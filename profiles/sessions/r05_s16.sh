# round 5, session 16: BDPT's connection kernel built for 2 (236 registers, as it was) / 3 (168 + 220 B of scratch) / 4 (128 + 344 B) waves per SIMD
O=gpurun_out/s16; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for lib in tuturenderer_amd/libtutu_hip.so build/libtutu_bd3.so build/libtutu_bd4.so; do
  for scene in cornell_box veach_room; do
    wh="--width 800 --height 800"; [ $scene = veach_room ] && wh="--width 800 --height 600"
    TUTU_HIP_LIB=$PWD/$lib timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu --scene $scene $wh 2> /dev/null | grep '"bdpt"' | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$lib $scene bdpt', round(d['value'],1), 'Ms/s', round(d['ms_per_frame_device'],2), 'ms')"
  done
done
done
