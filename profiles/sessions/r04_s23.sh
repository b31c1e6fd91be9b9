set -x
mkdir -p gpurun_out/s23
export TMPDIR=/tmp
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/s23/bench_${c}_$tag.log 2>gpurun_out/s23/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s23/bench_${c}_$tag.log
}
L=$PWD/tuturenderer_amd/libtutu_hip_s8.so
for c in c2 c5; do
run base $c 6 TUTU_SETS=4
run base_q8 $c 6 TUTU_SETS=4 GPU_MAX_HW_QUEUES=8
run m8_s4 $c 6 TUTU_HIP_LIB=$L TUTU_SETS=4
run m8_s8 $c 6 TUTU_HIP_LIB=$L TUTU_SETS=8
run m8_s8_q8 $c 6 TUTU_HIP_LIB=$L TUTU_SETS=8 GPU_MAX_HW_QUEUES=8
run m8_s6_q8 $c 6 TUTU_HIP_LIB=$L TUTU_SETS=6 GPU_MAX_HW_QUEUES=8
run m8_s4_q8 $c 6 TUTU_HIP_LIB=$L TUTU_SETS=4 GPU_MAX_HW_QUEUES=8
done
