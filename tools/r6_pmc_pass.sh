# One gpurun call of round 6: LDS / matrix-pipe counters of the hidden-layer kernel (this build and the no-staging diagnostic build),
# its HBM traffic, the environment kernels' HBM traffic, then the headline bench under the round's base build and under this build.
export RUBIKS_HIP_LIB_B=/root/repo/rl-rubiks_amd/lib/ab/abl1.so
bash tools/bench_rocprof.sh r6 gemm_lds
unset RUBIKS_HIP_LIB RUBIKS_HIP_LIB_B
bash tools/bench_rocprof.sh r6 gemm_traffic
bash tools/bench_rocprof.sh r6 env_traffic
cat gpurun_out/r6_gemm_lds_pmc.txt gpurun_out/r6_gemm_traffic.txt
for v in base_r6 tree; do
  if [ $v = tree ]; then unset RUBIKS_HIP_LIB; else export RUBIKS_HIP_LIB=/root/repo/rl-rubiks_amd/lib/ab/$v.so; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --legs f32s --extra-legs "" --no-cpu-baseline --no-env-roofline > gpurun_out/r6e_bench_$v.json 2> gpurun_out/r6e_bench_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/r6e_bench_$v.json').read().strip().splitlines()[-1]); r=d['config']['results']; print('$v', d['value'], d['ms_per_step'], r['value_run_to_completion'], r['value_pool_run'], d['roofline']['frac'])"
done
