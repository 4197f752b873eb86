set -e
cd $GRAFT_REPO_ROOT
# (record of the A/B that settled it; the build knob LH_ACC0_WAVES it toggled is gone: both variants measured 25.07 ms)
# A/B of the lazy-coordinate bucket accumulation (ec.cuh add_mixed_lazy): occupancy forced to 4 waves (default build, a few
# spilled registers) against the allocator's own choice (LH_ACC0_WAVES=0: 137 registers, 3 waves)
line() { python bench.py --no-inflight --no-cpu-baseline --steps 8 --warmup 3 2>> gpurun_out/ab_acc0.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], [(k['name'],k['ms']) for k in d['kernels'] if 'accumulate0' in k['name']])"; }
python -m pytest tests/test_gpu_parity_large.py tests/test_gpu_parity.py -m gpu -q -x -k "msm or g1 or commit" 2>&1 | tail -2
line "waves=4"
line "waves=4"
cd halo2-lasso_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics -DLH_ACC0_WAVES_UNUSED=0 -c msm.hip -o msm.o 2>> ../../gpurun_out/ab_acc0.err
make 2>> ../../gpurun_out/ab_acc0.err | tail -1
cd ../..
line "waves=auto"
line "waves=auto"
touch halo2-lasso_amd/csrc/msm.hip
make -C halo2-lasso_amd/csrc 2>> gpurun_out/ab_acc0.err | tail -1
