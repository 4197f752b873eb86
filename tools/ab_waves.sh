set -e
cd $GRAFT_REPO_ROOT
# A/B of the occupancy hint on the degree-2 streaming round kernels: the build as committed (4 waves per SIMD), then
# LH_SC_WAVES_D2=0 (no hint: the register allocator settles on 3)
python bench.py --no-inflight --no-cpu-baseline --steps 8 > gpurun_out/r02_ab_w4.json 2> gpurun_out/r02_ab.err
cd halo2-lasso_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics -DLH_SC_WAVES_D2=0 -c kernels_sumcheck.hip -o kernels_sumcheck.o 2>> ../../gpurun_out/r02_ab.err
make 2>> ../../gpurun_out/r02_ab.err | tail -1
cd ../..
python bench.py --no-inflight --no-cpu-baseline --steps 8 > gpurun_out/r02_ab_base.json 2>> gpurun_out/r02_ab.err
python -m pytest tests/test_gpu_parity_large.py -m gpu -q -k "sum_check or lasso" 2>&1 | tail -2
# restore the default build: the tree's library must never stay the variant (later bench / profile lines would be
# measured on the wrong build)
touch halo2-lasso_amd/csrc/kernels_sumcheck.hip
make -C halo2-lasso_amd/csrc 2>> gpurun_out/r02_ab.err | tail -1
