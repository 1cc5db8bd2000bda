# Experiment builds of the library, one per cache policy of the streaming world-matrix stores (run HERE, on the CPU box;
# the .so files travel to the GPU box), then tools/ab_cache_policy_run.sh on the GPU.
set -e
cd yume_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fno-vectorize"
mkdir -p ab
for P in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc $F -DYH_STORE_POLICY=$P -shared -o ab/libyume_hip_store$P.so yume_hip.hip -ldl 2>&1 | grep -E "error" || true
done
ls -la ab/
