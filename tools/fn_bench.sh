cd "$(dirname "$0")/.."
mkdir -p tools/ubench/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -shared "$@" tools/ubench/fn_bench.hip -o tools/ubench/build/libfn_bench.so && echo built
