# build the library of a git revision as smm.jl_amd/csrc/libsmmhip_a.so (A/B reference): tools/exp/build_ref.sh [rev]
set -e
rev=${1:-HEAD}
rm -rf /tmp/refsrc && mkdir -p /tmp/refsrc
git -C /root/repo archive $rev smm.jl_amd/csrc include | tar -x -C /tmp/refsrc
cd /tmp/refsrc/smm.jl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -shared -o /root/repo/smm.jl_amd/csrc/libsmmhip_a.so smmhip.hip
echo built libsmmhip_a.so from $rev
