#!/bin/bash
# tools/isa_of.sh <gen/file.hip> [extra hipcc flags]: ISA of the non-profiling instance of a generated kernel -> /tmp/isa/<name>.s
f=$1; shift
n=$(basename $f .hip); mkdir -p /tmp/isa/$n; cd /tmp/isa/$n
flags=$(sed -n '1s/^\/\/ hipcc-flags://p' /root/repo/deepqmc_amd/csrc/$f)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include $flags "$@" -save-temps -c /root/repo/deepqmc_amd/csrc/$f -o x.o 2>/dev/null
S=$(ls *gfx950.s | head -1)
awk '/^_ZN4dqmc.*ILb0E.*:/{p=1} p{print} /s_endpgm/{if(p){exit}}' $S | grep -v "^\s*;" > /tmp/isa/$n.s
grep -E "\.set.*ILb0E.*(num_vgpr|num_agpr|numbered_sgpr|private_seg_size)" $S | awk '{print $2, $3}' | sed 's/.*\.//' | tr '\n' ' '; echo
wc -l < /tmp/isa/$n.s
