#!/bin/bash
# Build the ABLATION library tools/probe_lin.sh times: the product sources with -DDQMC_LIN_PROBE (kernel_linear.hip's cfg_probe hooks) in a
# scratch copy of csrc/ -> deepqmc_amd/csrc/probe/libdqmc_probe.so (git-ignored; travels to the GPU box with the snapshot).  Never loaded by the package.
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/dqmc_probe_build && mkdir -p /tmp/dqmc_probe_build/deepqmc_amd && cp -r deepqmc_amd/csrc /tmp/dqmc_probe_build/deepqmc_amd/ && cp -r include /tmp/dqmc_probe_build/
cd /tmp/dqmc_probe_build/deepqmc_amd/csrc && rm -f *.o gen/*.o *.so
make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../../include -DDQMC_LIN_PROBE" > /tmp/dqmc_probe_build/make.log 2>&1
mkdir -p "$OLDPWD/deepqmc_amd/csrc/probe" && cp libdqmc_hip.so "$OLDPWD/deepqmc_amd/csrc/probe/libdqmc_probe.so"
ls -la "$OLDPWD/deepqmc_amd/csrc/probe/libdqmc_probe.so"
