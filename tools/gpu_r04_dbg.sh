#!/bin/bash
# backtrace of the host-side segfault seen in tests/test_gpu_parity.py::test_f64_every_buffer (third case)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for k in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_f64_every_buffer" > gpurun_out/dbg_run$k.log 2>&1; echo "run $k rc=$?"
done
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop" -ex run -ex bt -ex "info sharedlibrary libdqmc" --args python -m pytest tests/test_gpu_parity.py -x -q -k "test_f64_every_buffer" > gpurun_out/dbg_gdb.log 2>&1
grep -n "SIGSEGV\|^#" gpurun_out/dbg_gdb.log | head -40
