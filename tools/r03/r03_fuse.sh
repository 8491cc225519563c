#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_api.py -q -k "fused_init or scalar or csr or CSR" > $O/pytest_fuse.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 3 $O/pytest_fuse.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/factor_fuse_ab.py 1:64 1:256 > $O/factor_fuse_ab3.txt 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $O/factor_fuse_ab3.txt
