cd /root/repo
python -m pytest tests/test_hip_forward.py tests/test_hip_kernels.py -m gpu -q -k "long_sequence or generation_modes or insert_loglik or golden" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | grep "assert \|AssertionError\|passed\|failed\|worst" | cut -c1-200 | head -20
