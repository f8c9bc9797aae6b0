python tools/diag_inner_persist.py 2>&1 | grep -v amdgpu.ids | tail -12
