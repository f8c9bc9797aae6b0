import sys, os, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
from graphslim_amd.structure import _SymEig
from graphslim_amd._lib import ptr, stream
torch.cuda.set_device(0)
e = _SymEig.get()
has = hasattr(e.solver, "rocsolver_ssyevdj")
print("ssyevdj available:", has)
e.solver.rocsolver_ssyevdj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
import os
if os.environ.get("GS_EIG_PREALLOC"):
    e.blas.rocblas_set_device_memory_size.argtypes = [C.c_void_p, C.c_size_t]
    print("rocblas_set_device_memory_size ->", e.blas.rocblas_set_device_memory_size(e.handle, int(os.environ["GS_EIG_PREALLOC"]) << 20))
for n in (70, 153, 454):
    a = torch.randn(n, n, device="cuda"); a = (a + a.T) / 2
    lam, work, info = torch.empty(n, device="cuda"), torch.empty(n, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ref = torch.linalg.eigvalsh(a.double().cpu())
    for name in ("ssyevd", "ssyevdj"):
        def run():
            A = a.clone()
            if name == "ssyevd":
                e.eigh(A, lam, work, info)
            else:
                e.blas.rocblas_set_stream(e.handle, stream())
                rc = e.solver.rocsolver_ssyevdj(e.handle, 211, 121, n, ptr(A), n, ptr(lam), ptr(info))
                assert rc == 0, rc
            return A
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): A = run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
        err = float((lam.double().cpu() - ref).abs().max())
        rec = A.t() @ torch.diag(lam) @ A          # rows are eigenvectors: Q = A^T
        print(f"n={n} {name}: {dt:.2f} ms  eig err {err:.2e}  reconstruction err {float((rec - a).abs().max()):.2e}")
