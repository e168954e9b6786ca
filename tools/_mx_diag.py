import sys, torch
sys.path.insert(0, "/root/repo")
from mellow_amd.engine import Engine
eng = Engine(device=0, max_positions=1024, precision="fp8")
torch.manual_seed(0)
M, N, K = 256, 256, 576
A = torch.randn(M, K); W = torch.randn(N, K) * 0.05
ref = A.double() @ W.double().T
for rep in range(3):
    C, _ = eng.debug_gemm_fp8(A, W)
    bad = ~torch.isfinite(C)
    cols = sorted(set(bad.nonzero()[:, 1].tolist()))
    print("rep", rep, "nonfinite", int(bad.sum()), "cols", cols)
    for c in cols[:2]:
        rows = bad[:, c].nonzero()[:, 0].tolist()
        print("  col", c, "bad rows", rows[:40], "n bad", len(rows))
        print("  values", C[rows[:6], c].tolist(), "good neighbours", C[[0, 2, 4], c].tolist(), "ref", ref[[0, 2, 4], c].tolist())
        w = W[c]; print("  W row amax", float(w.abs().max()), "argmax k", int(w.abs().argmax()), "A at that k for bad rows", A[rows[:6], int(w.abs().argmax())].tolist(), "good rows", A[[0, 2, 4], int(w.abs().argmax())].tolist())
# same data, zero row scales? single column test: W with only one nonzero row
