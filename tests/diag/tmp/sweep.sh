R="timeout -k 10 100 python tests/diag/ddim_ab.py 3"
$R 2>/dev/null && TTK_GEMM_TILE=1 $R 2>/dev/null && TTK_GEMM_TILE=2 $R 2>/dev/null && TTK_NO_FUSED_GN=1 $R 2>/dev/null && timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
