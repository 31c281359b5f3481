R="timeout -k 10 100 python tests/diag/ddim_ab.py 3"
$R 2>/dev/null && TTK_ATTN_QT=2 $R 2>/dev/null
