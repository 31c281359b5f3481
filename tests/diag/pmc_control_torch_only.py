"""Control for the rocprofv3 --pmc fault (profiles/r04_pmc_eager_probe.log): N tiny kernel launches issued by torch alone -- libttk is not even loaded.
   rocprofv3 --pmc FETCH_SIZE -- python3 -X faulthandler tests/diag/pmc_control_torch_only.py 80000"""
import sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
x = torch.zeros(4096, device="cuda:0")
for i in range(n):
	x.add_(1.0)
	if i % 10000 == 0:
		torch.cuda.synchronize()
		print("launched", i, flush=True)
torch.cuda.synchronize()
print("done", n, float(x[0]), flush=True)
